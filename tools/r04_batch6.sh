O=gpurun_out/r04_f; mkdir -p $O
{
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear_encoder_sized or matcher_golden" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_train_ops_gpu.py -x -q -k "big_tile" 2>&1 | tail -8
timeout 600 python tools/big_gemm_bench.py
timeout 600 python - <<PY
import json, sys, types
sys.argv=["bench.py"]
import bench, os
args = types.SimpleNamespace(dtype="bf16")
print("S512_D512 new path:", json.dumps(bench._retrieval_512(args)))
PY
} > $O/big.txt 2>&1
grep -v amdgpu.ids $O/big.txt
