O=gpurun_out/r04_e; mkdir -p $O
{
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "xpool_attention or matcher" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "two_pass" 2>&1 | tail -15
timeout 300 python tools/xattn_bench.py
timeout 300 python tools/xattn_bench.py 8192 64 96 256
timeout 600 python - <<PY
import json, sys, types
sys.argv=["bench.py"]
import bench, os
args = types.SimpleNamespace(dtype="bf16")
print("S512_D512 new path:", json.dumps(bench._retrieval_512(args)))
os.environ["MADE_XPOOL_ATTN"]="0"
print("S512_D512 old chain:", json.dumps(bench._retrieval_512(args)))
PY
} > $O/xattn.txt 2>&1
grep -v amdgpu.ids $O/xattn.txt
