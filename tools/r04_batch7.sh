O=gpurun_out/r04_g; mkdir -p $O
{
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "big_tile or linear_encoder_sized" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "two_pass or retrieval" 2>&1 | tail -5
timeout 600 python tools/big_gemm_bench.py 2>&1 | head -3
timeout 600 python tools/retr512_breakdown.py
timeout 600 python bench.py --workload retrieval --no-cpu-baseline > $O/bench_retrieval.json 2>$O/bench_retrieval.err; python -c "
import json; d=json.load(open('$O/bench_retrieval.json')); print('retrieval', d['ms_per_step'], 'ms', d['value'], d['unit'], json.dumps(d['config'].get('S512_D512')))"
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
