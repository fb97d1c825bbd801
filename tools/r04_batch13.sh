O=gpurun_out/r04_n; mkdir -p $O
{
for v in mgsv_amd/libmade_hip.so; do
  echo "### $v"
  MADE_LIB_PATH=$PWD/$v timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -k "xpool_fused or retrieval" 2>&1 | tail -2
  MADE_BENCH_RETRIEVAL_512=0 MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload retrieval --no-cpu-baseline --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], 'ms', [(k, round(v2['ms_per_step'],2)) for k, v2 in d['kernels'].items()][:3])"
  MADE_LIB_PATH=$PWD/$v timeout 300 python tools/xpool_stamps.py 96 2>&1 | grep -v amdgpu | grep -A4 "wave 0\|wave 4" | grep "wave\|it 5\|it 6"
done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
