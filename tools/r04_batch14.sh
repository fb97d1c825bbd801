O=gpurun_out/r04_o; mkdir -p $O
{
for rep in 1 2 3; do for v in tools/_ab/lib_prev.so mgsv_amd/libmade_hip.so tools/_ab/lib_late0.so; do
  MADE_BENCH_RETRIEVAL_512=0 MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload retrieval --no-cpu-baseline --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], 'ms', [(k, round(v2['ms_per_step'],2)) for k, v2 in d['kernels'].items()][:1])"
done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
