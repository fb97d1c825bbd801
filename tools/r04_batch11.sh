O=gpurun_out/r04_k; mkdir -p $O
{
for v in tools/_ab/lib_prev.so tools/_ab/lib_pb4.so tools/_ab/lib_pb8.so tools/_ab/lib_pb16.so mgsv_amd/libmade_hip.so; do echo $v; MADE_LIB_PATH=$PWD/$v timeout 300 python tools/pool_bwd_bench.py 2>&1 | grep -v amdgpu; done
echo "### step A/B"
for rep in 1 2; do for v in tools/_ab/lib_prev.so tools/_ab/lib_pb4.so tools/_ab/lib_pb8.so tools/_ab/lib_pb16.so; do MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-28s' % '$v', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')')"; done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
