O=gpurun_out/r04_r; mkdir -p $O
{
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -x -q -k "attention or train_step or gradients" 2>&1 | tail -3
for rep in 1 2; do for v in tools/_ab/lib_prev.so mgsv_amd/libmade_hip.so; do MADE_LIB_PATH=$PWD/$v timeout 300 python tools/attn_pmc_target.py 2>&1 | grep -v amdgpu; done; done
echo "### step A/B"
for rep in 1 2; do for v in tools/_ab/lib_prev.so mgsv_amd/libmade_hip.so; do MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-28s' % '$v', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in d['kernels'].items() if 'attention' in k])"; done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
