O=gpurun_out/r04_p; mkdir -p $O
{
echo "### step A/B: product build vs a build whose dropout hash is one multiply (timing probe, wrong masks)"
for rep in 1 2; do for v in mgsv_amd/libmade_hip.so tools/_ab/lib_freerng.so; do MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-28s' % '$v', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in d['kernels'].items()][:8])"; done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
