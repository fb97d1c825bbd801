# round 4, batch 4: the library without packed-FP32 instructions against the packed build (training step A/B on one box), determinism probes, GPU suite
O=gpurun_out/r04_d; mkdir -p $O
ab() { MADE_LIB_PATH=$1 timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-28s' % '$2', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in list(d['kernels'].items())[:6]])"; }
{
for rep in 1 2; do ab $PWD/tools/_ab/lib_packed.so packed; ab $PWD/mgsv_amd/libmade_hip.so no-packed-fp32; done
echo "### step, MADE_RET_SPLIT=0 (tape), product build"; MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
echo "### micro, product build"; timeout 600 python tools/dec_corun_probe.py micro 2000
echo "### eval A/B"
for rep in 1 2; do for v in tools/_ab/lib_packed.so mgsv_amd/libmade_hip.so; do MADE_LIB_PATH=$PWD/$v timeout 300 python bench.py --workload forward --no-cpu-baseline --steps 100 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['value'])"; done; done
echo "### retrieval A/B (xpool_fused keeps its packed math in both)"
} > $O/ab_packed.txt 2>&1
grep -v amdgpu.ids $O/ab_packed.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
