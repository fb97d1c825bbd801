O=gpurun_out/r04_h; mkdir -p $O
{
echo "### new tests"
timeout 1200 python -m pytest tests/test_corun_determinism_gpu.py tests/test_train_dp_gpu.py tests/test_multi_gpu_nccl.py -x -q 2>&1 | tail -6
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "pack_music or matcher_golden" 2>&1 | tail -4
echo "### step A/B: big 128 x 256 tiles forced for every launch they support"
for rep in 1 2; do
for kv in "" "MADE_LINEAR_TILE=256"; do env $kv timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-24s' % '${kv:-default}', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in list(d['kernels'].items())[:4]], d['config'].get('executed_gflop_per_step'), d['config'].get('mfma_frac_over_step'))"; done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
