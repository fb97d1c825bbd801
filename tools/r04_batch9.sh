O=gpurun_out/r04_i; mkdir -p $O
{
timeout 900 python -m pytest tests/test_train_ops_gpu.py -x -q -k "attention_forward_dropout" 2>&1 | tail -6
for rep in 1 2; do BITS=0 timeout 300 python tools/attn_pmc_target.py; BITS=1 timeout 300 python tools/attn_pmc_target.py; done
echo "### step A/B: keep bits"
for rep in 1 2; do
for kv in "MADE_ATTN_BITS=0" "MADE_ATTN_BITS=1"; do env $kv timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-24s' % '${kv:-default}', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in d['kernels'].items() if 'attention' in k])"; done; done
timeout 1200 python -m pytest tests/test_trainer_gpu.py -x -q 2>&1 | tail -4
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
