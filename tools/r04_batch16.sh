O=gpurun_out/r04_q; mkdir -p $O
{
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_ops_gpu.py -x -q -k "linear" 2>&1 | tail -3
echo "### step A/B: MADE_LINEAR_BIG_TRAIN = least live 128 x 256 tiles for the big-tile kernel with the straight-line training epilogue (0 = off)"
for rep in 1 2; do for v in 0 256 400 700; do MADE_LINEAR_BIG_TRAIN=$v timeout 300 python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('BIG_TRAIN=%-5s' % '$v', d['ms_per_step'], 'ms (eager', d['config'].get('eager_ms_per_step'), ')', [(k, round(v2['ms_per_step'],3)) for k, v2 in d['kernels'].items() if 'linear' in k][:5])"; done; done
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt
