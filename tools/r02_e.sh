R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_train_ops_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -3
python tools/xpool_qk_bench.py 2>&1 | tail -8
python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in list(d['kernels'].items())[:9]]"
python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eval1', d['value'], d['ms_per_step'])"
python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eval2', d['value'], d['ms_per_step'])"
