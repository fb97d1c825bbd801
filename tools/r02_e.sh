R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 2>/dev/null | python -c "$P"
python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
