R=$GRAFT_REPO_ROOT; cd $R
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
python bench.py --workload train --steps 20 --warmup 5 2>/dev/null | python -c "$P"
python bench.py 2>/dev/null | python -c "$P"
python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
python tools/xpool_qk_bench.py 2>&1 | head -10
