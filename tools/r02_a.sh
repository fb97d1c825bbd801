set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear" > $O/a_test_linear.txt 2>&1; tail -3 $O/a_test_linear.txt
timeout 600 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/a_train_ring128.json 2>$O/a_err1.txt
MADE_LINEAR_TILE=1000 timeout 600 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/a_train_r01.json 2>$O/a_err2.txt
MADE_LINEAR_TILE=2256 timeout 600 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/a_train_ring256.json 2>$O/a_err3.txt
timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline > $O/a_eval_ring128.json 2>$O/a_err4.txt
MADE_LINEAR_TILE=1000 timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline > $O/a_eval_r01.json 2>$O/a_err5.txt
MADE_LINEAR_TILE=2256 timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline > $O/a_eval_ring256.json 2>$O/a_err6.txt
timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 > $O/a_eval1_ring128.json 2>$O/a_err7.txt
MADE_LINEAR_TILE=2256 timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 > $O/a_eval1_ring256.json 2>$O/a_err8.txt
for f in $O/a_*.json; do echo $f; python - <<PY
import json
d=json.load(open("$f"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"] if d.get("roofline") else None, d["roofline"]["frac"] if d.get("roofline") else None)
for k,v in list(d.get("kernels",{}).items())[:8]: print("   ",k,v)
PY
done
tail -5 $O/a_err*.txt
