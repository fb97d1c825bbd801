R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "dec_stage or linear" > $O/b_test_ops.txt 2>&1; tail -5 $O/b_test_ops.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q > $O/b_test_engine.txt 2>&1; tail -15 $O/b_test_engine.txt
timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 > $O/b_eval1.json 2>$O/b_err1.txt
timeout 600 python bench.py --workload forward --steps 40 --warmup 5 --no-cpu-baseline > $O/b_eval2.json 2>$O/b_err2.txt
for f in $O/b_eval1.json $O/b_eval2.json; do echo $f; python - <<PY
import json
d=json.load(open("$f"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"] if d.get("roofline") else None, d["roofline"]["frac"] if d.get("roofline") else None)
for k,v in list(d.get("kernels",{}).items())[:10]: print("   ",k,v)
PY
done
tail -n 5 $O/b_err1.txt $O/b_err2.txt
