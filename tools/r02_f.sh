R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/f_tests.txt 2>&1; tail -5 $O/f_tests.txt
timeout 900 python bench.py > $O/f_bench_all.json 2>$O/f_bench_all.err; tail -c 600 $O/f_bench_all.err
python - <<PY
import json
d=json.load(open("$O/f_bench_all.json"))
print("train", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
r=d["retrieval"]; print("retrieval", r["value"], r["ms_per_step"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"].get("traffic"))
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in d["eval_fwd"].items() if isinstance(v, dict)})
PY
