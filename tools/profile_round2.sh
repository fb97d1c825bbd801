# The round's profile set (run through gpurun from the repo root):  bash tools/profile_round2.sh <tag>
# -> gpurun_out/<tag>/: bench lines (default = train + retrieval + eval sub-objects), rocprofv3 --kernel-trace --stats of the three workloads,
#    PMC passes (FETCH_SIZE / WRITE_SIZE, separately) of the training step and the retrieval pass -> pmc_summary.json
set -x
TAG=${1:-r02x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2>$O/bench_default.err
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2>/dev/null
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) adam_update_kernel 70 > $O/train_trace_summary.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --workload forward --steps 10 --warmup 3 --no-cpu-baseline --launch eager --in-flight 1 > $O/bench_eval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eval_eager_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) sine_pe_kernel 50 > $O/eval_trace_summary.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/bench.py --workload retrieval --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_retrieval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_retrieval_bf16.csv
rm -rf /tmp/p4; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p4 -- python3 $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/p5; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p5 -- python3 $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $(find /tmp/p5 -name "*counter_collection.csv" | head -1) $O/pmc_train.json > /dev/null
rm -rf /tmp/p6; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p6 -- python3 $R/bench.py --workload forward --steps 2 --warmup 1 --no-cpu-baseline --launch eager --in-flight 1 > /dev/null 2>&1
rm -rf /tmp/p7; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p7 -- python3 $R/bench.py --workload forward --steps 2 --warmup 1 --no-cpu-baseline --launch eager --in-flight 1 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p6 -name "*counter_collection.csv" | head -1) $(find /tmp/p7 -name "*counter_collection.csv" | head -1) $O/pmc_eval.json > /dev/null
rm -rf /tmp/p8; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p8 -- python3 $R/bench.py --workload retrieval --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/p9; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p9 -- python3 $R/bench.py --workload retrieval --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p8 -name "*counter_collection.csv" | head -1) $(find /tmp/p9 -name "*counter_collection.csv" | head -1) $O/pmc_retrieval.json > /dev/null
python3 - <<PY
import json
out = {}
for leg in ("train", "eval", "retrieval"):
    try:
        out[leg] = json.load(open("$O/pmc_%s.json" % leg))
    except Exception as e:
        out[leg] = {"error": str(e)}
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
PY
ls -la $O
python3 -c "
import json
d=json.load(open('$O/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
print(d['retrieval']['value'], d['retrieval']['ms_per_step'], d['retrieval']['roofline'])
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['eval_fwd'].items() if isinstance(v, dict)})
"
