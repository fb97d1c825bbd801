R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/d_train.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$O/d_train.json")); print("train", d["value"], d["ms_per_step"])
PY
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/d_train_under_rocprof.json 2>/dev/null
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/d_kernel_stats_train_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) 15 70 > $O/d_train_trace_summary.txt
head -100 $O/d_train_trace_summary.txt
