# The round-6 profile set (the round-4 recipe under r06 names; run through gpurun from the repo root):  bash tools/profile_round5.sh <tag>
# -> gpurun_out/<tag>/: the default bench line, rocprofv3 --kernel-trace --stats of the three legs (+ per-kind average launch durations:
#    kernel_avg_us.json, what bench.py quotes as avg_launch_us_rocprof_committed), PMC passes (FETCH_SIZE / WRITE_SIZE, separately) of
#    the training step, the eval forward and the retrieval pass -> pmc_summary.json, and the north-star micro-benchmark
#    (tools/xpool_qk_bench.py) under --stats and the two PMC passes.
set -x
TAG=${1:-r06x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2>$O/bench_default.err
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2>/dev/null
T1=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_bf16.csv; python3 $R/tools/trace_summary.py $T1 adam_update_kernel 90 > $O/train_trace_summary.txt
python3 $R/tools/trace_timeline.py $T1 adam_update_kernel -7 > $O/train_step_timeline_tape_under_rocprof.txt
python3 $R/tools/kernel_avg.py train $T1 $O/kernel_avg_us.json > /dev/null
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --workload forward --steps 10 --warmup 3 --no-cpu-baseline --launch eager --in-flight 1 > $O/bench_eval_under_rocprof.json 2>/dev/null
T2=$(find /tmp/p2 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eval_eager_bf16.csv; python3 $R/tools/trace_summary.py $T2 sine_pe_kernel 50 > $O/eval_trace_summary.txt
python3 $R/tools/kernel_avg.py eval $T2 $O/kernel_avg_us.json > /dev/null
rm -rf /tmp/p3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/bench.py --workload retrieval --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_retrieval_under_rocprof.json 2>/dev/null
T3=$(find /tmp/p3 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_retrieval_bf16.csv
python3 $R/tools/kernel_avg.py retrieval $T3 $O/kernel_avg_us.json > /dev/null
for leg in train eval retrieval; do
  case $leg in
    train) A="--workload train --steps 2 --warmup 1 --no-cpu-baseline --launch eager";;
    eval) A="--workload forward --steps 2 --warmup 1 --no-cpu-baseline --launch eager --in-flight 1";;
    retrieval) A="--workload retrieval --steps 1 --warmup 1 --no-cpu-baseline";;
  esac
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py $A > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) $O/pmc_$leg.json > /dev/null
done
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
# north-star micro-benchmark: time table, rocprofv3 --stats, HBM traffic of the wide-attention launches
python3 $R/tools/xpool_qk_bench.py > $O/xpool_qk_microbench.txt 2>&1
rm -rf /tmp/q1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q1 -- python3 $R/tools/xpool_qk_bench.py > /dev/null 2>&1
cp $(find /tmp/q1 -name "*kernel_stats.csv" | head -1) $O/xpool_qk_kernel_stats.csv
rm -rf /tmp/qf /tmp/qw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/qf -- python3 $R/tools/xpool_qk_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/qw -- python3 $R/tools/xpool_qk_bench.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/qf -name "*counter_collection.csv" | head -1) $(find /tmp/qw -name "*counter_collection.csv" | head -1) $O/xpool_qk_pmc.json > /dev/null
ls -la $O
# round 4 additions: clocks / socket power under the retrieval launch (the 53 k x 4 k made_xpool_fused launch runs at the power limit), the
# SQ counters of the attention kernels and of the retrieval kernels (MFMA busy, VALU per MFMA, LDS conflicts)
bash $R/tools/clock_probe.sh > $O/retrieval_clock_power.txt 2>&1
bash $R/tools/pmc_sq_round4.sh $TAG > $O/pmc_sq.log 2>&1 || true
ls -la $O
