# rocprofv3 --kernel-trace of the training leg alone -> gpurun_out/<tag>/: train_trace_summary.txt, train_step_timeline_tape_under_rocprof.txt
TAG=${1:-r04x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2>/dev/null
T1=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_bf16.csv; python3 $R/tools/trace_summary.py $T1 adam_update_kernel 90 > $O/train_trace_summary.txt
python3 $R/tools/trace_timeline.py $T1 ${MARK:-adam_update_kernel} -7 > $O/train_step_timeline_tape_under_rocprof.txt
