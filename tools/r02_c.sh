R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --workload forward --steps 10 --warmup 3 --no-cpu-baseline --launch eager --in-flight 1 > $O/c_eval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/c_kernel_stats_eval_eager_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 16 60 > $O/c_eval_trace_summary.txt
head -90 $O/c_eval_trace_summary.txt
