# timeline of one eval step (hipGraph replay and eager) under rocprofv3 --kernel-trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evaltl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/e1; rocprofv3 --kernel-trace --output-format csv -d /tmp/e1 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --launch eager > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $(find /tmp/e1 -name "*kernel_trace.csv" | head -1) sine_pe_kernel 5 > $O/eval_timeline_eager.txt 2>&1
rm -rf /tmp/e2; rocprofv3 --kernel-trace --output-format csv -d /tmp/e2 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $(find /tmp/e2 -name "*kernel_trace.csv" | head -1) sine_pe_kernel 5 > $O/eval_timeline_graph.txt 2>&1
