R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sk; rocprofv3 --kernel-trace --output-format csv -d /tmp/sk -- python3 $R/tools/small_kernels_bench.py > /dev/null 2>&1
python3 $R/tools/trace_summary.py $(find /tmp/sk -name "*kernel_trace.csv" | head -1) 6 30 | grep blocks | grep -v "elementwise\|arange"
