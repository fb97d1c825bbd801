R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for nb in 512 256 128 64; do
export MADE_LNBWD_NB=$nb
rm -rf /tmp/ln; rocprofv3 --kernel-trace --output-format csv -d /tmp/ln -- python3 $R/tools/lnbwd_bench.py > /dev/null 2>&1
echo "nb=$nb"; python3 $R/tools/trace_summary.py $(find /tmp/ln -name "*kernel_trace.csv" | head -1) 8 30 | grep "layernorm_bwd" | grep blocks
done
