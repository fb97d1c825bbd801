# kernel-level timing of made_attention_wide at the north_star shape: streaming kernel vs general kernel, per key split
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for ns in 1 2 4; do
  rm -rf /tmp/wa; rocprofv3 --kernel-trace --output-format csv -d /tmp/wa -- python3 $R/tools/wide_only.py $ns > /dev/null 2>&1
  echo "streaming n_split=$ns"; python3 $R/tools/trace_summary.py $(find /tmp/wa -name "*kernel_trace.csv" | head -1) 6 8 | grep -i "wide"
done
export MADE_WIDE_GENERAL=1
for ns in 1 2 4; do
  rm -rf /tmp/wb; rocprofv3 --kernel-trace --output-format csv -d /tmp/wb -- python3 $R/tools/wide_only.py $ns > /dev/null 2>&1
  echo "general n_split=$ns"; python3 $R/tools/trace_summary.py $(find /tmp/wb -name "*kernel_trace.csv" | head -1) 6 8 | grep -i "wide"
done
