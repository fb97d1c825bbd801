# SQ counters of the dominant made_linear kernel on an encoder-sized problem (tools/linear_tiles_bench.py), one --pmc pass per group
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lin; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pl$i -- python3 $R/tools/linear_tiles_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_kernel.py /tmp/pl$i "linear_glds_kernel" >> $O/sq_counters_linear_glds.txt 2>&1
done
cat $O/sq_counters_linear_glds.txt
