# SQ counters (MFMA busy, VALU busy, waits, LDS bank conflicts, instruction mix) of the step's flash-attention kernels and of made_xpool_fused,
# one --pmc pass per counter group (rocprofv3 --pmc only: no trace domains beside it).  usage: bash tools/pmc_sq_round4.sh <tag>
export MADE_DEBUG_VARIANTS=1          # (measurement knobs are honoured only under this switch)
TAG=${1:-r04_a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE")
run_target() {   # name, kernel-substrings (space separated), command...
  local name=$1 subs=$2; shift 2
  local i=0
  : > $O/sq_counters_$name.txt
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1)); rm -rf /tmp/p4_$i
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/p4_$i -- "$@" > /tmp/p4_$i.log 2>&1
  done
  python3 $R/tools/pmc_sq_table.py "$subs" /tmp/p4_[0-9]* > $O/sq_counters_$name.txt 2>&1
  grep -v amdgpu.ids /tmp/p4_1.log | tail -2 >> $O/sq_counters_$name.txt
  rm -rf /tmp/p4_[0-9]*
}
run_target attention "attention_kernel,attn_bwd_fused_kernel,attn_bwd_dq_kernel,attn_bwd_dkv_kernel" python3 $R/tools/attn_pmc_target.py
run_target xpool_fused "xpool_fused_persist_kernel" python3 $R/tools/xpool_only.py 8192 512
# the training step's dominant family (encoder-sized Linears, tools/linear_tiles_bench.py: plain / residual / gathered rows) and the opt-in retrieval kernel
MADE_LINEAR_TILE=64 run_target linear_glds "linear_glds_kernel" python3 $R/tools/linear_tiles_bench.py
run_target xpool_sims "xpool_sims32_kernel" python3 $R/tools/xpool_sims_bench.py 8192 512 96
cat $O/sq_counters_attention.txt $O/sq_counters_xpool_fused.txt $O/sq_counters_linear_glds.txt $O/sq_counters_xpool_sims.txt
