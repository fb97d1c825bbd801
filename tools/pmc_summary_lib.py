"""Kernel symbol -> KernelTimer kind of mgsv_amd/ops.py (shared by pmc_summary.py and kernel_avg.py)."""


def kind(name):
    """-> (KernelTimer kind of mgsv_amd/ops.py / ops_train.py, counts_as_launch).  A made_* entry point that launches several kernels
    (made_attention_bwd: delta + dq + dkv; made_gemm_tn: either of its two kernels) sums their bytes; one of them counts the launches."""
    if "linear_big_kernel" in name:
        return ("linear_big_kernel<256>" if ("ILi256E" in name or "<256" in name) else "linear_big_kernel<128>"), True
    if "xpool_attn_kernel" in name:
        return "xpool_attention", True
    if "xpool_sims32_kernel" in name or "xpool_sims_kernel" in name:
        return "xpool_sims", True
    if "linear_glds_kernelILi1" in name or "linear_glds_kernel<1" in name:
        return ("linear_glds_kernel<1,.,64>" if ("ELi64E" in name or ", 64>" in name) else "linear_glds_kernel<1,.,128>"), True
    if "linear_glds_kernel" in name:
        return "linear_glds_kernel<3,.,128>", True
    if "linear_skinny_kernel" in name:
        return "linear_skinny_kernel", True
    if "linear_t16_kernel" in name:
        return "linear_t16_kernel", True
    if "linear_tiny_kernel" in name:
        return "linear_tiny_kernel", True
    if "linear_kernelIDF16bDF16b" in name:
        return "linear_kernel<bf16,bf16>", True
    if "linear_kernelIfDF16b" in name:
        return "linear_f32in_bf16", True
    if "linear_kernelIff" in name or "linear_kernel<float, float>" in name:
        return "linear_f32", True
    if "attn_bwd_dkv" in name:
        return "made_attention_bwd", True
    if "attn_bwd_dq" in name or "attn_delta" in name:
        return "made_attention_bwd", False
    if "gemm_tn" in name:
        return "made_gemm_tn", True
    if "attention_wide_kernel" in name:
        return "attention_wide_bf16", True
    if "attention_wide_combine" in name:
        return "attention_wide_bf16", False
    if "attention_kernel" in name:
        return "attention_bf16", True
    for k in ("dec_stage", "layernorm_bwd", "layernorm_add", "layernorm_kernel", "splitk_finish", "masked_mean", "xpool_tail", "xpool_fused"):
        if k in name:
            return k, True
    return None, False
