"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), grouped by the kernel kinds bench.py times.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB  (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, sys


def kind(name):
    """-> (KernelTimer kind of mgsv_amd/ops.py / ops_train.py, counts_as_launch).  A made_* entry point that launches several kernels
    (made_attention_bwd: delta + dq + dkv; made_gemm_tn: either of its two kernels) sums their bytes; one of them counts the launches."""
    if "linear_ring_kernel" in name:
        return "linear_ring_kernel<128,128>", True
    if "linear_glds_kernelILi1" in name or "linear_glds_kernel<1" in name:
        return ("linear_glds_kernel<1,.,64>" if ("ELi64E" in name or ", 64>" in name) else "linear_glds_kernel<1,.,128>"), True
    if "linear_glds_kernel" in name:
        return "linear_glds_kernel<3,.,128>", True
    if "linear_skinny_kernel" in name:
        return "linear_skinny_kernel", True
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


def read(path, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k, counts = kind(r["Kernel_Name"])
        if k is None:
            continue
        tot[k] += float(r["Counter_Value"]); n[k] += 1 if counts else 0
    return tot, n


fetch, nf = read(sys.argv[1], "FETCH_SIZE")
write, nw = read(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch):
    launches = nf[k]
    b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0 / max(launches, 1)
    out[k] = {"launches_profiled": launches, "hbm_bytes_per_launch": int(b), "fetch_kib_total": fetch[k], "write_kib_total": write.get(k, 0.0)}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1)[:1500])
