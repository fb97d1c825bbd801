"""rocprofv3 --pmc passes (one directory per counter group) -> one table per kernel: every counter's per-dispatch average, and the derived
MFMA / VALU / LDS utilisation.  usage: pmc_sq_table.py <kernel substring>[,<substring>...] <dir> [<dir> ...]

Derived (MI355X_MICROARCH.md, rocprofv3 PMC section): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the dispatch lasted GRBM_GUI_ACTIVE / 8
shader cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_32x32x16_bf16), so
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave."""
import collections, csv, glob, sys
subs = sys.argv[1].split(",")
tot = {s: collections.Counter() for s in subs}
cnt = {s: collections.Counter() for s in subs}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for s in subs:
                if s in r["Kernel_Name"]:
                    tot[s][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[s][r["Counter_Name"]] += 1
for s in subs:
    c = {k: tot[s][k] / cnt[s][k] for k in tot[s]}
    print(f"== {s}: per-dispatch averages ({max(cnt[s].values()) if cnt[s] else 0} dispatches per pass)")
    for k in sorted(c):
        print(f"   {k:28s} {c[k]:18.0f}")
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc:
        simd_cycles = cyc * 256 * 4
        print(f"   -> duration {cyc:.0f} shader cycles (GRBM_GUI_ACTIVE / 8)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: print(f"   -> MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles * 100:.1f} % of the chip's SIMD cycles")
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
            if k in c: print(f"   -> {k} / SQ_WAVE_CYCLES = {c[k] / w * 100:.1f} %")
    if c.get("SQ_INSTS_MFMA"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
            if k in c: print(f"   -> {k} per MFMA = {c[k] / c['SQ_INSTS_MFMA']:.2f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   -> LDS bank-conflict cycles / LDS active cycles = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'] * 100:.1f} %")
