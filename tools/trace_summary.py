"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) launches, average duration and time per step.
usage: python tools/trace_summary.py <kernel_trace.csv> <steps | marker-kernel-substring> [top]
(a marker = a kernel launched exactly once per step, e.g. adam_update_kernel / sine_pe_kernel: the step count is its launch count)"""
import collections, csv, re, sys

def short(name):
    m = re.search(r"(\w+_kernel|copyBuffer|elementwise_kernel|fillBuffer)", name)
    base = m.group(1) if m else name[:40]
    t = re.search(r"_kernelI([^E]*)E", name)
    if "<" in name:
        base += name[name.index("<"):name.index(">") + 1][:24]
    elif t:
        base += "<" + t.group(1) + ">"
    return base

rows = list(csv.DictReader(open(sys.argv[1])))
try:
    steps = float(sys.argv[2])
except ValueError:
    steps = float(sum(1 for r in rows if sys.argv[2] in r["Kernel_Name"]))
    print(f"steps in the trace ({sys.argv[2]} launches): {steps:.0f}")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
byname = collections.defaultdict(float)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Workgroup_Size_X"]) or 1
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[k][0] += 1; agg[k][1] += d
    byname[k[0]] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time per step: {tot / steps / 1e3:.3f} ms")
for n, d in sorted(byname.items(), key=lambda kv: -kv[1])[:20]:
    print(f"  {n:50s} {d / steps / 1e3:8.3f} ms")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:46s} blocks=({k[1]},{k[2]},{k[3]}) n/step={v[0] / steps:6.1f} avg={v[1] / v[0]:8.1f}us  {v[1] / steps / 1e3:7.3f} ms/step")
