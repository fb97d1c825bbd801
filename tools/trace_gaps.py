"""Where the GPU runs NO kernel inside a step, over every step of a rocprofv3 kernel_trace.csv: (kernel before, kernel after) -> how often a hole of
more than 5 us sits there and how long it is.  Tells a systematic hole (a dependency, the host) from one step's hiccup.
usage: python tools/trace_gaps.py <kernel_trace.csv> <marker kernel substring> [longest step span in us to count: leaves the eager steps out]"""
import collections, csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
max_span = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]


def short(r):
    name = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel|copyBuffer|fillBuffer)", name)
    base = m.group(1) if m else name[:40]
    wg = int(r["Workgroup_Size_X"]) or 1
    return f"{base}({int(r['Grid_Size_X']) // wg},{r['Grid_Size_Y']},{r['Grid_Size_Z']})"


holes = collections.defaultdict(list)
idle_per_step, spans = [], []
for a, b in zip(idx[2:-1], idx[3:]):                       # (the first steps are warm-up)
    if (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 > max_span:
        continue
    busy_end = int(rows[a]["Start_Timestamp"])
    prev = None
    idle = 0.0
    for k, r in enumerate(rows[a:b]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - busy_end) / 1e3
        if gap > 0:
            idle += gap
        if gap > 3 and prev is not None:
            holes[(k, short(prev), short(r))].append(gap)
        if e > busy_end:
            busy_end, prev = e, r
    idle_per_step.append(idle)
    spans.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
n = len(idle_per_step)
print(f"{n} steps; step span median {statistics.median(spans):.1f} us; GPU idle per step: median {statistics.median(idle_per_step):.1f} us, min {min(idle_per_step):.1f}, max {max(idle_per_step):.1f}")
print("holes > 3 us by position (launch number in the step, kernel that ended last, kernel that starts): in how many steps, median, max")
for (k, p, q), v in sorted(holes.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  #{k:3d} {p:42s} -> {q:42s} {len(v):3d}/{n}  median {statistics.median(v):6.1f}  max {max(v):6.1f}  total/step {sum(v) / n:6.1f}")
