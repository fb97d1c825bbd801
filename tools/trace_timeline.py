"""Timeline of ONE step out of a rocprofv3 kernel_trace.csv: start offset, duration, queue, kernel, grid -- shows what overlaps what
and where the GPU idles.  usage: python tools/trace_timeline.py <kernel_trace.csv> <marker kernel substring> <which occurrence (negative: from the end)>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker, occ = sys.argv[2], int(sys.argv[3])
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if occ < 0:
    occ += len(idx) - 1                     # negative: counted from the last complete step
a, b = idx[occ], idx[occ + 1]


def short(name):
    m = re.search(r"(\w+_kernel|copyBuffer|fillBuffer)", name)
    base = m.group(1) if m else name[:40]
    if "<" in name:
        base += name[name.index("<"):name.index(">") + 1][:20]
    return base


t0 = int(rows[a]["Start_Timestamp"])
queues = {}
busy_end = t0
idle = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r["Queue_Id"], len(queues))
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    busy_end = max(busy_end, e)
    wg = int(r["Workgroup_Size_X"]) or 1
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {'gap %.1f' % gap if gap > 1 else '':9s} {short(r['Kernel_Name']):44s} "
          f"({int(r['Grid_Size_X']) // wg},{r['Grid_Size_Y']},{r['Grid_Size_Z']})")
print(f"step span {(busy_end - t0) / 1e3:.1f} us, GPU idle (no kernel running) {idle:.1f} us")
