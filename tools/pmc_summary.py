"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), grouped by the kernel kinds bench.py times.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB  (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, sys


import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary_lib import kind  # noqa: E402


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
