"""Per-kernel-kind average launch duration (us) of a rocprofv3 kernel_trace.csv, keyed by the kinds bench.py's KernelTimer reports
(the mapping of tools/pmc_summary.py): the numbers bench.py quotes as `avg_launch_us_rocprof_committed` from profiles/r03_kernel_avg_us.json.
usage: python tools/kernel_avg.py <leg> <kernel_trace.csv> <out.json>   (merges into out.json under the key <leg>)"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary_lib import kind

leg, path, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    k, counts = kind(r["Kernel_Name"])
    if k is None:
        continue
    tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n[k] += 1 if counts else 0
out = json.load(open(out_path)) if os.path.isfile(out_path) else {}
out[leg] = {k: round(tot[k] / max(n[k], 1), 2) for k in sorted(tot)}
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(out[leg], indent=1))
