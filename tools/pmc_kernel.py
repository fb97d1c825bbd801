"""Sums rocprofv3 --pmc counter rows for kernels whose name contains a substring.  usage: pmc_kernel.py <dir> <substr>"""
import collections, csv, glob, sys
tot, n = collections.Counter(), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k] / n[k]:16.0f}  (avg of {n[k]} dispatches)")
