import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    if "attention_kernel" in r["Kernel_Name"] and "wide" not in r["Kernel_Name"]:
        d[int(r["Grid_Size_X"]) // 256].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v2 = sorted(v)
    print(k, len(v), "min %.1f med %.1f max %.1f" % (v2[0], v2[len(v2) // 2], v2[-1]), ["%.0f" % x for x in v[:24]])
