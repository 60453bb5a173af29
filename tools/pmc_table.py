#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (…_counter_collection.csv, one per pass) -> one markdown table: rows = kernels matching the
filters, columns = counters averaged per launch.  usage: pmc_table.py out.md filter[,filter...] csv [csv...]"""
import collections
import csv
import sys

out, filt, paths = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in paths:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if not any(f in k for f in filt):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for d in agg.values() for c in d})
with open(out, "w") as f:
    f.write("| kernel | us (profiled) | " + " | ".join(names) + " |\n|---|---|" + "---|" * len(names) + "\n")
    for k, d in agg.items():
        f.write("| `%s` | %.1f | " % (k[:70], sum(dur[k]) / len(dur[k])) +
                " | ".join("%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in names) + " |\n")
print(open(out).read())
