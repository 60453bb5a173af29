#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv per kernel (name filter optional)."""
import collections, csv, sys
path = sys.argv[1]; filt = sys.argv[2:] 
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if filt and not any(f in k for f in filt): continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k[:100], "| n=%d avg_us=%.1f" % (len(dur[k]) // max(len(d), 1), sum(dur[k]) / len(dur[k])))
    for c, v in sorted(d.items()):
        print("   %-36s %14.1f" % (c, sum(v) / len(v)))
