#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) as the
markdown table kept under profiles/.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of
wide coalesced reads); both counters are in KiB, the table is in MB (1e6 bytes).  The optional header becomes the table's
first line: bench.py looks there for "config: <dtype> B=.. T=.. L=.. R=.. S=.. wt=.." to decide whether a tracked table
covers the configuration it is running.  Usage: hbm_traffic.py <fetch.csv> <write.csv> <steps> <out.md> [header]"""
import collections
import csv
import sys


def load(path, name):
    val, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        val[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return val, dur


def main(fetch, write, steps, out, header=None):
    fv, fd = load(fetch, "FETCH_SIZE")
    wv, _ = load(write, "WRITE_SIZE")
    rows = []
    for k in fv:
        n = len(fv[k])
        rd = 2.0 * sum(fv[k]) / n * 1.024e-3    # KiB -> MB, x2 correction
        wr = sum(wv.get(k, [0.0])) / max(len(wv.get(k, [0.0])), 1) * 1.024e-3
        us = sum(fd[k]) / n
        rows.append((n * (rd + wr), k, n, rd, wr, us))
    rows.sort(reverse=True)
    tot_r = sum(r[2] * r[3] for r in rows) / steps / 1e3
    tot_w = sum(r[2] * r[4] for r in rows) / steps / 1e3
    with open(out, "w") as f:
        if header:
            f.write(header.strip() + "\n\n")
        f.write("| kernel | calls | read MB/launch | write MB/launch | us/launch (profiled) | GB/s |\n|---|---|---|---|---|---|\n")
        for _, k, n, rd, wr, us in rows:
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.0f |\n" % (k[:90], n, rd, wr, us, (rd + wr) / us * 1e3))
        f.write("\nper training step: %.2f GB read + %.2f GB written = %.2f GB\n" % (tot_r, tot_w, tot_r + tot_w))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
