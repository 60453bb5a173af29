#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue, short kernel name,
plus the time during which 0 / 1 / 2+ kernels were running.  Usage: timeline.py <kernel_trace.csv> [step-from-end]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", n)
    return (m.group(1) if m else n)[:28] + (" " + re.sub(r"[^0-9a-z, ]", "", n[n.find("<"):n.find(">")])[:24] if "<" in n else "")


def main(path, back=1):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at the first-layer input convolution kernel
    starts = [i for i, r in enumerate(rows) if "causal_conv" in r["Kernel_Name"] and "wgrad" not in r["Kernel_Name"]]
    a, b = starts[-1 - back], starts[-back]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    ev = []
    for r in step:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%9.1f %8.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
        ev += [(s, 1), (e, -1)]
    ev.sort()
    busy = {0: 0, 1: 0, 2: 0}
    cur, last = 0, 0
    for t, d in ev:
        busy[min(cur, 2)] += t - last
        cur += d
        last = t
    tot = int(rows[b]["Start_Timestamp"]) - t0
    print("step %.1f us: idle %.1f, one kernel %.1f, two or more %.1f, (gap to next step %.1f)" % (
        tot / 1e3, busy[0] / 1e3, busy[1] / 1e3, busy[2] / 1e3, (tot - last) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
