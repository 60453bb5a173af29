#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) into the markdown summary kept under profiles/."""
import csv
import sys


def main(path, steps, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("| kernel | calls | avg us | total ms | % | per-step ms |\n|---|---|---|---|---|---|\n")
        for r in rows:
            t = float(r["TotalDurationNs"])
            if t / tot < 0.0005:
                continue
            f.write("| `%s` | %s | %.1f | %.2f | %.1f | %.3f |\n" % (
                r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, t / 1e6, 100 * t / tot, t / 1e6 / steps))
        f.write("\ntotal kernel time %.2f ms over %d steps = %.3f ms/step\n" % (tot / 1e6, steps, tot / 1e6 / steps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
