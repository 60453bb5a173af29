#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (…_results.db from `rocprofv3 --kernel-trace --stats`) into the markdown
table kept under profiles/.  usage: rocpd_summary.py results.db steps out.md [--list substring]
--list prints every dispatch of the kernels whose name contains the substring, in launch order (us)."""
import sqlite3
import sys


def tables(cur, prefix):
    return [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like '%s%%'" % prefix)][0]


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kd, ks = tables(cur, "rocpd_kernel_dispatch"), tables(cur, "rocpd_info_kernel_symbol")
    q = ("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, "
         "s.arch_vgpr_count, s.accum_vgpr_count from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks))
    return list(cur.execute(q))


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = load(path)
    if "--list" in sys.argv:
        sub = sys.argv[sys.argv.index("--list") + 1]
        for n, s, e, gx, wx, lds, v, a in rows:
            if sub in n:
                print("%-60s %8.1f us  grid %d x %d lds %d vgpr %d+%d" % (n[:60], (e - s) / 1e3, gx // max(wx, 1), wx, lds, v, a))
        return
    agg = {}
    for n, s, e, *_ in rows:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        f.write("| kernel | calls | avg us | total ms | % | per-step ms |\n|---|---|---|---|---|---|\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if t / tot < 0.0005:
                continue
            f.write("| `%s` | %d | %.1f | %.2f | %.1f | %.3f |\n" % (n[:110], c, t / c / 1e3, t / 1e6, 100 * t / tot, t / 1e6 / steps))
        f.write("\ntotal kernel time %.2f ms over %d steps = %.3f ms/step\n" % (tot / 1e6, steps, tot / 1e6 / steps))


if __name__ == "__main__":
    main()
