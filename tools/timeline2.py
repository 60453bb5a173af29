#!/usr/bin/env python3
"""Step timeline from a rocprofv3 rocpd database of `bench.py` (graph replay): per step, wall time, the union of
kernel-busy time, the sum of kernel durations and the largest gaps.  usage: timeline2.py results.db [first_kernel_substr]"""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from rocpd_summary import load

rows = load(sys.argv[1])
# the step's first kernel: the input conv of its own launch (rounds 1-3), or -- fused into the first forward group kernel
# (round 4) -- whatever follows the re-pack that ends the previous step
if len(sys.argv) > 2:
    starts = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
else:
    starts = [i for i, r in enumerate(rows) if "causal_conv_cin1" in r[0]]
    if len(starts) < 3:
        starts = [i + 1 for i, r in enumerate(rows) if "pack_gather_kernel" in r[0] and i + 1 < len(rows)]
# the last complete step
if len(starts) < 3:
    raise SystemExit("not enough steps")
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0, t1 = step[0][1], max(r[2] for r in step)
print("kernels in step: %d   wall %.1f us   sum of durations %.1f us" % (len(step), (t1 - t0) / 1e3, sum(r[2] - r[1] for r in step) / 1e3))
ev = sorted((r[1], r[2]) for r in step)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy %.1f us   idle gaps total %.1f us (%d gaps)" % (busy / 1e3, sum(g for g, _ in gaps) / 1e3, len(gaps)))
for g, at in sorted(gaps, reverse=True)[:8]:
    print("   gap %.1f us at +%.1f us" % (g / 1e3, at / 1e3))
# concurrency: time with >= 2 kernels running
pts = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
n, last, over = 0, pts[0][0], 0
for t, d in pts:
    if n >= 2:
        over += t - last
    n += d
    last = t
print("time with two or more kernels in flight: %.1f us" % (over / 1e3))
for r in step:
    print("  %8.1f %8.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:70]))
