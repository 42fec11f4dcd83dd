#!/usr/bin/env python3
"""GPU dev tool: per-kernel time of ONE steady-state step out of a rocprofv3 --kernel-trace CSV: the window between the last two
launches of a marker kernel (default: iaf_adamax_ema_kernel, one per training step).
usage: python tools/step_breakdown.py <kernel_trace.csv> [marker substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "iaf_adamax_ema_kernel"
key_s = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
key_e = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
rows.sort(key=lambda r: int(r[key_s]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("marker %r seen %d times" % (marker, len(marks)))
a, b = marks[-2], marks[-1]
win = rows[a + 1:b + 1]
t0, t1 = int(rows[a][key_e]), int(rows[b][key_e])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    nm = r["Kernel_Name"]
    nm = nm[:nm.index("(")] if "(" in nm else nm
    agg[nm][0] += 1
    agg[nm][1] += (int(r[key_e]) - int(r[key_s])) / 1e3
busy = sum(v[1] for v in agg.values())
print("step window %.1f us, %d launches, kernel time %.1f us (%.0f %% of the window)" % ((t1 - t0) / 1e3, len(win), busy, 100 * busy / ((t1 - t0) / 1e3)))
for nm, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us %5.1f %%  x%-4d avg %7.1f us  %s" % (us, 100 * us / busy, n, us / n, nm[:120]))
