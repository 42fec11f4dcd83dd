#!/usr/bin/env python3
"""GPU dev tool: per-kernel time of ONE steady-state step out of a rocprofv3 --kernel-trace CSV: the window between the last two
launches of a marker kernel (default: iaf_adamax_ema_kernel, one per training step).
usage: python tools/step_breakdown.py <kernel_trace.csv> [marker substring] [--json out.json]
--json: the top launches and the kernel families (conv = forward-shaped convs incl. data gradients and the one-launch step, wgrad =
weight-gradient kernels + their packs and reduces, rest) for bench.py's training rooflines (profiles/train_kernels_<mode>.json)."""
import csv, sys, collections, json
argv = [a for a in sys.argv[1:]]
jout = None
if "--json" in argv:
    i = argv.index("--json"); jout = argv[i + 1]; del argv[i:i + 2]
rows = list(csv.DictReader(open(argv[0])))
marker = argv[1] if len(argv) > 1 else "iaf_adamax_ema_kernel"
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

if jout:
    def family(nm):
        if "wgrad" in nm or "pack_pixmajor" in nm:
            return "wgrad"
        if "iaf_conv_bf3_kernel" in nm or "iaf_conv_kernel" in nm or "iaf_step_fused_kernel" in nm:
            return "conv"
        return "rest"
    fam = collections.defaultdict(lambda: [0, 0.0])
    for nm, (n, us) in agg.items():
        fam[family(nm)][0] += n; fam[family(nm)][1] += us
    top = [{"kernel": nm.replace("void ", "")[:100], "launches_per_step": n, "avg_us": us / n, "us_per_step": us, "pct_of_step_kernel_time": 100 * us / busy}
           for nm, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]]
    json.dump({"source": "rocprofv3 --kernel-trace, one steady-state step (window between two %s launches): %s" % (marker, argv[0].split("/")[-1]),
               "step_window_us": (t1 - t0) / 1e3, "launches": len(win), "kernel_time_us": busy, "top": top,
               "families": [{"family": f, "launches_per_step": n, "us_per_step": us} for f, (n, us) in sorted(fam.items())]},
              open(jout, "w"), indent=1)
