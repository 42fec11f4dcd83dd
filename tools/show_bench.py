#!/usr/bin/env python3
"""print the interesting fields of a bench.py JSON line:  python tools/show_bench.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("%.4f ms/step  %.0f %s  dominant %.2f us frac %.3f  step frac %s" % (d["ms_per_step"], d["value"], d["unit"], r.get("avg_launch_us", 0), r.get("frac", 0),
                                                                          r.get("step", {}).get("frac")))
print(d["config"].get("kernels_chosen"))
for k in r.get("kernels", []):
    if "latent" in k:
        print("  ", k["latent"], k["layer"][:70], k["kernel"], "%.2f us %.1f TF" % (k["us"], k["live_tflops"]))
    else:               # (training lines: the top launches of the committed kernel trace)
        print("  ", k.get("kernel", "")[:90], "x%s" % k.get("launches_per_step"), "%.1f us" % k.get("avg_us", 0))
for x in r.get("extended_unit", []):
    print("   posterior block", x["latent"], "%.1f us" % x["us"])
