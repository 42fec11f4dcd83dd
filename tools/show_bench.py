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
    print("  ", k["latent"], k["layer"][:70], k["kernel"], "%.2f us %.1f TF" % (k["us"], k["live_tflops"]))
for x in r.get("extended_unit", []):
    print("   posterior block", x["latent"], "%.1f us" % x["us"])
