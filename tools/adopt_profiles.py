#!/usr/bin/env python3
"""Adopt what `gpurun -- 'ROUND=rNN bash tools/refresh_profiles.sh'` measured:  python tools/adopt_profiles.py rNN
copies gpurun_out/rNN/* into profiles/rNN/ (tracked) and the two JSON files bench.py reads -- roofline.rocprof and roofline.traffic --
from that SAME directory to profiles/ (the top-level copy once came from an earlier refresh than the CSV it cites).
tests/test_bench_contract.py::test_profile_json_matches_the_csv_it_cites holds the result."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", rnd), os.path.join(ROOT, "profiles", rnd)
if not os.path.isdir(src):
    sys.exit("no " + src)
for dp, dn, fn in os.walk(src):
    rel = os.path.relpath(dp, src)
    os.makedirs(os.path.join(dst, rel), exist_ok=True)
    for f in fn:
        shutil.copy2(os.path.join(dp, f), os.path.join(dst, rel, f))
for f in ("rocprof_dominant_kernel.json", "pmc_dominant_kernel.json", "train_kernels_model.json", "train_kernels_layers.json"):
    if os.path.exists(os.path.join(dst, f)):
        shutil.copy2(os.path.join(dst, f), os.path.join(ROOT, "profiles", f))
        print("profiles/%s <- profiles/%s/%s" % (f, rnd, f))
