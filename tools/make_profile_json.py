#!/usr/bin/env python3
"""Write the two JSON files bench.py reads (roofline.rocprof, roofline.traffic) from what tools/refresh_profiles.sh
measured:  python tools/make_profile_json.py gpurun_out/r03 r03
  <dir>/rocprof_dominant_kernel.json  <- <dir>/bench_kernel_stats.csv   (rocprofv3 --kernel-trace --stats of bench.py)
  <dir>/pmc_dominant_kernel.json      <- <dir>/pmc/step_*_counter_collection.csv (separate --pmc passes over tools/run_step.py)
Copy both to profiles/ afterwards (tracked).  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports
half the bytes of wide coalesced reads); WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import os
import sys

d, rnd = sys.argv[1], sys.argv[2]
prec = os.environ.get("PMC_PREC", "f16x2")
KERNEL = "iaf_step_fused_kernel<10, 2, 2, 16, 2, 0"       # (TF statement; the halo-exchange form carries one more template argument)


def stats_row(path, needle):
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return r
    return None


out = {}
ks = os.path.join(d, "bench_kernel_stats.csv")
if os.path.exists(ks):
    r = stats_row(ks, KERNEL)
    if r:
        out = {"kernel": r["Name"], "calls": int(r["Calls"]), "avg_launch_us": float(r["AverageNs"]) / 1e3,
               "source": "profiles/%s/bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline "
                         "(tools/refresh_profiles.sh, written by tools/make_profile_json.py)" % rnd}
        json.dump(out, open(os.path.join(d, "rocprof_dominant_kernel.json"), "w"), indent=1)
        print("rocprof:", out)

acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "pmc", "step_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "iaf_step_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    fetch, write = m.get("FETCH_SIZE"), m.get("WRITE_SIZE")
    alg = 4.0 * (3 * 32 + 160) * 32 * 256 + 4.0 * (9 * (32 * 160 + 160 * 160 + 2 * 160 * 32) + 2 * (160 + 160 + 64))
    pj = {"kernel": "iaf_step_fused_kernel<NHT=10,NZT=2,DEPTH=2,W=16,R=2,VAR=0 (TF statement),XCH=1 (halo rows exchanged)> (one IAF step: masked convs "
                    "32->160->160->64 + affine/log-det, B=32, 16x16)",
          "command": "rocprofv3 --kernel-trace --pmc <counter set> (separate passes) -- python tools/run_step.py --hw 16 --reps 10 "
                     "--precision %s (tools/refresh_profiles.sh; raw CSVs in profiles/%s/pmc/step_*)" % (prec, rnd),
          "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
          "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM "
                        "section); WRITE_SIZE as reported",
          "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None,
          "algorithmic_bytes_per_launch": alg,
          "sq": {k: v for k, v in m.items() if k.startswith("SQ_")}, "waves_per_launch": 1024}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_WAVE_CYCLES" in m:
        pj["mfma_busy_fraction_of_wave_time"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"])
        pj["mfma_per_wave"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / 16.0
    json.dump(pj, open(os.path.join(d, "pmc_dominant_kernel.json"), "w"), indent=1)
    print("pmc:", {k: pj[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "mfma_busy_fraction_of_wave_time") if k in pj})
