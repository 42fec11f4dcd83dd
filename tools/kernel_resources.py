#!/usr/bin/env python3
"""CPU dev tool: registers / scratch / LDS of every kernel of the build (hipcc -Rpass-analysis=kernel-resource-usage over the
translation units of iaf_amd/build.py, device code only).  A kernel with ScratchSize > 0 has spilled or -- worse -- keeps whole
register arrays in memory because a lambda of its K loop was not inlined (round 4: iaf_conv_bf3_kernel<4|5, PPW=4, ...> ran with
1.4 KB of scratch per lane until its lambdas were marked always_inline).  tests/test_build_resources.py keeps the hot kernels at 0.
usage: python tools/kernel_resources.py [substring of the object name ...]"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iaf_amd import build as B


def unit_resources(src, obj, flags):
    cmd = ["/opt/rocm/bin/hipcc"] + B.CFLAGS + flags + ["--cuda-device-only", "-c", src, "-o", os.devnull,
                                                        "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = dict(unit=os.path.basename(obj), name=m.group(1))
            rows.append(cur)
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def all_resources(filters=()):
    units = [u for u in B._units() if u[0].endswith(".hip") and (not filters or any(f in u[1] for f in filters))]
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda u: unit_resources(*u), units))
    return [r for rows in res for r in rows]


if __name__ == "__main__":
    rows = all_resources(sys.argv[1:])
    try:
        import subprocess as sp
        dem = sp.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(r["name"] for r in rows), stdout=sp.PIPE,
                     universal_newlines=True).stdout.splitlines()
    except Exception:
        dem = [r["name"] for r in rows]
    bad = 0
    for r, d in zip(rows, dem):
        flag = "  <-- SCRATCH" if r.get("scratch", 0) else ""
        bad += 1 if flag else 0
        print("%-22s vgpr %3d agpr %3d scratch %5d spill v%d s%d occ %d  %s%s" % (
            r["unit"], r.get("vgprs", -1), r.get("agprs", 0), r.get("scratch", 0), r.get("vspill", 0), r.get("sspill", 0),
            r.get("occ", 0), d[:110], flag))
    print("%d kernels, %d with scratch" % (len(rows), bad))
