"""Build the HIP engine in-tree for gfx950:  python -m iaf_amd.build [--force]
Output: iaf_amd/_lib/libiaf_hip.so (git-ignored; travels to the GPU box with the snapshot).

The masked-conv kernel is instantiated once per launch shape (pxt, wco, ks) in its own translation
unit (csrc/iaf_conv_inst.hip with -DIAF_PXT/-DIAF_WCO/-DIAF_KS); the units compile in parallel."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# IAF_BUILD_TAG=<tag> (with IAF_EXTRA_CFLAGS) builds an experiment library next to the product one, in _lib_<tag>/;
# load it with IAF_HIP_LIB=<path> (iaf_amd/_capi.py).  The product library is always _lib/libiaf_hip.so.
_TAG = os.environ.get("IAF_BUILD_TAG", "")
LIBDIR = os.path.join(HERE, "_lib" + ("_" + _TAG if _TAG else ""))
OBJDIR = os.path.join(LIBDIR, "obj")
OUT = os.path.join(LIBDIR, "libiaf_hip.so")
SHAPES = [(4, 1, 1), (4, 1, 2), (2, 2, 1), (2, 2, 2), (2, 1, 2), (2, 1, 4), (1, 1, 4), (1, 2, 2)]   # keep in sync with pick_kernel()
BF3_PLAIN_SHAPES = [(2, 1, 4, 1), (4, 1, 4, 1), (2, 1, 4, 2), (2, 1, 4, 3)]   # 9-tap plain convs: keep in sync with pick_bf3_plain()
BF3_SHAPES = [(4, 1, 4, 1), (2, 1, 4, 1), (1, 1, 4, 1), (1, 4, 1, 1), (2, 1, 4, 2), (1, 1, 4, 2)]   # (ppw, pxt, ks, wco): keep in sync with pick_bf3()
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
CFLAGS += os.environ.get("IAF_EXTRA_CFLAGS", "").split()       # dev experiments only (e.g. -DIAF_EXP_NOREFILL)
HEADERS = [os.path.join(ROOT, "include", "iaf_hip.h")] + sorted(
    os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))   # fallback: no dependency file yet -> every header counts


def _units():
    units = [(os.path.join(CSRC, "iaf_engine.hip"), os.path.join(OBJDIR, "iaf_engine.o"), []),
             # host-only: the RCCL gradient exchange (librccl is dlopen'ed at run time, no link dependency)
             (os.path.join(CSRC, "iaf_comm.cpp"), os.path.join(OBJDIR, "iaf_comm.o"), ["-x", "hip"]),
             # the weight gradient on the bf16 matrix cores (transposing LDS reads)
             (os.path.join(CSRC, "iaf_wgrad_bf3.hip"), os.path.join(OBJDIR, "iaf_wgrad_bf3.o"), [])]
    for pxt, wco, ks in SHAPES:
        units.append((os.path.join(CSRC, "iaf_conv_inst.hip"), os.path.join(OBJDIR, "iaf_conv_%d_%d_%d.o" % (pxt, wco, ks)),
                      ["-DIAF_PXT=%d" % pxt, "-DIAF_WCO=%d" % wco, "-DIAF_KS=%d" % ks]))
    for ppw, pxt, ks, wco in BF3_SHAPES:
        units.append((os.path.join(CSRC, "iaf_conv_bf3_inst.hip"), os.path.join(OBJDIR, "iaf_bf3_%d_%d_%d_%d.o" % (ppw, pxt, ks, wco)),
                      ["-DIAF_PPW=%d" % ppw, "-DIAF_PXT=%d" % pxt, "-DIAF_KS=%d" % ks, "-DIAF_WCO=%d" % wco]))
    for ppw, pxt, ks, wco in BF3_PLAIN_SHAPES:
        units.append((os.path.join(CSRC, "iaf_conv_bf3_plain_inst.hip"), os.path.join(OBJDIR, "iaf_bf3p_%d_%d_%d_%d.o" % (ppw, pxt, ks, wco)),
                      ["-DIAF_PPW=%d" % ppw, "-DIAF_PXT=%d" % pxt, "-DIAF_KS=%d" % ks, "-DIAF_WCO=%d" % wco]))
    # accumulators in architectural VGPRs: left to itself the register allocator puts them in AGPRs and rotates them through
    # VGPR copies inside the K loop (48 v_accvgpr moves per 162 MFMAs)
    for part in (0, 1, 2, 3, 4, 5, 6):
        units.append((os.path.join(CSRC, "iaf_step_fused_inst.hip"), os.path.join(OBJDIR, "iaf_step_fused_%d.o" % part),
                      ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DIAF_FUSED_PART=%d" % part]))
    return units


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _deps(obj, src):
    """headers the unit actually includes (hipcc -MD wrote obj + '.d' at the last compile), else every project header"""
    d = obj + ".d"
    if not os.path.exists(d):
        return [src] + HEADERS
    words = open(d).read().replace("\\\n", " ").split()
    mine = [w for w in words[1:] if w.startswith(ROOT) or not os.path.isabs(w)]      # project files only (not /opt/rocm)
    return [src] + mine


def build(force=False, verbose=True, jobs=None):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJDIR, exist_ok=True)
    todo = []
    for src, obj, defs in _units():
        if force or _stale(obj, _deps(obj, src)):
            todo.append([hipcc] + CFLAGS + defs + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])
    if todo:
        jobs = jobs or min(len(todo), os.cpu_count() or 4)
        if verbose:
            print("compiling %d translation unit(s) for gfx950 with %d job(s)" % (len(todo), jobs), flush=True)
        with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
            for cmd, rc in zip(todo, ex.map(lambda c: subprocess.run(c).returncode, todo)):
                if rc != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(cmd))
    objs = [obj for _, obj, _ in _units()]
    if force or todo or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
