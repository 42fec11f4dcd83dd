"""The C ABI used from plain C: tests/c_abi/iaf_c_client.c is compiled against include/iaf_hip.h, linked with
libiaf_hip.so and the C oracle, and run -- no Python or torch in the process that drives the engine.  This is the
binding a Go/Java/Rust/... host would make."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "iaf_c_client.c")
LIB = os.path.join(ROOT, "iaf_amd", "_lib")
ORACLE = os.path.join(ROOT, "oracle", "_build")


def _build(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "iaf_c_client")
    cmd = [hipcc, "-x", "c", SRC, "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-L" + LIB, "-liaf_hip", "-L" + ORACLE, "-liaf_oracle_c", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIB, "-Wl,-rpath," + ORACLE, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_compiles_against_the_header(tmp_path):
    """CPU: the header is valid C (not only C++) and every symbol the client uses links"""
    if not os.path.exists(os.path.join(ORACLE, "libiaf_oracle_c.so")) or not os.path.exists(os.path.join(LIB, "libiaf_hip.so")):
        pytest.skip("libraries not built (run __graft_entry__.build())")
    try:
        exe = _build(tmp_path)
    except subprocess.CalledProcessError as e:
        pytest.fail("C client does not build:\n" + e.stderr[-3000:])
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [(32, 160, 2, 4, 16, 16), (32, 64, 1, 3, 8, 8), (4, 8, 2, 2, 5, 5)],
                         ids=lambda a: "z%d_h%d_d%d_B%d_%dx%d" % a)
def test_c_client_matches_c_oracle(tmp_path, args):
    exe = _build(tmp_path)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
