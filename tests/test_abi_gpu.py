"""GPU: the C ABI called from a plain C program (tests/c_abi/smoke.c) -- hipMalloc'd buffers, include/rcf_hip.h, -lrcf_hip:
what a maintainer binding the library from cgo / JNI / a C++ extension would do, with no torch in the process."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_links_and_runs(tmp_path, report):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    libdir = os.path.join(ROOT, "rcf-unsupvideoseg_amd")
    assert os.path.exists(os.path.join(libdir, "librcf_hip.so")), "build the library first (python __graft_entry__.py)"
    exe = str(tmp_path / "c_abi_smoke")
    cmd = [hipcc, "-x", "c", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "smoke.c"), "-o", exe, "-L", libdir, "-lrcf_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    report("C ABI from C: " + " | ".join(l for l in r.stdout.strip().splitlines()))
    assert r.returncode == 0 and "C ABI smoke: OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
