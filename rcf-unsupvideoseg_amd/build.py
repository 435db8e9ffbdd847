"""Builds librcf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU) -- and librcf_hip_f16.so: the same objects,
except that the four sources which touch 16-bit tensors are compiled again with -DRCF_HALF_F16 (IEEE fp16 storage instead of
bf16: csrc/rcf_common.h)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librcf_hip.so")
LIB_F16 = os.path.join(HERE, "librcf_hip_f16.so")
HALF_SOURCES = ("bn.hip", "spatial.hip", "igemm_bf16.hip", "foldbn.hip")     # the sources that depend on the 16-bit storage type
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    """everything a source may include: csrc/*.h, csrc/*.inc, the public header"""
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(HERE, "..", "include", "rcf_hip.h")]


def stale():
    if not os.path.exists(LIB) or not os.path.exists(LIB_F16):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_F16))
    deps = sources() + headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, objs_f16 = [], []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in headers())
    for src in sources():
        base = os.path.basename(src)
        variants = [("", [])] + ([("_f16", ["-DRCF_HALF_F16"])] if base in HALF_SOURCES else [])
        for suffix, defs in variants:
            obj = os.path.join(HERE, "build", base[:-4] + suffix + ".o")
            if not suffix:
                objs.append(obj)
            if suffix or base not in HALF_SOURCES:
                objs_f16.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
                cmd = [hipcc] + FLAGS + defs + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, obs in ((LIB, objs), (LIB_F16, objs_f16)):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + obs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
