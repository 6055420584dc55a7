"""Build libpigs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpigs_hip.so")
SOURCES = ["pigs_k1.hip", "pigs_sampler.hip", "pigs_diag.hip", "pigs_cm.hip", "pigs_kernels.hip", "pigs_capi.hip", "pigs_comm.cpp", "pigs_tables.cpp"]
HEADERS = ["pigs_device.h", "pigs_k1_device.h", "pigs_kernels.h", "pigs_comm.h", "pigs_sampler_device.h"]
# -ffp-contract=off: every per-pair term must round exactly like the reference's x86-64
# build (no FMA); hipcc's default is fast contraction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function"]
OBJDIR = os.environ.get("PIGS_OBJDIR", "/tmp/pigs_hip_obj")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + \
           [os.path.join(ROOT, "include", "pigs_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """One object per source, compiled in parallel (the sampler and K1 dominate), then one link.  Objects live
    in a scratch directory keyed by the flags; `force` recompiles everything."""
    if not force and not stale() and not os.environ.get("PIGS_LIB_OUT"):
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("PIGS_EXTRA_FLAGS", "").split()          # experiment builds (e.g. -DPIGS_SWEEP_TIMING)
    cc = hipcc()
    inc = "-I" + os.path.join(ROOT, "include")
    odir = os.path.join(OBJDIR, hashlib.sha1(" ".join(FLAGS + extra + [CSRC]).encode()).hexdigest()[:12])
    os.makedirs(odir, exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in [os.path.join(CSRC, h) for h in HEADERS] +
                [os.path.join(ROOT, "include", "pigs_hip.h")])

    def compile_one(src):
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj
        cmd = [cc] + FLAGS + extra + [inc, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib = os.environ.get("PIGS_LIB_OUT", LIB)          # experiment builds go next to the product library
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
