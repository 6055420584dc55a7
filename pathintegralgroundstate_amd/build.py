"""Build libpigs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpigs_hip.so")
SOURCES = ["pigs_k1.hip", "pigs_sampler.hip", "pigs_kernels.hip", "pigs_capi.hip", "pigs_comm.cpp", "pigs_tables.cpp"]
HEADERS = ["pigs_device.h", "pigs_k1_device.h", "pigs_kernels.h", "pigs_comm.h"]
# -ffp-contract=off: every per-pair term must round exactly like the reference's x86-64
# build (no FMA); hipcc's default is fast contraction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


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
    if not force and not stale():
        return LIB
    extra = os.environ.get("PIGS_EXTRA_FLAGS", "").split()          # experiment builds (e.g. -DPIGS_SWEEP_TIMING)
    cmd = [hipcc()] + FLAGS + extra + ["-I" + os.path.join(ROOT, "include")] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
