#!/usr/bin/env python3
"""Two sampling contexts of W/2 walkers on ONE GPU against one context of W walkers (N=256, 161 beads, stock schedule, moves only):
the shards' TranslateChain kernels are chained device-wide (tuning key "cm_shared"), so one shard's TranslateChain (H = 3) runs on the
CUs the other shard's bisection phase leaves idle.  Prints ms per MC step of all W walkers and checks that every walker's trajectory is
the one the single context gives.   usage (GPU box): [WALKERS=128] [H=3] python scripts/k6_stagger.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload
from pathintegralgroundstate_amd import SystemConfig, api
cfg = SystemConfig(dim=3, Np=256, Nb=80, Nlev=4, Nstag=5, Lstag=32, CMFreq=1, delta_cm=0.12)
VT, WF = api.build_tables(cfg)
W = int(os.environ.get("WALKERS", 128))
H = int(os.environ.get("H", 3))
G = int(os.environ.get("SHARDS", 2))
NWARM, NSTEP = 3, int(os.environ.get("STEPS", 10))
Paths, _ = make_workload(cfg, W, 1, 1982)

def run(groups):
    ctxs = []
    for lo, hi in groups:
        c = api.PigsContext(cfg, VT, WF, n_walkers=hi - lo)
        c.upload_all(Paths[lo:hi])
        c.sampler_init()
        if len(groups) > 1:
            c.set_tuning("cm_shared", 1)
            c.set_tuning("cm_split", H)
        for w in range(lo, hi):
            c.sampler_seed(w - lo, 1982 + w)
        ctxs.append(c)
    for i in range(NWARM):
        for c in ctxs:
            c.sampler_step(1 + i)
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for i in range(NSTEP):
        for c in ctxs:
            c.sampler_step(1 + NWARM + i)
    for c in ctxs:
        c.sync()
    dt = (time.perf_counter() - t0) / NSTEP
    out = np.concatenate([c.download_all() for c in ctxs])
    cnt = np.concatenate([c.sampler_counters16() for c in ctxs])
    for c in ctxs:
        c.close()
    return dt, out, cnt

res = {}
extra = None
if os.environ.get("EXTRA") == "1":          # an idle third context, as in bench.py (its main context lives through the extra legs)
    extra = api.PigsContext(cfg, VT, WF, n_walkers=W)
    extra.upload_all(Paths)
    extra.sampler_init()
    extra.sampler_step(1)
    extra.sync()
shards = [(g * W // G, (g + 1) * W // G) for g in range(G)]
for name, groups in (("one context", [(0, W)]), ("two shards", shards), ("one context", [(0, W)]), ("two shards", shards)):
    dt, out, cnt = run(groups)
    res[name] = (out, cnt)
    print(f"{name if len(groups) == 1 else str(G) + ' shards, H=' + str(H):14s}: {dt * 1e3:.2f} ms per MC step of {W} walkers -> {W / dt:.0f} walker-sweeps/s", flush=True)
print("same worldlines:", np.array_equal(res["one context"][0], res["two shards"][0]), " same counters:", np.array_equal(res["one context"][1], res["two shards"][1]))
