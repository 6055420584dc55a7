#!/usr/bin/env python3
"""A/B of one sampler tuning key in ONE process (same box, same clocks): KEY=<name> python scripts/k6_ab.py  -> ms per MC step
for value 1 / 0 alternately at N=256, 161 beads, 128 walkers, plus a check that the trajectories are identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload
from pathintegralgroundstate_amd import SystemConfig, api
key = os.environ.get("KEY", "sweep_prio")
cfg = SystemConfig(dim=3, Np=256, Nb=80, Nlev=4, Nstag=5, Lstag=32, CMFreq=1, delta_cm=0.12)
VT, WF = api.build_tables(cfg)
W = int(os.environ.get("WALKERS", 128))
Paths, _ = make_workload(cfg, W, 1, 1982)
res = {}
for val in (1, 0, 1, 0):
    ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
    ctx.upload_all(Paths)
    ctx.sampler_init()
    ctx.set_tuning(key, val)
    for w in range(W):
        ctx.sampler_seed(w, 1982 + w)
    for i in range(3):
        ctx.sampler_step(1 + i)
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(10):
        ctx.sampler_step(4 + i)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    res[val] = (ctx.download_all(), ctx.sampler_counters16())
    print(f"{key}={val}: {dt * 1e3:.2f} ms per MC step -> {W / dt:.0f} walker-sweeps/s", flush=True)
    ctx.close()
print("same worldlines:", np.array_equal(res[1][0], res[0][0]), " same counters:", np.array_equal(res[1][1], res[0][1]))
