#!/usr/bin/env python3
"""Time the device-resident sampler (K6) at the BASELINE shape: N=256, 161 beads, W walkers."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402
from pathintegralgroundstate_amd import SystemConfig, api  # noqa: E402


def main():
    Np = int(os.environ.get("NP", 256))
    Nb = int(os.environ.get("NB", 80))
    nsteps = int(os.environ.get("STEPS", 3))
    for W in [int(x) for x in os.environ.get("WALKERS", "128").split(",")]:
        cfg = SystemConfig(dim=3, Np=Np, Nb=Nb, Nlev=int(os.environ.get('NLEV', 4)), Nstag=int(os.environ.get('NSTAG', 5)),
                           Lstag=32, CMFreq=int(os.environ.get('CMFREQ', 1)), delta_cm=0.12)
        VT, WF = api.build_tables(cfg)
        Paths, _ = make_workload(cfg, W, 1, 1982)
        ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
        ctx.upload_all(Paths)
        cworm = float(os.environ.get('CWORM', 0))
        ctx.sampler_init(CWorm=cworm, swapping=cworm > 0, Nobdm=int(os.environ.get('NOBDM', 10)))
        if cworm > 0:       # every walker starts closed; xend = bead Nb of the last particle, like the reference's init
            xe = np.repeat(Paths[:, cfg.Nb, cfg.Np - 1][:, None, :], 2, axis=1)
            ctx.sampler_set_worm(np.zeros(W, np.int32), np.zeros(W, np.int32), xe)
        if os.environ.get('SWEEP_SPLIT'):
            ctx.set_tuning('sweep_split', int(os.environ['SWEEP_SPLIT']))
        if os.environ.get('CM_SPLIT'):
            ctx.set_tuning('cm_split', int(os.environ['CM_SPLIT']))
        if os.environ.get('SWEEP_THREADS'):
            ctx.set_tuning('sweep_threads', int(os.environ['SWEEP_THREADS']))
        for w in range(W):
            ctx.sampler_seed(w, 1982 + w)
        ctx.sampler_step(1)
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            ctx.sampler_step(2 + i)
        ctx.sync()
        dt = (time.perf_counter() - t0) / nsteps
        acc = ctx.sampler_counters().sum(0) / (W * (nsteps + 1))
        if os.environ.get('TIMING'):     # library built with PIGS_EXTRA_FLAGS=-DPIGS_SWEEP_TIMING
            c16 = ctx.sampler_counters16()[:, 8:].mean(0) / (nsteps + 1)
            if os.environ.get('SWEEP_SPLIT') == '1':     # stage machine (pigs_diag.hip)
                names = ['ctl: all', 'task: load wait', 'task: pair evaluations', 'task: reduce+store', 'task: descriptor', 'task: pipe_task', 'control step total', 'bis: wait for slowest']
            else:                                        # one-launch kernel (pigs_sampler.hip), bisection moves
                names = ['move: segment + chain + end guess (wave 0)', 'move: barrier', 'end bead: Delta S', 'end bead: Metropolis', 'level: Gaussians + midpoints (wave 0)', 'level: barrier', 'level: Delta S', 'level: Metropolis']
            print('   shader-clock cycles per MC step per walker (thread 0): ' + ', '.join(f'{n} {v / 1e3:.0f}k' for n, v in zip(names, c16)), flush=True)
        if cworm > 0:
            c16 = ctx.sampler_counters16().sum(0)
            print(f"   worm: open {c16[5]}/{c16[4]}, close {c16[7]}/{c16[6]}, swap {c16[13]}/{c16[12]}, "
                  f"open walkers now {int(ctx.sampler_get_worm()[0].sum())}", flush=True)
        t1 = time.perf_counter()
        E, Ec, Ep = ctx.therm_energy_batch()
        t2 = time.perf_counter()
        ctx.local_energy_batch(0)
        ctx.local_energy_batch(2 * cfg.Nb)
        t3 = time.perf_counter()
        print(f"   estimators for {W} walkers: ThermEnergy {1e3 * (t2 - t1):.1f} ms, 2 x LocalEnergy {1e3 * (t3 - t2):.1f} ms", flush=True)
        # Delta-S items per sweep per walker (upper bound schedule): measured acceptance below
        print(f"W={W}: {dt * 1e3:8.1f} ms per MC step -> {W / dt:9.1f} walker-sweeps/s;  accepted per sweep per walker "
              f"(cm, head, tail, bis) = {acc.round(1)};  <Et/N> = {np.mean(E) / Np:.4f}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
