#!/usr/bin/env python3
"""Soak: the 8-wave form of the device sampler (speculative proposals, TranslateChain kernel, end-bead tasks) against the
4-wave form (none of these) over many MC steps: identical generator states, counters, worm flags; worldlines to rounding."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402
from pathintegralgroundstate_amd import SystemConfig, api  # noqa: E402


def main():
    steps = int(os.environ.get("STEPS", 600))
    for Np, Nb, dens in ((130, 40, 0.365), (256, 20, 0.365)):
        cfg = SystemConfig(dim=3, Np=Np, Nb=Nb, density=dens, dt=5e-3, Rm=1.2, Nlev=4, Nstag=2, Lstag=8, CMFreq=1,
                           delta_cm=0.3, CWorm=0.5, Nobdm=3, Nbin=50, Npw=0)
        VT, WF = api.build_tables(cfg)
        W = 4
        res = {}
        for form in ("8 waves", "4 waves"):
            ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
            ctx.sampler_init(CWorm=cfg.CWorm, swapping=True, Nobdm=cfg.Nobdm, Nbin=cfg.Nbin, Npw=0)
            if form == "4 waves":
                ctx.set_tuning("sweep_threads", 256)
                ctx.set_tuning("cm_split", 0)
            Paths, _ = make_workload(cfg, W, 1, 4000)
            for w in range(W):
                ctx.sampler_seed(w, 4000 + w)
            ctx.upload_all(Paths)
            xe = np.repeat(Paths[:, cfg.Nb, cfg.Np - 1][:, None, :], 2, axis=1)
            ctx.sampler_set_worm(np.zeros(W, np.int32), np.zeros(W, np.int32), xe)
            for istep in range(1, steps + 1):
                ctx.sampler_step(istep)
            res[form] = (ctx.download_all(), ctx.sampler_counters16(), [ctx.sampler_get_rng(w) for w in range(W)], ctx.sampler_get_worm())
            ctx.close()
        a, b = res["8 waves"], res["4 waves"]
        L = np.asarray(cfg.Lbox[:3])
        d = a[0] - b[0]
        worst = np.max(np.abs(d - L * np.round(d / L)))
        same = np.array_equal(a[1], b[1]) and all(x[0] == y[0] and np.array_equal(np.asarray(x[1]), np.asarray(y[1])) for x, y in zip(a[2], b[2])) \
            and np.array_equal(a[3][0], b[3][0])
        print(f"N={Np} beads={2 * Nb + 1} steps={steps}: decisions identical: {same}; worldlines max |d| = {worst:.2e}; "
              f"accepted (cm, head, tail, bis) per walker-step = {(a[1][:, :4].sum(0) / (W * steps)).round(1)}", flush=True)
        assert same and worst < 1e-8


if __name__ == "__main__":
    main()
