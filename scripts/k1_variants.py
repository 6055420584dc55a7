#!/usr/bin/env python3
"""A/B timing of the K1 kernel variants on the bench workload (interleaved rounds, one process),
each checked against variant 1 (plain statement) and a sample against the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_workload  # noqa: E402
from pathintegralgroundstate_amd import SystemConfig, api  # noqa: E402

NAMES = {1: "v1 plain", 2: "v2 shortdiv", 3: "v2 +LDS table", 4: "v2 +compact", 5: "v2 +LDS+compact", 6: "v2 +prefetch", 7: "fast", 8: "fast +prefetch", 9: "fast +LDS table", 10: "fast +LDS +prefetch", 11: "pipe", 12: "pipe2"}


def main():
    Np = int(os.environ.get("NP", 256))
    Nb = int(os.environ.get("NB", 80))
    W = int(os.environ.get("WALKERS", 128))
    variants = [int(v) for v in os.environ.get("VARIANTS", "1,2,3,4,5").split(",")]
    cfg = SystemConfig(dim=3, Np=Np, Nb=Nb)
    VT, WF = api.build_tables(cfg)
    Paths, sets = make_workload(cfg, W, 4, 1982)
    dev = torch.device("cuda", 0)
    ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
    ctx.upload_all(Paths)
    bad, n = ctx.selftest_fastmath(2048, 512)
    print("selftest_fastmath: mismatches", bad, "of", n, flush=True)
    dsets = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in s) for s in sets]
    n_items = len(sets[0][0])
    outs = {v: torch.zeros(n_items, dtype=torch.float64, device=dev) for v in variants}
    torch.cuda.synchronize()
    ks = torch.cuda.ExternalStream(ctx.stream(), device=dev)

    def run(v, i, out):
        w, ip, ib, xn, xo = dsets[i % len(dsets)]
        ctx.delta_action_batch_dev(n_items, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(),
                                   xo.data_ptr(), out.data_ptr())

    for v in variants:
        ctx.set_tuning("k1_variant", v)
        run(v, 0, outs[v])
    ctx.sync()
    ref = outs[variants[0]].cpu().numpy()
    from oracle.pyoracle import Oracle, System
    S = System(dim=3, Np=Np, Nb=Nb)
    w, ip, ib, xn, xo = sets[0]
    sel = np.arange(0, n_items, 53)
    want = Oracle().delta_action_batch(S, WF, VT, Paths, w[sel], ip[sel], ib[sel], xn[sel], xo[sel])
    for v in variants:
        got = outs[v].cpu().numpy()
        e_or = np.max(np.abs(got[sel] - want) / (np.abs(want) + 1e-6))
        e_v1 = np.max(np.abs(got - ref) / (np.abs(ref) + 1e-6))
        print(f"variant {v} ({NAMES[v]}): max rel err vs oracle {e_or:.2e}, vs first variant {e_v1:.2e}", flush=True)
    rounds, reps = int(os.environ.get('ROUNDS', 7)), int(os.environ.get('REPS', 50))
    times = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            ctx.set_tuning("k1_variant", v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ks)
            for i in range(reps):
                run(v, i, outs[v])
            e1.record(ks)
            ctx.sync()
            times[v].append(e0.elapsed_time(e1) / reps * 1e3)
    alg = n_items * (3 * Np * 8 + 48 + 8)
    for v in variants:
        t = np.array(times[v])
        print(f"variant {v} ({NAMES[v]:18s}): median {np.median(t):8.2f} us  min {t.min():8.2f} us  "
              f"-> {alg / np.median(t) / 1e3:7.1f} GB/s algorithmic, {n_items * (Np - 1) / np.median(t) / 1e3:7.2f} G evals/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
