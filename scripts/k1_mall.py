#!/usr/bin/env python3
"""Does the Infinity Cache serve K1's slices?  Same 20 608 items per launch, over 128 / 64 / 32 / 16 walkers
(working set 127 / 63 / 32 / 16 MB, every slice read 1 / 2 / 4 / 8 times per launch by different items)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_workload  # noqa: E402
from pathintegralgroundstate_amd import SystemConfig, api  # noqa: E402


def main():
    cfg = SystemConfig(dim=3, Np=256, Nb=80)
    VT, WF = api.build_tables(cfg)
    dev = torch.device("cuda", 0)
    for W in (128, 64, 32, 16):
        rep = 128 // W
        Paths, sets = make_workload(cfg, W, rep, 1982)
        cat = [np.concatenate([s[i] for s in sets]) for i in range(5)]
        ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
        ctx.upload_all(Paths)
        d = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in cat)
        n = len(cat[0])
        out = torch.zeros(n, dtype=torch.float64, device=dev)
        ks = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        for v in (12, 2):
            ctx.set_tuning("k1_variant", v)
            w, ip, ib, xn, xo = d
            for _ in range(10):
                ctx.delta_action_batch_dev(n, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(), xo.data_ptr(), out.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ks)
            for _ in range(200):
                ctx.delta_action_batch_dev(n, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(), xo.data_ptr(), out.data_ptr())
            e1.record(ks)
            ctx.sync()
            print(f"walkers {W:4d} (working set {W * 0.989:6.1f} MB, {n} items)  variant {v:2d}: {e0.elapsed_time(e1) * 5:7.2f} us per launch", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
