#!/usr/bin/env python3
"""K1 stage launches replayed from a hipGraph vs plain stream launches (dispatch overhead between dependent kernels)."""
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
    W = 128
    Paths, sets = make_workload(cfg, W, 4, 1982)
    dev = torch.device("cuda", 0)
    ctx = api.PigsContext(cfg, VT, WF, n_walkers=W)
    ctx.upload_all(Paths)
    d = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in s) for s in sets]
    n = len(sets[0][0])
    outs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in d]
    ks = torch.cuda.ExternalStream(ctx.stream(), device=dev)

    def launch(i):
        w, ip, ib, xn, xo = d[i % len(d)]
        ctx.delta_action_batch_dev(n, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(), xo.data_ptr(), outs[i % len(d)].data_ptr())

    for i in range(8):
        launch(i)
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ks)
    for i in range(200):
        launch(i)
    e1.record(ks)
    ctx.sync()
    print(f"stream launches: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per stage", flush=True)
    ref = outs[0].clone()
    K = 40
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(ks):
        with torch.cuda.graph(g, stream=ks):
            for i in range(K):
                launch(i)
    ctx.sync()
    g.replay()
    ctx.sync()
    e0.record(ks)
    with torch.cuda.stream(ks):
        for _ in range(10):
            g.replay()
    e1.record(ks)
    ctx.sync()
    print(f"graph replay ({K} stages per graph): {e0.elapsed_time(e1) / (10 * K) * 1e3:.2f} us per stage; same results: {bool(torch.equal(ref, outs[0]))}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
