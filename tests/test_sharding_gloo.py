"""The N>1 path on CPU: world_size-2 gloo processes shard walkers with no data-path collective
and sum the block-estimator vector once (the same code path bench.py / the sampler use with
backend "nccl" == RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest

from pathintegralgroundstate_amd.sharding import (EstimatorVector, allreduce_estimators,
                                                  shard_walkers, walker_seed)


def test_shard_walkers_partitions_exactly():
    for W in (0, 1, 7, 128, 1024, 1025):
        for R in (1, 2, 3, 8):
            got = [w for r in range(R) for w in shard_walkers(W, r, R)]
            assert got == list(range(W))
            sizes = [len(shard_walkers(W, r, R)) for r in range(R)]
            assert max(sizes) - min(sizes) <= 1
    assert list(shard_walkers(1024, 3, 8)) == list(range(384, 512))
    with pytest.raises(ValueError):
        shard_walkers(4, 2, 2)
    assert walker_seed(1982, 5) == 1987


def test_estimator_vector_layout():
    ev = EstimatorVector(Nbin=100, Nk=50, dim=3, Npw=0)
    assert ev.size == 13 + 100 + 150 + 100 + 20
    ev.add("E", 2.5)
    ev.add("gr", np.ones(100))
    ev["acc_cm"] = 7
    assert ev["E"][0] == 2.5 and ev["gr"].sum() == 100 and ev["acc_cm"][0] == 7
    assert ev.data.sum() == 2.5 + 100 + 7


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_walkers(W, rank, world)
        ev = EstimatorVector()
        # each walker contributes a deterministic "block result"; no communication until here
        for w in mine:
            rng = np.random.default_rng(walker_seed(1982, w))
            ev.add("n_diag", 1.0)
            ev.add("E", rng.normal())
            ev.add("gr", rng.integers(0, 5, 100).astype(float))
            ev.add("acc_cm", float(w))
        tot = allreduce_estimators(ev.data)
        q.put((rank, list(mine), tot))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_estimator_allreduce():
    import torch.multiprocessing as mp
    W, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] + res[1][1] == list(range(W))
    # expected = single-process sum over all walkers
    ev = EstimatorVector()
    for w in range(W):
        rng = np.random.default_rng(walker_seed(1982, w))
        ev.add("n_diag", 1.0)
        ev.add("E", rng.normal())
        ev.add("gr", rng.integers(0, 5, 100).astype(float))
        ev.add("acc_cm", float(w))
    for _, _, tot in res:
        assert np.allclose(tot, ev.data, rtol=0, atol=1e-12)
        assert tot[ev.fields["n_diag"]][0] == W
