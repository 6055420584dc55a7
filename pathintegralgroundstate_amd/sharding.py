"""Walker sharding across the GPUs of one node and the once-per-block estimator reduction.

Walkers are independent Markov chains: rank r of R owns the contiguous block of
walkers ``shard_walkers(W, r, R)``, holds only their worldlines, and exchanges nothing during
sampling.  The single collective is a sum of the concatenated block-estimator vector
(``EstimatorVector``), through ``torch.distributed`` (backend "nccl" == RCCL over xGMI on the
GPUs, "gloo" in CPU tests) or, from a Fortran host, through ``pigs_estimators_allreduce``.
"""
from __future__ import annotations

import numpy as np


def shard_walkers(n_walkers: int, rank: int, world: int):
    """Contiguous, balanced partition: the first ``n_walkers % world`` ranks get one extra."""
    if not (0 <= rank < world) or n_walkers < 0:
        raise ValueError((n_walkers, rank, world))
    base, extra = divmod(n_walkers, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def walker_seed(seed0: int, global_walker: int) -> int:
    """Walker w runs the reference's chain for seed ``seed0 + w``."""
    return int(seed0) + int(global_walker)


class EstimatorVector:
    """Flat fp64 layout of everything that is summed over walkers once per block
    (reference accumulators: vpi.f90:456-469 energies, sample_mod.f90:392-526 g(r), S(k), n(r),
    and the move counters of vpi.f90:250-275)."""

    ENERGY = ("n_diag", "E", "K", "V", "Et", "Kt", "Vt", "E2", "K2", "V2", "Et2", "Kt2", "Vt2")
    COUNTERS = ("try_cm", "acc_cm", "try_stag", "acc_bd", "acc_head", "acc_tail", "try_cm_half",
                "acc_cm_half", "try_stag_half", "acc_bd_half", "acc_head_half", "acc_tail_half",
                "try_open", "acc_open", "try_close", "acc_close", "try_swap", "acc_swap",
                "n_obdm", "ngr")

    def __init__(self, Nbin=100, Nk=50, dim=3, Npw=0):
        self.fields = {}
        off = 0
        for name, n in ([(k, 1) for k in self.ENERGY] + [("gr", Nbin), ("Sk", dim * Nk),
                                                         ("nrho", (Npw + 1) * Nbin)] +
                        [(k, 1) for k in self.COUNTERS]):
            self.fields[name] = slice(off, off + n)
            off += n
        self.size = off
        self.data = np.zeros(off)

    def __getitem__(self, name):
        return self.data[self.fields[name]]

    def __setitem__(self, name, value):
        self.data[self.fields[name]] = value

    def add(self, name, value):
        self.data[self.fields[name]] += value


def allreduce_estimators(vec: np.ndarray, device=None):
    """Sum ``vec`` over all ranks of the default torch.distributed group (in place semantics:
    returns the summed copy).  One small message per block: latency-bound, no bucketing."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(vec, dtype=np.float64, copy=True)
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
