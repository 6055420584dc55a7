#!/usr/bin/env python3
"""Fuzz of the hot path through the C ABI against the pinned CPU oracle (test infrastructure: the checker): random system
shapes -- dim 1..3, 2..300 particles (every remainder mod 64 and mod 8), 3..81 beads, periodic and trapped, table and
analytic trial function, odd and even Nmax, coarse and fine tables, several walkers -- and for each: a random Delta-S batch
on every K1 variant the shape admits, PotentialEnergy of a few slices, ThermEnergy, LocalEnergy at both ends, the one-call and
the overlapped estimator entry points, commit + download.  Tolerances as tests/test_gpu_parity.py.
usage (GPU box): python scripts/k1_fuzz.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import delta_s_tolerance, same_bits, term_scales  # noqa: E402
from oracle.pyoracle import Oracle, System, build_oracle  # noqa: E402
from pathintegralgroundstate_amd import SystemConfig, api  # noqa: E402

build_oracle()
o = Oracle()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
bad = 0
t0 = time.time()
for case in range(ncase):
    dim = int(rng.integers(1, 4))
    trap = bool(rng.random() < 0.25)
    Np = int(rng.choice([2, 3, 7, 8, 9, 15, 16, 31, 33, 63, 64, 65, 100, 127, 128, 129, 191, 193, 255, 256, 257, 300]))
    if trap:
        Np = min(Np, 33)
    Nb = int(rng.integers(1, 41))
    Nmax = int(rng.choice([63, 64, 1000, 4001, 10000]))
    wf_table = bool(rng.random() < 0.8)
    dens = float(rng.choice([0.05, 0.2, 0.365])) if dim == 3 else float(rng.choice([0.05, 0.2]))
    a_ho = [1.0, 1.3, 0.8][:dim] if trap else None
    kw = dict(dim=dim, Np=Np, Nb=Nb, Nmax=Nmax, density=dens, dt=float(rng.choice([5e-3, 2e-2])), Rm=1.2, trap=trap)
    if trap:
        kw["a_ho"] = a_ho
    try:
        cfg = SystemConfig(wf_table=wf_table, **kw)
        S = System(wf_table=wf_table, **{k: v for k, v in kw.items()})
    except Exception as e:      # noqa: BLE001
        print("case", case, "skipped (config):", e)
        continue
    VT, WF = api.build_tables(cfg)
    W = int(rng.integers(1, 4))
    L = np.asarray(S.Lbox[:dim])
    Paths = []
    far = (not trap) and rng.random() < 0.35          # separations beyond 1.5 L: the reference folds once and leaves them outside the cutoff
    for w in range(W):
        P, _ = o.init_path(S, 100 + case * 7 + w)
        P = P + rng.normal(0, 0.12 if not trap else 0.3, P.shape)
        if not trap:
            P = np.where(P > L / 2, P - L, P)
            P = np.where(P < -L / 2, P + L, P)
            if far:                      # beads left outside the box by ONE fold of a far proposal (pbc_mod.f90:20-21)
                P = P + L * rng.integers(-3, 4, P.shape) * (rng.random(P.shape) < 0.15)
        Paths.append(P)
    Paths = np.stack(Paths)
    n = 300
    wv = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    ib[:8] = [0, 2 * Nb, 0, 2 * Nb, min(1, 2 * Nb), max(2 * Nb - 1, 0), Nb, Nb]          # end beads, odd and even ones
    xold = Paths[wv, ib, ip - 1].copy()
    xnew = xold + rng.normal(0, 0.1, xold.shape) * (1.0 if not far else np.where(rng.random((n, 1)) < 0.5, 1.0, 30.0 * float(L.max())))
    if not trap:
        xnew = np.where(xnew > L / 2, xnew - L, xnew)
        xnew = np.where(xnew < -L / 2, xnew + L, xnew)
    want = o.delta_action_batch(S, WF, VT, Paths, wv, ip, ib, xnew, xold)
    fin = np.isfinite(want)
    variants = [0, 1, 2, 14] + ([] if trap else [7, 8]) + ([12, 13] if (not trap and Np <= 256 and Nmax % 2 == 0) else [])
    msgs = []
    with api.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        if not same_bits(ctx.download_all(), Paths):
            msgs.append("upload/download round trip")
        tol = None
        for v in variants:
            ctx.set_tuning("k1_variant", v)
            got = ctx.delta_action_batch(wv, ip, ib, xnew, xold)
            if not np.array_equal(np.isnan(got), np.isnan(want)):
                msgs.append(f"variant {v}: NaN pattern")
                continue
            if tol is None:
                tols = []
                for w in range(W):
                    sel = np.flatnonzero(wv == w)
                    sv, sf, su = term_scales(S, VT, WF, Paths[w], ip[sel], ib[sel], xnew[sel], xold[sel])
                    t = np.zeros(n); t[sel] = delta_s_tolerance(S, sv, sf, su)
                    tols.append(t)
                tol = np.sum(tols, 0)
            err = np.abs(got - want)[fin]
            if err.size and np.any(err > 3 * tol[fin] + 1e-13 * np.abs(want[fin])):
                msgs.append(f"variant {v}: Delta S off by {np.max(err / (tol[fin] + 1e-300)):.1f} x tolerance")
        ctx.set_tuning("k1_variant", 0)
        # estimators
        E, Ec, Ep = ctx.therm_energy_batch()
        for w in range(W):
            te = np.array(o.therm_energy(S, VT, Paths[w]))
            if np.all(np.isfinite(te)) and not np.allclose([E[w], Ec[w], Ep[w]], te, rtol=1e-10, atol=0):
                msgs.append(f"ThermEnergy walker {w}: {[E[w], Ec[w], Ep[w]]} vs {te}")
            for end in (0, 2 * Nb):
                le = ctx.LocalEnergy(w, end)
                lo = np.array(o.local_energy(S, WF, VT, Paths[w][end]))
                sc = abs(lo[1]) + abs(lo[2])
                if np.all(np.isfinite(lo)) and not (abs(le[0] - lo[0]) <= 1e-10 * sc and abs(le[1] - lo[1]) <= 1e-10 * sc and abs(le[2] - lo[2]) <= 1e-10 * abs(lo[2]) + 1e-300):
                    msgs.append(f"LocalEnergy walker {w} slice {end}: {le} vs {lo}")
        one = ctx.diagonal_estimators(20, cfg.rcut / 20, 4)
        ctx.diagonal_estimators_begin(20, cfg.rcut / 20, 4)
        two = ctx.diagonal_estimators_end()
        for k in ("E1", "K1", "V1", "E2", "K2", "V2", "Et", "Kt", "Vt"):
            if not same_bits(one[k], two[k]):
                msgs.append(f"overlapped estimators differ from the one-call form in {k}")
        if not same_bits(one["Et"], E):
            msgs.append("one-call ThermEnergy differs from the separate call")
        # commit a few beads and read them back
        k = 40
        ctx.commit_beads(wv[:k], ip[:k], ib[:k], xnew[:k])
        Q = Paths.copy()
        for i in range(k):
            Q[wv[i], ib[i], ip[i] - 1] = xnew[i]
        if not same_bits(ctx.download_all(), Q):
            msgs.append("commit_beads")
    tag = f"case {case}: dim={dim} Np={Np} Nb={Nb} Nmax={Nmax} trap={trap} wf_table={wf_table} W={W} far={far} variants={variants}"
    if msgs:
        bad += 1
        print("FAIL", tag, "|", "; ".join(msgs), flush=True)
    elif case % 10 == 0:
        print("ok  ", tag, flush=True)
print(f"{ncase} cases, {bad} failing, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
