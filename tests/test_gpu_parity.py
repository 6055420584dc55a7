"""Parity of the HIP path (through the C ABI, libpigs_hip.so) with the reference.

Checkers, in order of authority: the committed golden vectors (generated from the unmodified
reference), the pinned C restatement (oracle/) on fresh seeded inputs, and size-independent
properties at the full BASELINE sizes.  The compiled reference stays in the build container:
nothing under -m gpu loads oracle/_ref (SURVEY.md 8c travel rule).

Tolerances (fp64 path; stated per SURVEY §7 hard-part 2 and BASELINE.json's 1e-10):
  * per-pair terms are bit-identical to the reference's, only the summation order differs, so
    |DeltaS_gpu - DeltaS_ref| <= 2e-13 * (sum of |terms|)       (helpers.delta_s_tolerance)
  * accept/reject decisions on a fixed uniform are identical
  * energy estimators (ThermEnergy, LocalEnergy, PotentialEnergy) agree to 1e-10 relative
  * index / layout work (upload, download, commit, swap) is bit-exact.
"""
import numpy as np
import pytest

from conftest import config_from_golden, load_golden, system_from_golden
from helpers import delta_s_tolerance, same_bits, term_scales

pytestmark = pytest.mark.gpu

REL = 1e-10


def _tables(d):
    if "VTable" in d:
        return d["VTable"], d["LogWF"]
    t = load_golden("tables_he4_n64")
    return t["VTable"], t["LogWF"]


def _close_rel(a, b, rel=REL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all(np.abs(a - b) <= rel * np.abs(b) + 1e-300))


@pytest.mark.parametrize("name", ["he4_n64_eq", "he4_n64_rnd", "ho1d_n2", "trap3d_n8", "pbc2d_n16"])
def test_delta_action_vs_golden(gpu_lib, name):
    d = load_golden(name)
    cfg, S = config_from_golden(d), system_from_golden(d)
    VT, WF = _tables(d)
    n = len(d["ip"])
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=2) as ctx:
        ctx.upload(1, d["Path"])
        assert same_bits(ctx.download(1), d["Path"])
        w = np.ones(n, np.int32)
        dS = ctx.delta_action_batch(w, d["ip"], d["ib"], d["xnew"], d["xold"])
        parts = ctx.delta_action_parts(w, d["ip"], d["ib"], d["xnew"], d["xold"])
    ref = d["DeltaS"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(dS), np.isnan(ref))            # NaN (r<dr, quirk Q4) parity
    sv, sf, su = term_scales(S, VT, WF, d["Path"], d["ip"], d["ib"], d["xnew"], d["xold"])
    tol = delta_s_tolerance(S, sv, sf, su)
    err = np.abs(dS - ref)[fin]
    assert np.all(err <= tol[fin]), (err.max(), (err / tol[fin]).max())
    # components: DeltaPot, DeltaF2, DeltaLogPsi
    pr = d["parts"]
    assert np.all(np.abs(parts[fin, 0] - pr[fin, 0]) <= 2e-13 * sv[fin] + 1e-300)
    assert np.all(np.abs(parts[fin, 1] - pr[fin, 1]) <= 8e-13 * sf[fin] ** 2 + 1e-300)
    assert np.all(np.abs(parts[fin, 2] - pr[fin, 2]) <= 2e-13 * su[fin] + 1e-300)
    # Metropolis decisions on a fixed uniform stream are identical
    u = np.random.default_rng(1).uniform(size=n)
    with np.errstate(over="ignore", invalid="ignore"):
        assert np.array_equal(np.exp(-dS) >= u, np.exp(-ref) >= u)


@pytest.mark.parametrize("name", ["he4_n64_eq", "he4_n64_rnd", "ho1d_n2", "trap3d_n8", "pbc2d_n16"])
def test_energies_vs_golden(gpu_lib, name):
    d = load_golden(name)
    cfg = config_from_golden(d)
    VT, WF = _tables(d)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=3) as ctx:
        for w in range(3):
            ctx.upload(w, d["Path"])
        pot_f2 = np.array([ctx.PotentialEnergy(2, ib, True) for ib in range(cfg.M)])
        pot0 = np.array([ctx.PotentialEnergy(0, ib, False)[0] for ib in range(cfg.M)])
        E, Ec, Ep = ctx.therm_energy_batch()
        le0 = ctx.local_energy_batch(0)
        le1 = ctx.local_energy_batch(2 * cfg.Nb, walkers=[1])
    fin = np.isfinite(d["pot_f2"]).all(1)
    assert _close_rel(pot_f2[fin], d["pot_f2"][fin]) and _close_rel(pot0[fin], d["pot_only"][fin])
    if np.all(np.isfinite(d["therm"])):
        for w in range(3):
            assert _close_rel([E[w], Ec[w], Ep[w]], d["therm"])
    if np.all(np.isfinite(d["local"])):
        assert _close_rel([le0[0][2], le0[1][2], le0[2][2]], d["local"][0])
        assert _close_rel([le1[0][0], le1[1][0], le1[2][0]], d["local"][1])


@pytest.mark.parametrize("tag", ["lj", "dipolar"])
def test_external_tables_vs_golden(gpu_lib, tag):
    """Config-2 (Lennard-Jones) and config-5 (dipolar) potentials, pinned at the table boundary."""
    base, d = load_golden("he4_n64_eq"), load_golden(f"he4_n64_table_{tag}")
    cfg, S = config_from_golden(base), system_from_golden(base)
    WF = load_golden("tables_he4_n64")["LogWF"]
    VT = d["VTable"]
    n = len(d["ip"])
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=1) as ctx:
        ctx.upload(0, base["Path"])
        dS = ctx.delta_action_batch(np.zeros(n, np.int32), d["ip"], d["ib"], d["xnew"], d["xold"])
        E, Ec, Ep = ctx.therm_energy_batch()
    sv, sf, su = term_scales(S, VT, WF, base["Path"], d["ip"], d["ib"], d["xnew"], d["xold"])
    assert np.all(np.abs(dS - d["DeltaS"]) <= delta_s_tolerance(S, sv, sf, su))
    assert _close_rel([E[0], Ec[0], Ep[0]], d["therm"])


def test_short_division_and_sqrt_are_exact(gpu_lib):
    """The kernels' 3-instruction division / fused sqrt forms are bit-identical to IEEE `/` and
    sqrt() (so per-pair terms stay bit-identical to the reference's): 2.7e8 random operands."""
    t = load_golden("tables_he4_n256")
    cfg = config_from_golden(t)
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=1) as ctx:
        bad, n = ctx.selftest_fastmath(2048, 512)
    assert n == 2048 * 256 * 512 and bad == [0, 0, 0, 0], bad


def test_stream_read_measurement_aid(gpu_lib):
    """pigs_selftest_stream_read: bytes per pass = the resident worldlines (padded SoA), a positive time, and the
    worldlines are still what was uploaded."""
    t = load_golden("tables_he4_n256")
    cfg = config_from_golden(t)
    rng = np.random.default_rng(3)
    P = rng.uniform(-1, 1, (4, cfg.M, cfg.Np, cfg.dim))
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=4) as ctx:
        ctx.upload_all(P)
        nbytes, sec = ctx.stream_read(5)
        assert nbytes >= P.nbytes and nbytes < 1.1 * P.nbytes and 0 < sec < 1e-2
        assert np.array_equal(ctx.download_all(), P)


@pytest.mark.parametrize("variant", [1, 2, 7, 8, 12, 13, 14])
@pytest.mark.parametrize("name", ["he4_n64_eq", "he4_n64_rnd", "pbc2d_n16", "trap3d_n8"])
def test_every_k1_variant_vs_golden(gpu_lib, name, variant):
    d = load_golden(name)
    cfg, S = config_from_golden(d), system_from_golden(d)
    VT, WF = _tables(d)
    n = len(d["ip"])
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=1) as ctx:
        ctx.set_tuning("k1_variant", variant)
        ctx.upload(0, d["Path"])
        dS = ctx.delta_action_batch(np.zeros(n, np.int32), d["ip"], d["ib"], d["xnew"], d["xold"])
    ref = d["DeltaS"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(dS), np.isnan(ref))
    sv, sf, su = term_scales(S, VT, WF, d["Path"], d["ip"], d["ib"], d["xnew"], d["xold"])
    tol = delta_s_tolerance(S, sv, sf, su)
    assert np.all(np.abs(dS - ref)[fin] <= tol[fin])
    u = np.random.default_rng(1).uniform(size=n)
    with np.errstate(over="ignore", invalid="ignore"):
        assert np.array_equal(np.exp(-dS) >= u, np.exp(-ref) >= u)


@pytest.mark.parametrize("Np,Nb,W,n", [(256, 80, 6, 30000), (64, 40, 3, 6000)])
def test_short_arithmetic_vs_exact_forms(gpu_lib, oracle, Np, Nb, W, n):
    """The library's default Delta-S arithmetic for periodic systems (variants 7, 8, 12, 13: rint minimum image, one
    Newton step after v_rsq_f64, interpolation in the normalised cell coordinate) against the variant that keeps
    the reference's rounding of every term (2): each part -- DeltaPot, DeltaF2, DeltaLogPsi -- agrees to 2e-13 of
    the sum of its terms' magnitudes (the tolerance the oracle tests use), the Metropolis decision on a common
    uniform is the same, and the cutoff membership is identical (r^2 keeps the reference's rounding sequence)."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(dim=3, Np=Np, Nb=Nb)
    cfg = SystemConfig(dim=3, Np=Np, Nb=Nb)
    VT, WF = oracle.tables(S)
    Paths = _worldlines(oracle, S, W, 5, 0.12)
    rng = np.random.default_rng(11)
    w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, n, 0.1)
    res = {}
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        for v in (2, 7, 8, 12, 13, 0):
            ctx.set_tuning("k1_variant", v)
            res[v] = (ctx.delta_action_batch(w, ip, ib, xnew, xold), ctx.delta_action_parts(w, ip, ib, xnew, xold))
    sv = np.zeros(n); sf = np.zeros(n); su = np.zeros(n)
    for k in range(W):
        m = w == k
        sv[m], sf[m], su[m] = term_scales(S, VT, WF, Paths[k], ip[m], ib[m], xnew[m], xold[m])
    tol = delta_s_tolerance(S, sv, sf, su)
    ex, exp_ = res[2]
    fin = np.isfinite(ex)
    u = rng.uniform(size=n)
    for v in (7, 8, 12, 13, 0):
        dS, parts = res[v]
        assert np.array_equal(np.isnan(dS), np.isnan(ex)), v
        assert np.all(np.abs(dS - ex)[fin] <= tol[fin]), (v, np.max((np.abs(dS - ex) / tol)[fin]))
        assert np.all(np.abs(parts[:, 0] - exp_[:, 0])[fin] <= 2e-13 * sv[fin] + 1e-300), v
        assert np.all(np.abs(parts[:, 2] - exp_[:, 2])[fin] <= 2e-13 * su[fin] + 1e-300), v
        with np.errstate(over="ignore", invalid="ignore"):
            assert np.array_equal(np.exp(-dS) >= u, np.exp(-ex) >= u), v
    # all short-arithmetic variants evaluate the same expressions: their results differ by summation order only
    assert np.all(np.abs(res[7][0] - res[12][0])[fin] <= tol[fin])
    # pipe2 and its plain-grid twin: identical bits
    assert same_bits(res[12][0], res[13][0])


def _random_batch(rng, S, Paths, n, sigma):
    W = Paths.shape[0]
    w = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, S.Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    ib[::9] = 0
    ib[4::9] = 2 * S.Nb
    xold = Paths[w, ib, ip - 1].copy()
    xnew = xold + rng.normal(0, sigma, xold.shape)
    L = np.asarray(S.Lbox[:S.dim])
    xnew = np.where(xnew > L / 2, xnew - L, xnew)
    xnew = np.where(xnew < -L / 2, xnew + L, xnew)
    return w, ip, ib, xnew, xold


def _worldlines(oracle, S, W, seed0, spread):
    """W seeded worldlines: reference-style init (all beads equal) + a Brownian-bridge-like spread."""
    rng = np.random.default_rng(seed0)
    Ps = []
    L = np.asarray(S.Lbox[:S.dim])
    for w in range(W):
        P, _ = oracle.init_path(S, 1982 + w)
        P = P + rng.normal(0, spread, P.shape)
        P = np.where(P > L / 2, P - L, P)
        P = np.where(P < -L / 2, P + L, P)
        Ps.append(P)
    return np.stack(Ps)


@pytest.mark.parametrize("Np,Nb,W,n", [(64, 40, 3, 4000), (256, 80, 4, 6000), (37, 5, 2, 1500)])
def test_delta_action_vs_oracle_seeded(gpu_lib, oracle, Np, Nb, W, n):
    """Fresh seeded batches over several walkers at the C2 / C3 shapes (+ a ragged Np)."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(dim=3, Np=Np, Nb=Nb)
    cfg = SystemConfig(dim=3, Np=Np, Nb=Nb)
    VT, WF = oracle.tables(S)                       # == the reference's tables bit for bit
    Paths = _worldlines(oracle, S, W, 77, 0.12)
    rng = np.random.default_rng(Np)
    w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, n, 0.1)
    want = oracle.delta_action_batch(S, WF, VT, Paths, w, ip, ib, xnew, xold)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        assert same_bits(ctx.download_all(), Paths)
        got = ctx.delta_action_batch(w, ip, ib, xnew, xold)
        E, Ec, Ep = ctx.therm_energy_batch()
        le = ctx.local_energy_batch(0)
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    sv = np.zeros(n); sf = np.zeros(n); su = np.zeros(n)
    for k in range(W):
        m = w == k
        sv[m], sf[m], su[m] = term_scales(S, VT, WF, Paths[k], ip[m], ib[m], xnew[m], xold[m])
    tol = delta_s_tolerance(S, sv, sf, su)
    assert np.all(np.abs(got - want)[fin] <= tol[fin])
    u = rng.uniform(size=n)
    with np.errstate(over="ignore", invalid="ignore"):
        assert np.array_equal(np.exp(-got) >= u, np.exp(-want) >= u)
    for k in range(W):
        te = oracle.therm_energy(S, VT, Paths[k])
        if np.all(np.isfinite(te)):
            assert _close_rel([E[k], Ec[k], Ep[k]], te)
        lo = oracle.local_energy(S, WF, VT, Paths[k][0])
        if np.all(np.isfinite(lo)):
            assert _close_rel([le[0][k], le[1][k], le[2][k]], lo)


def test_commit_swap_and_edge_cases(gpu_lib, oracle):
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(dim=3, Np=64, Nb=6)
    cfg = SystemConfig(dim=3, Np=64, Nb=6)
    VT, WF = oracle.tables(S)
    Paths = _worldlines(oracle, S, 2, 5, 0.1)
    rng = np.random.default_rng(2)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=2) as ctx:
        ctx.upload_all(Paths)
        # empty batch is a no-op
        assert ctx.delta_action_batch([], [], [], np.zeros((0, 3)), np.zeros((0, 3))).size == 0
        # commit: bit-exact scatter, last write wins in host order for distinct targets
        w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, 200, 0.2)
        key = (w.astype(np.int64) * 1000 + ip) * 1000 + ib
        _, first = np.unique(key, return_index=True)
        w, ip, ib, xnew = w[first], ip[first], ib[first], xnew[first]
        ctx.commit_beads(w, ip, ib, xnew)
        Paths[w, ib, ip - 1] = xnew
        assert same_bits(ctx.download_all(), Paths)
        # swap tails: beads Nb..2Nb of two particles exchange (reference Swap accept branch)
        ctx.swap_tails(1, 3, 17)
        t = Paths[1, S.Nb:, 2].copy()
        Paths[1, S.Nb:, 2] = Paths[1, S.Nb:, 16]
        Paths[1, S.Nb:, 16] = t
        assert same_bits(ctx.download_all(), Paths)
        # staged (pinned, zero-copy) forms give the same bits as the copying forms
        w2, ip2, ib2, xn2, xo2 = _random_batch(rng, S, Paths, 777, 0.2)
        assert same_bits(ctx.delta_action_staged(w2, ip2, ib2, xn2, xo2), ctx.delta_action_batch(w2, ip2, ib2, xn2, xo2))
        key = (w2.astype(np.int64) * 1000 + ip2) * 1000 + ib2
        _, first = np.unique(key, return_index=True)
        ctx.commit_staged(w2[first], ip2[first], ib2[first], xn2[first])
        Paths[w2[first], ib2[first], ip2[first] - 1] = xn2[first]
        assert same_bits(ctx.download_all(), Paths)
        # bad indices are refused, not faulted on
        with pytest.raises(gpu_lib.PigsError):
            ctx.delta_action_batch([0], [65], [0], np.zeros((1, 3)), np.zeros((1, 3)))
        with pytest.raises(gpu_lib.PigsError):
            ctx.delta_action_batch([2], [1], [0], np.zeros((1, 3)), np.zeros((1, 3)))
        with pytest.raises(gpu_lib.PigsError):
            ctx.commit_beads([0], [1], [13], np.zeros((1, 3)))
        # row ip of the slice is never read (aliasing contract): garbage there changes nothing
        w1, ip1, ib1, xn1, xo1 = _random_batch(rng, S, Paths, 64, 0.1)
        _, first = np.unique(w1.astype(np.int64) * 1000 + ib1, return_index=True)   # one item per slice
        w1, ip1, ib1, xn1, xo1 = w1[first], ip1[first], ib1[first], xn1[first], xo1[first]
        a = ctx.delta_action_batch(w1, ip1, ib1, xn1, xo1)
        ctx.commit_beads(w1, ip1, ib1, np.full((len(w1), 3), 1e300))
        b = ctx.delta_action_batch(w1, ip1, ib1, xn1, xo1)
        assert same_bits(a, b)


def test_full_size_properties(gpu_lib, oracle):
    """BASELINE config 3 (N=256, 161 beads, 128 walkers): properties that need no oracle run."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(dim=3, Np=256, Nb=80)
    cfg = SystemConfig(dim=3, Np=256, Nb=80)
    t = load_golden("tables_he4_n256")
    VT, WF = t["VTable"], t["LogWF"]
    W = 128
    rng = np.random.default_rng(123)
    L = S.Lbox[0]
    # jittered-lattice worldlines (no overlaps -> finite potentials), cheap to build at this size
    g = int(np.ceil(S.Np ** (1 / 3)))
    lat = (np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3)[:S.Np] + 0.5) * (L / g) - L / 2
    Paths = lat[None, None] + rng.normal(0, 0.08, (W, S.M, S.Np, 3))
    Paths = np.where(Paths > L / 2, Paths - L, Paths)
    Paths = np.where(Paths < -L / 2, Paths + L, Paths)
    n = W * S.M                                       # one full-chain stage: every slice once
    w = np.repeat(np.arange(W, dtype=np.int32), S.M)
    ib = np.tile(np.arange(S.M, dtype=np.int32), W)
    ip = np.repeat(rng.integers(1, S.Np + 1, W).astype(np.int32), S.M)
    xa = Paths[w, ib, ip - 1].copy()
    xb = xa + rng.normal(0, 0.05, xa.shape)
    xc = xa + rng.normal(0, 0.05, xa.shape)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        zero = ctx.delta_action_batch(w, ip, ib, xa, xa)
        ab = ctx.delta_action_batch(w, ip, ib, xb, xa)
        ba = ctx.delta_action_batch(w, ip, ib, xa, xb)
        bc = ctx.delta_action_batch(w, ip, ib, xc, xb)
        ac = ctx.delta_action_batch(w, ip, ib, xc, xa)
        perm = rng.permutation(n)
        ab_p = ctx.delta_action_batch(w[perm], ip[perm], ib[perm], xb[perm], xa[perm])
        # K1 vs K2+K5: DeltaPot of a move == PotentialEnergy(after commit) - PotentialEnergy(before)
        sel = rng.choice(n, 24, replace=False)
        parts = ctx.delta_action_parts(w[sel], ip[sel], ib[sel], xb[sel], xa[sel])
        before = np.array([ctx.PotentialEnergy(int(w[i]), int(ib[i]))[0] for i in sel])
        ctx.commit_beads(w[sel], ip[sel], ib[sel], xb[sel])
        after = np.array([ctx.PotentialEnergy(int(w[i]), int(ib[i]))[0] for i in sel])
        E, Ec, Ep = ctx.therm_energy_batch()
    assert np.all(zero == 0.0)                        # x -> x is exactly zero
    assert same_bits(ab, -ba)                         # exact antisymmetry
    assert same_bits(ab_p, ab[perm])                  # batch order does not matter (determinism)
    assert np.all(np.abs(ab + bc - ac) <= 1e-9 * (np.abs(ab) + np.abs(bc) + np.abs(ac)) + 1e-10)
    assert np.all(np.abs((after - before) - parts[:, 0]) <= 1e-10 * np.abs(before))
    assert np.all(np.isfinite(E)) and np.all(np.abs(Ec + Ep - E) <= 1e-9 * np.abs(E))
    # spot-check a sample of the full-size batch against the pinned oracle
    sel = rng.choice(n, 600, replace=False)
    Paths_before = Paths
    want = oracle.delta_action_batch(S, WF, VT, Paths_before, w[sel], ip[sel], ib[sel], xb[sel], xa[sel])
    assert np.all(np.abs(ab[sel] - want) <= 1e-10 * np.abs(want) + 1e-11)


def test_rccl_estimator_allreduce_single_rank(gpu_lib):
    """The RCCL path of pigs_estimators_allreduce (dlopen'ed librccl, communicator of one rank on this
    one-GPU box): sum over one rank is the identity; a second context gets its own communicator."""
    import ctypes as C
    t = load_golden("tables_he4_n64")
    cfg = config_from_golden(t)
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=1) as ctx:
        uid = gpu_lib.comm_unique_id()
        assert len(uid) == 128
        ctx.comm_init_rank(1, 0, uid)
        v = np.arange(390, dtype=float) * 0.5 - 7.0
        out = ctx.estimators_allreduce(v)
        assert same_bits(out, v)
        # single-process form used by a Fortran host with one thread per GPU
        L = gpu_lib.load_library()
        arr = (C.c_void_p * 1)(ctx.h)
        assert L.pigs_comm_init_all(arr, 1) == 0
        assert same_bits(ctx.estimators_allreduce(v), v)
    # without a communicator the call fails loudly
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=1) as ctx:
        with pytest.raises(gpu_lib.PigsError, match="communicator"):
            ctx.estimators_allreduce(np.ones(4))


def test_a_shard_on_a_missing_gpu_fails_at_once(gpu_lib):
    """Multi-GPU readiness on a box with fewer GPUs than shards: a context asked for device n_devices (what the front
    end's shard n_devices+1 would ask for with &gpu n_gpus too large, or a mis-set LOCAL_RANK) is refused by
    pigs_ctx_create with a message -- at once, before any communicator is set up, so no rank can be left waiting in
    ncclCommInitAll / ncclCommInitRank for a peer that never comes."""
    import time
    t = load_golden("tables_he4_n64")
    cfg = config_from_golden(t)
    nd = gpu_lib.device_count()
    t0 = time.perf_counter()
    with pytest.raises(gpu_lib.PigsError, match="device"):
        gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=1, device_id=nd)
    with pytest.raises(gpu_lib.PigsError, match="device"):
        gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=1, device_id=-1)
    assert time.perf_counter() - t0 < 5.0


def test_host_pointer_batch_rate_is_reported(gpu_lib):
    """PCIe-inclusive form (host pointers): only checks that it works at the bench size and prints its
    rate for DESIGN.md; the bench's `value` is always the resident-input rate."""
    import time
    cfg = SystemConfigC3()
    t = load_golden("tables_he4_n256")
    W = 16
    rng = np.random.default_rng(0)
    L = cfg.Lbox[0]
    Paths = rng.uniform(-L / 2, L / 2, (W, cfg.M, cfg.Np, 3))
    w = np.repeat(np.arange(W, dtype=np.int32), cfg.M)
    ib = np.tile(np.arange(cfg.M, dtype=np.int32), W)
    ip = np.ones(W * cfg.M, np.int32)
    xo = Paths[w, ib, 0].copy()
    xn = xo + 0.01
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        a = ctx.delta_action_batch(w, ip, ib, xn, xo)
        t0 = time.perf_counter()
        for _ in range(20):
            a = ctx.delta_action_batch(w, ip, ib, xn, xo)
        dt1 = (time.perf_counter() - t0) / 20
        b = ctx.delta_action_staged(w, ip, ib, xn, xo)
        t0 = time.perf_counter()
        for _ in range(20):
            b = ctx.delta_action_staged(w, ip, ib, xn, xo)
        dt2 = (time.perf_counter() - t0) / 20
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.all((a == b) | np.isnan(a))
    print(f"host-pointer batch of {len(w)} items: {dt1 * 1e6:.0f} us (copying form), {dt2 * 1e6:.0f} us (staged form)")


def SystemConfigC3():
    from pathintegralgroundstate_amd import SystemConfig
    return SystemConfig(dim=3, Np=256, Nb=80)


def test_structure_estimators_vs_oracle(gpu_lib, oracle):
    """K7: g(r) histogram increments are exact (integer counts); S(k) agrees to 1e-12 (device sin/cos)."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    for Np, Nb in ((64, 6), (37, 4)):
        S = System(dim=3, Np=Np, Nb=Nb)
        cfg = SystemConfig(dim=3, Np=Np, Nb=Nb)
        VT, WF = oracle.tables(S)
        Paths = _worldlines(oracle, S, 3, 21, 0.2)
        with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=3) as ctx:
            ctx.upload_all(Paths)
            gr, Sk = ctx.structure_batch(Nb, S.Nbin, S.rbin, 50)
            gr1, Sk1 = ctx.structure_batch(1, S.Nbin, S.rbin, 50, walkers=[2])
        for w in range(3):
            assert same_bits(gr[w], oracle.pair_correlation(S, Paths[w][Nb]))
            want = oracle.structure_factor(S, 50, Paths[w][Nb])
            assert np.all(np.abs(Sk[w] - want) <= 1e-12 * (np.abs(want) + Np))
        assert same_bits(gr1[0], oracle.pair_correlation(S, Paths[2][1]))
        assert gr.sum() > 0


@pytest.mark.parametrize("kw", [dict(dim=1, Np=2, Nb=3, density=0.2), dict(dim=1, Np=5, Nb=2, density=0.3),
                                dict(dim=2, Np=3, Nb=1, density=0.1), dict(dim=3, Np=300, Nb=2),
                                dict(dim=3, Np=65, Nb=3, Nmax=64), dict(dim=2, Np=130, Nb=2, trap=True, a_ho=[2.0, 2.5])])
def test_edge_shapes_all_kernels(gpu_lib, oracle, kw):
    """Ragged / minimal / oversized shapes: Np=2, Np not a multiple of the wave or of the padding,
    Np > 256 (more passes than any variant's fast path), a coarse table, 1D and 2D boxes, a wide trap."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(**kw)
    cfg = SystemConfig(**{k: v for k, v in kw.items()})
    VT, WF = oracle.tables(S)
    rng = np.random.default_rng(S.Np * 7 + S.dim)
    W = 2
    if S.trap:
        Paths = rng.normal(0, 1.5, (W, S.M, S.Np, S.dim))
    else:
        L = np.asarray(S.Lbox[:S.dim])
        Paths = rng.uniform(-0.5, 0.5, (W, S.M, S.Np, S.dim)) * L
    n = 400
    w = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, S.Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    xold = Paths[w, ib, ip - 1].copy()
    xnew = xold + rng.normal(0, 0.3, xold.shape)
    if not S.trap:
        xnew = np.where(xnew > L / 2, xnew - L, xnew)
        xnew = np.where(xnew < -L / 2, xnew + L, xnew)
    want = oracle.delta_action_batch(S, WF, VT, Paths, w, ip, ib, xnew, xold)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        assert same_bits(ctx.download_all(), Paths)
        for variant in (1, 2, 14):
            ctx.set_tuning("k1_variant", variant)
            got = ctx.delta_action_batch(w, ip, ib, xnew, xold)
            assert np.array_equal(np.isnan(got), np.isnan(want)), variant
            fin = np.isfinite(want)
            assert np.all(np.abs(got - want)[fin] <= 1e-9 * np.abs(want[fin]) + 1e-9 * np.max(np.abs(want[fin]), initial=0)), variant
        E, Ec, Ep = ctx.therm_energy_batch()
        le = ctx.local_energy_batch(2 * S.Nb)
        pot = [ctx.PotentialEnergy(1, ib_, True) for ib_ in range(S.M)]
    for k in range(W):
        te = np.array(oracle.therm_energy(S, VT, Paths[k]))
        if np.all(np.isfinite(te)):
            assert _close_rel([E[k], Ec[k], Ep[k]], te, 1e-9)
        lo = np.array(oracle.local_energy(S, WF, VT, Paths[k][2 * S.Nb]))
        if np.all(np.isfinite(lo)):
            assert _close_rel([le[0][k], le[1][k], le[2][k]], lo, 1e-9)
    for ib_ in range(S.M):
        p = np.array(oracle.potential_energy(S, VT, Paths[1][ib_], True))
        if np.all(np.isfinite(p)):
            assert _close_rel(pot[ib_], p, 1e-9)


def test_therm_energy_many_walkers_lds_table_kernel(gpu_lib, oracle):
    """ThermEnergy of 16 walkers at N=64, Nb=80: 2 560 slices in one launch take the persistent LDS-table form of K2
    (>= 8 slices per CU); every walker against the oracle, and against the per-slice kernel through the test hook."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(dim=3, Np=64, Nb=80)
    cfg = SystemConfig(dim=3, Np=64, Nb=80)
    VT, WF = oracle.tables(S)
    W = 16
    Paths = _worldlines(oracle, S, W, 31, 0.1)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        E, Ec, Ep = ctx.therm_energy_batch()
        sub = ctx.therm_energy_batch(walkers=[3, 7])               # 320 slices: the per-slice kernel
        pot_f2 = [ctx.potential_energy_slice(5, ib, True) for ib in (0, 1, 80, 159)]
    for k in range(W):
        te = oracle.therm_energy(S, VT, Paths[k])
        assert _close_rel([E[k], Ec[k], Ep[k]], te), k
    assert _close_rel([sub[0][0], sub[1][0], sub[2][0]], [E[3], Ec[3], Ep[3]])
    assert _close_rel([sub[0][1], sub[1][1], sub[2][1]], [E[7], Ec[7], Ep[7]])
    for (pot, f2), ib in zip(pot_f2, (0, 1, 80, 159)):
        want = oracle.potential_energy(S, VT, Paths[5][ib], True)
        assert _close_rel([pot, f2], want)


@pytest.mark.parametrize("kw,W", [(dict(dim=2, Np=37, Nb=40, density=0.05), 64), (dict(dim=1, Np=11, Nb=24, density=0.2), 48),
                                  (dict(dim=3, Np=200, Nb=16, density=0.365), 80)])
def test_persistent_kernels_on_ragged_shapes(gpu_lib, oracle, kw, W):
    """The persistent LDS-table kernels (K1 pipe2 and its plain-grid twin, K2 LDS form) on shapes they were not tuned for: 1D / 2D,
    particle counts that do not fill a wave or a pass, short chains.  DeltaS against the oracle, ThermEnergy of every
    walker against the oracle."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(**kw)
    cfg = SystemConfig(**kw)
    VT, WF = oracle.tables(S)
    Paths = _worldlines(oracle, S, W, 3, 0.1)
    rng = np.random.default_rng(S.Np)
    n = 24000
    w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, n, 0.1)
    sel = np.arange(0, n, 17)
    want = oracle.delta_action_batch(S, WF, VT, Paths, w[sel], ip[sel], ib[sel], xnew[sel], xold[sel])
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        got = {}
        for v in (13, 12, 0, 2):
            ctx.set_tuning("k1_variant", v)
            got[v] = ctx.delta_action_batch(w, ip, ib, xnew, xold)
        ctx.set_tuning("k1_variant", 0)
        E, Ec, Ep = ctx.therm_energy_batch()
    assert W * 2 * S.Nb >= 8 * 256                                   # enough slices for the LDS form of K2
    sv = np.zeros(len(sel)); sf = np.zeros(len(sel)); su = np.zeros(len(sel))
    for k in range(W):
        m = w[sel] == k
        if m.any():
            sv[m], sf[m], su[m] = term_scales(S, VT, WF, Paths[k], ip[sel][m], ib[sel][m], xnew[sel][m], xold[sel][m])
    tol = delta_s_tolerance(S, sv, sf, su)
    fin = np.isfinite(want)
    for v in (13, 12, 0, 2):
        assert np.array_equal(np.isnan(got[v][sel]), np.isnan(want)), v
        assert np.all(np.abs(got[v][sel] - want)[fin] <= tol[fin]), (v, np.max((np.abs(got[v][sel] - want) / tol)[fin]))
    for k in range(0, W, 7):
        te = oracle.therm_energy(S, VT, Paths[k])
        if np.all(np.isfinite(te)):
            assert _close_rel([E[k], Ec[k], Ep[k]], te), k


@pytest.mark.parametrize("name", ["he4_n64_eq", "he4_n64_rnd", "he4_n64_table_lj", "he4_n64_table_dipolar", "pbc2d_n16",
                                  "ho1d_n2", "trap3d_n8"])
def test_reference_order_kernel_is_bit_identical(gpu_lib, name):
    """BASELINE config 2 taken literally -- "pair-action kernel vs CPU bit-compare": k1_variant = 14 computes every
    per-partner term with the exact-term arithmetic and adds them in the reference's jp order, so Delta S, DeltaPot,
    DeltaF2 and DeltaLogPsi carry the reference's bits (fixtures generated by the compiled reference: equilibrated and
    overlapping 4He worldlines, Lennard-Jones and dipolar tables, 2D PBC, 1D and 3D traps; NaN where the reference
    gives NaN)."""
    d = load_golden(name)
    if "Path" not in d:                                       # table-boundary fixtures: the equilibrated worldline,
        base = load_golden("he4_n64_eq")                      # the reference's Jastrow table, their own VTable
        d = dict(base, **d)
        d["LogWF"] = load_golden("tables_he4_n64")["LogWF"]
    cfg = config_from_golden(d)
    VT, WF = _tables(d)
    n = len(d["ip"])
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=1) as ctx:
        ctx.set_tuning("k1_variant", 14)
        ctx.upload(0, d["Path"])
        w = np.zeros(n, np.int32)
        dS = ctx.delta_action_batch(w, d["ip"], d["ib"], d["xnew"], d["xold"])
        parts = ctx.delta_action_parts(w, d["ip"], d["ib"], d["xnew"], d["xold"])
    assert same_bits(dS, d["DeltaS"]), int(np.sum(dS.view(np.uint64) != d["DeltaS"].view(np.uint64)))
    assert same_bits(parts[:, 0], d["parts"][:, 0])
    odd = d["ib"] % 2 == 1
    assert same_bits(parts[odd, 1], d["parts"][odd, 1])
    end = (d["ib"] == 0) | (d["ib"] == 2 * cfg.Nb)
    assert same_bits(parts[end, 2], d["parts"][end, 2])


def test_delta_s_bits_do_not_depend_on_the_launch(gpu_lib, oracle):
    """An item's Delta S must not depend on how many other items share its launch (a walker's Metropolis chain in the
    host-driven sampler would otherwise depend on the number of walkers in its stage, or on how walkers are sharded
    over GPUs): the same 300 items alone, in launches of 1..64 items, and inside a 24 000-item launch (which takes the
    persistent LDS-table kernel) give identical bits.  Periodic Np <= 256 (pipe2 / grid), periodic Np > 256 and a trap."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    for kw, W in ((dict(dim=3, Np=256, Nb=16), 4), (dict(dim=3, Np=64, Nb=40), 4), (dict(dim=3, Np=300, Nb=8), 2),
                  (dict(dim=2, Np=30, Nb=8, trap=True, a_ho=[1.0, 1.2]), 2)):
        S = System(**kw)
        cfg = SystemConfig(**kw)
        VT, WF = oracle.tables(S)
        rng = np.random.default_rng(S.Np)
        if S.trap:
            Paths = rng.normal(0, 1.5, (W, S.M, S.Np, S.dim))
        else:
            Paths = _worldlines(oracle, S, W, 9, 0.1)
        n = 24000
        w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, n, 0.1) if not S.trap else _trap_batch(rng, S, Paths, n)
        with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
            ctx.upload_all(Paths)
            big = ctx.delta_action_batch(w, ip, ib, xnew, xold)
            sel = np.arange(0, n, 80)
            one_by_one = np.array([ctx.delta_action_batch(w[i:i + 1], ip[i:i + 1], ib[i:i + 1], xnew[i:i + 1], xold[i:i + 1])[0]
                                   for i in sel])
            chunk = ctx.delta_action_batch(w[:64], ip[:64], ib[:64], xnew[:64], xold[:64])
        assert same_bits(one_by_one, big[sel]), kw
        assert same_bits(chunk, big[:64]), kw


def _trap_batch(rng, S, Paths, n):
    W = Paths.shape[0]
    w = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, S.Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    xold = Paths[w, ib, ip - 1].copy()
    xnew = xold + rng.normal(0, 0.3, xold.shape)
    return w, ip, ib, xnew, xold


@pytest.mark.parametrize("kw", [dict(dim=3, Np=64, Nb=12), dict(dim=2, Np=20, Nb=8, trap=True, a_ho=[1.0, 1.3])])
def test_analytic_trial_function_wf_table_false(gpu_lib, oracle, kw):
    """wf_table = F, the reference's DEFAULT (vpi_mod.f90:59): LogPsi evaluated analytically (McMillan, system_mod.f90:38-66)
    in UpdateWf (end beads of K1, every kernel variant incl. the reference-order one, which must stay bit-identical) and
    in LocalEnergy (K4: dudr, d2udr2), against the oracle's wf_table = F branch."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    S = System(wf_table=False, **kw)
    cfg = SystemConfig(wf_table=False, **kw)
    VT, WF = oracle.tables(S)
    rng = np.random.default_rng(5)
    W = 2
    if S.trap:
        Paths = rng.normal(0, 1.2, (W, S.M, S.Np, S.dim))
        w, ip, ib, xnew, xold = _trap_batch(rng, S, Paths, 26000)
    else:
        Paths = _worldlines(oracle, S, W, 5, 0.1)
        w, ip, ib, xnew, xold = _random_batch(rng, S, Paths, 26000, 0.1)
    ib[::3] = 0
    ib[1::3] = 2 * S.Nb                                        # two thirds end beads
    xold = Paths[w, ib, ip - 1].copy()
    sel = np.arange(0, len(w), 13)
    want = oracle.delta_action_batch(S, WF, VT, Paths, w[sel], ip[sel], ib[sel], xnew[sel], xold[sel])
    fin = np.isfinite(want)
    with gpu_lib.PigsContext(cfg, VT, None, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        for v in (1, 2, 7, 13, 0, 14):
            ctx.set_tuning("k1_variant", v)
            got = ctx.delta_action_batch(w, ip, ib, xnew, xold)[sel] if v in (0, 12) else \
                ctx.delta_action_batch(w[sel], ip[sel], ib[sel], xnew[sel], xold[sel])
            assert np.array_equal(np.isnan(got), np.isnan(want)), v
            if v == 14:
                assert same_bits(got, want)
            else:
                assert np.all(np.abs(got - want)[fin] <= 1e-10 * np.abs(want[fin]) + 1e-10 * np.max(np.abs(want[fin]))), v
        for slot in (0, 2 * S.Nb):
            E, K, Pp = ctx.local_energy_batch(slot)
            for k in range(W):
                lo = np.array(oracle.local_energy(S, WF, VT, Paths[k][slot]))
                if np.all(np.isfinite(lo)):
                    assert _close_rel([E[k], K[k], Pp[k]], lo), (slot, k)


def test_commit_list_is_a_sequence_last_value_wins(gpu_lib):
    """A commit list is the caller's program order of `Path(:,ip,ib) = x`: a bead that appears several times must end
    up with the LAST value (the kernel writes entries in parallel; earlier duplicates are masked on the host).  The
    host-driven sampler queues a worm's bead Nb twice before one flush (re-selection of xend between half-chain moves)."""
    t = load_golden("tables_he4_n64")
    cfg = config_from_golden(t)
    rng = np.random.default_rng(8)
    W = 5
    Paths = rng.uniform(-1, 1, (W, cfg.M, cfg.Np, 3))
    with gpu_lib.PigsContext(cfg, t["VTable"], t["LogWF"], n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        n = 4000
        w = rng.integers(0, W, n).astype(np.int32)
        ip = rng.integers(1, 4, n).astype(np.int32)          # few particles, few beads: many duplicates
        ib = rng.integers(0, 6, n).astype(np.int32)
        x = rng.normal(0, 1, (n, 3))
        ctx.commit_beads(w, ip, ib, x)
        got = ctx.download_all()
    want = Paths.copy()
    for i in range(n):
        want[w[i], ib[i], ip[i] - 1] = x[i]
    assert same_bits(got, want)


def test_device_log_is_the_host_libm_log_bit_for_bit(gpu_lib):
    """pigs_selftest_log: the log() inside the device sampler's Box-Muller transform (csrc/pigs_log_host.h, glibc's
    algorithm + constants with the x86-64 FMA build's fusions) against THIS host's libm on 2^30 arguments of the
    sampler's domain (stream uniforms, polar radii, random mantissas, the near-one branch): zero mismatches, which is
    what makes the device-resident sampler's worldlines bit-identical to the reference's."""
    from pathintegralgroundstate_amd import SystemConfig
    cfg = SystemConfig(dim=3, Np=8, Nb=2)
    VT, WF = gpu_lib.build_tables(cfg)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=1) as ctx:
        bad, x = ctx.selftest_log(1 << 30, seed=20261004)
    assert bad == 0, (bad, float(x).hex())


@pytest.mark.parametrize("pot", ["dipolar", "lj"])
def test_infinite_table_head_gives_the_references_infinities(gpu_lib, oracle, pot):
    """Singular potentials tabulate V(0) = +Inf (vpi_mod.f90:96-110 fills cell 1 at r = 0).  A pair closer than 3 dr then has
    an infinite V (r < 2 dr) or an infinite dV/dr (the derivative stencil reaches cell 1 up to r < 3 dr), and the reference's
    Delta S is -Inf for a move AWAY from such a pair (accepted without a uniform), +Inf towards one (rejected), NaN between
    two such places.  Rounds 1-2 returned NaN for all of them -- the exact short divisions and the one-product interpolation
    turn Inf into Inf - Inf -- so a particle that started within 3 dr of another could never move again (round 3's sampler
    fuzz: 1D random starts).  Every K1 variant now returns the reference's value, infinities and NaNs included."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    cfg = SystemConfig(dim=1, Np=21, Nb=12, density=0.2, dt=5e-3, Rm=1.1, Nmax=4000)
    S = System(dim=1, Np=21, Nb=12, density=0.2, dt=5e-3, Rm=1.1, Nmax=4000)
    VT, WF = gpu_lib.build_tables(cfg, pot)
    assert np.isinf(VT[1])
    L, dr = S.Lbox[0], S.dr
    P = np.zeros((S.M, S.Np, 1))
    P[:, :, 0] = (np.arange(S.Np) - 10) * 4.9                       # a regular line ...
    for j, gap in enumerate((0.5, 1.5, 2.5, 3.5)):                  # ... with four close pairs: inside and just outside the zone
        P[:, 2 * j + 1, 0] = P[:, 2 * j, 0] + gap * dr
    ip, ib, xn, xo = [], [], [], []
    for j in range(4):
        for b in (0, 1, 2, 2 * S.Nb):                                # end, odd, even, end
            p = 2 * j + 2                                            # 1-based index of the second particle of pair j
            for target in (1.0, 0.7 * dr + P[b, 2 * j, 0] - P[b, 2 * j + 1, 0]):     # away from the pair / onto another place inside the zone
                ip.append(p); ib.append(b); xo.append(P[b, p - 1].copy()); xn.append(P[b, p - 1] + target)
    # and a far particle moved INTO the zone of pair 0
    for b in (0, 1, 2):
        ip.append(21); ib.append(b); xo.append(P[b, 20].copy()); xn.append(P[b, 0] + 0.3 * dr)
    ip, ib = np.array(ip, np.int32), np.array(ib, np.int32)
    xn, xo = np.array(xn), np.array(xo)
    w = np.zeros(len(ip), np.int32)
    want = oracle.delta_action_batch(S, WF, VT, P[None], w, ip, ib, xn, xo)
    assert np.isneginf(want).any() and np.isposinf(want).any()
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=1) as ctx:
        ctx.upload_all(P[None])
        for v in (0, 1, 2, 7, 8, 12, 13, 14):
            ctx.set_tuning("k1_variant", v)
            got = ctx.delta_action_batch(w, ip, ib, xn, xo)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (v, got, want)
            assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want)), (v, got, want)
            fin = np.isfinite(want)
            assert np.all(np.abs(got[fin] - want[fin]) <= 1e-9 * np.abs(want[fin]) + 1e-12), v


@pytest.mark.parametrize("dim,Np", [(3, 2), (3, 4), (2, 3), (1, 5)])
def test_beads_far_outside_the_box_fold_once_like_the_reference(gpu_lib, oracle, dim, Np):
    """pbc_mod.f90:40-41 folds a separation ONCE (two compares), and BoundaryConditions (pbc_mod.f90:20-21) folds a proposal
    once: in a small box the long free segment of a head / tail move (Nlev' up to 7 -> 128 links) puts beads several box
    lengths away, and a separation beyond 1.5 L stays outside the cutoff in the reference.  The short-arithmetic minimum
    image of rounds 1-2 (v - L rint(v/L)) folded it all the way and found a pair inside the cutoff that the reference
    does not see (round 3's WIDE sampler fuzz, Np = 2, Nlev = 7: one decision in 40 000 items).  Every K1 variant against
    the oracle with resident beads up to 3 L and proposals up to 5 L from the origin."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    kw = dict(dim=dim, Np=Np, Nb=16, density=0.2, dt=0.03)
    S = System(**kw)
    cfg = SystemConfig(**kw)
    VT, WF = oracle.tables(S)
    L = np.asarray(S.Lbox[:dim])
    rng = np.random.default_rng(100 * dim + Np)
    W, n = 3, 3000
    Paths = rng.uniform(-3.0, 3.0, (W, S.M, Np, dim)) * L
    Paths[:, ::3] = rng.uniform(-0.5, 0.5, Paths[:, ::3].shape) * L          # a third of the slices inside the box
    w = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    xold = Paths[w, ib, ip - 1].copy()
    xnew = rng.uniform(-5.0, 5.0, xold.shape) * L
    xnew[::2] = rng.uniform(-1.0, 1.0, xnew[::2].shape) * L
    want = oracle.delta_action_batch(S, WF, VT, Paths, w, ip, ib, xnew, xold)
    fin = np.isfinite(want)
    assert fin.sum() > n // 2 and np.count_nonzero(want[fin]) > n // 8
    u = rng.uniform(size=n)
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        for v in (0, 1, 2, 7, 8, 12, 13, 14):
            ctx.set_tuning("k1_variant", v)
            got = ctx.delta_action_batch(w, ip, ib, xnew, xold)
            assert np.array_equal(np.isnan(got), np.isnan(want)), v
            assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), v
            bad = np.abs(got - want)[fin] > 1e-9 * np.abs(want[fin]) + 1e-10
            assert not bad.any(), (v, int(bad.sum()), got[fin][bad][:3], want[fin][bad][:3])
            with np.errstate(over="ignore", invalid="ignore"):
                assert np.array_equal(np.exp(-got) >= u, np.exp(-want) >= u), v


@pytest.mark.parametrize("W", [2, 96])
def test_estimators_near_an_infinite_table_head(gpu_lib, oracle, W):
    """The estimators on worldlines with pairs inside the +Inf head of a singular table (see
    test_infinite_table_head_gives_the_references_infinities): PotentialEnergy / ThermEnergy / LocalEnergy return what the
    reference's arithmetic returns -- +-Inf where it does, NaN where it does (Inf - Inf), the finite values elsewhere.
    W = 2: the per-slice kernels; W = 96 (2 304 slices): the persistent LDS-table form of K2, whose one-product interpolation
    F0 + f (F1 - F0) turned the head into NaN until round 3."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    kw = dict(dim=1, Np=21, Nb=12, density=0.2, dt=5e-3, Rm=1.1, Nmax=4000)
    cfg, S = SystemConfig(**kw), System(**kw)
    VT, WF = gpu_lib.build_tables(cfg, "dipolar")
    assert np.isinf(VT[1])
    dr = S.dr
    rng = np.random.default_rng(5)
    Paths = np.zeros((W, S.M, S.Np, 1))
    for w in range(W):
        Paths[w, :, :, 0] = (np.arange(S.Np) - 10) * 4.9 + rng.normal(0, 0.05, (S.M, S.Np))
        if w % 2 == 0:                                                   # every other walker: close pairs on some slices
            for j, gap in enumerate((0.5, 1.5, 2.5, 3.5)):
                for b in (0, 3, 4, 2 * S.Nb, S.Nb):
                    Paths[w, (b + j) % S.M, 2 * j + 1, 0] = Paths[w, (b + j) % S.M, 2 * j, 0] + gap * dr
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        E, Ec, Ep = ctx.therm_energy_batch()
        le0 = ctx.local_energy_batch(0)
    seen_inf = False
    for w in list(range(min(W, 6))) + ([W - 2, W - 1] if W > 6 else []):
        want = np.array(oracle.therm_energy(S, VT, Paths[w]))
        got = np.array([E[w], Ec[w], Ep[w]])
        seen_inf |= bool(np.any(~np.isfinite(want)))
        assert np.array_equal(np.isnan(got), np.isnan(want)), (w, got, want)
        assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want)), (w, got, want)
        fin = np.isfinite(want)
        assert np.all(np.abs(got[fin] - want[fin]) <= 1e-10 * np.abs(want[fin])), (w, got, want)
        lo = np.array(oracle.local_energy(S, WF, VT, Paths[w][0]))
        lg = np.array([le0[0][w], le0[1][w], le0[2][w]])
        assert np.array_equal(np.isnan(lg), np.isnan(lo)) and np.array_equal(np.isposinf(lg), np.isposinf(lo)), (w, lg, lo)
        fin = np.isfinite(lo)
        assert np.all(np.abs(lg[fin] - lo[fin]) <= 1e-10 * (np.abs(lo[1]) + np.abs(lo[2]))), (w, lg, lo)
    assert seen_inf


def test_trap_pairs_beyond_the_table_are_clamped_not_read_out_of_bounds(gpu_lib, oracle):
    """A trapped system has no cutoff (vpi_mod.f90:2699-2722 evaluate every pair) and its tables end at rcut = 30 a_ho
    (vpi.f90:84-92): for a pair farther apart the reference reads VTable / LogWF out of bounds -- undefined.  The product clamps
    the cell index to the last cell (lerp_setup / flerp_setup: an identity wherever the reference is defined) instead of reading
    whatever lies behind the table; the oracle does the same, so that the two can be compared there at all (round 3's fuzz with
    TranslateChain shifts of 9 a_ho).  Delta S on every trap variant, ThermEnergy and LocalEnergy."""
    from oracle.pyoracle import System
    from pathintegralgroundstate_amd import SystemConfig
    kw = dict(dim=1, Np=6, Nb=4, density=0.05, dt=5e-3, Rm=1.4, Nmax=4000, trap=True, a_ho=[1.0])
    cfg, S = SystemConfig(**kw), System(**kw)
    VT, WF = oracle.tables(S)
    assert abs(S.rcut - 30.0) < 1e-12
    rng = np.random.default_rng(67)
    W, n = 2, 400
    Paths = (np.arange(S.Np) - 2.5)[None, None, :, None] * 17.0 + rng.uniform(-3.0, 3.0, (W, S.M, S.Np, 1))   # spans 85 a_ho: pairs up to 2.8 rcut apart
    w = rng.integers(0, W, n).astype(np.int32)
    ip = rng.integers(1, S.Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    xold = Paths[w, ib, ip - 1].copy()
    xnew = rng.uniform(-60.0, 60.0, xold.shape)
    want = oracle.delta_action_batch(S, WF, VT, Paths, w, ip, ib, xnew, xold)
    fin = np.isfinite(want)
    assert fin.sum() > n // 2
    with gpu_lib.PigsContext(cfg, VT, WF, n_walkers=W) as ctx:
        ctx.upload_all(Paths)
        for v in (0, 1, 2, 14):
            ctx.set_tuning("k1_variant", v)
            got = ctx.delta_action_batch(w, ip, ib, xnew, xold)
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), v
            assert np.all(np.abs(got - want)[fin] <= 1e-10 * np.abs(want[fin]) + 1e-9), (v, np.max(np.abs(got - want)[fin]))
        E, Ec, Ep = ctx.therm_energy_batch()
        le = ctx.local_energy_batch(0)
    for k in range(W):
        te = np.array(oracle.therm_energy(S, VT, Paths[k]))
        if np.all(np.isfinite(te)):
            assert np.all(np.abs(np.array([E[k], Ec[k], Ep[k]]) - te) <= 1e-10 * np.abs(te).max()), k
        lo = np.array(oracle.local_energy(S, WF, VT, Paths[k][0]))
        if np.all(np.isfinite(lo)):
            assert np.all(np.abs(np.array([le[0][k], le[1][k], le[2][k]]) - lo) <= 1e-10 * (abs(lo[1]) + abs(lo[2]))), k
