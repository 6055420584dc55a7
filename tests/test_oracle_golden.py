"""The C restatement (oracle/pigs_oracle.c) against the committed golden vectors that
tests/golden/make_golden.py generated from the unmodified reference: bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, system_from_golden
from helpers import same_bits

CASES = ["he4_n64_eq", "he4_n64_rnd", "ho1d_n2", "trap3d_n8", "pbc2d_n16"]


def _tables(name, d):
    if "VTable" in d:
        return d["VTable"], d["LogWF"]
    t = load_golden("tables_he4_n64")
    return t["VTable"], t["LogWF"]


@pytest.mark.parametrize("name", CASES)
def test_update_action_bit_exact(oracle, name):
    d = load_golden(name)
    S = system_from_golden(d)
    VT, WF = _tables(name, d)
    n = len(d["ip"])
    got = np.array([oracle.update_action(S, WF, VT, d["Path"], int(d["ip"][i]), int(d["ib"][i]),
                                         d["xnew"][i], d["xold"][i]) for i in range(n)])
    assert same_bits(got, d["DeltaS"])
    # batched form == per-call form
    W = np.zeros(n, np.int32)
    got_b = oracle.delta_action_batch(S, WF, VT, d["Path"][None], W, d["ip"], d["ib"], d["xnew"], d["xold"])
    assert same_bits(got_b, d["DeltaS"])


@pytest.mark.parametrize("name", CASES)
def test_update_pot_wf_parts_bit_exact(oracle, name):
    d = load_golden(name)
    S = system_from_golden(d)
    VT, WF = _tables(name, d)
    for i in range(0, len(d["ip"]), 3):
        ip, ib = int(d["ip"][i]), int(d["ib"][i])
        odd = ib % 2 == 1
        dp, df = oracle.update_pot(S, VT, ip, d["Path"][ib], d["xnew"][i], d["xold"][i], odd)
        assert same_bits([dp, df], d["parts"][i, :2])
        if ib in (0, 2 * S.Nb):
            dw = oracle.update_wf(S, WF, ip, d["Path"][ib], d["xnew"][i], d["xold"][i])
            assert same_bits([dw], d["parts"][i, 2:])


@pytest.mark.parametrize("name", CASES)
def test_energies_bit_exact(oracle, name):
    d = load_golden(name)
    S = system_from_golden(d)
    VT, WF = _tables(name, d)
    P = d["Path"]
    pe = np.array([oracle.potential_energy(S, VT, P[ib], True) for ib in range(S.M)])
    pe0 = np.array([oracle.potential_energy(S, VT, P[ib], False)[0] for ib in range(S.M)])
    assert same_bits(pe, d["pot_f2"]) and same_bits(pe0, d["pot_only"])
    le = np.array([oracle.local_energy(S, WF, VT, P[0]), oracle.local_energy(S, WF, VT, P[2 * S.Nb])])
    assert same_bits(le, d["local"])
    assert same_bits(np.array(oracle.therm_energy(S, VT, P)), d["therm"])


@pytest.mark.parametrize("tag", ["lj", "dipolar"])
def test_external_table_boundary(oracle, tag):
    """LJ / dipolar potentials are not active code in the reference: parity is
    pinned at the table boundary -- same externally filled VTable through both."""
    base = load_golden("he4_n64_eq")
    d = load_golden(f"he4_n64_table_{tag}")
    S = system_from_golden(base)
    WF = load_golden("tables_he4_n64")["LogWF"]
    VT, P = d["VTable"], base["Path"]
    got = np.array([oracle.update_action(S, WF, VT, P, int(d["ip"][i]), int(d["ib"][i]),
                                         d["xnew"][i], d["xold"][i]) for i in range(len(d["ip"]))])
    assert same_bits(got, d["DeltaS"])
    assert same_bits(np.array(oracle.therm_energy(S, VT, P)), d["therm"])
    assert same_bits(np.array([oracle.potential_energy(S, VT, P[ib], True) for ib in range(S.M)]), d["pot_f2"])


def test_tables_bit_exact(oracle):
    for name in ("tables_he4_n64", "tables_he4_n256"):
        t = load_golden(name)
        S = system_from_golden(t)
        VT, WF = oracle.tables(S)
        assert same_bits(VT, t["VTable"]) and same_bits(WF, t["LogWF"])
        # quirk Q4: entry 1 is V(0)=NaN / u(0)=-inf, ghost cells mirror entries 2 and Nmax
        assert np.isnan(VT[1]) and WF[1] == -np.inf
        assert VT[0] == VT[2] and VT[S.Nmax + 1] == VT[S.Nmax]


def test_primitives_bit_exact(oracle):
    p = load_golden("primitives_n64")
    t = load_golden("tables_he4_n64")
    S = system_from_golden(t)
    for o in (0, 1, 2):
        got = np.array([oracle.interpolate(o, S.Nmax, S.dr, t["VTable"], x) for x in p["x"]])
        assert same_bits(got, p[f"interp{o}"])
    gf = np.array([[oracle.green_function(o, ib, S.Nb, 5e-3, 1.2345678901234, -9.87654321)
                    for ib in (0, 1, 2, 39, 40, 79, 80)] for o in (0, 1)])
    assert same_bits(gf, p["green"])


def test_rng_bit_exact(oracle):
    r = load_golden("rng_seed1982")
    g = oracle.rng(1982)
    u = np.array([oracle.grnd(g) for _ in range(len(r["grnd"]))])
    assert same_bits(u, r["grnd"])
    assert g.mti == int(r["mti_after"]) and np.array_equal(np.array(g.mt[:], np.uint32), r["mt_after"])
    gs = np.array([oracle.rangauss(g, 1.0, 0.0) for _ in range(len(r["rangauss"]))])
    assert same_bits(gs, r["rangauss"])
    assert g.mti == int(r["mti_after_gauss"]) and np.array_equal(np.array(g.mt[:], np.uint32), r["mt_after_gauss"])


def test_mixed_estimator_is_ill_conditioned_at_one_ulp(oracle):
    """Why both samplers have to deliver BIT-identical worldlines (helpers.MIXED_TOL = 1e-10).  LocalEnergy takes d2u/dr2
    as a second difference of the linear interpolant divided by dr^2 (interpolate.f90:36-42) with dr = rcut/9999: rounding
    is amplified by ~1e7.  Moving every coordinate of a slice by ONE ulp moves the reference's own E and Kin by several
    1e-10 of |K|+|V| on the worm-busy run's worldline (6e-11 on the equilibrated N=64 one; Pot stays at 1e-14), so a
    sampler whose coordinates differ from the reference's in the last bit (the device-resident one of rounds 1-2: device
    log() in Box-Muller) cannot match the mixed estimator to 1e-10 step by step, whatever it computes; one whose
    worldline is bit-identical (the host-driven one; the device-resident one since its log() is the host libm's,
    csrc/pigs_log_host.h) does."""
    import os
    from conftest import GOLDEN
    from helpers import MIXED_TOL
    from oracle.pyoracle import System
    S = System(dim=3, Np=16, Nb=8, density=0.365, dt=2e-2)
    VT, WF = oracle.tables(S)
    P = np.load(os.path.join(GOLDEN, "vpi_runs", "he4_wormbusy_s7", "final_worldline.npz"))["Path"]
    rng = np.random.default_rng(3)
    worst = np.zeros(3)
    for ib in (0, 2 * S.Nb):
        R = P[ib]
        e0 = np.array(oracle.local_energy(S, WF, VT, R))
        scale = abs(e0[1]) + abs(e0[2])
        for _ in range(100):
            R2 = np.nextafter(R, R + rng.choice([-1.0, 1.0], R.shape))
            worst = np.maximum(worst, np.abs(np.array(oracle.local_energy(S, WF, VT, R2)) - e0) / scale)
    assert worst[2] < 1e-13                                   # the potential energy is well conditioned
    assert MIXED_TOL < worst[0] < 2e-9 and MIXED_TOL < worst[1] < 2e-9, worst      # one ulp already breaks the contract


def test_interpolate_beyond_the_table_is_clamped_not_out_of_bounds(oracle):
    """interpolate.f90 reads F(ix+1) for any x; beyond the table's last cell (a trapped system's pair farther apart than
    rcut: no cutoff there) that is out of bounds -- undefined in the reference.  The oracle clamps the cell index as the
    product does: inside the table nothing changes, beyond it the last cell's values are used with the
    unclamped offsets, and nothing behind the array is read (the sentinel stays unread)."""
    N, dx = 50, 0.1
    rng = np.random.default_rng(16)
    F = rng.normal(size=N + 2)
    G = np.concatenate([F, np.full(64, np.nan)])                 # a table followed by poison
    for opt in (0, 1, 2):
        for x in (0.31, 2.5, 4.89, N * dx - 1e-9):               # inside: same value with or without what lies behind the table
            assert oracle.interpolate(opt, N, dx, F, x) == oracle.interpolate(opt, N, dx, G, x)
        for x in (N * dx + 0.03, 7.77, 123.456):                 # beyond: finite, from the last cell
            v = oracle.interpolate(opt, N, dx, G, x)
            assert np.isfinite(v)
            ix = int(x / dx) + 1
            a1 = x - (ix - 1) * dx
            a2 = dx - a1
            i = N
            if opt == 0:
                want = (a1 * F[i] + a2 * F[i - 1]) / dx
            else:
                fb, fc, fa = (a1 * F[i - 1] + a2 * F[i - 2]) / dx, (a1 * F[i] + a2 * F[i - 1]) / dx, (a1 * F[i + 1] + a2 * F[i]) / dx
                want = 0.5 * (fa - fb) / dx if opt == 1 else (fa - 2.0 * fc + fb) / (dx * dx)
            assert v == want, (opt, x, v, want)
