"""The C restatement against the unmodified reference itself (oracle/_ref/libvpiref.so, built
from /root/reference by oracle/Makefile) on fresh seeded inputs: bit-exact.  Skipped where the
reference build is absent (then tests/test_oracle_golden.py carries the pin)."""
import numpy as np
import pytest

from helpers import same_bits
from oracle.pyoracle import System


def _wrap(x, L):
    x = np.where(x > L / 2, x - L, x)
    return np.where(x < -L / 2, x + L, x)


@pytest.mark.parametrize("kw", [
    dict(dim=3, Np=64, Nb=40),
    dict(dim=3, Np=37, Nb=5),                       # ragged particle count
    dict(dim=2, Np=20, Nb=4, density=0.3),
    dict(dim=1, Np=5, Nb=3, density=0.4),
    dict(dim=3, Np=9, Nb=4, trap=True, a_ho=[0.9, 1.1, 1.4]),
    dict(dim=1, Np=2, Nb=10, trap=True, a_ho=[1.0]),
])
def test_hot_path_vs_reference(oracle, ref, kw):
    S = System(**kw)
    VT, WF = ref.tables(S)
    VTo, WFo = oracle.tables(S)
    assert same_bits(VT, VTo) and same_bits(WF, WFo)
    ref.set_system(S)
    seed = 4242 + S.Np
    P, _ = ref.init(seed)
    Po, _ = oracle.init_path(S, seed)
    assert same_bits(P, Po)
    rng = np.random.default_rng(S.Np * 1000 + S.dim)
    P = P + rng.normal(0, 0.3, P.shape)
    if not S.trap:
        P = _wrap(P, S.Lbox[:S.dim])
    for _ in range(300):
        ip = int(rng.integers(1, S.Np + 1))
        ib = int(rng.choice([0, 2 * S.Nb, int(rng.integers(0, S.M))]))
        xold = P[ib, ip - 1].copy()
        xnew = xold + rng.normal(0, 0.4, S.dim)
        if not S.trap:
            xnew = _wrap(xnew, S.Lbox[:S.dim])
        a = ref.update_action(WF, VT, P, ip, ib, xnew, xold)
        b = oracle.update_action(S, WF, VT, P, ip, ib, xnew, xold)
        assert same_bits([a], [b]), (ip, ib, a, b)
    for ib in range(S.M):
        for w in (False, True):
            assert same_bits(ref.potential_energy(VT, P[ib], w), oracle.potential_energy(S, VT, P[ib], w))
    for ib in (0, 2 * S.Nb):
        assert same_bits(ref.local_energy(WF, VT, P[ib]), oracle.local_energy(S, WF, VT, P[ib]))
    assert same_bits(ref.therm_energy(VT, P), oracle.therm_energy(S, VT, P))


def test_structural_estimators_vs_reference(oracle, ref):
    S = System(dim=3, Np=64, Nb=4, Npw=2)
    ref.set_system(S)
    P, _ = ref.init(11)
    assert same_bits(ref.pair_correlation(P[S.Nb]), oracle.pair_correlation(S, P[S.Nb]))
    assert same_bits(ref.structure_factor(50, P[S.Nb]), oracle.structure_factor(S, 50, P[S.Nb]))
    rng = np.random.default_rng(5)
    for _ in range(50):
        xe = rng.uniform(-S.Lbox[0] / 2, S.Lbox[0] / 2, (2, 3))
        assert same_bits(ref.obdm(xe), oracle.obdm(S, xe))


def test_box_and_primitives_vs_reference(oracle, ref):
    for Np, dim, rho in ((64, 3, 0.365), (256, 3, 0.365), (20, 2, 0.3), (7, 1, 0.11)):
        assert ref.box_length(Np, dim, rho) == oracle.L.po_box_length(Np, dim, rho)
        S = System(dim=dim, Np=Np, Nb=2, density=rho)
        assert S.Lbox[0] == ref.box_length(Np, dim, rho)
    S = System(dim=3, Np=64, Nb=40)
    ref.set_system(S)
    rng = np.random.default_rng(3)
    for _ in range(500):
        x = rng.uniform(-1.5 * S.Lbox[0], 1.5 * S.Lbox[0], 3)
        a, r2a = ref.minimum_image(x)
        b, r2b = oracle.minimum_image(S, x)
        assert same_bits(a, b) and r2a == r2b
    for opt in (0, 1):
        for ib in (0, 1, 2, 40, 79, 80):
            for pot, f2 in ((1.7, -3.3), (-2e5, 9e9)):
                assert ref.green_function(opt, ib, 5e-3, pot, f2) == \
                    oracle.green_function(opt, ib, S.Nb, 5e-3, pot, f2)


def test_rng_vs_reference(oracle, ref):
    for seed in (1982, 1, 4357, 2**31 - 1):
        ref.sgrnd(seed)
        g = oracle.rng(seed)
        assert all(ref.grnd() == oracle.grnd(g) for _ in range(1500))
        assert all(ref.rangauss(0.7, 0.1) == oracle.rangauss(g, 0.7, 0.1) for _ in range(200))
        mti, mt = ref.rng_get_state()
        assert mti == g.mti and np.array_equal(mt, np.array(g.mt[:], np.uint32))
