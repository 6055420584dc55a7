"""Host logic of the Fortran sampler (pathintegralgroundstate_amd/host) against the reference's
own movers on identical MT19937 state: worldline, RNG state, worm state and acceptance must come
out bit-identical after every move.  The sampler evaluates Delta S through the C ABI; here (no
GPU) that ABI is served by tests/shim (CPU oracle, test infrastructure), so what is tested is the
host side: random streams, proposal arithmetic incl. the single-precision quirk Q7, accept /
restore / commit bookkeeping.  On the GPU box tests/test_gpu_host.py repeats the end-to-end run
against the real library."""
import numpy as np
import pytest

from helpers import same_bits
from hostlib import HostSampler, build_cpu_host
from oracle.pyoracle import System


@pytest.fixture(scope="module")
def libs():
    return build_cpu_host()


def test_rng_streams_match_reference_fixture(libs):
    import ctypes as C
    from conftest import load_golden
    H = C.CDLL(libs[1])
    r = load_golden("rng_seed1982")
    dp = C.POINTER(C.c_double)
    H.hs_uniform_stream.argtypes = [C.c_int, C.c_int, dp]
    H.hs_gauss_stream.argtypes = [C.c_int, C.c_int, dp]
    u = np.zeros(len(r["grnd"]))
    H.hs_uniform_stream(1982, len(u), u.ctypes.data_as(dp))
    assert same_bits(u, r["grnd"])


def _setup(ref, S, seed, sweeps=1):
    VT, WF = ref.tables(S)
    ref.set_system(S)
    P, xend = ref.init(seed)
    delta = 0.12 / S.density ** (1.0 / 3.0) if not S.trap else 0.12
    for _ in range(sweeps):                       # spread the beads with the reference itself
        for ip in range(1, S.Np + 1):
            ref.translate_chain(delta, WF, VT, ip, P)
            ref.diag_move("Bisection", WF, VT, 3, ip, P)
            ref.diag_move("MoveHeadBisection", WF, VT, 3, ip, P)
            ref.diag_move("MoveTailBisection", WF, VT, 3, ip, P)
    return VT, WF, P, xend, delta


def _sync_rng(ref, hs, w=0):
    mti, mt = ref.rng_get_state()
    hs.set_rng(w, mti, mt)


def _same_rng(ref, hs, w=0):
    a, b = ref.rng_get_state(), hs.get_rng(w)
    return a[0] == b[0] and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("kw", [dict(dim=3, Np=16, Nb=20, density=0.365),
                                dict(dim=2, Np=9, Nb=12, density=0.25),
                                dict(dim=3, Np=6, Nb=10, trap=True, a_ho=[1.0, 1.2, 0.9])])
def test_diagonal_movers_bit_exact(libs, ref, kw):
    S = System(**kw)
    VT, WF, P, xend, delta = _setup(ref, S, 77)
    hs = HostSampler(S, VT, WF, W=1, backend=libs[0], hostlib=libs[1])
    try:
        hs.set_path(0, P)
        hs.upload()
        _sync_rng(ref, hs)
        seq = [("TranslateChain", 0), ("Bisection", 4), ("MoveHeadBisection", 4), ("MoveTailBisection", 4),
               ("Staging", 8), ("MoveHead", 8), ("MoveTail", 8), ("Bisection", 3), ("MoveHeadBisection", 2)]
        nacc = 0
        for rep in range(6):
            for ip in range(1, S.Np + 1):
                for name, par in seq:
                    if name == "TranslateChain":
                        a = ref.translate_chain(delta, WF, VT, ip, P)
                        b = hs.move(name, ip, rpar=delta)[0][0]
                    else:
                        a = ref.diag_move(name, WF, VT, par, ip, P)
                        b = hs.move(name, ip, i1=par)[0][0]
                    assert a == b, (rep, ip, name)
                    nacc += a
                    assert _same_rng(ref, hs), (rep, ip, name)
            assert same_bits(hs.get_path(0), P), rep
            assert same_bits(hs.device_paths()[0], P), rep        # commits reached the "device"
        assert nacc > 20                                          # both branches were exercised
    finally:
        hs.close()


def test_worm_movers_bit_exact(libs, ref):
    S = System(dim=3, Np=12, Nb=16, density=0.365, CWorm=0.8)
    VT, WF, P, xend, delta = _setup(ref, S, 5, sweeps=6)
    hs = HostSampler(S, VT, WF, W=1, backend=libs[0], hostlib=libs[1])
    Lstag = 8
    try:
        hs.set_path(0, P)
        hs.upload()
        _sync_rng(ref, hs)
        isopen, iworm = False, 0
        hs.set_worm(0, isopen, iworm, xend)
        n_open = n_close = n_swap = 0
        rng = np.random.default_rng(1)
        for step in range(1500):
            if not isopen:
                iworm = int(rng.integers(1, S.Np + 1))
                isopen_r, a = ref.open_chain(WF, VT, Lstag, iworm, P, xend, isopen)
                hs.set_worm(0, isopen, iworm, hs.get_worm(0)[2])
                b = hs.move("OpenChain", iworm, i1=Lstag)[0][0]
                assert a == b, step
                isopen = isopen_r
                n_open += a
            else:
                for half in (1, 2):
                    a = ref.half_move("TranslateHalfChain", half, delta, WF, VT, Lstag, iworm, P, xend)
                    b = hs.move("TranslateHalfChain", iworm, i2=half, rpar=delta)[0][0]
                    assert a == b, (step, "thc", half)
                    for name in ("MoveHeadHalfChain", "MoveTailHalfChain", "StagingHalfChain"):
                        a = ref.half_move(name, half, delta, WF, VT, Lstag, iworm, P, xend)
                        b = hs.move(name, iworm, i1=Lstag, i2=half)[0][0]
                        assert a == b, (step, name, half)
                        assert _same_rng(ref, hs), (step, name, half)
                for _ in range(4):
                    iw2, ik, swapped, a = ref.swap(WF, VT, Lstag, iworm, P, xend)
                    acc, par, swp = hs.move("Swap", iworm, i1=Lstag)
                    assert a == acc[0] and bool(swp[0]) == swapped and (not swapped or par[0] == ik), step
                    n_swap += a
                    assert same_bits(hs.get_path(0), P) and same_bits(hs.get_worm(0)[2], xend), step
                if step % 9 == 0:
                    isopen_r, a = ref.close_chain(WF, VT, Lstag, iworm, P, xend, isopen)
                    b = hs.move("CloseChain", iworm, i1=Lstag)[0][0]
                    assert a == b, step
                    isopen = isopen_r
                    n_close += a
            o, iw, xe = hs.get_worm(0)
            assert o == isopen and same_bits(xe, xend), step
            assert _same_rng(ref, hs), step
            assert same_bits(hs.get_path(0), P), step
            # what other particles see on the device is the host mirror, bead for bead
            assert same_bits(hs.device_paths()[0], P), step
        assert n_open > 2 and n_close > 1 and n_swap >= 2, (n_open, n_close, n_swap)
    finally:
        hs.close()


def test_lockstep_walkers_are_independent(libs, ref):
    """W walkers advanced together == each advanced alone (own seed, own stream)."""
    S = System(dim=3, Np=10, Nb=12, density=0.365)
    VT, WF, P, xend, delta = _setup(ref, S, 9)
    W = 3
    hs = HostSampler(S, VT, WF, W=W, backend=libs[0], hostlib=libs[1])
    try:
        Ps = [P + 0.0 for _ in range(W)]
        for w in range(W):
            hs.set_path(w, Ps[w])
            hs.seed(w, 100 + w)
        hs.upload()
        for rep in range(3):
            for ip in range(1, S.Np + 1):
                ipo = np.array([ip, (ip % S.Np) + 1, ((ip + 4) % S.Np) + 1], np.int32)
                act = np.array([1, 1, rep != 1], np.int32)
                hs.move("TranslateChain", ipo, rpar=delta, active=act)
                hs.move("MoveHeadBisection", ipo, i1=4, active=act)
                hs.move("MoveTailBisection", ipo, i1=4, active=act)
                hs.move("Bisection", ipo, i1=4, active=act)
        got = [hs.get_path(w) for w in range(W)]
        assert same_bits(hs.device_paths(), np.stack(got))
        # replay every walker alone with the reference
        for w in range(W):
            ref.sgrnd(100 + w)
            Pw = P + 0.0
            for rep in range(3):
                for ip in range(1, S.Np + 1):
                    if w == 2 and rep == 1:
                        continue
                    ipw = [ip, (ip % S.Np) + 1, ((ip + 4) % S.Np) + 1][w]
                    ref.translate_chain(delta, WF, VT, ipw, Pw)
                    ref.diag_move("MoveHeadBisection", WF, VT, 4, ipw, Pw)
                    ref.diag_move("MoveTailBisection", WF, VT, 4, ipw, Pw)
                    ref.diag_move("Bisection", WF, VT, 4, ipw, Pw)
            assert same_bits(got[w], Pw), w
    finally:
        hs.close()
