"""Test-side numerics helpers (numpy only)."""
import numpy as np


def min_image(d, Lbox, trap):
    if trap:
        return d
    d = np.where(d > Lbox / 2, d - Lbox, d)
    return np.where(d < -Lbox / 2, d + Lbox, d)


def term_scales(S, VT, WF, Path, ip, ib, xnew, xold):
    """Per-case magnitude of the sums UpdateAction forms: sum|V|, sum|dV/dr|, sum|u| over both
    distances of every partner.  Used only to scale rounding-error tolerances (the GPU sums the
    same bit-identical terms in a different order)."""
    dim, L = S.dim, np.asarray(S.Lbox[:S.dim])
    n = len(ip)
    sv, sf, su = np.zeros(n), np.zeros(n), np.zeros(n)
    aV, aW = np.abs(np.nan_to_num(VT, nan=0.0, posinf=0.0, neginf=0.0)), np.abs(np.nan_to_num(WF, nan=0.0, posinf=0.0, neginf=0.0))
    dV = np.abs(np.gradient(np.nan_to_num(VT, nan=0.0, posinf=0.0, neginf=0.0))) / S.dr
    for i in range(n):
        R = np.delete(Path[ib[i]], ip[i] - 1, axis=0)
        for x in (xnew[i], xold[i]):
            d = min_image(x[None, :dim] - R, L, S.trap)
            r = np.sqrt((d * d).sum(1))
            if not S.trap:
                r = r[r * r <= S.rcut2]
            idx = np.clip((r / S.dr).astype(int) + 1, 1, S.Nmax)
            sv[i] += np.maximum(aV[idx], aV[idx - 1]).sum()
            su[i] += np.maximum(aW[idx], aW[idx - 1]).sum()
            sf[i] += np.maximum(dV[np.clip(idx + 1, 0, S.Nmax + 1)], dV[idx - 1]).sum() * 2.0
    return sv, sf, su


def delta_s_tolerance(S, sv, sf, su, eps=2e-13):
    dt = S.dt
    return eps * (su + (4.0 / 3.0) * dt * (sv + dt * dt * (sf * sf) / 6.0)) + 1e-300


def ulp_diff(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(all="ignore"):
        u = np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return np.where(both_nan | same_inf, 0.0, u)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, float), np.ascontiguousarray(b, float)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or \
        bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


# ---- fixtures written by tests/golden/ref_driver.py (driver.npz next to a reference-program run) ----
def fold_maxnorm(d, Lbox, trap=False):
    """max |d| after folding every coordinate difference by the box length (a last-bit difference may sit on
    either side of the wrap at +-L/2)."""
    d = np.asarray(d, float)
    if not trap:
        L = np.asarray(Lbox[:d.shape[-1]], float)
        d = d - L * np.round(d / L)
    return float(np.max(np.abs(d))) if d.size else 0.0


def read_hex_blocks(path):
    """e_vpi.hex of pigs_vpi: one line per diagonal block = block number + E K V Et Kt Vt (per particle) as
    64-bit hex -> (blocks int array, (n,6) float array)."""
    import struct
    blocks, rows = [], []
    for ln in open(path):
        t = ln.split()
        if not t:
            continue
        blocks.append(int(t[0]))
        rows.append([struct.unpack(">d", bytes.fromhex(h))[0] for h in t[1:7]])
    return np.array(blocks, int), np.array(rows, float).reshape(-1, 6)


def driver_blocks(drv):
    """(blocks, (n,6)) of a driver.npz in the e_vpi.hex column order: E K V Et Kt Vt."""
    be, bt = drv["block_e"], drv["block_t"]
    return be[:, 0].astype(int), np.concatenate([be[:, 1:4], bt[:, 1:4]], axis=1)


def sha256_of(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, float).tobytes()).digest(), np.uint8)


def check_worldline_vs_driver(P, drv, Lbox, trap=False, tol=0.0):
    """A final worldline (M,Np,dim) against a driver.npz: tol = 0 asks for identical bits (SHA-256 of the whole
    array), tol > 0 for the L-folded max-norm on the stored beads and on every bead's coordinate sums."""
    P = np.ascontiguousarray(P, float)
    assert tuple(P.shape) == tuple(int(x) for x in drv["Path_shape"]), (P.shape, drv["Path_shape"])
    if tol == 0.0:
        assert np.array_equal(sha256_of(P), drv["Path_sha256"]), "final worldline differs from the reference's (a decision flipped)"
        return 0.0
    worst = 0.0
    if "Path_sub" in drv:
        st = int(drv["bead_stride"])
        worst = fold_maxnorm(P[::st] - drv["Path_sub"], Lbox, trap)
        assert worst < tol, worst
    # every bead enters through its coordinate sums (folded by L: wraps shift a sum by multiples of L)
    s = fold_maxnorm(P.sum(axis=1) - drv["bead_sums"], Lbox, trap)
    assert s < tol * P.shape[1], s
    return max(worst, s / P.shape[1])


def block_energy_errors(rows, want):
    """Errors of block rows [E K V Et Kt Vt] scaled the way their rounding errors scale: E = K + V and Et = Kt + Vt are
    sums of opposite-sign terms (E per particle can be a small remainder), so E, K are measured against |K|+|V| and
    Et, Kt against |Kt|+|Vt|; V, Vt against themselves.  Returns (err_mixed (n,2), err_rest (n,4))."""
    rows, want = np.atleast_2d(rows), np.atleast_2d(want)
    sc_e = np.abs(want[:, 1]) + np.abs(want[:, 2])
    sc_t = np.abs(want[:, 4]) + np.abs(want[:, 5])
    d = np.abs(rows - want)

    def ratio(num, den):                  # identical values are zero error also where the scale vanishes (two particles
        with np.errstate(all="ignore"):   # beyond each other's cutoff: V = 0 exactly)
            return np.where(num == 0, 0.0, num / den)
    mixed = ratio(d[:, 0:2], sc_e[:, None])
    rest = np.stack([ratio(d[:, 2], np.abs(want[:, 2])), ratio(d[:, 3], sc_t), ratio(d[:, 4], sc_t),
                     ratio(d[:, 5], np.abs(want[:, 5]))], 1)
    return mixed, rest


# The mixed estimator (LocalEnergy: E, K columns) is ill-conditioned at one ulp: the reference's own LocalEnergy moves by
# up to 6e-10 (|K|+|V|) when every coordinate moves by ONE ulp -- its second derivative of log psi is a second difference
# of the table divided by dr^2 = (rcut/9999)^2 (interpolate.f90:36-42), which amplifies rounding by ~1e7 (pinned on the
# oracle: tests/test_oracle_golden.py::test_mixed_estimator_is_ill_conditioned_at_one_ulp).  It can therefore meet the
# 1e-10 contract only on a worldline that is BIT-identical to the reference's.  Both samplers deliver that: the
# host-driven one since round 1, the device-resident one since its Box-Muller log() became the host libm's bit for bit
# (csrc/pigs_log_host.h, round 3; rounds 1-2 allowed 2e-9 here).
MIXED_TOL = 1e-10
