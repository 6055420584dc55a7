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
