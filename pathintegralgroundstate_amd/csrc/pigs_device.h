// pigs_device.h -- device-side primitives of the PIGS hot path for gfx950 (wave64).
//
// fp64 VALU only (no MFMA: there is no dense contraction on this path).  Every
// expression keeps the reference's operand order and is compiled with
// -ffp-contract=off, so each per-pair TERM is bit-identical to the reference's; only
// the order in which terms are summed differs (lane-strided partial sums + a fixed
// butterfly, deterministic run to run, no atomics).
//
// Reference restated here: interpolate.f90:1-45 (Interpolate), pbc_mod.f90:29-52
// (MinimumImage), global_mod.f90:19-72 (GreenFunction), system_mod.f90:213-252
// (TrapPsi/TrapPot).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pigs {

constexpr int kWave = 64;

// Kernel-argument block (lives in SGPRs / kernarg segment).
struct DevParams {
    int32_t dim, Np, NpPad, Nb;
    int32_t M, Nmax, trap, wf_table;
    int32_t v_table, nW, pad1, pad2;     // nW = resident walkers (device-side range check)
    double  dr, rcut2, dt, Rm;
    double  Lbox[3], LboxHalf[3], a_ho[3];
};

// Resident worldline layout in HBM ("bead-major, SoA inside a slice"):
//   paths[walker][ib][k][jp]   jp fastest, padded to NpPad (multiple of 8 doubles = 64 B)
// so that lane jp of a wave reads x/y/z of partner jp with unit stride (512 B per
// wave-load), and one (walker, ib) slice is one contiguous dim*NpPad*8-byte block.
__host__ __device__ inline size_t slice_doubles(int dim, int NpPad) { return (size_t)dim * NpPad; }

// ---- wave64 reductions (fixed butterfly: deterministic) ---------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// ---- minimum image: single wrap per coordinate, returns r^2, mutates x (Q14) -------
template <int DIM>
__device__ __forceinline__ double min_image(double (&x)[DIM], const DevParams &P)
{
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        double v = x[k];
        if (v >  P.LboxHalf[k]) v = v - P.Lbox[k];
        if (v < -P.LboxHalf[k]) v = v + P.Lbox[k];
        x[k] = v;
        r2 = r2 + v * v;
    }
    return r2;
}

template <int DIM>
__device__ __forceinline__ double plain_r2(const double (&x)[DIM])
{
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) r2 = r2 + x[k] * x[k];
    return r2;
}

// ---- table lookup (interpolate.f90).  F points at F(0); entries 0..Nmax+1. --------
// The index is clamped to the table only to keep loads in bounds where the reference
// itself would read outside its array (r<dr: its result is NaN there anyway, Q4; trap
// branch beyond the table, Q5); inside the table the clamp is the identity.
struct Lerp {
    int    ix;    // table cell, <= Nmax
    int    im2;   // ix-2, clamped at 0 (the reference reads F(-1) when r<dr: undefined there)
    double a1, a2;
};

__device__ __forceinline__ Lerp lerp_setup(double x, double dx, int Nmax)
{
    Lerp L;
    int ix = (int)(x / dx) + 1;                 // interpolate.f90:13
    L.a1   = x - (double)(ix - 1) * dx;         // :14
    L.a2   = dx - L.a1;                         // :15
    L.ix   = min(ix, Nmax);                     // keeps ix+1 <= Nmax+1 (identity when r<=rcut)
    L.im2  = max(L.ix - 2, 0);
    return L;
}

__device__ __forceinline__ double interp0(const double *__restrict__ F, const Lerp &L, double dx)
{
    return (L.a1 * F[L.ix] + L.a2 * F[L.ix - 1]) / dx;                    // :21
}

__device__ __forceinline__ double interp1(const double *__restrict__ F, const Lerp &L, double dx)
{
    double Fbefore = (L.a1 * F[L.ix - 1] + L.a2 * F[L.im2]) / dx;       // :25
    double Fafter  = (L.a1 * F[L.ix + 1] + L.a2 * F[L.ix]) / dx;           // :26
    return 0.5 * (Fafter - Fbefore) / dx;                                  // :28
}

__device__ __forceinline__ double interp2(const double *__restrict__ F, const Lerp &L, double dx)
{
    double Fbefore = (L.a1 * F[L.ix - 1] + L.a2 * F[L.im2]) / dx;       // :32
    double Fcurr   = (L.a1 * F[L.ix] + L.a2 * F[L.ix - 1]) / dx;           // :33
    double Fafter  = (L.a1 * F[L.ix + 1] + L.a2 * F[L.ix]) / dx;           // :34
    return (Fafter - 2.0 * Fcurr + Fbefore) / (dx * dx);                   // :36
}

// ---- Chin weights (global_mod.f90:19-72) -------------------------------------------
__host__ __device__ inline double green_function(int opt, int ib, int Nb, double dt, double Pot, double F2)
{
    double g = 0.0;
    if (opt == 0) {
        double Ve = Pot;
        double Vc = Pot + dt * dt * F2 / 6.0;
        if (ib == 0 || ib == 2 * Nb) g = dt * Ve / 3.0;
        else if ((ib & 1) == 0)      g = 2.0 * dt * Ve / 3.0;
        else                         g = 4.0 * dt * Vc / 3.0;
    } else {
        double dVe = Pot;
        double dVc = Pot + dt * dt * F2 / 2.0;
        if (ib == 0 || ib == 2 * Nb) g = dVe / 3.0;
        else if ((ib & 1) == 0)      g = 2.0 * dVe / 3.0;
        else                         g = 4.0 * dVc / 3.0;
    }
    return g;
}

// ---- one-body trap terms (system_mod.f90:213-252) -----------------------------------
__host__ __device__ inline double trap_pot(int opt, double a, double x)
{
    const double a4 = a * a * a * a;       // a_osc**4 as a left-to-right product (matches the reference build)
    return opt == 0 ? 0.5 * (x * x) / a4 : x / a4;
}

__host__ __device__ inline double trap_psi(int opt, double a, double x)
{
    if (opt == 0) { double q = x / a; return -0.5 * (q * q); }
    if (opt == 1) return -(x / (a * a));
    return -1.0 / (a * a);
}

} // namespace pigs
