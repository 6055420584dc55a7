// pigs_device.h -- device-side primitives of the PIGS hot path for gfx950 (wave64).
//
// fp64 VALU only (no MFMA: there is no dense contraction on this path), compiled with -ffp-contract=off
// (every fused multiply-add is written out).  Two forms of the pair arithmetic live here:
//   * exact-term: every expression keeps the reference's operand order and rounding, so each per-pair TERM is
//     bit-identical to the reference's; only the order in which terms are summed differs (lane-strided partial
//     sums + a fixed reduction, deterministic run to run, no atomics).  min_image*, lerp_setup / interp*, and
//     the exact short forms div_by / sqrt_rinv / flerp_* / finterp*.
//   * short: the same quantities to ~1 ulp per term with the cutoff decision unchanged (min_image_rn, fcell_setup,
//     FastTab; the branch-free PipeTab form is in pigs_k1_device.h) -- the default for periodic systems.
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
    double  rdr;                         // RN(1/dr), for the exact constant division below
    double  hrdr;                        // 0.5/dr and 1/L: the short-arithmetic path (pigs_k1_device.h, FastTab)
    double  rLbox[3];
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

// ---- exact fp64 division / sqrt in few instructions ---------------------------------------
// The compiler's IEEE `a/b` costs ~12 instructions incl. a quarter-rate v_rcp_f64, and the hot
// loop has up to 16 of them per bead pair.  Both forms below return the correctly rounded
// quotient (so every term still rounds exactly like the reference's `/`):
//   * division by a loop constant d with rd = RN(1/d) precomputed on the host (Markstein:
//     q=a*rd; r=fma(-q,d,a); q'=fma(r,rd,q) is RN(a/d) when rd is the correctly rounded
//     reciprocal);
//   * several divisions by the same r sharing one refined reciprocal y ~ 1/r (error << 1 ulp):
//     the same residual-correction step, which is also how the compiler's own expansion ends.
// Operands here are normal-range (0 < r <= rcut, |numerators| far from over/underflow), so the
// v_div_scale / v_div_fixup range handling of the general expansion is not needed; NaN/Inf
// inputs still propagate to NaN.  pigs_selftest_fastmath() checks both against `/` and sqrt()
// bit for bit on the GPU.
__device__ __forceinline__ double div_by(double a, double d, double rd)
{
    const double q = a * rd;
    const double r = __builtin_fma(-q, d, a);
    return __builtin_fma(r, rd, q);
}

// The same for a numerator that may be +-Inf (a table head of a singular potential: V(0) = +Inf for r^-3 or r^-12): IEEE `/`
// gives +-Inf, the residual step above Inf - Inf = NaN.  The reference's Delta S is then -Inf (a move AWAY from an overlap
// below dr: accepted without a uniform) or +Inf (towards one: rejected), and NaN instead would freeze the particle for ever
// (round 3 fuzz: 1D starts with two particles within dr).  One compare + select more; used where an infinite numerator can
// arrive -- the force terms of the exact-term forms and the Chin-weight epilogue of every kernel -- not in the table-index path.
__device__ __forceinline__ double div_by_inf(double a, double d, double rd)
{
    const double q = a * rd;
    const double r = __builtin_fma(-q, d, a);
    const double z = __builtin_fma(r, rd, q);
    return __builtin_fabs(q) == __builtin_inf() ? q : z;
}

// s = RN(sqrt(x)) and y ~ 1/s from ONE v_rsq_f64 seed (Goldschmidt step + residual corrections).
__device__ __forceinline__ void sqrt_rinv(double x, double &s, double &y)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0;
    double h = 0.5 * y0;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d1, h, g);
    s = g;
    double yy = h + h;
    const double e = __builtin_fma(-g, yy, 1.0);
    y = __builtin_fma(e, yy, yy);
}

__device__ __forceinline__ double sqrt_exact(double x)
{
    double s, y;
    sqrt_rinv(x, s, y);
    return s;
}

// table lookup with the exact constant division (same results as lerp_setup/interp* above)
struct FLerp {
    int    ix, im2;
    double a1, a2;
};

__device__ __forceinline__ FLerp flerp_setup(double x, const DevParams &P)
{
    FLerp L;
    const int ix = (int)div_by(x, P.dr, P.rdr) + 1;
    L.a1  = x - (double)(ix - 1) * P.dr;
    L.a2  = P.dr - L.a1;
    L.ix  = min(ix, P.Nmax);
#ifdef PIGS_EXPERIMENT_HOT_TABLE
    L.ix  = (L.ix & 63) + 2;                 // timing experiment only: every lookup hits the same few lines
#endif
    L.im2 = max(L.ix - 2, 0);
    return L;
}

template <typename TabPtr>
__device__ __forceinline__ double finterp0(TabPtr F, const FLerp &L, const DevParams &P)
{
    return div_by_inf(L.a1 * F[L.ix] + L.a2 * F[L.ix - 1], P.dr, P.rdr);      // (an infinite table head stays infinite)
}

template <typename TabPtr>
__device__ __forceinline__ void finterp01(TabPtr F, const FLerp &L, const DevParams &P, double &v0, double &v1)
{
    const double fm2 = F[L.im2], fm1 = F[L.ix - 1], f0 = F[L.ix], fp1 = F[L.ix + 1];
    v0 = div_by_inf(L.a1 * f0 + L.a2 * fm1, P.dr, P.rdr);
    const double Fbefore = div_by_inf(L.a1 * fm1 + L.a2 * fm2, P.dr, P.rdr);
    const double Fafter  = div_by_inf(L.a1 * fp1 + L.a2 * f0, P.dr, P.rdr);
    v1 = div_by_inf(0.5 * (Fafter - Fbefore), P.dr, P.rdr);
}

// ---- minimum image, branch-free form: d - copysign(L,d) where |d| > L/2.  Identical to the
// two compares of pbc_mod.f90:40-41 for every input (x+L == x-(-L); the second wrap can never
// follow the first because LboxHalf == L/2 exactly); NaN stays NaN.
template <int DIM>
__device__ __forceinline__ double min_image_fast(double (&x)[DIM], const DevParams &P)
{
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double v = x[k];
        const double w = v - __builtin_copysign(P.Lbox[k], v);
        const double u = (__builtin_fabs(v) > P.LboxHalf[k]) ? w : v;
        x[k] = u;
        r2 = r2 + u * u;
    }
    return r2;
}

// ---- short-arithmetic forms (K1 variant "fast", selected by passing the table as a FastTab) ----
// Same quantities as above to ~1 ulp per term instead of the reference's exact rounding sequence:
//   * minimum image as v - sign(v)*L*rint(min(|v|/L,1)): identical result to the two compares except for |v|
//     within an ulp of L/2 exactly, where the pair is outside the cutoff anyway (r >= L/2 >= rcut);
//   * r^2 keeps the reference's rounding sequence, so the cutoff decision r2 <= rcut2 is unchanged;
//   * sqrt / 1/r from one v_rsq_f64 + one coupled Newton step (r to < 1 ulp, 1/r to ~1e-14);
//   * interpolation in the normalised cell coordinate f = r/dr - int(r/dr): f*F[i+1] + (1-f)*F[i]
//     instead of (a1*F(ix) + a2*F(ix-1))/dx -- the same straight line, rounded differently.
// ~50 fp64-rate instructions per in-cutoff distance on odd beads instead of ~105.
struct FastTab {
    const double *p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};
template <typename T> struct is_fast_tab { static constexpr bool value = false; };
template <> struct is_fast_tab<FastTab> { static constexpr bool value = true; };

template <int DIM>
__device__ __forceinline__ double min_image_rn(double (&x)[DIM], const DevParams &P)
{
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double v = x[k];
        // ONE fold at most, as pbc_mod.f90:40-41: |n| = rint(min(|v|/L, 1)) is 0 or 1 (the min/max pair is the VOP3 clamp
        // modifier of the multiply).  A plain rint(v/L) folds a |v| > 1.5 L twice where the reference leaves it outside
        // the cutoff -- reachable in a small box by the long free segments of a head / tail move (round 3 fuzz).
        const double t = __builtin_fmin(__builtin_fmax(__builtin_fabs(v) * P.rLbox[k], 0.0), 1.0);
        const double n = __builtin_copysign(__builtin_rint(t), v);      // one v_bfi_b32 on the high word
        const double u = __builtin_fma(-P.Lbox[k], n, v);
        x[k] = u;
        r2 = r2 + u * u;                                          // the reference's rounding sequence
    }
    return r2;
}

struct FCell {
    int    i0;        // int(r/dr) = ix-1 of the reference
    double f, omf;    // position inside the cell, 1-f
    double r, rinv;
};

__device__ __forceinline__ FCell fcell_setup(double r2, const DevParams &P)
{
    FCell C;
    const double y0 = __builtin_amdgcn_rsq(r2);
    double g = r2 * y0;
    double h = 0.5 * y0;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    const double d0 = __builtin_fma(-g, g, r2);
    g = __builtin_fma(d0, h, g);
    C.r    = g;
    C.rinv = h + h;
    const double t = g * P.rdr;
    C.i0  = (int)t;
    C.f   = t - (double)C.i0;
    C.omf = 1.0 - C.f;
    return C;
}

// ---- several accumulators reduced over the wave with one shared butterfly: at each of the
// first log2(N) steps a lane keeps half of its values and ships the other half, so N values
// cost N-1 + (6-log2 N) shuffles instead of 6N.  Afterwards value i's total sits in every
// lane whose bits {32,16,8,...} spell i (MSB first); lane_of(i) gives one such lane.
template <int N>
__device__ __forceinline__ void wave_reduce_multi(double (&v)[N], int lane)
{
    int stride = 32;
#pragma unroll
    for (int cnt = N; cnt > 1; cnt >>= 1, stride >>= 1) {
        const bool upper = (lane & stride) != 0;
#pragma unroll
        for (int i = 0; i < cnt / 2; ++i) {
            const double send = upper ? v[i] : v[i + cnt / 2];
            const double keep = upper ? v[i + cnt / 2] : v[i];
            v[i] = keep + __shfl_xor(send, stride, kWave);
        }
    }
#pragma unroll
    for (; stride >= 1; stride >>= 1) v[0] += __shfl_xor(v[0], stride, kWave);
}

// ---- N accumulators x 64 lanes summed through a per-wave LDS transpose ---------------------
// scratch: N rows of kRedStride doubles (stride 65 keeps both the row writes and the strided
// reads bank-conflict free).  Each lane first adds up N entries of value (lane % N), then the
// 64/N partials per value are folded with log2(64/N) shuffles: ~3N+5*log2(64/N) instructions
// instead of ~50 per value for a plain butterfly.  Afterwards every lane l holds the total of
// value l % N.  DS instructions of one wave execute in issue order, so no barrier is needed
// between the writes and the reads (wave_barrier only pins the compiler's order).
constexpr int kRedStride = 65;

template <int N>
__device__ __forceinline__ double wave_reduce_lds(const double (&v)[N], double *scratch, int lane)
{
    static_assert(N == 2 || N == 4 || N == 8, "N must divide 64");
#pragma unroll
    for (int i = 0; i < N; ++i) scratch[i * kRedStride + lane] = v[i];
    __builtin_amdgcn_wave_barrier();
    const int vi = lane % N, g = lane / N;
    const double *row = scratch + vi * kRedStride + g * N;
    double s = row[0];
#pragma unroll
    for (int k = 1; k < N; ++k) s = s + row[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int stride = N; stride < kWave; stride <<= 1) s = s + __shfl_xor(s, stride, kWave);
    return s;
}

template <int N>
__device__ __forceinline__ constexpr int lane_of(int i)
{
    // bits of i, MSB first, land on lane bits 32,16,8,...
    int lane = 0, stride = 32;
    for (int cnt = N; cnt > 1; cnt >>= 1, stride >>= 1)
        if (i & (cnt >> 1)) lane |= stride;
    return lane;
}

__device__ __forceinline__ double read_lane(double v, int lane)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

// value of lane (i + N) of the same row of 16 lanes (DPP row_shl:N; 0 beyond the row): no LDS round trip
template <int N>
__device__ __forceinline__ double row_shl(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x100 + N, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x100 + N, 0xf, 0xf, true);
    return r.d;
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

// GreenFunction(0, ...) for the kernels' epilogues: the same expressions with the divisions by 3 and 6 done as
// exact short divisions (div_by with RN(1/3), RN(1/6): bit-identical to `/`, checked over 3e8 operands) -- the
// IEEE expansions were ~500 dependent cycles at the end of every item and of every sampler stage.
__device__ __forceinline__ double green_function_action(int ib, int Nb, double dt_in, double Pot, double F2)
{
    constexpr double r3 = 1.0 / 3.0, r6 = 1.0 / 6.0;
    // dt made opaque: hoisted out of a persistent kernel's item loop, the products 2*dt, 4*dt, dt*dt each cost a
    // VGPR pair (and spilled at the 128-VGPR budget of K1's pipe kernels); recomputing them per item is free
    double dt = dt_in;
    asm volatile("" : "+v"(dt));
    if (ib == 0 || ib == 2 * Nb) return div_by_inf(dt * Pot, 3.0, r3);
    if ((ib & 1) == 0)           return div_by_inf(2.0 * dt * Pot, 3.0, r3);
    const double Vc = Pot + div_by_inf(dt * dt * F2, 6.0, r6);
    return div_by_inf(4.0 * dt * Vc, 3.0, r3);
}

// ---- analytic trial function, wf_table = F (the reference's DEFAULT, vpi_mod.f90:59): McMillan u(r) = -0.5 (Rm/r)^5 and
// its derivatives (system_mod.f90:38-66).  (Rm/r)**5 is a left-to-right product in the reference build (pinned on the
// oracle); the divisions are IEEE.  Used by the end-bead terms of K1 / K6 and by K4.
__host__ __device__ inline double log_psi(int opt, double Rm, double r)
{
    const double q = Rm / r;
    const double q5 = q * q * q * q * q;
    if (opt == 0) return -0.5 * q5;
    if (opt == 1) return 2.5 * q5 / r;
    return -15.0 * q5 / (r * r);
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
