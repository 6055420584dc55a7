// pigs_k1_device.h -- the per-item Delta-S evaluation of K1 as device functions, shared by the
// batched kernel (pigs_k1.hip) and the device-resident sampler (pigs_sampler.hip).
// One wave64 evaluates one item; see pigs_k1.hip for the design notes.
#pragma once

#include "pigs_device.h"

namespace pigs {

constexpr int kWaveLds = 8 * kRedStride * (int)sizeof(double);   // 4160 B per wave

// bead classes of UpdateAction (vpi_mod.f90:2509-2525): what is accumulated per pair
enum BeadClass { CLS_EVEN = 0, CLS_ODD = 1, CLS_END = 2 };

// per-lane accumulators of one item
template <int DIM, int CLS>
struct Acc {
    double potN = 0.0, potO = 0.0;
    double psiN = 0.0, psiO = 0.0;      // CLS_END only
    double fN[DIM], fO[DIM];            // CLS_ODD only
    __device__ __forceinline__ Acc()
    {
#pragma unroll
        for (int k = 0; k < DIM; ++k) { fN[k] = 0.0; fO[k] = 0.0; }
    }
};

// in-cutoff work of ONE distance: table cell, V (opt 0), dV/dr (opt 1) and the force terms
// (dv*xij(k))/rij on odd beads, u (opt 0 of LogWF) on end beads.
template <int DIM, int CLS, bool IS_OLD, typename VTab>
__device__ __forceinline__ void pair_accumulate(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                                double r2, const double (&d)[DIM], Acc<DIM, CLS> &A,
                                                bool pot_on = true)
{
    if constexpr (is_fast_tab<VTab>::value) {
        // short arithmetic (pigs_device.h): cells are never clamped here -- PBC only, so r <= rcut keeps
        // i0+2 <= Nmax+1; only i0-1 needs the lower clamp (r < dr, where the table head is NaN anyway, Q4)
        const FCell C = fcell_setup(r2, P);
        const double *V = VT.p + C.i0;
        const double F0 = V[0], F1 = V[1];
        if (CLS == CLS_ODD) {
            const double Fm = VT.p[max(C.i0 - 1, 0)], Fp = V[2];
            const double v  = __builtin_fma(C.f, F1, C.omf * F0);
            const double Fb = __builtin_fma(C.f, F0, C.omf * Fm);
            const double Fa = __builtin_fma(C.f, Fp, C.omf * F1);
            const double s  = ((Fa - Fb) * P.hrdr) * C.rinv;       // (dV/dr)/r
            if (IS_OLD) A.potO = A.potO + v; else A.potN = A.potN + v;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                if (IS_OLD) A.fO[k] = __builtin_fma(s, d[k], A.fO[k]); else A.fN[k] = __builtin_fma(s, d[k], A.fN[k]);
            }
        } else {
            if (pot_on) {
                const double v = __builtin_fma(C.f, F1, C.omf * F0);
                if (IS_OLD) A.potO = A.potO + v; else A.potN = A.potN + v;
            }
            if (CLS == CLS_END) {
                double u;
                if (P.wf_table) { const double *U = WF + C.i0; u = __builtin_fma(C.f, U[1], C.omf * U[0]); }
                else            u = log_psi(0, P.Rm, C.r);                  // analytic trial function (wf_table = F)
                if (IS_OLD) A.psiO = A.psiO + u; else A.psiN = A.psiN + u;
            }
        }
        return;
    }
    double r, rinv;
    sqrt_rinv(r2, r, rinv);
    const FLerp L = flerp_setup(r, P);
    if (CLS == CLS_ODD) {
        double v, dv;
        finterp01(VT, L, P, v, dv);
        if (IS_OLD) A.potO = A.potO + v; else A.potN = A.potN + v;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            const double f = div_by_inf(dv * d[k], r, rinv);        // (dv*xij(k))/rij  (dv = +-Inf below dr of a singular table)
            if (IS_OLD) A.fO[k] = A.fO[k] + f; else A.fN[k] = A.fN[k] + f;
        }
    } else {
        if (pot_on) {
            const double v = finterp0(VT, L, P);
            if (IS_OLD) A.potO = A.potO + v; else A.potN = A.potN + v;
        }
        if (CLS == CLS_END) {
            const double u = P.wf_table ? finterp0(WF, L, P) : log_psi(0, P.Rm, r);   // vpi_mod.f90:2599-2645
            if (IS_OLD) A.psiO = A.psiO + u; else A.psiN = A.psiN + u;
        }
    }
}

// reduce the class's accumulators over the wave, apply the Chin weight, store
// `out` / `parts` point at THIS item's result slot (global or LDS)
template <int DIM, int CLS>
__device__ __forceinline__ void finish_item(const DevParams &P, int lane, int b, const Acc<DIM, CLS> &A,
                                            double *red, double *out, double *parts)
{
    // after the reduction lane q holds the total of column q; the neighbours' columns reach lane 0 by DPP row shifts
    // (no v_readlane -> scalar -> vector round trips); the results below are meaningful in lane 0 only
    double dPot, dF2 = 0.0, dPsi = 0.0;
    if (CLS == CLS_ODD) {
        double v[8] = {A.potN, A.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DIM; ++k) { v[2 + k] = A.fN[k]; v[5 + k] = A.fO[k]; }
        const double t = wave_reduce_lds<8>(v, red, lane);
        const double sq = t * t;
        double f2 = sq;                                             // lane 2: |Fnew|^2, lane 5: |Fold|^2 (vpi_mod.f90:2831-2832)
        if (DIM > 1) f2 = f2 + row_shl<1>(sq);
        if (DIM > 2) f2 = f2 + row_shl<2>(sq);
        dPot = t - row_shl<1>(t);                                   // :2838
        dF2  = row_shl<2>(f2) - row_shl<5>(f2);                     // :2835
    } else if (CLS == CLS_END) {
        const double v[4] = {A.potN, A.potO, A.psiN, A.psiO};
        const double t = wave_reduce_lds<4>(v, red, lane);
        dPot = t - row_shl<1>(t);
        dPsi = row_shl<2>(dPot);                                    // lane 2: PsiNew - PsiOld (:2653)
    } else {
        const double v[2] = {A.potN, A.potO};
        const double t = wave_reduce_lds<2>(v, red, lane);
        dPot = t - row_shl<1>(t);
    }
    if (lane == 0) {
        *out = -dPsi + green_function_action(b, P.Nb, P.dt, dPot, dF2);      // :2527
        if (parts) {
            parts[0] = dPot;
            parts[1] = dF2;
            parts[2] = dPsi;
        }
    }
}

// the reference's one-body terms of a trapped system enter the sums once (vpi_mod.f90:2688-2695,
// 2555-2560): lane 0 of the wave that owns the item's first pass adds them
template <int DIM, bool TRAP, int CLS>
__device__ __forceinline__ void trap_terms(const DevParams &P, const double (&xn)[DIM], const double (&xo)[DIM],
                                           Acc<DIM, CLS> &A)
{
    if (!TRAP) return;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        A.potN = A.potN + trap_pot(0, P.a_ho[k], xn[k]);
        A.potO = A.potO + trap_pot(0, P.a_ho[k], xo[k]);
        if (CLS == CLS_ODD) {
            A.fO[k] = trap_pot(1, P.a_ho[k], xo[k]);
            A.fN[k] = trap_pot(1, P.a_ho[k], xn[k]);
        }
        if (CLS == CLS_END) {
            A.psiO = A.psiO + trap_psi(0, P.a_ho[k], xo[k]);
            A.psiN = A.psiN + trap_psi(0, P.a_ho[k], xn[k]);
        }
    }
}

// partner at rj[] of the moved particle: both distances, cutoff tests, accumulate
template <int DIM, bool TRAP, int CLS, typename VTab>
__device__ __forceinline__ void partner_accumulate_at(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                                      const double (&rj)[DIM], const double (&xn)[DIM],
                                                      const double (&xo)[DIM], Acc<DIM, CLS> &A)
{
    double dnew[DIM], dold[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        dnew[k] = xn[k] - rj[k];                                      // :2706-2707
        dold[k] = xo[k] - rj[k];
    }
    double r2n, r2o;
    if (TRAP) { r2o = plain_r2<DIM>(dold); r2n = plain_r2<DIM>(dnew); }
    else if constexpr (is_fast_tab<VTab>::value) { r2o = min_image_rn<DIM>(dold, P); r2n = min_image_rn<DIM>(dnew, P); }
    else      { r2o = min_image_fast<DIM>(dold, P); r2n = min_image_fast<DIM>(dnew, P); }
    if (TRAP || r2n <= P.rcut2)                                       // :2723 (Q5) / :2771
        pair_accumulate<DIM, CLS, false>(P, VT, WF, r2n, dnew, A);
    const bool in_o = r2o <= P.rcut2;                                 // :2745 / :2795
    if (in_o) pair_accumulate<DIM, CLS, true>(P, VT, WF, r2o, dold, A);
    else if (TRAP && CLS == CLS_END)                                  // UpdateWf's trap branch has no cutoff
        pair_accumulate<DIM, CLS, true>(P, VT, WF, r2o, dold, A, false);
}

// partner j of the moved particle p (row p itself is never read: vpi_mod.f90:2699)
template <int DIM, bool TRAP, int CLS, typename VTab>
__device__ __forceinline__ void partner_accumulate(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                                   const double *__restrict__ S, int p, int j,
                                                   const double (&xn)[DIM], const double (&xo)[DIM],
                                                   Acc<DIM, CLS> &A)
{
    if (j < P.Np && j != p) {
        double rj[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) rj[k] = S[(size_t)k * P.NpPad + j];
        partner_accumulate_at<DIM, TRAP, CLS>(P, VT, WF, rj, xn, xo, A);
    }
}

// same item with the partner coordinates of up to 4 passes (Np <= 256) requested up front, so that
// one memory latency is exposed per item instead of one per pass
template <int DIM, bool TRAP, int CLS, typename VTab>
__device__ __forceinline__ void item_direct_prefetch(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                                     const double *__restrict__ S, int p, const double (&xn)[DIM],
                                                     const double (&xo)[DIM], int lane, int b, double *red,
                                                     double *out, double *parts)
{
    constexpr int MAXP = 4;
    Acc<DIM, CLS> A;
    if (TRAP && lane == 0) trap_terms<DIM, TRAP, CLS>(P, xn, xo, A);
    double rj[MAXP][DIM];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        const int j = m * kWave + lane;
        const int jj = j < P.Np ? j : 0;                              // in-bounds dummy for idle lanes
#pragma unroll
        for (int k = 0; k < DIM; ++k) rj[m][k] = S[(size_t)k * P.NpPad + jj];
    }
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        const int j = m * kWave + lane;
        if (j < P.Np && j != p) partner_accumulate_at<DIM, TRAP, CLS>(P, VT, WF, rj[m], xn, xo, A);
    }
    finish_item<DIM, CLS>(P, lane, b, A, red, out, parts);
}

// one item, every partner visited by its lane (no compaction)
template <int DIM, bool TRAP, int CLS, typename VTab>
__device__ __forceinline__ void item_direct(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                            const double *__restrict__ S, int p, const double (&xn)[DIM],
                                            const double (&xo)[DIM], int lane, int b, double *red,
                                            double *out, double *parts)
{
    Acc<DIM, CLS> A;
    if (TRAP && lane == 0) trap_terms<DIM, TRAP, CLS>(P, xn, xo, A);
    for (int j0 = 0; j0 < P.Np; j0 += kWave)
        partner_accumulate<DIM, TRAP, CLS>(P, VT, WF, S, p, j0 + lane, xn, xo, A);
    finish_item<DIM, CLS>(P, lane, b, A, red, out, parts);
}

// ---- split form (device-resident sampler, few beads per stage): one wave does ONE pass of 64
// partners of an item and leaves its 8 wave totals in tot8[]; item_finish_split() adds the passes
// of an item in pass order and applies the Chin weight.  Totals layout: potN potO fN[3] fO[3]
// (odd) / potN potO psiN psiO (end) / potN potO (even).
template <int DIM, bool TRAP, int CLS, typename VTab>
__device__ __forceinline__ void item_pass_cls(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                              const double *__restrict__ S, int p, int m,
                                              const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                              double *red, double *tot8)
{
    Acc<DIM, CLS> A;
    if (TRAP && m == 0 && lane == 0) trap_terms<DIM, TRAP, CLS>(P, xn, xo, A);
    partner_accumulate<DIM, TRAP, CLS>(P, VT, WF, S, p, m * kWave + lane, xn, xo, A);
    if (CLS == CLS_ODD) {
        double v[8] = {A.potN, A.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DIM; ++k) { v[2 + k] = A.fN[k]; v[5 + k] = A.fO[k]; }
        const double t = wave_reduce_lds<8>(v, red, lane);
        if (lane < 8) tot8[lane] = t;
    } else if (CLS == CLS_END) {
        const double v[4] = {A.potN, A.potO, A.psiN, A.psiO};
        const double t = wave_reduce_lds<4>(v, red, lane);
        if (lane < 4) tot8[lane] = t;
    } else {
        const double v[2] = {A.potN, A.potO};
        const double t = wave_reduce_lds<2>(v, red, lane);
        if (lane < 2) tot8[lane] = t;
    }
}

template <int DIM, bool TRAP, typename VTab>
__device__ __forceinline__ void item_pass(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                          const double *__restrict__ S, int p, int b, int m,
                                          const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                          double *red, double *tot8)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       item_pass_cls<DIM, TRAP, CLS_ODD>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
    else if (endb) item_pass_cls<DIM, TRAP, CLS_END>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
    else           item_pass_cls<DIM, TRAP, CLS_EVEN>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
}

// one thread: Delta S of an item from its npass x 8 wave totals
template <int DIM>
__device__ __forceinline__ double item_finish_split(const DevParams &P, int b, int npass, const double *tot)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    // all eight columns are summed (static register indexing); the class decides which ones mean anything
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < npass; ++m) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = s[q] + tot[m * 8 + q];
    }
    double dF2 = 0.0, dPsi = 0.0;
    const double dPot = s[0] - s[1];
    if (odd) {
        double fn2 = 0.0, fo2 = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) { fn2 = fn2 + s[2 + k] * s[2 + k]; fo2 = fo2 + s[5 + k] * s[5 + k]; }
        dF2 = fn2 - fo2;
    } else if (endb) {
        dPsi = s[2] - s[3];
    }
    return -dPsi + green_function_action(b, P.Nb, P.dt, dPot, dF2);
}

// the same by the lanes of ONE wave for up to 8 items at once: lane = item * 8 + column adds its column over the passes
// (in pass order, as above), the neighbours' columns come by DPP row shifts (no LDS round trip, no loop on one lane);
// Delta S of item i is returned in lane 8 i.  Same operations in the same order as item_finish_split: same bits.
template <int DIM>
__device__ __forceinline__ double item_finish_split_lanes(const DevParams &P, int n, const int *ibs, int npass,
                                                          const double *tot, int lane)
{
    const bool on = lane < n * 8;
    const int i = on ? lane >> 3 : 0;
    double cs = 0.0;
    {
        const double *T = tot + (size_t)i * npass * 8 + (lane & 7);
        for (int m = 0; m < npass; ++m) cs = cs + (on ? T[m * 8] : 0.0);
    }
    const int b = ibs[i];
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    const double d01 = cs - row_shl<1>(cs);                           // column 0: DeltaPot; column 2: PsiNew - PsiOld
    const double sq  = cs * cs;
    double f2 = sq;                                                   // column 2: |Fnew|^2, column 5: |Fold|^2 (DIM terms, in order)
    if (DIM > 1) f2 = f2 + row_shl<1>(sq);
    if (DIM > 2) f2 = f2 + row_shl<2>(sq);
    const double fn2 = row_shl<2>(f2), fo2 = row_shl<5>(f2), dps = row_shl<2>(d01);
    return -(endb ? dps : 0.0) + green_function_action(b, P.Nb, P.dt, d01, odd ? fn2 - fo2 : 0.0);
}

template <int DIM, bool TRAP, typename VTab>
__device__ __forceinline__ void item_eval_prefetch(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                                   const double *__restrict__ S, int p, int b, const double (&xn)[DIM],
                                                   const double (&xo)[DIM], int lane, double *red, double *out, double *parts)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       item_direct_prefetch<DIM, TRAP, CLS_ODD>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else if (endb) item_direct_prefetch<DIM, TRAP, CLS_END>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else           item_direct_prefetch<DIM, TRAP, CLS_EVEN>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
}

// UpdateAction's three cases, chosen per item (wave-uniform)
template <int DIM, bool TRAP, typename VTab>
__device__ __forceinline__ void item_eval(const DevParams &P, VTab VT, const double *__restrict__ WF,
                                          const double *__restrict__ S, int p, int b, const double (&xn)[DIM],
                                          const double (&xo)[DIM], int lane, double *red, double *out, double *parts)
{
    const bool odd  = (b & 1) != 0;                         // force term on odd beads
    const bool endb = (b == 0) || (b == 2 * P.Nb);          // UpdateWf only on the two end beads
    if (odd)       item_direct<DIM, TRAP, CLS_ODD>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else if (endb) item_direct<DIM, TRAP, CLS_END>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else           item_direct<DIM, TRAP, CLS_EVEN>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
}

// Row a lane loads in the branch-free forms: its own partner, or -- for a lane without one (beyond Np, or the moved
// particle's own row, which the aliasing contract says is never read: vpi_mod.f90:2699) -- another particle's row,
// whose (finite) coordinates are then masked out through the table's zero cell.
__device__ __forceinline__ int pipe_row(const DevParams &P, int j, int p)
{
    return (j < P.Np && j != p) ? j : (p == 0 && P.Np > 1 ? 1 : 0);
}

// Image of the VTable used by the branch-free evaluation (LDS copy in K1's pipe kernels, a global copy for
// the device-resident sampler): [VT(0)] VT(0) .. VT(Nmax+1) [0 0 0 0]
//   * the leading copy of VT(0) stands for the reference's clamp max(ix-2,0) at r < dr;
//   * the trailing zeros are the "zero cell": a lane whose distance is outside the cutoff (or that has no
//     partner) looks up cell zc = Nmax+3 and so contributes exactly 0 to every sum -- no weights, no selects
//     on the results.
struct PipeTab {
    const double *p;      // -> VT(0) inside LDS
    int zc;
};

// r2 of a lane measuring the moved particle against its own row is exactly 0 and v_rsq_f64(0) = +Inf would poison the
// (masked) lane's lookup with NaN.  r2 + 1e-300 is r2 itself, bit for bit, for every r2 >= 1e-284 (the addend is far below
// half an ulp) and 1e-300 for r2 = 0: ONE v_add_f64 where fmax(r2, 1e-300) compiles to two v_max_f64 (the first
// canonicalises a possible signalling NaN) -- 2 of the 74 (even) / 102 (odd) instructions of a pass.
__device__ __forceinline__ double floor_r2(double r2) { return r2 + 1e-300; }

// one distance, branch-free and weight-free (see PipeTab).  r2 must be finite and > 0 on every lane (the caller
// floors it at 1e-300: a lane measuring the moved particle against its own row has r2 = 0).
//   r   = sqrt(r2) from v_rsq_f64 (2^-24) + one coupled Newton step + one residual correction (< 1 ulp)
//   t   = r/dr,  i0 = int(t) = ix-1 of the reference,  f = fract(t)
//   V   = F0 + f (F1-F0);   dV/dr * dr = (F1-Fm) + f ((Fp-F1) - (F0-Fm))   [= Fafter - Fbefore of interpolate.f90]
//   force term (dV/dr)/r * x_k with 1/r = 2h from the same Newton step
template <int DIM, int CLS, bool IS_OLD>
__device__ __forceinline__ void pipe_pair(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                          double r2, bool in, const double (&d)[DIM], Acc<DIM, CLS> &A)
{
    const double y0 = __builtin_amdgcn_rsq(r2);
    double g = r2 * y0;
    double h = 0.5 * y0;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    const double d0 = __builtin_fma(-g, g, r2);
    g = __builtin_fma(d0, h, g);
    const double t  = g * P.rdr;
    const int    it = (int)t;
    const int    i0 = in ? it : VT.zc;
    const double f  = __builtin_amdgcn_fract(t);
    const double *V = VT.p + i0;
    const double F0 = V[0], F1 = V[1];
    // f F1 + (1-f) F0 as two products: a table head of +Inf (singular potentials, r < 2 dr) then gives +Inf as the reference's
    // (a1 F(ix) + a2 F(ix-1))/dx does, where F0 + f (F1 - F0) gave Inf - Inf = NaN (round 3 fuzz); one instruction more per distance
    const double omf = 1.0 - f;
    const double v   = __builtin_fma(f, F1, omf * F0);
    if (IS_OLD) A.potO = A.potO + v; else A.potN = A.potN + v;
    if (CLS == CLS_ODD) {
        const double Fm = V[-1], Fp = V[2];
        const double D  = __builtin_fma(f, Fp, omf * F1) - __builtin_fma(f, F0, omf * Fm);    // Fafter - Fbefore of interpolate.f90, in cell units
        const double s  = D * (h * P.rdr);                                // ((Fafter-Fbefore)*0.5/dr) * (2h)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            if (IS_OLD) A.fO[k] = __builtin_fma(s, d[k], A.fO[k]); else A.fN[k] = __builtin_fma(s, d[k], A.fN[k]);
        }
    }
    if (CLS == CLS_END) {                                                 // LogWF stays in global memory (2 of 161 beads)
        double u;
        if (P.wf_table) {
#ifdef PIGS_EXP_NOWF                                                          // timing experiment only: no LogWF gather
            u = in ? f : 0.0;
#else
            const double *U = WF + (in ? it : 0);
            const double u0 = U[0], u1 = U[1];
            u = in ? __builtin_fma(f, u1, omf * u0) : 0.0;                    // (1-f)*(-Inf) keeps the -Inf head (Q4)
#endif
        } else {
            u = in ? log_psi(0, P.Rm, g) : 0.0;                               // analytic trial function (wf_table = F)
        }
        if (IS_OLD) A.psiO = A.psiO + u; else A.psiN = A.psiN + u;
    }
}


// ---- branch-free item evaluation on a PipeTab (periodic systems, Np <= 256) --------------------------------
// one pass of 64 partners: both distances as independent chains; rjm = this lane's partner coordinates
template <int DIM, int CLS>
__device__ __forceinline__ void pipe_pass_at(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                             int p, int j, const double (&rjm)[DIM], const double (&xn)[DIM],
                                             const double (&xo)[DIM], Acc<DIM, CLS> &A)
{
    const bool valid = j < P.Np && j != p;                            // row p itself never enters (vpi_mod.f90:2699)
    double dn[DIM], dold[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        // opaque per class: keeps the optimiser from hoisting the distance arithmetic of all passes above the
        // class branch (it did: 24 live doubles more, spills)
        double rj = rjm[k];
        asm volatile("; class %1 pass" : "+v"(rj) : "n"(CLS));
        dn[k] = xn[k] - rj; dold[k] = xo[k] - rj;
    }
    const double r2o = min_image_rn<DIM>(dold, P);
    const double r2n = min_image_rn<DIM>(dn, P);
    pipe_pair<DIM, CLS, false>(P, VT, WF, floor_r2(r2n), valid && r2n <= P.rcut2, dn, A);
    pipe_pair<DIM, CLS, true>(P, VT, WF, floor_r2(r2o), valid && r2o <= P.rcut2, dold, A);
    __builtin_amdgcn_sched_barrier(0);                                // two chains in flight (VGPRs)
}

template <int DIM, int CLS>
__device__ __forceinline__ void item_direct_pipe(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                                 const double *__restrict__ S, int p, const double (&xn)[DIM],
                                                 const double (&xo)[DIM], int lane, int b, double *red,
                                                 double *out, double *parts)
{
    constexpr int MAXP = 4;
    Acc<DIM, CLS> A;
    double rj[MAXP][DIM];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        const int j = m * kWave + lane;
        const int jj = pipe_row(P, j, p);
#pragma unroll
        for (int k = 0; k < DIM; ++k) rj[m][k] = S[(size_t)k * P.NpPad + jj];
    }
#pragma unroll
    for (int m = 0; m < MAXP; ++m) pipe_pass_at<DIM, CLS>(P, VT, WF, p, m * kWave + lane, rj[m], xn, xo, A);
    finish_item<DIM, CLS>(P, lane, b, A, red, out, parts);
}

template <int DIM>
__device__ __forceinline__ void item_eval_pipe(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                               const double *__restrict__ S, int p, int b, const double (&xn)[DIM],
                                               const double (&xo)[DIM], int lane, double *red, double *out, double *parts)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       item_direct_pipe<DIM, CLS_ODD>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else if (endb) item_direct_pipe<DIM, CLS_END>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
    else           item_direct_pipe<DIM, CLS_EVEN>(P, VT, WF, S, p, xn, xo, lane, b, red, out, parts);
}

// split form (see item_pass_cls): ONE pass of an item, wave totals into tot8[]
template <int DIM, int CLS>
__device__ __forceinline__ void item_pass_pipe_cls(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                                   const double *__restrict__ S, int p, int m,
                                                   const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                                   double *red, double *tot8)
{
    Acc<DIM, CLS> A;
    const int j = m * kWave + lane;
    const int jj = pipe_row(P, j, p);
    double rj[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) rj[k] = S[(size_t)k * P.NpPad + jj];
    pipe_pass_at<DIM, CLS>(P, VT, WF, p, j, rj, xn, xo, A);
    if (CLS == CLS_ODD) {
        double v[8] = {A.potN, A.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DIM; ++k) { v[2 + k] = A.fN[k]; v[5 + k] = A.fO[k]; }
        const double t = wave_reduce_lds<8>(v, red, lane);
        if (lane < 8) tot8[lane] = t;
    } else if (CLS == CLS_END) {
        const double v[4] = {A.potN, A.potO, A.psiN, A.psiO};
        const double t = wave_reduce_lds<4>(v, red, lane);
        if (lane < 4) tot8[lane] = t;
    } else {
        const double v[2] = {A.potN, A.potO};
        const double t = wave_reduce_lds<2>(v, red, lane);
        if (lane < 2) tot8[lane] = t;
    }
}

template <int DIM>
__device__ __forceinline__ void item_pass_pipe(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                               const double *__restrict__ S, int p, int b, int m,
                                               const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                               double *red, double *tot8)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       item_pass_pipe_cls<DIM, CLS_ODD>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
    else if (endb) item_pass_pipe_cls<DIM, CLS_END>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
    else           item_pass_pipe_cls<DIM, CLS_EVEN>(P, VT, WF, S, p, m, xn, xo, lane, red, tot8);
}

// ---- task form for the device-resident sampler's replicated stages (pigs_sampler.hip) ----------------------
// A task is a run of `np` 64-partner passes of one proposal bead, starting at pass m0, for the new distance
// (sides & 1), the old one (sides & 2) or both; its wave totals go to tot8[] in the layout of item_pass_cls.
// np, m0 and sides are wave-uniform (scalar branches); partner coordinates of up to four passes are
// requested up front.  A side that is not evaluated leaves zeros in its columns, so item_finish_split() can
// add the tasks of a bead column by column whatever their shape.
template <int DIM, int CLS>
__device__ __forceinline__ void pipe_task_cls(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                              const double *__restrict__ S, int p, int m0, int np, int sides,
                                              const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                              double *red, double *tot8)
{
    constexpr int MAXP = 4;
    Acc<DIM, CLS> A;
    for (int mb = m0; mb < m0 + np; mb += MAXP) {                     // one trip for Np <= 256
        const int nn = m0 + np - mb < MAXP ? m0 + np - mb : MAXP;
        double rj[MAXP][DIM];
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            if (m < nn) {
                const int j = (mb + m) * kWave + lane;
                const int jj = pipe_row(P, j, p);
#pragma unroll
                for (int k = 0; k < DIM; ++k) rj[m][k] = S[(size_t)k * P.NpPad + jj];
            }
        }
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            if (m < nn) {
                const int j = (mb + m) * kWave + lane;
                const bool valid = j < P.Np && j != p;                // row p itself never enters (vpi_mod.f90:2699)
                if (sides & 1) {
                    double d[DIM];
#pragma unroll
                    for (int k = 0; k < DIM; ++k) d[k] = xn[k] - rj[m][k];
                    const double r2 = min_image_rn<DIM>(d, P);
                    pipe_pair<DIM, CLS, false>(P, VT, WF, floor_r2(r2), valid && r2 <= P.rcut2, d, A);
                }
                if (sides & 2) {
                    double d[DIM];
#pragma unroll
                    for (int k = 0; k < DIM; ++k) d[k] = xo[k] - rj[m][k];
                    const double r2 = min_image_rn<DIM>(d, P);
                    pipe_pair<DIM, CLS, true>(P, VT, WF, floor_r2(r2), valid && r2 <= P.rcut2, d, A);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (CLS == CLS_ODD) {
        double v[8] = {A.potN, A.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DIM; ++k) { v[2 + k] = A.fN[k]; v[5 + k] = A.fO[k]; }
        const double t = wave_reduce_lds<8>(v, red, lane);
        if (lane < 8) tot8[lane] = t;
    } else if (CLS == CLS_END) {
        const double v[4] = {A.potN, A.potO, A.psiN, A.psiO};
        const double t = wave_reduce_lds<4>(v, red, lane);
        if (lane < 4) tot8[lane] = t;
    } else {
        const double v[2] = {A.potN, A.potO};
        const double t = wave_reduce_lds<2>(v, red, lane);
        if (lane < 2) tot8[lane] = t;
    }
}

// The same task with the passes as a rolled loop (one pass per trip: ~1/8 of the code of pipe_task_cls).  The stage
// machine's tasks are 1-4 passes long, and its stage loop must fit the 64 KB instruction cache that two CUs share.
template <int DIM, int CLS>
__device__ __forceinline__ void pipe_task_rolled_cls(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                                               const double *__restrict__ S, int p, int m0, int np, int sides,
                                                               const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                                               double *red, double *tot8, unsigned long long *tsub = nullptr)
{
    Acc<DIM, CLS> A;
    double rn[DIM];
    unsigned long long q0 = 0, q1 = 0, q2 = 0;
    if (tsub) q0 = __builtin_amdgcn_s_memtime();
    {
        const int jj = pipe_row(P, m0 * kWave + lane, p);
#pragma unroll
        for (int k = 0; k < DIM; ++k) rn[k] = S[(size_t)k * P.NpPad + jj];
    }
    if (tsub) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q1 = __builtin_amdgcn_s_memtime(); }
#pragma nounroll
    for (int m = m0; m < m0 + np; ++m) {
        double rj[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) rj[k] = rn[k];
        if (m + 1 < m0 + np) {                                        // the next pass's partners are on their way
            const int jj = pipe_row(P, (m + 1) * kWave + lane, p);
#pragma unroll
            for (int k = 0; k < DIM; ++k) rn[k] = S[(size_t)k * P.NpPad + jj];
        }
        const int j = m * kWave + lane;
        const bool valid = j < P.Np && j != p;                        // row p itself never enters (vpi_mod.f90:2699)
        if (sides & 1) {
            double d[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) d[k] = xn[k] - rj[k];
            const double r2 = min_image_rn<DIM>(d, P);
            pipe_pair<DIM, CLS, false>(P, VT, WF, floor_r2(r2), valid && r2 <= P.rcut2, d, A);
        }
        if (sides & 2) {
            double d[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) d[k] = xo[k] - rj[k];
            const double r2 = min_image_rn<DIM>(d, P);
            pipe_pair<DIM, CLS, true>(P, VT, WF, floor_r2(r2), valid && r2 <= P.rcut2, d, A);
        }
    }
    if (tsub) { asm volatile("" :: "v"(A.potN), "v"(A.potO)); q2 = __builtin_amdgcn_s_memtime(); }
    if (CLS == CLS_ODD) {
        double v[8] = {A.potN, A.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DIM; ++k) { v[2 + k] = A.fN[k]; v[5 + k] = A.fO[k]; }
        const double t = wave_reduce_lds<8>(v, red, lane);
        if (lane < 8) tot8[lane] = t;
    } else if (CLS == CLS_END) {
        const double v[4] = {A.potN, A.potO, A.psiN, A.psiO};
        const double t = wave_reduce_lds<4>(v, red, lane);
        if (lane < 4) tot8[lane] = t;
    } else {
        const double v[2] = {A.potN, A.potO};
        const double t = wave_reduce_lds<2>(v, red, lane);
        if (lane < 2) tot8[lane] = t;
    }
    if (tsub && threadIdx.x == 0) { const unsigned long long q3 = __builtin_amdgcn_s_memtime(); tsub[0] += q1 - q0; tsub[1] += q2 - q1; tsub[2] += q3 - q2; }
}

template <int DIM>
__device__ __forceinline__ void pipe_task_rolled(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                                 const double *__restrict__ S, int p, int b, int m0, int np, int sides,
                                                 const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                                 double *red, double *tot8, unsigned long long *tsub = nullptr)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       pipe_task_rolled_cls<DIM, CLS_ODD>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8, tsub);
    else if (endb) pipe_task_rolled_cls<DIM, CLS_END>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8, tsub);
    else           pipe_task_rolled_cls<DIM, CLS_EVEN>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8, tsub);
}

template <int DIM>
__device__ __forceinline__ void pipe_task(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                          const double *__restrict__ S, int p, int b, int m0, int np, int sides,
                                          const double (&xn)[DIM], const double (&xo)[DIM], int lane,
                                          double *red, double *tot8)
{
    const bool odd  = (b & 1) != 0;
    const bool endb = (b == 0) || (b == 2 * P.Nb);
    if (odd)       pipe_task_cls<DIM, CLS_ODD>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8);
    else if (endb) pipe_task_cls<DIM, CLS_END>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8);
    else           pipe_task_cls<DIM, CLS_EVEN>(P, VT, WF, S, p, m0, np, sides, xn, xo, lane, red, tot8);
}

} // namespace pigs
