// pigs_k1.hip -- K1: batched Delta S of proposal beads on gfx950.
//
// Replaces the reference's `call UpdateAction` (vpi_mod.f90:2491-2530) with its callees
// UpdatePot (2660-2841), UpdateWf (2534-2656) and GreenFunction opt 0 (global_mod.f90:19-72),
// batched over (walker, particle, bead) proposal items.
//
// Work decomposition: ONE wave64 per item (an item's 255 partners are 4 lane-strided passes
// over one contiguous 6 KB SoA slice: 512-B coalesced loads per coordinate); no workgroup
// barrier anywhere in the item loop.  Kernels in this file (pigs_set_tuning("k1_variant"); measurements in
// DESIGN.md section 4 and profiles/; the A/B history of the forms that lost -- LDS table in the plain grid,
// in-cutoff compaction, prefetch-only, the first persistent kernel -- is archived in profiles/r01_k1_variants_ab*.txt):
//    1 v1    plain statement (IEEE `/`, sqrt(), tables gathered from global memory)
//    2 v2    exact-term: exact short division / fused sqrt+1/r (pigs_device.h), one LDS-transpose reduction for all
//            accumulators.  v1 and v2 sum bit-identical per-pair terms; only the summation order differs.
//    7 / 8   v2 with the short arithmetic on the global table (FastTab; 8: partner loads issued up front) -- the
//            form for Np > 256
//   12 pipe2 persistent, one 1024-thread workgroup per CU, table image with its zero cell in LDS, branch-free short
//            arithmetic, per-workgroup item queue, item records and partner coordinates requested ahead: the default
//            for large launches of a periodic system with Np <= 256
//   13 grid  the SAME per-item arithmetic as pipe2 (pipe_pair on the global table image, passes and sides in the same
//            order) on a plain grid: what small launches of such a system run, so that an item's Delta S does not
//            depend on how many other items share its launch
//   14 reference order (validation): exact-term arithmetic, per-partner terms parked in LDS and added by one lane
//            per accumulator in the reference's jp order -- every bit of Delta S equals the reference's
#include "pigs_device.h"
#include "pigs_k1_device.h"
#include "pigs_kernels.h"

namespace pigs {

// =====================================================================================
// K1 v1 (plain form, kept as the readable reference of the kernel and for A/B timing):
// one wave64 per proposal item; lane l visits partners jp = l, l+64, ... of the
// item's slice (unit-stride 512-B loads per coordinate), accumulates its partial sums
// in registers, then one butterfly per accumulator.  Wave-uniform control flow per item
// (bead parity / end bead), so no divergence except the physical cutoff test.
// Algorithmic bytes per item: dim*Np*8 (slice) + 2*dim*8 (xnew,xold) + 8 (DeltaS)
// (+12 B of indices), i.e. 6 200 B at Np=256 for Np-1=255 bead-pair evaluations.
// =====================================================================================
template <int DIM, bool TRAP>
__global__ __launch_bounds__(256) void k_delta_action_v1(
    DevParams P, const double *__restrict__ paths, const double *__restrict__ VT,
    const double *__restrict__ WF, int n_items, const int32_t *__restrict__ walker,
    const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
    const double *__restrict__ xnew, const double *__restrict__ xold,
    double *__restrict__ out, double *__restrict__ parts)
{
    const int lane  = threadIdx.x & (kWave - 1);
    const int wid   = threadIdx.x >> 6;
    const int nwave = gridDim.x * (blockDim.x >> 6);
    const size_t sl = slice_doubles(DIM, P.NpPad);

    for (int item = blockIdx.x * (blockDim.x >> 6) + wid; item < n_items; item += nwave) {
        const int it = __builtin_amdgcn_readfirstlane(item);
        const int w  = walker[it];
        const int p  = ipv[it] - 1;          // 0-based moved particle
        const int b  = ibv[it];
        if ((unsigned)w >= (unsigned)P.nW || (unsigned)p >= (unsigned)P.Np || (unsigned)b >= (unsigned)P.M) {
            if (lane == 0) out[it] = __builtin_nan("");      // bad index: never touch memory with it
            continue;
        }
        double xn[DIM], xo[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            xn[k] = xnew[(size_t)it * DIM + k];
            xo[k] = xold[(size_t)it * DIM + k];
        }
        const double *S   = paths + ((size_t)w * P.M + b) * sl;
        const bool odd    = (b & 1) != 0;                    // UpdateAction: force term on odd beads
        const bool endb   = (b == 0) || (b == 2 * P.Nb);     // UpdateWf only on the two end beads

        double potN = 0.0, potO = 0.0, psiN = 0.0, psiO = 0.0;
        double fN[DIM], fO[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) { fN[k] = 0.0; fO[k] = 0.0; }

        if (TRAP && lane == 0) {                              // vpi_mod.f90:2688-2695, 2555-2560
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                potN = potN + trap_pot(0, P.a_ho[k], xn[k]);
                potO = potO + trap_pot(0, P.a_ho[k], xo[k]);
                fO[k] = trap_pot(1, P.a_ho[k], xo[k]);
                fN[k] = trap_pot(1, P.a_ho[k], xn[k]);
                if (endb) {
                    psiO = psiO + trap_psi(0, P.a_ho[k], xo[k]);
                    psiN = psiN + trap_psi(0, P.a_ho[k], xn[k]);
                }
            }
        }

        for (int j0 = 0; j0 < P.Np; j0 += kWave) {
            const int j = j0 + lane;
            if (j < P.Np && j != p) {                        // vpi_mod.f90:2699: never read row ip
                double dnew[DIM], dold[DIM];
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    const double rj = S[(size_t)k * P.NpPad + j];
                    dnew[k] = xn[k] - rj;                      // :2706
                    dold[k] = xo[k] - rj;                      // :2707
                }
                double r2n, r2o;
                if (TRAP) { r2o = plain_r2<DIM>(dold); r2n = plain_r2<DIM>(dnew); }
                else      { r2o = min_image<DIM>(dold, P); r2n = min_image<DIM>(dnew, P); }

                if (TRAP || r2n <= P.rcut2) {                // :2723 (Q5) / :2771
                    const double r = sqrt(r2n);
                    const Lerp L = lerp_setup(r, P.dr, P.Nmax);
                    potN = potN + interp0(VT, L, P.dr);
                    if (odd) {
                        const double dv = interp1(VT, L, P.dr);
#pragma unroll
                        for (int k = 0; k < DIM; ++k) fN[k] = fN[k] + dv * dnew[k] / r;   // :2784
                    }
                    if (endb) psiN = psiN + (P.wf_table ? interp0(WF, L, P.dr) : log_psi(0, P.Rm, r));   // :2637-2641
                }
                if (r2o <= P.rcut2) {                        // :2745 / :2795
                    const double r = sqrt(r2o);
                    const Lerp L = lerp_setup(r, P.dr, P.Nmax);
                    potO = potO + interp0(VT, L, P.dr);
                    if (odd) {
                        const double dv = interp1(VT, L, P.dr);
#pragma unroll
                        for (int k = 0; k < DIM; ++k) fO[k] = fO[k] + dv * dold[k] / r;   // :2808
                    }
                    if (endb) psiO = psiO + (P.wf_table ? interp0(WF, L, P.dr) : log_psi(0, P.Rm, r));   // :2623-2627
                } else if (TRAP && endb) {
                    // UpdateWf's trap branch has no cutoff on either distance (vpi_mod.f90:2595-2615)
                    const double r = sqrt(r2o);
                    const Lerp L = lerp_setup(r, P.dr, P.Nmax);
                    psiO = psiO + (P.wf_table ? interp0(WF, L, P.dr) : log_psi(0, P.Rm, r));
                }
            }
        }

        potN = wave_sum(potN);
        potO = wave_sum(potO);
        double dF2 = 0.0, dPsi = 0.0;
        if (odd) {
            double fn2 = 0.0, fo2 = 0.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                const double a = wave_sum(fN[k]);
                const double c = wave_sum(fO[k]);
                fn2 = fn2 + a * a;                            // :2831
                fo2 = fo2 + c * c;                            // :2832
            }
            dF2 = fn2 - fo2;                                  // :2835
        }
        if (endb) {
            psiN = wave_sum(psiN);
            psiO = wave_sum(psiO);
            dPsi = psiN - psiO;                               // :2653
        }
        if (lane == 0) {
            const double dPot = potN - potO;                  // :2838
            out[it] = -dPsi + green_function(0, b, P.Nb, P.dt, dPot, dF2);   // :2527
            if (parts) {
                parts[(size_t)it * 3 + 0] = dPot;
                parts[(size_t)it * 3 + 1] = dF2;
                parts[(size_t)it * 3 + 2] = dPsi;
            }
        }
    }
}


// =====================================================================================
// K1 v2 (exact-term; FAST: short arithmetic on the global table)
// =====================================================================================
// LDS (dynamic): per wave kWaveLds bytes of reduction scratch (8 x 65 doubles)
template <int DIM, bool TRAP, int BLOCK, bool PREFETCH = false, bool FAST = false>
__global__ __launch_bounds__(BLOCK) void k_delta_action_v2(
    DevParams P, const double *__restrict__ paths, const double *__restrict__ VT,
    const double *__restrict__ WF, int n_items, const int32_t *__restrict__ walker,
    const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
    const double *__restrict__ xnew, const double *__restrict__ xold,
    double *__restrict__ out, double *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane  = threadIdx.x & (kWave - 1);
    const int wid   = threadIdx.x >> 6;
    const int nwave = gridDim.x * (BLOCK >> 6);
    const size_t sl = slice_doubles(DIM, P.NpPad);
    double *red = reinterpret_cast<double *>(smem + (size_t)wid * kWaveLds);

    for (int item = blockIdx.x * (BLOCK >> 6) + wid; item < n_items; item += nwave) {
        const int it = __builtin_amdgcn_readfirstlane(item);
        const int w  = walker[it];
        const int p  = ipv[it] - 1;
        const int b  = ibv[it];
        if ((unsigned)w >= (unsigned)P.nW || (unsigned)p >= (unsigned)P.Np || (unsigned)b >= (unsigned)P.M) {
            if (lane == 0) out[it] = __builtin_nan("");
            continue;
        }
        double xn[DIM], xo[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            xn[k] = xnew[(size_t)it * DIM + k];
            xo[k] = xold[(size_t)it * DIM + k];
        }
        const double *S = paths + ((size_t)w * P.M + b) * sl;
        double *o = out + it;
        double *q = parts ? parts + (size_t)it * 3 : nullptr;
        if (FAST && !TRAP) {                                    // short arithmetic (pigs_device.h, FastTab)
            if (PREFETCH) item_eval_prefetch<DIM, TRAP>(P, FastTab{VT}, WF, S, p, b, xn, xo, lane, red, o, q);
            else          item_eval<DIM, TRAP>(P, FastTab{VT}, WF, S, p, b, xn, xo, lane, red, o, q);
        } else if (PREFETCH) {
            item_eval_prefetch<DIM, TRAP>(P, VT, WF, S, p, b, xn, xo, lane, red, o, q);
        } else {
            item_eval<DIM, TRAP>(P, VT, WF, S, p, b, xn, xo, lane, red, o, q);
        }
    }
}

// =====================================================================================
// K1 "grid" (variant 13): pipe2's per-item arithmetic on a plain grid -- pipe_pair on the global table image
// (pigs_k1_device.h: item_eval_pipe = the four passes in order, new distance then old, the same accumulators and
// the same reduction as pipe2_item).  Small launches of a periodic system run this, large ones pipe2: the bits of an
// item's Delta S do not depend on the launch it travels in (a walker's Metropolis chain in the host-driven sampler
// must not depend on how many walkers share the stage or on how they are sharded over GPUs).
// =====================================================================================
template <int DIM>
__global__ __launch_bounds__(256) void k_delta_action_grid(
    DevParams P, const double *__restrict__ paths, const double *__restrict__ VTimg,
    const double *__restrict__ WF, int n_items, const int32_t *__restrict__ walker,
    const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
    const double *__restrict__ xnew, const double *__restrict__ xold,
    double *__restrict__ out, double *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane  = threadIdx.x & (kWave - 1);
    const int wid   = threadIdx.x >> 6;
    const int nwave = gridDim.x * 4;
    const size_t sl = slice_doubles(DIM, P.NpPad);
    double *red = reinterpret_cast<double *>(smem + (size_t)wid * kWaveLds);
    const PipeTab VT{VTimg + 2, P.Nmax + 3};

    for (int item = blockIdx.x * 4 + wid; item < n_items; item += nwave) {
        const int it = __builtin_amdgcn_readfirstlane(item);
        const int w  = walker[it];
        const int p  = ipv[it] - 1;
        const int b  = ibv[it];
        if ((unsigned)w >= (unsigned)P.nW || (unsigned)p >= (unsigned)P.Np || (unsigned)b >= (unsigned)P.M) {
            if (lane == 0) out[it] = __builtin_nan("");
            continue;
        }
        double xn[DIM], xo[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            xn[k] = xnew[(size_t)it * DIM + k];
            xo[k] = xold[(size_t)it * DIM + k];
        }
        item_eval_pipe<DIM>(P, VT, WF, paths + ((size_t)w * P.M + b) * sl, p, b, xn, xo, lane, red, out + it,
                            parts ? parts + (size_t)it * 3 : nullptr);
    }
}

// =====================================================================================
// K1 "reference order" (variant 14, validation): one wave per item, one item per workgroup.  Every partner's terms
// are computed with the exact-term arithmetic (bit-identical to the reference's per-pair terms) and parked in LDS
// (Np x 8 doubles); then lane q adds column q over jp = 1..Np in the reference's order (vpi_mod.f90:2697-2823:
// PotNew, PotOld, Fnew(k), Fold(k); UpdateWf 2580-2650: PsiNew, PsiOld), starting from the trap's one-body terms
// exactly where the reference starts from them.  A partner outside the cutoff parks +0.0, which leaves a sum's bits
// alone.  Delta S then equals the reference's bit for bit -- BASELINE config 2's "pair-action kernel vs CPU
// bit-compare" taken literally (tests/test_gpu_parity.py::test_reference_order_kernel_is_bit_identical).
// =====================================================================================
template <int DIM, bool TRAP, int CLS>
__device__ __forceinline__ void reforder_item(const DevParams &P, const double *__restrict__ VT, const double *__restrict__ WF,
                                              const double *__restrict__ S, int p, int b, const double (&xn)[DIM],
                                              const double (&xo)[DIM], int lane, double *T, double *out, double *parts)
{
    // columns: 0 potN 1 potO 2..4 fN 5..7 fO (odd) / 2 psiN 3 psiO (end)
    for (int j = lane; j < P.Np; j += kWave) {
        Acc<DIM, CLS> A;
        if (j != p) {
            double rj[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) rj[k] = S[(size_t)k * P.NpPad + j];
            partner_accumulate_at<DIM, TRAP, CLS>(P, VT, WF, rj, xn, xo, A);
        }
        double *t = T + (size_t)j * 8;
        t[0] = A.potN; t[1] = A.potO;
        if (CLS == CLS_ODD) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) { t[2 + k] = A.fN[k]; t[5 + k] = A.fO[k]; }
        } else if (CLS == CLS_END) {
            t[2] = A.psiN; t[3] = A.psiO;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    double s = 0.0;
    if (lane < 8) {
        if (TRAP) {                                               // the one-body terms come first: vpi_mod.f90:2688-2695, 2555-2560
            Acc<DIM, CLS> A0;
            trap_terms<DIM, TRAP, CLS>(P, xn, xo, A0);
            double col[8] = {A0.potN, A0.potO, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (CLS == CLS_ODD) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) { col[2 + k] = A0.fN[k]; col[5 + k] = A0.fO[k]; }
            } else if (CLS == CLS_END) {
                col[2] = A0.psiN; col[3] = A0.psiO;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (lane == q) s = col[q];
        }
        for (int j = 0; j < P.Np; ++j) {
            if (j == p) continue;                                 // vpi_mod.f90:2699
            s = s + T[(size_t)j * 8 + lane];
        }
    }
    const double c0 = read_lane(s, 0), c1 = read_lane(s, 1);
    double dF2 = 0.0, dPsi = 0.0;
    if (CLS == CLS_ODD) {
        double fn2 = 0.0, fo2 = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            const double a = read_lane(s, 2 + k), c = read_lane(s, 5 + k);
            fn2 = fn2 + a * a;                                    // :2831-2832
            fo2 = fo2 + c * c;
        }
        dF2 = fn2 - fo2;                                          // :2835
    } else if (CLS == CLS_END) {
        dPsi = read_lane(s, 2) - read_lane(s, 3);                 // :2653
    }
    const double dPot = c0 - c1;                                  // :2838
    if (lane == 0) {
        *out = -dPsi + green_function(0, b, P.Nb, P.dt, dPot, dF2);   // :2527 (plain IEEE divisions)
        if (parts) { parts[0] = dPot; parts[1] = dF2; parts[2] = dPsi; }
    }
}

template <int DIM, bool TRAP>
__global__ __launch_bounds__(64) void k_delta_action_reforder(
    DevParams P, const double *__restrict__ paths, const double *__restrict__ VT,
    const double *__restrict__ WF, int n_items, const int32_t *__restrict__ walker,
    const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
    const double *__restrict__ xnew, const double *__restrict__ xold,
    double *__restrict__ out, double *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *T = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x;
    const size_t sl = slice_doubles(DIM, P.NpPad);
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const int w = walker[it], p = ipv[it] - 1, b = ibv[it];
        if ((unsigned)w >= (unsigned)P.nW || (unsigned)p >= (unsigned)P.Np || (unsigned)b >= (unsigned)P.M) {
            if (lane == 0) out[it] = __builtin_nan("");
            continue;
        }
        double xn[DIM], xo[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            xn[k] = xnew[(size_t)it * DIM + k];
            xo[k] = xold[(size_t)it * DIM + k];
        }
        const double *S = paths + ((size_t)w * P.M + b) * sl;
        double *o = out + it;
        double *q = parts ? parts + (size_t)it * 3 : nullptr;
        const bool odd  = (b & 1) != 0;
        const bool endb = (b == 0) || (b == 2 * P.Nb);
        if (odd)       reforder_item<DIM, TRAP, CLS_ODD>(P, VT, WF, S, p, b, xn, xo, lane, T, o, q);
        else if (endb) reforder_item<DIM, TRAP, CLS_END>(P, VT, WF, S, p, b, xn, xo, lane, T, o, q);
        else           reforder_item<DIM, TRAP, CLS_EVEN>(P, VT, WF, S, p, b, xn, xo, lane, T, o, q);
        __syncthreads();
    }
}

// =====================================================================================
// Pieces of the persistent kernel: the LDS image of the VTable
// =====================================================================================
__device__ __forceinline__ size_t pipe_tab_bytes(int nt) { return ((size_t)(nt + 6) * sizeof(double) + 15) & ~(size_t)15; }

// Staging the table image with the LDS-DMA form of the global load (global_load_lds_dwordx4: 16 bytes per lane
// straight into LDS at M0 + lane*16, no VGPRs, no ds_write): issued as the very first instructions of the
// kernel, ahead of the HBM requests of the first items; completion is a vmcnt matter, so ONE `s_waitcnt vmcnt(0)`
// before the workgroup barrier covers it (the first item needs its own loads by then anyway).  A wave issues the
// chunk [u*blockDim + wid*64, +64) only if it starts inside the table; lanes past the end are masked off.

__device__ __forceinline__ void pipe_table_dma_issue(unsigned char *smem, const double *__restrict__ VTg, int nt, int wid, int lane)
{
#ifndef PIGS_EXPERIMENT_NO_STAGE
    const double2 *src = reinterpret_cast<const double2 *>(VTg);
    const int n2 = nt / 2;
    for (int cb = wid * kWave; cb < n2; cb += (int)blockDim.x) {      // wave-uniform
        const int idx = cb + lane;
        auto *l = reinterpret_cast<__attribute__((address_space(3))) void *>(
            (__attribute__((address_space(3))) unsigned char *)smem + 16 + (size_t)cb * 16);
        // lanes past the end of the table are masked off (EXEC): they neither load nor write their LDS slot
        if (idx < n2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx), l, 16, 0, 0);
    }
#endif
}

// after this and a workgroup barrier the image is complete
__device__ __forceinline__ PipeTab pipe_table_dma_finish(unsigned char *smem, const double *__restrict__ VTg, int nt)
{
    double *base = reinterpret_cast<double *>(smem);          // base[1] = leading copy, base[2..] = table
    double *tab  = base + 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        if (nt & 1) tab[nt - 1] = VTg[nt - 1];
        base[0] = 0.0; base[1] = tab[0];                          // wave 0 staged chunk 0 itself: tab[0] has landed
        tab[nt] = 0.0; tab[nt + 1] = 0.0; tab[nt + 2] = 0.0; tab[nt + 3] = 0.0;
    }
    return PipeTab{tab, nt + 1};
}

// =====================================================================================
// K1 "pipe2": the short-arithmetic item evaluation as a persistent, software-pipelined kernel (PBC, Np <= 256).
// One 1024-thread workgroup per CU keeps the VTable image in LDS -- the table gather was what kept the texture
// addresser 60-65 % busy in v2 and held every wave on s_waitcnt; LDS gathers count on lgkmcnt, so they never wait
// for outstanding global loads -- and its 16 waves draw the workgroup's items (blockIdx + k*gridDim: one walker's
// consecutive beads spread over all CUs) from an LDS counter, because odd beads cost more than even ones.  The two
// distances of a pass are branch-free, independent chains (masked lanes look up the table's zero cell).  A wave
// that started every item with its dependent chain -- scalar loads of the item record, then the slice loads whose
// address they give, ~3 us -- reached 78 % VALU occupancy; here (a) the record of the item AFTER the next one is requested with vector loads
// (lanes 0..2: walker/ip/ib, lanes 0..2*DIM-1: xnew,xold; vmcnt is in order, unlike scalar loads it does not
// force an lgkmcnt drain at the next LDS gather) and decoded one item later with v_readlane, (b) the
// partner coordinates run two passes ahead: pass m+2 of this item, then passes 0/1 of the next item, are
// requested before pass m is evaluated, in the same 4 x DIM registers.
// =====================================================================================
template <int DIM>
struct ItemMeta {                                   // wave-uniform (SGPRs)
    int    p, b, ok;
    const double *S;
    double xn[DIM], xo[DIM];
};

struct ItemRaw {
    int    i;      // lane 0 walker, 1 ip, 2 ib
    double x;      // lanes 0..DIM-1 xnew, DIM..2*DIM-1 xold
};

template <int DIM>
__device__ __forceinline__ ItemRaw pipe2_request(int it, const int32_t *__restrict__ walker,
                                                 const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
                                                 const double *__restrict__ xnew, const double *__restrict__ xold, int lane)
{
    ItemRaw r;
    const int32_t *ai = lane == 0 ? walker : (lane == 1 ? ipv : ibv);
    r.i = ai[it];
    const int k = lane < DIM ? lane : (lane < 2 * DIM ? lane - DIM : 0);
    const double *ax = lane < DIM ? xnew : xold;
    r.x = ax[(size_t)it * DIM + k];
    return r;
}

__device__ __forceinline__ double bcast_lane(double v, int lane_id)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane_id);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane_id);
    return u.d;
}

template <int DIM>
__device__ __forceinline__ void pipe2_decode(const DevParams &P, const double *__restrict__ paths, const ItemRaw &r,
                                             size_t sl, ItemMeta<DIM> &R)
{
    const int w = __builtin_amdgcn_readlane(r.i, 0);
    R.p = __builtin_amdgcn_readlane(r.i, 1) - 1;
    R.b = __builtin_amdgcn_readlane(r.i, 2);
    R.ok = (unsigned)w < (unsigned)P.nW && (unsigned)R.p < (unsigned)P.Np && (unsigned)R.b < (unsigned)P.M;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        R.xn[k] = bcast_lane(r.x, k);
        R.xo[k] = bcast_lane(r.x, DIM + k);
    }
    R.S = paths + (R.ok ? ((size_t)w * P.M + R.b) * sl : 0);
}

// only the slice address of a record (what the coordinate lookahead needs mid-item; the full decode waits
// for the item's own turn so that two decoded records never compete for SGPRs)
template <int DIM>
__device__ __forceinline__ const double *pipe2_slice(const DevParams &P, const double *__restrict__ paths, const ItemRaw &r, size_t sl)
{
    const int w = __builtin_amdgcn_readlane(r.i, 0);
    const int p = __builtin_amdgcn_readlane(r.i, 1) - 1;
    const int b = __builtin_amdgcn_readlane(r.i, 2);
    const bool ok = (unsigned)w < (unsigned)P.nW && (unsigned)p < (unsigned)P.Np && (unsigned)b < (unsigned)P.M;
    return paths + (ok ? ((size_t)w * P.M + b) * sl : 0);
}

template <int DIM>
__device__ __forceinline__ void pipe2_load(const DevParams &P, const double *__restrict__ S, int p, int m, int lane,
                                           double (&rj)[DIM])
{
    const int j  = m * kWave + lane;
    const int jj = pipe_row(P, j, p);                                 // never the moved particle's own row
#pragma unroll
    for (int k = 0; k < DIM; ++k) rj[k] = S[(size_t)k * P.NpPad + jj];
}

template <int DIM, int CLS, int M>
__device__ __forceinline__ void pipe2_pass(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                           const ItemMeta<DIM> &R, const double (&rjm)[DIM], int lane, Acc<DIM, CLS> &A)
{
    const int j = M * kWave + lane;
    const bool valid = j < P.Np && j != R.p;
    double dn[DIM], dold[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        double rj = rjm[k];
        asm volatile("; class %1 pass %2" : "+v"(rj) : "n"(CLS), "n"(M));     // opaque per (class, pass): keeps the optimiser
                                                                              // from hoisting all four passes' distance arithmetic above the class branch (spills)
        dn[k] = R.xn[k] - rj; dold[k] = R.xo[k] - rj;
    }
    const double r2o = min_image_rn<DIM>(dold, P);
    const double r2n = min_image_rn<DIM>(dn, P);
    pipe_pair<DIM, CLS, false>(P, VT, WF, floor_r2(r2n), valid && r2n <= P.rcut2, dn, A);
    pipe_pair<DIM, CLS, true>(P, VT, WF, floor_r2(r2o), valid && r2o <= P.rcut2, dold, A);
    __builtin_amdgcn_sched_barrier(0);
}

// per-wave pipeline state that crosses items
template <int DIM>
struct PipeState {
    ItemRaw raw_next;         // the next item's record (requested one item ago; decoded when its turn comes)
    ItemRaw raw_nn;           // record of the item after it (requested mid-item)
    int k_nn;                 // queue index of that item
    bool has_next;            // the wave has a next item
    const double *idle;       // what the look-ahead reads when it has not
    double a0[DIM], a1[DIM];  // passes 0/1: of the current item on entry, of the next item on exit
};

template <int DIM, int CLS>
__device__ __forceinline__ void pipe2_item(const DevParams &P, PipeTab VT, const double *__restrict__ WF,
                                           const double *__restrict__ paths, size_t sl, int n_local, int *queue,
                                           const int32_t *__restrict__ walker, const int32_t *__restrict__ ipv,
                                           const int32_t *__restrict__ ibv, const double *__restrict__ xnew,
                                           const double *__restrict__ xold,
                                           const ItemMeta<DIM> &R, PipeState<DIM> &st, int k_self, int lane, double *red,
                                           double *out, double *parts)
{
    Acc<DIM, CLS> A;
    double b0[DIM], b1[DIM];
    pipe2_load<DIM>(P, R.S, R.p, 2, lane, b0);
    pipe2_pass<DIM, CLS, 0>(P, VT, WF, R, st.a0, lane, A);
    pipe2_load<DIM>(P, R.S, R.p, 3, lane, b1);
    pipe2_pass<DIM, CLS, 1>(P, VT, WF, R, st.a1, lane, A);
    // the next item's record arrived an item ago; the one after it is drawn from the queue and requested now
    // no next item: the look-ahead loads still run (a branch here costs 40 spilled VGPRs) but read `idle`, a
    // region every CU keeps hot in L2 (the table in global memory): no HBM traffic for nothing
    const double *Snext = st.has_next ? pipe2_slice<DIM>(P, paths, st.raw_next, sl) : st.idle;
    {
        int kn = 0;
        if (lane == 0) kn = atomicAdd(queue, 1);
        kn = __builtin_amdgcn_readfirstlane(kn);
        st.k_nn = kn;
        const int kq = kn < n_local ? kn : k_self;                    // past the end: this item's own record again (its
                                                                      // slice is in L2: the look-ahead loads cost no HBM traffic)
        st.raw_nn = pipe2_request<DIM>((int)blockIdx.x + kq * (int)gridDim.x, walker, ipv, ibv, xnew, xold, lane);
    }
    const int pnext = __builtin_amdgcn_readlane(st.raw_next.i, 1) - 1;    // (a wave without a next item re-reads a stale record: any row does)
    pipe2_load<DIM>(P, Snext, pnext, 0, lane, st.a0);
    pipe2_pass<DIM, CLS, 2>(P, VT, WF, R, b0, lane, A);
    pipe2_load<DIM>(P, Snext, pnext, 1, lane, st.a1);
    pipe2_pass<DIM, CLS, 3>(P, VT, WF, R, b1, lane, A);
    finish_item<DIM, CLS>(P, lane, R.b, A, red, out, parts);
}

template <int DIM>
__global__ __launch_bounds__(1024) void k_delta_action_pipe2(
    DevParams P, const double *__restrict__ paths, const double *__restrict__ VTg,
    const double *__restrict__ WF, int n_items, const int32_t *__restrict__ walker,
    const int32_t *__restrict__ ipv, const int32_t *__restrict__ ibv,
    const double *__restrict__ xnew, const double *__restrict__ xold,
    double *__restrict__ out, double *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int next_local;                                  // this workgroup's item queue
    const int lane  = threadIdx.x & (kWave - 1);
    const int wid   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t sl = slice_doubles(DIM, P.NpPad);
    const int n_local = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // first two items of this wave: static (k = wid, wid + 16); the queue starts behind them
    int k_cur = wid, k_nx = wid + 16;
    ItemMeta<DIM> cur;
    PipeState<DIM> st;
    const int nt = P.Nmax + 2;
    pipe_table_dma_issue(smem, VTg, nt, wid, lane);              // L2 hits, issued ahead of the HBM requests below
    {
        const ItemRaw r0 = pipe2_request<DIM>((int)blockIdx.x + (k_cur < n_local ? k_cur : 0) * (int)gridDim.x,
                                              walker, ipv, ibv, xnew, xold, lane);
        st.raw_next = pipe2_request<DIM>((int)blockIdx.x + (k_nx < n_local ? k_nx : (k_cur < n_local ? k_cur : 0)) * (int)gridDim.x,
                                         walker, ipv, ibv, xnew, xold, lane);
        pipe2_decode<DIM>(P, paths, r0, sl, cur);
        pipe2_load<DIM>(P, cur.S, cur.p, 0, lane, st.a0);
        pipe2_load<DIM>(P, cur.S, cur.p, 1, lane, st.a1);
    }

    const PipeTab VT = pipe_table_dma_finish(smem, VTg, nt);
    if (threadIdx.x == 0) next_local = 32;
    double *red = reinterpret_cast<double *>(smem + pipe_tab_bytes(nt) + (size_t)wid * kWaveLds);
    __syncthreads();                                            // the only workgroup barrier

    st.idle = (size_t)(P.Nmax + 2) >= sl ? VTg : paths;         // any readable region of at least one slice
    while (k_cur < n_local) {                                   // wave-uniform
        st.has_next = k_nx < n_local;
        const int it = (int)blockIdx.x + k_cur * (int)gridDim.x;
        double *o = out + it;
        double *q = nullptr;                                     // the (DeltaPot, DeltaF2, DeltaPsi) diagnostic runs on the plain grid
        if (!cur.ok) {
            // malformed item: NaN result; keep the pipeline moving without evaluating anything
            if (lane == 0) *o = __builtin_nan("");
            const double *Snext = st.has_next ? pipe2_slice<DIM>(P, paths, st.raw_next, sl) : st.idle;
            int kn = 0;
            if (lane == 0) kn = atomicAdd(&next_local, 1);
            kn = __builtin_amdgcn_readfirstlane(kn);
            st.k_nn = kn;
            st.raw_nn = pipe2_request<DIM>((int)blockIdx.x + (kn < n_local ? kn : k_cur) * (int)gridDim.x,
                                           walker, ipv, ibv, xnew, xold, lane);
            const int pnext = __builtin_amdgcn_readlane(st.raw_next.i, 1) - 1;
            pipe2_load<DIM>(P, Snext, pnext, 0, lane, st.a0);
            pipe2_load<DIM>(P, Snext, pnext, 1, lane, st.a1);
        } else {
            const bool odd  = (cur.b & 1) != 0;
            const bool endb = (cur.b == 0) || (cur.b == 2 * P.Nb);
            if (odd)       pipe2_item<DIM, CLS_ODD>(P, VT, WF, paths, sl, n_local, &next_local, walker, ipv, ibv, xnew, xold, cur, st, k_cur, lane, red, o, q);
            else if (endb) pipe2_item<DIM, CLS_END>(P, VT, WF, paths, sl, n_local, &next_local, walker, ipv, ibv, xnew, xold, cur, st, k_cur, lane, red, o, q);
            else           pipe2_item<DIM, CLS_EVEN>(P, VT, WF, paths, sl, n_local, &next_local, walker, ipv, ibv, xnew, xold, cur, st, k_cur, lane, red, o, q);
        }
        pipe2_decode<DIM>(P, paths, st.raw_next, sl, cur);       // arrived long ago: readlanes only
        k_cur = k_nx;
        k_nx = st.k_nn;
        st.raw_next = st.raw_nn;
    }
}

// bit-for-bit check of the short division / sqrt forms against the compiler's IEEE ones
__global__ void k_selftest_fastmath(DevParams P, unsigned long long seed, int iters, unsigned long long *bad)
{
    unsigned long long s = seed ^ (0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1));
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < iters; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u1 = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u2 = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        const double rmax2 = 4.0 * P.rcut2;
        const double x = (i & 1) ? u1 * rmax2 : u1 * u1 * 1e-3 + 1e-12;       // r^2 samples
        double sq, y;
        sqrt_rinv(x, sq, y);
        if (sq != sqrt(x)) ++b0;
        const double num = (u2 - 0.5) * ((i & 2) ? 1e6 : 3.0);
        if (div_by(num, sq, y) != num / sq) ++b1;
        if (div_by(sq, P.dr, P.rdr) != sq / P.dr) ++b2;
        if (div_by(num, P.dr, P.rdr) != num / P.dr) ++b3;
    }
    atomicAdd(&bad[0], b0); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2); atomicAdd(&bad[3], b3);
}

// =====================================================================================
// launcher
// =====================================================================================
#define PIGS_DISPATCH(P, CALL)                                              \
    do {                                                                    \
        if ((P).trap) {                                                     \
            if ((P).dim == 1) { CALL(1, true); }                            \
            else if ((P).dim == 2) { CALL(2, true); }                       \
            else { CALL(3, true); }                                         \
        } else {                                                            \
            if ((P).dim == 1) { CALL(1, false); }                           \
            else if ((P).dim == 2) { CALL(2, false); }                      \
            else { CALL(3, false); }                                        \
        }                                                                   \
    } while (0)

static int k1_grid(int n_items, int waves_per_block, int cap_blocks)
{
    int blocks = (n_items + waves_per_block - 1) / waves_per_block;
    return blocks < cap_blocks ? (blocks > 0 ? blocks : 1) : cap_blocks;
}

// persistent grid of the pipelined kernel: one 1024-thread workgroup per CU
static int k1_pipe_blocks()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            n = pr.multiProcessorCount;
        else
            n = 256;
    }
    return n;
}

template <typename K>
static hipError_t set_lds(K kern, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_delta_action(const DevParams &P, int variant, const double *paths, const double *VT,
                               const double *VTimg, const double *WF, int n_items, const int32_t *walker,
                               const int32_t *ip, const int32_t *ib, const double *xnew,
                               const double *xold, double *out, double *parts, hipStream_t st)
{
    if (n_items <= 0) return hipSuccess;
    const size_t pipe_bytes = (((size_t)(P.Nmax + 2 + 6) * sizeof(double)) + 15) & ~(size_t)15;   // pipe_tab_bytes()
    const bool can_pipe = !P.trap && P.Np <= 256 && VTimg && pipe_bytes + 16 * kWaveLds <= 160 * 1024;
    if (variant == K1_AUTO) {
        // The arithmetic form follows from the SYSTEM, never from the size of the launch: trapped systems the exact-term
        // form; periodic systems with Np <= 256 the pipe arithmetic -- persistent LDS-table kernel once a launch fills its
        // 16 waves per CU several times over, the same per-item code on a plain grid below that (identical bits);
        // periodic systems beyond 256 particles the short arithmetic on the global table.
        if (P.trap) variant = K1_V2;
        else if (can_pipe) variant = (n_items >= 16 * k1_pipe_blocks() && !parts) ? K1_PIPE2 : K1_GRID;
        else variant = K1_FAST;
    }
    if ((variant == K1_PIPE2 || variant == K1_GRID) && !can_pipe) variant = P.trap ? K1_V2 : K1_FAST;
    if (variant == K1_PIPE2 && parts) variant = K1_GRID;        // pipe2 has no registers to spare for the diagnostic output: same arithmetic on the grid
    if (variant == K1_FAST_PREFETCH && P.Np > 256) variant = K1_FAST;
    if ((variant == K1_FAST || variant == K1_FAST_PREFETCH) && P.trap) variant = K1_V2;   // no cutoff in the trap: exact path
    hipError_t e = hipSuccess;
    switch (variant) {
    case K1_V1: {
#define CALL(D, T)                                                                              \
    hipLaunchKernelGGL((k_delta_action_v1<D, T>), dim3(k1_grid(n_items, 4, 2048)), dim3(256), 0, st, P, \
                       paths, VT, WF, n_items, walker, ip, ib, xnew, xold, out, parts)
        PIGS_DISPATCH(P, CALL);
#undef CALL
        break;
    }
    case K1_V2:
    case K1_FAST:
    case K1_FAST_PREFETCH: {
        const size_t lds = 4 * kWaveLds;
        const int grid = k1_grid(n_items, 4, 1 << 22);
#define CALL(D, T)                                                                                             \
    do {                                                                                                       \
        if (variant == K1_V2)                                                                                  \
            hipLaunchKernelGGL((k_delta_action_v2<D, T, 256, false, false>), dim3(grid), dim3(256), lds, st, P, \
                               paths, VT, WF, n_items, walker, ip, ib, xnew, xold, out, parts);                \
        else if (variant == K1_FAST)                                                                           \
            hipLaunchKernelGGL((k_delta_action_v2<D, T, 256, false, true>), dim3(grid), dim3(256), lds, st, P, \
                               paths, VT, WF, n_items, walker, ip, ib, xnew, xold, out, parts);                \
        else                                                                                                   \
            hipLaunchKernelGGL((k_delta_action_v2<D, T, 256, true, true>), dim3(grid), dim3(256), lds, st, P,  \
                               paths, VT, WF, n_items, walker, ip, ib, xnew, xold, out, parts);                \
    } while (0)
        PIGS_DISPATCH(P, CALL);
#undef CALL
        break;
    }
    case K1_GRID: {
        const size_t lds = 4 * kWaveLds;
        const int grid = k1_grid(n_items, 4, 1 << 22);
#define CALLP(D)                                                                                        \
    hipLaunchKernelGGL((k_delta_action_grid<D>), dim3(grid), dim3(256), lds, st, P, paths, VTimg, WF,   \
                       n_items, walker, ip, ib, xnew, xold, out, parts)
        if (P.dim == 1) CALLP(1); else if (P.dim == 2) CALLP(2); else CALLP(3);
#undef CALLP
        break;
    }
    case K1_PIPE2: {
        const size_t lds = pipe_bytes + 16 * kWaveLds;
        int blocks = (n_items + 15) / 16;
        if (blocks > k1_pipe_blocks()) blocks = k1_pipe_blocks();
#define CALLP(D)                                                                                        \
    do {                                                                                                \
        e = set_lds(k_delta_action_pipe2<D>, lds);                                                      \
        if (e == hipSuccess)                                                                            \
            hipLaunchKernelGGL((k_delta_action_pipe2<D>), dim3(blocks), dim3(1024), lds, st, P, paths,  \
                               VT, WF, n_items, walker, ip, ib, xnew, xold, out, parts);                \
    } while (0)
        if (P.dim == 1) CALLP(1); else if (P.dim == 2) CALLP(2); else CALLP(3);
#undef CALLP
        break;
    }
    case K1_REFORDER: {
        const size_t lds = (size_t)P.Np * 8 * sizeof(double);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        const int grid = n_items < (1 << 16) ? n_items : (1 << 16);
#define CALL(D, T)                                                                                      \
    do {                                                                                                \
        e = set_lds(k_delta_action_reforder<D, T>, lds);                                                \
        if (e == hipSuccess)                                                                            \
            hipLaunchKernelGGL((k_delta_action_reforder<D, T>), dim3(grid), dim3(64), lds, st, P, paths, VT, WF, \
                               n_items, walker, ip, ib, xnew, xold, out, parts);                        \
    } while (0)
        PIGS_DISPATCH(P, CALL);
#undef CALL
        break;
    }
    default:
        return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_selftest_fastmath(const DevParams &P, unsigned long long seed, int blocks, int iters,
                                    unsigned long long *d_bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_selftest_fastmath, dim3(blocks), dim3(256), 0, st, P, seed, iters, d_bad);
    return hipGetLastError();
}

} // namespace pigs
