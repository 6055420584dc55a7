// pigs_diag.hip -- K6d: the diagonal moves of one MC step of a periodic system with sampling = 'bis', one workgroup
// per walker, as a STAGE MACHINE:
//   TranslateChain for every (non-worm) particle (vpi_mod.f90:313-379), then
//   Nstag x Np x { MoveHeadBisection, MoveTailBisection, Bisection } (vpi_mod.f90:864-1372).
// (The open / close attempt before and the worm moves after these run in pigs_sampler.hip's kernel: three launches
// per MC step; trapped systems and sampling = 'sta' run everything there.)
//
// Why a machine: a walker's moves are one dependent chain of ~10 000 short stages per MC step (a bisection level has
// 1-8 proposal beads), and ONE wave of a CU issues at most one instruction every ~4 cycles -- the chain is bound by
// the number of instructions on its critical path, not by arithmetic or bandwidth.  So:
//   * control step (wave 0 alone between two workgroup barriers): finish the pending stage -- column sums of its
//     tasks' totals, Chin weight, the Metropolis question asked as a >= log(u) with log(u) tabulated by the producer
//     wave (no exp on the critical path) --, commit an accepted move, pick the next stage, take its Gaussians out
//     of ONE window of 64 precomputed polar Box-Muller candidates, build every proposal bead of the stage in
//     parallel lanes, publish a stage descriptor.  All stream positions live in registers; ring indices are masks.
//   * task phase (all waves): the stage's Delta S cut into tasks of (bead, run of 64-partner passes, new | old
//     distance), one per wave (pipe_task, pigs_k1_device.h), so that a lone bead keeps eight waves busy; a
//     TranslateChain stage hands whole beads to the waves.  In the first phase of a move every wave touches a few
//     slices of the segment (one load per slice warms its lines in L2 for the deeper levels); the last wave tops
//     up the random stream; all threads fetch the next visit's particle chain a phase ahead.
// The moved particle's whole chain lives in LDS; proposals reach the resident worldline only when accepted.
// Random stream, arithmetic and decisions are those of pigs_sampler.hip (same helpers, pigs_sampler_device.h).
#include "pigs_device.h"
#include "pigs_k1_device.h"
#include "pigs_kernels.h"
#include "pigs_sampler_device.h"

namespace pigs {

#ifdef PIGS_SWEEP_TIMING
#define DSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define DACC(slot, a, b) do { if (tid == 0) tacc[slot] += (b) - (a); } while (0)
#else
#define DSTAMP(v) do { } while (0)
#define DACC(slot, a, b) do { } while (0)
#endif

namespace {

// cold paths of the control step, kept out of line so that the stage loop stays compact (instruction cache)
__device__ __attribute__((noinline)) void cold_refill(const Rng &R, int pos, int lane)
{
    if (lane == 0) R.ctl[0] = pos;
    __builtin_amdgcn_wave_barrier();
    rng_background(R, lane);
}
__device__ __attribute__((noinline)) int cold_gaussians(const Rng &R, int pos, int G, double *gbuf, int lane)
{
    if (lane == 0) R.ctl[0] = pos;
    __builtin_amdgcn_wave_barrier();
    wave_gaussians(R, G, gbuf, lane);
    return R.ctl[0];
}
__device__ __attribute__((noinline)) int cold_metropolis_nan(const Rng &R, int pos, double a, int lane)
{
    if (lane == 0) { R.ctl[0] = pos; (void)metropolis(R, a); }
    __builtin_amdgcn_wave_barrier();
    return R.ctl[0];
}

// workgroup barrier that drains only this wave's LDS / scalar traffic: global loads issued before it (slice touches,
// the next visit's chain) stay in flight across it.  Global STORES that other waves read later (an accepted move's
// beads) are drained by the storing wave itself (stores_done) before it arrives here.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct DiagLds {
    size_t mt, ctl, Gc, Lc, pc0, pc1, dS, cn, gbuf, sxn, sxo, sb, dxs, sgn, sgb, tots, red, tab, total;
};

__host__ __device__ inline DiagLds diag_layout(const DevParams &P, const SweepParams &sp, int nwaves)
{
    const size_t M = P.M, D = P.dim;
    const size_t nseg = (size_t)1 << sp.Nlev, nbmax = nseg / 2;
    const size_t ntm = nbmax > 16 ? nbmax : 16;
    DiagLds L;
    size_t b = 0;
    auto take = [&](size_t bytes) { const size_t at = b; b += (bytes + 15) & ~(size_t)15; return at; };
    L.mt   = take(kRing * 4);
    L.ctl  = take(64 * 4);                       // [0..15] stream control, [16..31] counters, [32..47] stage descriptor
    L.Gc   = take(kGRing * 8);
    L.Lc   = take(kGRing * 8);
    L.pc0  = take(M * D * 8);
    L.pc1  = take(M * D * 8);
    L.dS   = take(M * 8);
    L.cn   = take((nseg + 1) * D * 8);
    L.gbuf = take((nbmax * D > 64 ? nbmax * D : 64) * 8);
    L.sxn  = take(nbmax * D * 8);
    L.sxo  = take(nbmax * D * 8);
    L.sb   = take((nbmax > 4 ? nbmax : 4) * 4);
    L.dxs  = take(4 * 8);
    L.sgn  = take((nseg + 1) * 8);
    L.sgb  = take(8 * 8);
    L.tots = take(ntm * 8 * 8);
    L.red  = take((size_t)nwaves * kWaveLds);
    L.tab  = take(((size_t)P.Nmax + 2 + 6) * 8);
    L.total = b;
    return L;
}

} // namespace

template <int DIM, int NT>
__global__ __launch_bounds__(NT, 1) void k_diag(
    DevParams P, SweepParams sp, double *__restrict__ paths, const double *__restrict__ VTimg,
    const double *__restrict__ WF, uint32_t *__restrict__ rng, unsigned long long *__restrict__ counters,
    const double *__restrict__ worm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / kWave;
    // waves that take the tasks of a bisection stage: ONE per SIMD.  A CU issues one instruction per SIMD every four
    // cycles whatever the number of its waves, and a task carries ~140 instructions of set-up and reduction next to
    // its ~60 per distance-pass: eight small tasks on two waves per SIMD take longer than four twice as big.
#ifndef PIGS_DIAG_TASK_WAVES
#define PIGS_DIAG_TASK_WAVES 4
#endif
    constexpr int NWT = PIGS_DIAG_TASK_WAVES < NW ? PIGS_DIAG_TASK_WAVES : NW;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = blockIdx.x;
    const int M = P.M, Nb = P.Nb, MD = M * DIM;
    const DiagLds L = diag_layout(P, sp, NW);

    uint32_t *mt  = reinterpret_cast<uint32_t *>(smem + L.mt);
    int      *ctl = reinterpret_cast<int *>(smem + L.ctl);
    unsigned int *cnt = reinterpret_cast<unsigned int *>(ctl + 16);
    int      *sd  = ctl + 32;
    double   *Gc  = reinterpret_cast<double *>(smem + L.Gc);
    double   *Lc  = reinterpret_cast<double *>(smem + L.Lc);
    double   *pcb[2] = {reinterpret_cast<double *>(smem + L.pc0), reinterpret_cast<double *>(smem + L.pc1)};
    double   *dS   = reinterpret_cast<double *>(smem + L.dS);
    double   *cn   = reinterpret_cast<double *>(smem + L.cn);
    double   *gbuf = reinterpret_cast<double *>(smem + L.gbuf);
    double   *sxn  = reinterpret_cast<double *>(smem + L.sxn);
    double   *sxo  = reinterpret_cast<double *>(smem + L.sxo);
    int      *sb   = reinterpret_cast<int *>(smem + L.sb);
    double   *dxs  = reinterpret_cast<double *>(smem + L.dxs);
    double   *sgn  = reinterpret_cast<double *>(smem + L.sgn);
    double   *sgb  = reinterpret_cast<double *>(smem + L.sgb);
    double   *tots = reinterpret_cast<double *>(smem + L.tots);
    double   *red  = reinterpret_cast<double *>(smem + L.red + (size_t)wid * kWaveLds);
    double   *lt   = reinterpret_cast<double *>(smem + L.tab);
    const Rng R{mt, Gc, Lc, ctl};

    const size_t sl = slice_doubles(DIM, P.NpPad);
    double *Pw = paths + (size_t)w * M * sl;
    const int npass = (P.Np + kWave - 1) / kWave;
    const int nseg = 1 << sp.Nlev;

    // generator state in the reference's block form (what mtsavef holds): words of the current block + index
    for (int t = tid; t < MT_N; t += NT) mt[t] = rng[(size_t)w * kRngWords + MT_N + t];
    if (tid == 0) {
        ctl[0] = (int)rng[(size_t)w * kRngWords + 2 * MT_N];
        ctl[11] = MT_N;
        ctl[12] = ctl[0];
    }
    if (tid < kCounters) cnt[tid] = 0;
    // proposal widths (the same IEEE sqrt of the same arguments the reference takes): sgn[n] = sqrt(n dt) (free end
    // guess), sgb[k] = sqrt(0.5 * (0.5 * 2^k * dt)) (bisection level with delta_ib = 2^k)
    for (int t = tid; t <= nseg; t += NT) {
        sgn[t] = sqrt((double)t * P.dt);
        if (t < 8) sgb[t] = sqrt(0.5 * (0.5 * (double)(1 << t) * P.dt));
    }
    {   // table image [0 VT(0)] VT(0..Nmax+1) [0 0 0 0] (pigs_k1_device.h, PipeTab)
        const int nimg = P.Nmax + 2 + 6;
        const double2 *src = reinterpret_cast<const double2 *>(VTimg);
        double2 *dst = reinterpret_cast<double2 *>(lt);
        for (int t = tid; t < nimg / 2; t += NT) dst[t] = src[t];
    }
    const PipeTab VTp{lt + 2, P.Nmax + 3};
    const bool isopen = sp.worm ? (int)worm[(size_t)w * kWormDoubles] != 0 : false;
    const int  pworm  = sp.worm ? (int)worm[(size_t)w * kWormDoubles + 1] - 1 : -1;
#ifdef PIGS_SWEEP_TIMING
    __shared__ unsigned long long tacc[8];
    if (tid < 8) tacc[tid] = 0;
#endif

    const bool prefetch_ok = MD <= 4 * NT;
    const int V = ((sp.do_cm ? 1 : 0) + sp.Nstag) * P.Np;
    int lgp2 = 0;                                     // passes rounded up to a power of two: tasks are addressed with shifts
    while ((1 << lgp2) < npass) ++lgp2;
    {
        int v0 = 0;
        if (isopen && v0 % P.Np == pworm) ++v0;
        if (v0 < V) {
            const int p0 = v0 % P.Np;
            for (int e = tid; e < MD; e += NT) pcb[0][e] = Pw[(size_t)(e / DIM) * sl + (size_t)(e % DIM) * P.NpPad + p0];
        }
    }
    __syncthreads();
    if (wid == NW - 1) rng_background(R, lane);
    __syncthreads();

    // control state (meaningful in wave 0)
    int cv = -1, cmv = 3, clev = 0, cnl = 0, cii = 0, cseg = 0, ccur = 1, cp = 0;
    int c_nbd = 0, c_lgtpb = 0, c_kind = -1, c_phase = 0, c_pn = -1;
    bool cv_seen = false;
    int pos = ctl[0];
    double nx[4] = {0.0, 0.0, 0.0, 0.0};
    int sink0 = 0, sink1 = 0, sink2 = 0;              // destinations of the slice-touch loads (see the task phase)

    for (;;) {
        DSTAMP(tA);
        if (wid == 0) {
            // ================= control step =================
            DSTAMP(c0);
            // the producer only works in task phases: its marks are stable here.  One refill makes sure that the two
            // uniforms and the one window of candidates a step can take are there (anything beyond goes the slow way).
            int gd = ctl[12];
            if (gd < pos + 2 * kWave + 4) {
                cold_refill(R, pos, lane);
                gd = ctl[12];
            }
            // everything the step may read from LDS whatever the pending decision turns out to be is requested now, in one go
            double pre_tv[4];
            {
                const int tpb0 = 1 << c_lgtpb;
                const bool on = c_kind == 0 && lane < c_nbd * 8 && tpb0 <= 4;
                const double *Tb = tots + (on ? (lane >> 3) * tpb0 * 8 + (lane & 7) : 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) pre_tv[t] = Tb[(on && t < tpb0 ? t : 0) * 8];
            }
            const int    pre_b  = sb[(c_kind == 0 && lane < c_nbd * 8) ? (lane >> 3) : 0];
            const double pre_lu = Lc[pos & (kGRing - 1)];                     // log u of the pending Metropolis question
            bool ok = false;
            if (c_kind == 0) {
                // ---- finish the pending bisection stage: column sums of each bead's tasks (lane = bead*8 + column)
                const int tpb = 1 << c_lgtpb;
                double tsum;
                if (c_nbd > 8) {                                              // Nlev > 4: a lane per bead
                    double dSi = 0.0;
                    if (lane < c_nbd) dSi = item_finish_split<DIM>(P, sb[lane], tpb, tots + (size_t)lane * tpb * 8);
                    tsum = 0.0;
                    for (int i = 0; i < c_nbd; ++i) tsum = tsum + read_lane(dSi, i);
                } else if (tpb > 4) {                                         // Np > 256 or fewer task waves: columns through LDS
                    double cs = 0.0;
                    if (lane < c_nbd * 8) {
                        const double *Tb = tots + (lane >> 3) * tpb * 8 + (lane & 7);
                        for (int t = 0; t < tpb; ++t) cs = cs + Tb[t * 8];
                    }
                    red[lane] = cs;
                    __builtin_amdgcn_wave_barrier();
                    double dSi = 0.0;
                    if (lane < c_nbd) dSi = item_finish_split<DIM>(P, sb[lane], 1, red + lane * 8);
                    __builtin_amdgcn_wave_barrier();
                    tsum = 0.0;
                    for (int i = 0; i < c_nbd; ++i) tsum = tsum + read_lane(dSi, i);
                } else {
                    // lane = bead*8 + column: the column sums of the bead's (<= 4) tasks, then Delta S in the bead's first
                    // lane with its neighbours' columns fetched by DPP row shifts (no LDS round trip), in the reference's
                    // order: DeltaPot = c0 - c1, Fnew2 = (f0^2 + f1^2) + f2^2 (vpi_mod.f90:2825-2838)
                    const bool on = lane < c_nbd * 8;
                    double cs = 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) cs = cs + ((on && t < tpb) ? pre_tv[t] : 0.0);
                    const double d01 = cs - row_shl<1>(cs);                   // column 0: DeltaPot; column 2: PsiNew - PsiOld
                    const double sq  = cs * cs;
                    const double f2  = (sq + row_shl<1>(sq)) + row_shl<2>(sq);   // column 2: |Fnew|^2; column 5: |Fold|^2
                    const double fn2 = row_shl<2>(f2), fo2 = row_shl<5>(f2), dps = row_shl<2>(d01);
                    const int b = pre_b;
                    const bool odd  = (b & 1) != 0;
                    const bool endb = (b == 0) || (b == 2 * Nb);
                    const double dSi = -(endb ? dps : 0.0) + green_function_action(b, Nb, P.dt, d01, odd ? fn2 - fo2 : 0.0);
                    tsum = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (i < c_nbd) tsum = tsum + read_lane(dSi, 8 * i);
                }
                // Metropolis (vpi_mod.f90:960-969) on exp(-tsum), asked as -tsum >= log(u)
                const double a = -tsum;
                if (a >= -0x1p-54) ok = true;                                 // exp(a) rounds to >= 1: no uniform is drawn
                else if (a == a) { ok = a >= pre_lu; ++pos; }
                else pos = cold_metropolis_nan(R, pos, a, lane);              // NaN: the plain form (draws a uniform, rejects)
                if (ok && clev == cnl) {
                    // accepted: the generated beads go to the chain in LDS and to the resident worldline
                    const int j0 = cmv == 0 ? 0 : 1, j1 = cmv == 1 ? cseg : cseg - 1;
                    for (int e = lane + j0 * DIM; e < (j1 + 1) * DIM; e += kWave) {
                        const int jj = e / DIM, kk = e - jj * DIM;
                        const double nw = cn[e];
                        pcb[ccur][(cii + jj) * DIM + kk] = nw;
                        Pw[(size_t)(cii + jj) * sl + (size_t)kk * P.NpPad + cp] = nw;
                    }
                    stores_done();
                    if (lane == 0) ++cnt[1 + cmv];
                    clev = cnl + 1;                                           // move finished
                } else if (ok) {
                    ++clev;
                } else {
                    clev = cnl + 1;                                           // rejected: move finished
                }
            } else if (c_kind == 1) {
                // ---- finish the pending TranslateChain stage (vpi_mod.f90:354-374)
                double t = 0.0;
                for (int b = lane; b < M; b += kWave) t = t + dS[b];
                t = wave_sum(t);
                const double a = -t;
                if (a >= -0x1p-54) ok = true;
                else if (a == a) { ok = a >= pre_lu; ++pos; }
                else pos = cold_metropolis_nan(R, pos, a, lane);
                if (ok) {
                    const double *pcur = pcb[ccur];
                    for (int e = lane; e < MD; e += kWave) {
                        const int b = e / DIM, kk = e - b * DIM;
                        const double Lq = kk == 0 ? P.Lbox[0] : (kk == 1 ? P.Lbox[1] : P.Lbox[2]);
                        const double Hq = kk == 0 ? P.LboxHalf[0] : (kk == 1 ? P.LboxHalf[1] : P.LboxHalf[2]);
                        double nw = pcur[e] + dxs[kk];
                        if (nw >  Hq) nw = nw - Lq;
                        if (nw < -Hq) nw = nw + Lq;
                        pcb[ccur][e] = nw;
                        Pw[(size_t)b * sl + (size_t)kk * P.NpPad + cp] = nw;
                    }
                    stores_done();
                }
                if (lane == 0) { ++cnt[14]; cnt[0] += ok; }
                cmv = 3;                                                      // visit finished
            }
            DSTAMP(c1);
            // ---- the next stage
            int flags = 0;
            bool quit = false;
            if (c_kind != 0 || clev > cnl) {                                  // a new move (or visit)
                if (c_kind == 0) ++cmv;
                if (cmv >= 3) {                                               // new visit
                    ++cv;
                    if (cv < V && isopen && cv % P.Np == pworm) ++cv;
                    if (cv >= V) quit = true;
                    else {
                        // Two particles, one of them the open worm: every visit is to the SAME particle.  Its chain buffer is
                        // kept current by the commits, whereas a copy fetched "a visit ahead" -- during this visit's first
                        // phases -- would be stale by the time it is used (round 3 fuzz: Np = 2, Nlev = 5, CWorm > 0).  So a
                        // visit to the particle just visited keeps the buffer, and nothing is fetched ahead for it.
                        const int cp_prev = cv_seen ? cp : -1;
                        cp = cv % P.Np;
                        cv_seen = true;
                        if (cp != cp_prev) ccur ^= 1;
                        c_phase = 0;
                        int vn = cv + 1;
                        if (vn < V && isopen && vn % P.Np == pworm) ++vn;
                        c_pn = (vn < V && vn % P.Np != cp) ? vn % P.Np : -1;
                        cmv = (sp.do_cm && cv < P.Np) ? -1 : 0;               // -1: a TranslateChain visit
                        if (cmv == 0 && lane == 0) ++cnt[15];
                    }
                }
                if (!quit && cmv >= 0) {
                    // segment choice (vpi_mod.f90:890, 1023)
                    const double u = mt_real(mt[ring_w(pos)]);
                    ++pos;
                    cnl = sp.Nlev;
                    if (cmv == 2) {
                        cii = (int)((double)(2 * Nb - nseg + 1) * u);
                        if (cii > 2 * Nb - nseg) cii = 2 * Nb - nseg;         // u == 1.0 (quirk Q15)
                    } else {
                        cnl = (int)((double)(sp.Nlev - 1) * u) + 2;
                        if (cnl > sp.Nlev) cnl = sp.Nlev;                     // u == 1.0 (Q15): the reference draws Nlev+1
                        cii = cmv == 0 ? 0 : 2 * Nb - (1 << cnl);
                    }
                    cseg = 1 << cnl;
                    clev = cmv == 2 ? 1 : 0;
                }
            }
            if (quit) {
                c_kind = 2;
                if (lane == 0) { sd[0] = 2; ctl[0] = pos; }
            } else if (cmv < 0) {
                // ---- TranslateChain stage: every bead shifted by the same random vector
                if (lane < DIM) dxs[lane] = sp.delta_cm * (2.0 * mt_real(mt[ring_w(pos + lane)]) - 1.0);
                pos += DIM;
                c_kind = 1;
                if (c_pn >= 0) flags = 3;                                     // fetch and park the next chain in this phase
                if (lane == 0) { sd[0] = 1; sd[1] = cp; sd[5] = flags; sd[6] = c_pn; sd[7] = ccur; ctl[0] = pos; }
            } else {
                // ---- bisection-type stage clev of move cmv on beads cii..cii+cseg:
                //   clev == 0: the free guess of the end bead of a head / tail move with its own test (Q12)
                //   clev >= 1: the 2^(clev-1) midpoints of the level (vpi_mod.f90:903-971)
                DSTAMP(c2);
                const int nbd = clev == 0 ? 1 : 1 << (clev - 1);
                const int G = nbd * DIM;
                // G Gaussians = the first G accepted polar pairs at pos, pos+2, ... (= G sequential rangauss calls)
                {
                    const double g = Gc[(pos + 2 * lane) & (kGRing - 1)];
                    const bool acc = g == g;
                    const unsigned long long m = __ballot(acc);
                    if (__builtin_popcountll(m) >= G && G <= kWave) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        if (acc && rank < G) gbuf[rank] = g;
                        const unsigned long long sel = __ballot(acc && rank == G - 1);
                        pos += 2 * (__builtin_ctzll(sel) + 1);
                        __builtin_amdgcn_wave_barrier();
                    } else {                                                  // (rare) more than one window
                        pos = cold_gaussians(R, pos, G, gbuf, lane);
                    }
                }
                DSTAMP(c3);
                const double *pcur = pcb[ccur];
                const int dib = clev == 0 ? 0 : cseg >> (clev - 1);
                const double sigma = clev == 0 ? sgn[cseg] : sgb[cnl - clev + 1];
                for (int Lx = lane; Lx < G; Lx += kWave) {                    // one trip up to 21 beads (Nlev <= 5)
                    const int il = Lx / DIM, kl = Lx - il * DIM;
                    const double Lk = kl == 0 ? P.Lbox[0] : (kl == 1 ? P.Lbox[1] : P.Lbox[2]);
                    const double Hk = kl == 0 ? P.LboxHalf[0] : (kl == 1 ? P.LboxHalf[1] : P.LboxHalf[2]);
                    int jc, ja, jb;
                    if (clev == 0) { jc = cmv == 0 ? 0 : cseg; ja = cmv == 0 ? cseg : 0; jb = ja; }
                    else           { ja = il * dib; jb = ja + dib; jc = ja + dib / 2; }
                    // an anchor: a bead generated earlier in this move, or an untouched end of the segment
                    const bool gen_a = clev != 0 && (ja == 0 ? cmv == 0 : (ja == cseg ? cmv == 1 : true));
                    const bool gen_b = clev != 0 && (jb == 0 ? cmv == 0 : (jb == cseg ? cmv == 1 : true));
                    const double o  = pcur[(cii + jc) * DIM + kl];
                    const double va = gen_a ? cn[ja * DIM + kl] : pcur[(cii + ja) * DIM + kl];
                    const double vb = gen_b ? cn[jb * DIM + kl] : pcur[(cii + jb) * DIM + kl];
                    // nearest images of the anchors (vpi_mod.f90:925-937): o + wrap(a - o), o - wrap(o - b)
                    double da = va - o, db = o - vb;
                    if (da < -Hk) da = da + Lk;
                    if (da >  Hk) da = da - Lk;
                    if (db < -Hk) db = db + Lk;
                    if (db >  Hk) db = db - Lk;
                    const double xp = o + da, xq = o - db;
                    double xm;
                    if (clev == 0) xm = cmv == 0 ? xq : xp;                   // the one anchor is the next (head) / previous (tail) bead
                    else           xm = 0.5 * (xp + xq);
                    double nw = xm + sigma * gbuf[Lx];
                    if (nw >  Hk) nw = nw - Lk;
                    if (nw < -Hk) nw = nw + Lk;
                    cn[jc * DIM + kl] = nw;
                    sxn[Lx] = nw; sxo[Lx] = o;
                    if (kl == 0) sb[il] = cii + jc;
                }
                // task shape: k = distance-passes per task (1: one side of one pass, 2: both sides of one pass, 4,
                // 8, ...: both sides of k/2 passes), as few as keep every task on a wave of its own
                const int U = nbd << (lgp2 + 1);
                int lgk = 0;
                while ((NWT << lgk) < U && lgk < lgp2 + 1) ++lgk;
                c_lgtpb = lgp2 + 1 - lgk;
                c_nbd = nbd; c_kind = 0;
                if (c_phase == 0 && c_pn >= 0) flags = 1;                     // fetch the next visit's chain ...
                else if (c_phase == 1 && c_pn >= 0) flags = 2;                // ... park it one phase later
                if (clev == (cmv == 2 ? 1 : 0)) flags |= 4;                   // first stage of a move: touch the segment's slices
                if (lane == 0) {
                    sd[0] = 0; sd[1] = cp; sd[2] = nbd << c_lgtpb; sd[3] = c_lgtpb; sd[4] = lgk; sd[5] = flags;
                    sd[6] = c_pn; sd[7] = ccur; sd[8] = cii; sd[9] = cseg;
                    ctl[0] = pos;
                }
                ++c_phase;
                DSTAMP(c4);
                DACC(0, c0, c4);
            }
        }
        DSTAMP(tB);
        lds_barrier();                                                        // descriptor, proposals, commits visible
        DSTAMP(tC);
        const int4 d0 = *reinterpret_cast<const int4 *>(sd), d1 = *reinterpret_cast<const int4 *>(sd + 4);
        const int kind = d0.x;
        if (kind == 2) break;
        const int p = d0.y, flags = d1.y, pn = d1.z, cur = d1.w;
        const double *pcur = pcb[cur];
        if ((flags & 1) && prefetch_ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + q * NT;
                if (e < MD) nx[q] = Pw[(size_t)(e / DIM) * sl + (size_t)(e % DIM) * P.NpPad + pn];
            }
        }
        if (kind == 0) {
            const int ntask = d0.z, lgtpb = d0.w, lgk = d1.x;
            for (int t = wid; t < ntask; t += NWT) {
                const int i = t >> lgtpb, q = t & ((1 << lgtpb) - 1);
                const int b = sb[i];
                double a[DIM], c[DIM];
#pragma unroll
                for (int kk = 0; kk < DIM; ++kk) { a[kk] = sxn[i * DIM + kk]; c[kk] = sxo[i * DIM + kk]; }
                int sides = 3, m0, np;
                if (lgk == 0) { sides = 1 + (q & 1); m0 = q >> 1; np = 1; }
                else          { np = 1 << (lgk - 1); m0 = q * np; }
                if (m0 + np > npass) np = npass - m0;                         // (passes were rounded up to a power of two)
                DSTAMP(k0);
#ifndef PIGS_DIAG_SKIP_TASKS
                #ifdef PIGS_SWEEP_TIMING
                pipe_task_rolled<DIM>(P, VTp, WF, Pw + (size_t)b * sl, p, b, m0, np, sides, a, c, lane, red, tots + (size_t)t * 8, tacc + 1);
#else
                pipe_task_rolled<DIM>(P, VTp, WF, Pw + (size_t)b * sl, p, b, m0, np, sides, a, c, lane, red, tots + (size_t)t * 8);
#endif
#else
                if (lane < 8) tots[(size_t)t * 8 + lane] = a[0] * 1e-3 + c[0] * (double)(np + sides + m0);   // timing experiment: no Delta S
#endif
                DSTAMP(k1);
                DACC(4, tC, k0); DACC(5, k0, k1);
            }
        } else {
            double dx[DIM];
#pragma unroll
            for (int kk = 0; kk < DIM; ++kk) dx[kk] = dxs[kk];
            // bead r*NW + (wid + r) mod NW in round r: with an even number of waves a plain stride would give a wave
            // only odd or only even beads of a chain, and odd beads (force terms) cost 1.3x the even ones
            for (int r = 0; r * NW < M; ++r) {
                int i = wid + r;
                i = r * NW + (i >= NW ? i % NW : i);
                if (i >= M) continue;
                double a[DIM], c[DIM];
#pragma unroll
                for (int kk = 0; kk < DIM; ++kk) { c[kk] = pcur[i * DIM + kk]; a[kk] = wrap_coord(P, false, kk, c[kk] + dx[kk]); }
                if (P.Np <= 256) item_eval_pipe<DIM>(P, VTp, WF, Pw + (size_t)i * sl, p, i, a, c, lane, red, &dS[i], nullptr);
                else {
                    double *t8 = red + 8 * kRedStride - 8;                    // tail of this wave's own scratch
                    pipe_task<DIM>(P, VTp, WF, Pw + (size_t)i * sl, p, i, 0, npass, 3, a, c, lane, red, t8);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) dS[i] = item_finish_split<DIM>(P, i, 1, t8);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (flags & 2) {
            if (prefetch_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + q * NT;
                    if (e < MD) pcb[cur ^ 1][e] = nx[q];
                }
            } else {
                for (int e = tid; e < MD; e += NT) pcb[cur ^ 1][e] = Pw[(size_t)(e / DIM) * sl + (size_t)(e % DIM) * P.NpPad + pn];
            }
        }
        if (wid == NW - 1) cold_refill(R, ctl[0], lane);                      // nobody consumes random numbers in this phase
        if (kind == 0 && (flags & 4)) {
            // first phase of a move: one load instruction per slice of the segment warms all its lines in L2 (a lane per
            // 128-byte line) for the deeper levels.  Issued behind the wave's own work and never waited for in this phase.
            // The compiler does not see these loads, so their destination registers (sink0..2, live around the whole
            // loop) are released only after an explicit wait -- placed here, a move later, when they landed long ago.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(sink0), "v"(sink1), "v"(sink2));
            const int t_ii = sd[8], t_seg = sd[9];
            const int nline = (int)((sl * sizeof(double) + 127) / 128);
            const int lo = (lane < nline ? lane : 0) * 32;
            const int *q0 = reinterpret_cast<const int *>(Pw + (size_t)(t_ii + (wid <= t_seg ? wid : 0)) * sl) + lo;
            const int *q1 = reinterpret_cast<const int *>(Pw + (size_t)(t_ii + (wid + NW <= t_seg ? wid + NW : 0)) * sl) + lo;
            const int *q2 = reinterpret_cast<const int *>(Pw + (size_t)(t_ii + (wid + 2 * NW <= t_seg ? wid + 2 * NW : 0)) * sl) + lo;
            asm volatile("global_load_dword %0, %1, off" : "=v"(sink0) : "v"(q0) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(sink1) : "v"(q1) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(sink2) : "v"(q2) : "memory");
        }
        DSTAMP(tD);
        lds_barrier();                                                        // totals, parked chain, stream visible
        DSTAMP(tE);
        DACC(6, tA, tB); if (kind == 0) DACC(7, tD, tE);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(sink0), "v"(sink1), "v"(sink2));
    // generator state back in block form: every word of the block pos lies in (the consumed ones are still in the ring)
    __syncthreads();
    const int bs_save = block_start(ctl[0]);
    if (wid == 0) {
        while (ctl[11] < bs_save + MT_N) rng_produce(R, kRing, lane);         // wave-uniform
    }
    __syncthreads();
    for (int t = tid; t < MT_N; t += NT) rng[(size_t)w * kRngWords + MT_N + t] = mt[ring_w(bs_save + t)];
    if (tid == 0) {
        rng[(size_t)w * kRngWords + 2 * MT_N] = (uint32_t)(ctl[0] - bs_save);
#pragma unroll
        for (int q = 0; q < kCounters; ++q) counters[(size_t)w * kCounters + q] += cnt[q];
#ifdef PIGS_SWEEP_TIMING
        for (int q = 0; q < 8; ++q) counters[(size_t)w * kCounters + 8 + q] += tacc[q];
#endif
    }
}

bool diag_supported(const DevParams &P, const SweepParams &sp)
{
    // (Nlev >= 2: with Nlev = 1 the head / tail moves still bisect 2^2 beads -- vpi_mod.f90:1023 -- and this kernel sizes its
    // buffers by 2^Nlev; the one-launch kernel serves that case)
    return !P.trap && !sp.staging && !(P.Nmax & 1) && sp.Nlev >= 2 && sp.Nlev <= 7 && (1 << sp.Nlev) <= 2 * P.Nb &&
           diag_form(P, sp, 1024) != 0;
}

int diag_form(const DevParams &P, const SweepParams &sp, int threads)
{
    if (threads >= 512 && diag_layout(P, sp, 8).total <= 160 * 1024) return 512;
    return 0;
}

hipError_t launch_diag(const DevParams &P, const SweepParams &sp, int threads, double *paths, const double *VTimg,
                       const double *WF, uint32_t *rng, unsigned long long *counters, const double *worm, hipStream_t st)
{
    const int nt = diag_form(P, sp, threads);
    if (nt == 0 || !diag_supported(P, sp)) return hipErrorInvalidValue;
    const size_t lds = diag_layout(P, sp, nt / kWave).total;
    hipError_t e = hipSuccess;
#define CALLD(D)                                                                                               \
    do {                                                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_diag<D, 512>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        if (e == hipSuccess)                                                                                   \
            hipLaunchKernelGGL((k_diag<D, 512>), dim3(P.nW), dim3(512), lds, st, P, sp, paths, VTimg, WF, rng, \
                               counters, worm);                                                                \
    } while (0)
    if (P.dim == 1) CALLD(1); else if (P.dim == 2) CALLD(2); else CALLD(3);
#undef CALLD
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace pigs
