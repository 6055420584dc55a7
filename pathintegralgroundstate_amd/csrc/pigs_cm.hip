// pigs_cm.hip -- K6c: the TranslateChain moves of one MC step of a periodic system (vpi_mod.f90:313-379; called for
// every particle at vpi.f90:331-339) by H COOPERATING WORKGROUPS PER WALKER.
//
// Why its own kernel: the device-resident sampler gives a walker one workgroup, hence one CU, and a TranslateChain move
// is the one stage of the step that is bound by arithmetic, not by latency -- Delta S of all M beads of the shifted
// particle against all its partners (M x Np x 2 distances: 21 M pair evaluations per walker and step at N=256, 161
// beads, a quarter of the step's time).  With W walkers on a chip of n_cu >= 2 W CUs the other CUs idle.  Here a
// walker's M beads are cut into H contiguous ranges, one workgroup (one CU) each:
//   * every workgroup replays the walker's random stream (the phase draws only uniforms: dim per move + at most one
//     for the Metropolis question) and takes the same decisions from the same bits;
//   * per move a workgroup evaluates Delta S of ITS beads (one bead per wave and round, the K1 pipe evaluation on the
//     LDS table image), publishes them in a global exchange buffer and collects the others' -- every value travels as
//     two 64-bit words, each half of the double next to a 32-bit sequence tag, written and polled with relaxed
//     agent-scope atomics: no fence, no cache flush, nothing to order;
//   * the M values are then added in bead order by one thread, exactly as the one-workgroup kernel does
//     (pigs_sampler.hip), so the trajectory does not depend on H -- bit for bit;
//   * an accepted move is committed by every workgroup to its own beads.  A workgroup only ever reads slices of its own
//     range (the pair action is local in imaginary time), so the worldline needs no cross-workgroup visibility
//     inside the launch.
// Residency: H x W <= CUs and a workgroup fills a CU's LDS (table image + per-wave scratch), so on a chip this context
// has to itself -- the library checks: one live context per device in the process -- every workgroup gets a CU of its
// own at once and the polling loops wait microseconds.  They are bounded all the same (another PROCESS may crowd the
// chip): a workgroup that waits in vain gives up after seconds and raises an error word the host checks at its next
// synchronisation.  (hipLaunchCooperativeKernel would state the residency requirement to the runtime; it is not used:
// same speed, and rocprofv3 of this ROCm release segfaults at exit after a cooperative launch.)
#include "pigs_device.h"
#include "pigs_k1_device.h"
#include "pigs_kernels.h"
#include "pigs_sampler_device.h"

#include <cstdlib>

namespace pigs {

namespace {

struct CmLds { size_t mt, ctl, pc, dS, red, tab, total; };

__host__ __device__ inline CmLds cm_layout(const DevParams &P, int nwaves, int H)
{
    const size_t M = P.M, D = P.dim;
    const size_t nbo = (M + H - 1) / H + 1;                          // beads of the largest range
    CmLds L;
    size_t b = 0;
    auto take = [&](size_t bytes) { const size_t at = b; b += (bytes + 15) & ~(size_t)15; return at; };
    L.mt  = take(kRing * 4);
    L.ctl = take(16 * 4);
    L.pc  = take(nbo * D * 8);
    L.dS  = take(M * 8);
    L.red = take((size_t)nwaves * kWaveLds);
    L.tab = take(((size_t)P.Nmax + 2 + 6) * 8);
    L.total = b;
    return L;
}

constexpr int kCmSpinLimit = 1 << 22;        // polls of one exchange before a workgroup gives up (seconds; a move takes ~25 us)

} // namespace

// first bead of range h of H over M beads
__host__ __device__ inline int cm_range(int M, int H, int h) { return (int)(((long long)M * h) / H); }

template <int DIM, int NT>
__global__ __launch_bounds__(NT, 1) void k_cm(
    DevParams P, SweepParams sp, int H, unsigned int seq0, double *__restrict__ paths, const double *__restrict__ VTimg,
    const double *__restrict__ WF, uint32_t *__restrict__ rng, unsigned long long *__restrict__ counters,
    const double *__restrict__ worm, unsigned long long *__restrict__ xch, int *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / kWave;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = blockIdx.x % P.nW, h = blockIdx.x / P.nW;          // ranges of a walker: blocks w, w + nW, ...
    const int M = P.M;
    const int b0 = cm_range(M, H, h), b1 = cm_range(M, H, h + 1), nbo = b1 - b0;

    const CmLds L = cm_layout(P, NW, H);
    uint32_t *mt  = reinterpret_cast<uint32_t *>(smem + L.mt);
    int      *ctl = reinterpret_cast<int *>(smem + L.ctl);           // [0] accepted [1] uniforms taken by the question [2] an exchange
                                                                     // timed out: nobody in this workgroup waits any more
    double   *pc  = reinterpret_cast<double *>(smem + L.pc);         // the particle's beads b0..b1-1, bead-major
    double   *dS  = reinterpret_cast<double *>(smem + L.dS);         // Delta S of every bead of the chain
    double   *red = reinterpret_cast<double *>(smem + L.red + (size_t)wid * kWaveLds);
    double   *lt  = reinterpret_cast<double *>(smem + L.tab);

    const size_t sl = slice_doubles(DIM, P.NpPad);
    double *Pw = paths + (size_t)w * M * sl;

    // generator state in the reference's block form (what mtsavef holds): words of the current block + index
    for (int t = tid; t < MT_N; t += NT) mt[t] = rng[(size_t)w * kRngWords + MT_N + t];
    int pos = (int)rng[(size_t)w * kRngWords + 2 * MT_N];            // next word, counted from the block's first
    {   // table image [0 VT(0)] VT(0..Nmax+1) [0 0 0 0] (pigs_k1_device.h, PipeTab)
        const int nimg = P.Nmax + 2 + 6;
        const double2 *src = reinterpret_cast<const double2 *>(VTimg);
        double2 *dst = reinterpret_cast<double2 *>(lt);
        for (int t = tid; t < nimg / 2; t += NT) dst[t] = src[t];
    }
    const PipeTab VTp{lt + 2, P.Nmax + 3};
    if (tid == 0) ctl[2] = 0;
    const bool isopen = sp.worm ? (int)worm[(size_t)w * kWormDoubles] != 0 : false;
    const int  pworm  = sp.worm ? (int)worm[(size_t)w * kWormDoubles + 1] - 1 : -1;
    __syncthreads();
    // every word the phase can take -- (dim + 1) per move -- and the rest of the block the last of them lies in (the
    // state saved at the end).  Word a + 624 comes from words a, a + 1, a + 397: 64 at a time.
    const int need = block_start(pos + P.Np * (DIM + 1)) + MT_N;
    if (wid == 0) {
        for (int f = MT_N; f < need; f += kWave) {
            const int a = f + lane;
            const uint32_t y = mt_next(mt[ring_w(a - MT_N)], mt[ring_w(a - MT_N + 1)], mt[ring_w(a - MT_N + MT_M)]);
            __builtin_amdgcn_wave_barrier();
            mt[ring_w(a)] = y;
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the first particle's beads of this range
    int pfirst = 0;
    if (isopen && pfirst == pworm) ++pfirst;
    if (tid < nbo * DIM && pfirst < P.Np)
        pc[tid] = Pw[(size_t)(b0 + tid / DIM) * sl + (size_t)(tid % DIM) * P.NpPad + pfirst];
    __syncthreads();

    unsigned int n_try = 0, n_acc = 0;
    for (int p = pfirst; p < P.Np; ) {
        int pn = p + 1;
        if (isopen && pn == pworm) ++pn;
        // the shift: dim uniforms (vpi_mod.f90:335-337)
        double dx[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) dx[k] = sp.delta_cm * (2.0 * mt_real(mt[ring_w(pos + k)]) - 1.0);
        pos += DIM;
        ++n_try;
        // the next particle's beads travel while this move is evaluated
        double nxv = 0.0;
        const bool mine = tid < nbo * DIM;
        const int eb = mine ? tid / DIM : 0, ek = mine ? tid - eb * DIM : 0;
        if (mine && pn < P.Np) nxv = Pw[(size_t)(b0 + eb) * sl + (size_t)ek * P.NpPad + pn];
        // Delta S of this range's beads: bead r*NW + (wid + r) mod NW in round r (a plain stride would give a wave only
        // odd or only even beads, and odd beads -- force terms -- cost 1.3x the even ones)
        // The range that holds the chain's LAST bead takes it FIRST: an end bead gathers the trial function's table from
        // global memory and costs a lone wave twice an inner bead (7 100 against 3 500 cycles), and with 81 (161) beads
        // on 16 waves it used to be the one bead of the last, otherwise empty round.  Which wave evaluates a bead when
        // changes no bit: every Delta S lands in dS[bead] and the sum runs in bead order.
        const bool last_first = b1 == M && nbo > 1;
        for (int r = 0; r * NW < nbo; ++r) {
            int i = wid + r;
            i = r * NW + (i >= NW ? i % NW : i);
            if (i >= nbo) continue;
            if (last_first) i = i == 0 ? nbo - 1 : i - 1;             // local order: nbo-1, 0, 1, ..., nbo-2
            double a[DIM], c[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) { c[k] = pc[i * DIM + k]; a[k] = wrap_coord(P, false, k, c[k] + dx[k]); }
            item_eval_pipe<DIM>(P, VTp, WF, Pw + (size_t)(b0 + i) * sl, p, b0 + i, a, c, lane, red, &dS[b0 + i], nullptr);
        }
        __syncthreads();
        // exchange: publish this range, collect the others'
        if (H > 1) {
            // two slots, taken in turn by consecutive moves: a workgroup overwrites a slot two moves later, and it gets
            // there only after the others published the move in between, i.e. after they finished reading this one
            const unsigned int tag = seq0 + n_try;
            unsigned long long *X = xch + ((size_t)w * 2 + (n_try & 1)) * (size_t)M * 2;
            const int spin_limit = (sp.cm_fault & 1) ? 2048 : kCmSpinLimit;
            if (tid < nbo && !((sp.cm_fault & 1) && h == H - 1)) {          // (cm_fault: this range's values never arrive)
                const unsigned long long v = (unsigned long long)__double_as_longlong(dS[b0 + tid]);
                const unsigned long long hi = (unsigned long long)tag << 32;
                __hip_atomic_store(&X[(size_t)(b0 + tid) * 2],     hi | (v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&X[(size_t)(b0 + tid) * 2 + 1], hi | (v >> 32),           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int j = tid; j < M - nbo; j += NT) {
                const int b = j < b0 ? j : j + nbo;                  // the beads outside [b0, b1)
                unsigned long long lo = 0, hi = 0;
                int spins = 0;
                for (;;) {
                    lo = __hip_atomic_load(&X[(size_t)b * 2],     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi = __hip_atomic_load(&X[(size_t)b * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((unsigned int)(lo >> 32) == tag && (unsigned int)(hi >> 32) == tag) || *(volatile int *)&ctl[2]) break;
                    if (++spins > spin_limit) {                     // give up once, for the whole workgroup and the rest of the launch
                        *(volatile int *)&ctl[2] = 1;
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                dS[b] = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
            }
            __syncthreads();
        }
        // Metropolis on exp(-sum) with the sum in bead order (vpi_mod.f90:354-364), by one thread as in pigs_sampler.hip
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < M; ++i) t = t + dS[i];
            const double a = -t;
            int ok = 1, took = 0;
            if (!(a >= -0x1p-54)) {                                  // else exp(a) rounds to >= 1: no uniform is drawn
                const double e = exp(a);
                if (!(e >= 1.0)) { ok = e >= mt_real(mt[ring_w(pos)]); took = 1; }
            }
            if (ctl[2]) ok = 0;                                       // an exchange timed out: the sum is garbage, nothing more is committed
            ctl[0] = ok; ctl[1] = took;
        }
        __syncthreads();
        const bool ok = ctl[0] != 0;
        pos += ctl[1];
        n_acc += ok;
        if (mine) {
            if (ok) Pw[(size_t)(b0 + eb) * sl + (size_t)ek * P.NpPad + p] = wrap_coord(P, false, ek, pc[tid] + dx[ek]);
            pc[tid] = nxv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // later moves read the committed rows: stores drained
        __syncthreads();
        p = pn;
    }

    // generator state back in block form, counters: the walker's first workgroup
    if (h == 0) {
        const int bs_save = block_start(pos);
        for (int t = tid; t < MT_N; t += NT) rng[(size_t)w * kRngWords + MT_N + t] = mt[ring_w(bs_save + t)];
        if (tid == 0) {
            rng[(size_t)w * kRngWords + 2 * MT_N] = (uint32_t)(pos - bs_save);
            counters[(size_t)w * kCounters + 14] += n_try;
            counters[(size_t)w * kCounters + 0] += n_acc;
        }
    }
}

// workgroup size for H ranges per walker: sixteen waves (four per SIMD, 128 registers each) where their scratch fits next
// to the table image, eight otherwise; a thread per coordinate of the range's beads; 0: does not fit
static int cm_threads(const DevParams &P, int H)
{
    static const int want = getenv("PIGS_CM_THREADS") ? atoi(getenv("PIGS_CM_THREADS")) : 1024;
    const int rows = (cm_range(P.M, H, 1) + 1) * P.dim;
    if (want >= 1024 && cm_layout(P, 16, H).total <= 160 * 1024 && rows <= 1024) return 1024;
    if (cm_layout(P, 8, H).total <= 160 * 1024 && rows <= 512) return 512;
    return 0;
}

// the most workgroups per walker the chip holds together (1..4), 0 where the kernel does not apply: trapped systems,
// more than 256 particles (four 64-partner passes per bead), a step without TranslateChain, a worldline beyond the ring
// of random words or the LDS
int cm_helpers(const DevParams &P, const SweepParams &sp, int n_cu)
{
    if (P.trap || (P.Nmax & 1) || P.Np > 256 || !sp.do_cm) return 0;
    if (((MT_N + P.Np * (P.dim + 1)) / MT_N + 1) * MT_N + kWave > kRing) return 0;
    int H = n_cu / P.nW;
    if (H > 4) H = 4;
    if (H < 1) H = 1;
    if (cm_threads(P, H) == 0) return 0;         // (more workgroups = shorter ranges: if H does not fit, fewer do not either)
    return H;
}

// does the kernel fit with exactly H workgroups per walker?  (Fewer workgroups = longer ranges: H = 2 can fit where H = 1 does
// not -- 321 beads: 966 rows and 165 KB of LDS on one workgroup -- so a caller that lowers H re-checks.)
bool cm_fits(const DevParams &P, int H) { return H >= 1 && H <= 4 && cm_threads(P, H) != 0; }

size_t cm_exchange_words(const DevParams &P) { return (size_t)P.nW * 2 * P.M * 2; }

hipError_t launch_cm(const DevParams &P, const SweepParams &sp, int H, unsigned int seq0, double *paths, const double *VTimg,
                     const double *WF, uint32_t *rng, unsigned long long *counters, const double *worm,
                     unsigned long long *xch, int *err, hipStream_t st)
{
    if (H < 1 || H > 4 || P.trap || (P.Nmax & 1) || P.Np > 256) return hipErrorInvalidValue;
    const int nt = cm_threads(P, H);
    if (nt == 0) return hipErrorInvalidValue;
    const size_t lds = cm_layout(P, nt / kWave, H).total;
    hipError_t e = hipSuccess;
    DevParams Pk = P;
    SweepParams spk = sp;
    void *args[] = {&Pk, &spk, &H, &seq0, &paths, &VTimg, &WF, &rng, &counters, &worm, &xch, &err};
#define CALLC(D, NT)                                                                                               \
    do {                                                                                                           \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cm<D, NT>),                                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
        if (e == hipSuccess)                                                                                       \
            e = hipLaunchKernel(reinterpret_cast<const void *>(k_cm<D, NT>), dim3(P.nW * H), dim3(NT), args,       \
                                (unsigned int)lds, st);                                                            \
    } while (0)
    if (nt == 1024) { if (P.dim == 1) CALLC(1, 1024); else if (P.dim == 2) CALLC(2, 1024); else CALLC(3, 1024); }
    else            { if (P.dim == 1) CALLC(1, 512); else if (P.dim == 2) CALLC(2, 512); else CALLC(3, 512); }
#undef CALLC
    return e;
}

} // namespace pigs
