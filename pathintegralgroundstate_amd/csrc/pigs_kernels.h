// pigs_kernels.h -- host-callable launchers of the gfx950 kernels (pigs_kernels.hip).
// All launches are asynchronous on the given stream and allocate nothing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pigs_device.h"

namespace pigs {

// K1 kernel variants (pigs_k1.hip).  The numbers are part of the tuning interface (pigs_set_tuning "k1_variant");
// 3-6 and 9-11 were A/B forms of round 1 (LDS table / compaction / prefetch-only / first persistent kernel) that
// measured slower and are gone: profiles/r01_k1_variants_ab*.txt keeps their numbers.
enum K1Variant {
    K1_AUTO = 0,            // library's choice (by system, never by launch size: see launch_delta_action)
    K1_V1 = 1,              // plain statement
    K1_V2 = 2,              // exact-term: exact short division, fused sqrt/rinv, one LDS-transpose reduction
    K1_FAST = 7,            // v2 with the short arithmetic (~1 ulp per term instead of the reference's rounding; PBC)
    K1_FAST_PREFETCH = 8,   // + all partner loads of an item issued up front (Np <= 256)
    K1_PIPE2 = 12,          // persistent: LDS table image, branch-free short arithmetic, item queue, look-ahead loads (Np <= 256)
    K1_GRID = 13,           // pipe2's per-item arithmetic on a plain grid (global table image): identical bits, small launches
    K1_REFORDER = 14        // validation: exact terms added in the reference's jp order -- Delta S bit-identical to the reference
};
inline bool k1_variant_valid(int v)
{
    return v == K1_AUTO || v == K1_V1 || v == K1_V2 || v == K1_FAST || v == K1_FAST_PREFETCH || v == K1_PIPE2 ||
           v == K1_GRID || v == K1_REFORDER;
}

hipError_t launch_delta_action(const DevParams &P, int variant, const double *paths, const double *VT,
                               const double *VTimg, const double *WF, int n_items, const int32_t *walker,
                               const int32_t *ip, const int32_t *ib, const double *xnew,
                               const double *xold, double *out, double *parts, hipStream_t st);

// max_blocks > 0 caps the persistent form's grid (estimators that run NEXT TO the sampler take half the chip)
hipError_t launch_slice_energy(const DevParams &P, const double *paths, const double *VT, const double *VTimg,
                               int n_slots, const int32_t *slot_walker, const int32_t *slot_ib,
                               int force_mode, int want_spring, double *out, hipStream_t st, int max_blocks = 0);

hipError_t launch_therm_combine(const DevParams &P, int n, const double *slices, double *E,
                                double *Ec, double *Ep, hipStream_t st);

hipError_t launch_local_energy(const DevParams &P, const double *paths, const double *VT,
                               const double *WF, int n_slots, const int32_t *slot_walker, int ib,
                               double *out, hipStream_t st);

hipError_t launch_structure(const DevParams &P, const double *paths, int n_slots, const int32_t *slot_walker,
                            int ib, int Nbin, double rbin, int Nk, double *gr, double *Sk, hipStream_t st);

hipError_t launch_commit_beads(const DevParams &P, double *paths, int64_t n, const int32_t *walker,
                               const int32_t *ip, const int32_t *ib, const double *x, hipStream_t st);

hipError_t launch_swap_tails(const DevParams &P, double *paths, int walker, int iw, int ik,
                             hipStream_t st);

// K6: device-resident sampler (pigs_sampler.hip)
// per-walker generator state in global memory: 624 sliding words, 624 block-form words, position
constexpr int kRngWords = 2 * 624 + 1;
constexpr int kCounters = 16;       // per-walker move counters (pigs_sampler.hip)
constexpr int kWormDoubles = 8;     // isopen, iworm, xend(:,1), xend(:,2)
constexpr int kEvInts = 64;         // event log of one MC step: at least this many ints per walker (SweepParams.ev_ints)
struct SweepParams {
    int32_t Nlev, Nstag, Lstag, do_cm;
    int32_t open_attempt, parts, worm, swapping;  // worm: CWorm > 0 (open/close/swap sector sampled); parts: sections of the
                                                  // step a launch runs (1 open/close attempt, 2 diagonal moves, 4 worm moves)
    int32_t Nobdm, Nbin, Npw, staging;            // staging: sampling = 'sta' in the diagonal sector
    int32_t ev_ints, cm_fault;                    // ints per walker of the event log: max(kEvInts, 4 + 2*(1+Nobdm)); cm_fault: TEST
                                                  // ONLY (tuning key "cm_fault"): the last range of every walker in k_cm withholds
                                                  // its Delta S and waiting workgroups give up after 2048 polls (forces the time-out path)
    double  delta_cm, log_cworm_density, rbin;
};
hipError_t launch_sweep(const DevParams &P, const SweepParams &sp, int threads, double *paths, const double *VT,
                        const double *VTimg, const double *WF, uint32_t *rng, unsigned long long *counters, double *worm,
                        int *evlog, double *nrho, const double *dklog, hipStream_t st);
hipError_t launch_slice_gather(const DevParams &P, const double *paths, int ib, double *out, hipStream_t st);
size_t sweep_lds_bytes(const DevParams &P, const SweepParams &sp, int threads);
// pigs_diag.hip: the diagonal moves of a periodic system with sampling = 'bis' as a stage machine (one workgroup per walker)
bool diag_supported(const DevParams &P, const SweepParams &sp);
int diag_form(const DevParams &P, const SweepParams &sp, int threads);      // workgroup size launch_diag uses (0: does not fit)
hipError_t launch_diag(const DevParams &P, const SweepParams &sp, int threads, double *paths, const double *VTimg,
                       const double *WF, uint32_t *rng, unsigned long long *counters, const double *worm, hipStream_t st);
// pigs_cm.hip: the TranslateChain moves of a periodic system by H cooperating workgroups per walker
int cm_helpers(const DevParams &P, const SweepParams &sp, int n_cu);        // H the chip and the kernel allow (0: none)
bool cm_fits(const DevParams &P, int H);                                    // the kernel fits with exactly H workgroups per walker
size_t cm_exchange_words(const DevParams &P);                               // 64-bit words of the exchange buffer
hipError_t launch_cm(const DevParams &P, const SweepParams &sp, int H, unsigned int seq0, double *paths, const double *VTimg,
                     const double *WF, uint32_t *rng, unsigned long long *counters, const double *worm,
                     unsigned long long *xch, int *err, hipStream_t st);
int sweep_form(const DevParams &P, const SweepParams &sp, int threads);   // workgroup size launch_sweep uses for a request

hipError_t launch_stream_read(const double *a, size_t doubles, int blocks, double *sink, hipStream_t st);
// argument `idx` of the log self-test (pigs_selftest_log): the device sampler's domain -- a uniform of the stream
// k/(2^32-1); a polar radius u1^2+u2^2 <= 1 of two such uniforms; a random mantissa over 2^-69 .. 2; the near-one branch
__host__ __device__ inline double selftest_log_arg(unsigned long long idx, unsigned long long seed)
{
    unsigned long long z = (idx + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    unsigned long long y = (z + 0x632BE59BD9B4E019ull) * 0xD1342543DE82EF95ull; y ^= y >> 29;
    const double ua = (double)(unsigned int)z / 4294967295.0, ub = (double)(unsigned int)y / 4294967295.0;
    switch (idx & 3) {
    case 0: return ua;
    case 1: { const double u1 = 2.0 * ua - 1.0, u2 = 2.0 * ub - 1.0; const double q = u1 * u1 + u2 * u2; return q > 1.0 ? q - 1.0 : q; }
    case 2: { const unsigned long long ix = (z >> 12) | ((unsigned long long)(0x3ff - (y % 70)) << 52);
              double x; __builtin_memcpy(&x, &ix, 8); return x; }
    default: return 0.9375 + (double)(z >> 11) * (1.0 / 9007199254740992.0) * 0.13;
    }
}
hipError_t launch_selftest_log(unsigned long long first, unsigned long long n, unsigned long long seed, double *d_out, hipStream_t st);
hipError_t launch_selftest_fastmath(const DevParams &P, unsigned long long seed, int blocks, int iters,
                                    unsigned long long *d_bad, hipStream_t st);

hipError_t launch_pack(const DevParams &P, double *paths, const double *raw, int w0, int nw,
                       hipStream_t st);
hipError_t launch_unpack(const DevParams &P, const double *paths, double *raw, int w0, int nw,
                         hipStream_t st);

} // namespace pigs
