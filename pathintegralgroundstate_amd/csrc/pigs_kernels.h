// pigs_kernels.h -- host-callable launchers of the gfx950 kernels (pigs_kernels.hip).
// All launches are asynchronous on the given stream and allocate nothing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pigs_device.h"

namespace pigs {

// K1 kernel variants (pigs_k1.hip)
enum K1Variant {
    K1_AUTO = 0,            // library's choice
    K1_V1 = 1,              // plain statement
    K1_V2 = 2,              // exact short division, fused sqrt/rinv, shared butterfly
    K1_V2_LDS = 3,          // + VTable in LDS
    K1_V2_COMPACT = 4,      // + in-cutoff compaction (global table)
    K1_V2_LDS_COMPACT = 5,  // + both
    K1_V2_PREFETCH = 6,     // v2 with all partner loads of an item issued up front (Np <= 256)
    K1_FAST = 7,            // v2 with the short arithmetic (~1 ulp per term instead of the reference's rounding; PBC)
    K1_FAST_PREFETCH = 8,   // + prefetch
    K1_FAST_LDS = 9,        // fast with the VTable in LDS (one 1024-thread workgroup per CU)
    K1_FAST_LDS_PREFETCH = 10,
    K1_PIPE = 11,           // persistent: LDS table, branch-free short arithmetic, per-workgroup item queue (Np <= 256)
    K1_PIPE2 = 12           // + item records and partner coordinates requested one item / two passes ahead
};

hipError_t launch_delta_action(const DevParams &P, int variant, const double *paths, const double *VT,
                               const double *WF, int n_items, const int32_t *walker,
                               const int32_t *ip, const int32_t *ib, const double *xnew,
                               const double *xold, double *out, double *parts, hipStream_t st);

hipError_t launch_slice_energy(const DevParams &P, const double *paths, const double *VT, const double *VTimg,
                               int n_slots, const int32_t *slot_walker, const int32_t *slot_ib,
                               int force_mode, int want_spring, double *out, hipStream_t st);

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
constexpr int kEvInts = 64;         // event log of one MC step
struct SweepParams {
    int32_t Nlev, Nstag, Lstag, do_cm;
    int32_t open_attempt, parts, worm, swapping;  // worm: CWorm > 0 (open/close/swap sector sampled); parts: sections of the
                                                  // step a launch runs (1 open/close attempt, 2 diagonal moves, 4 worm moves)
    int32_t Nobdm, Nbin, Npw, staging;            // staging: sampling = 'sta' in the diagonal sector
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
int sweep_form(const DevParams &P, const SweepParams &sp, int threads);   // workgroup size launch_sweep uses for a request

hipError_t launch_selftest_fastmath(const DevParams &P, unsigned long long seed, int blocks, int iters,
                                    unsigned long long *d_bad, hipStream_t st);

hipError_t launch_pack(const DevParams &P, double *paths, const double *raw, int w0, int nw,
                       hipStream_t st);
hipError_t launch_unpack(const DevParams &P, const double *paths, double *raw, int w0, int nw,
                         hipStream_t st);

} // namespace pigs
