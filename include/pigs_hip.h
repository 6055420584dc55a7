/* pigs_hip.h -- C ABI of libpigs_hip.so: the MI355X (gfx950) drop-in for the PIGS
 * action / energy hot path of amaciarey/PathIntegralGroundState.
 *
 * The reference has no FFI: its boundary is the Fortran procedure interface
 * (SURVEY.md §8b).  Each entry point below names the reference procedure(s) it
 * replaces (file:line under the reference tree).  A Fortran host binds these with
 * ISO_C_BINDING (pathintegralgroundstate_amd/host/pigs_capi.f90).
 *
 * Conventions (identical to the reference so that a Fortran caller passes its arrays
 * unchanged): fp64 everywhere; arrays column-major; Path(dim,Np,0:2*Nb); tables
 * F(0:Nmax+1) with the pointer at element 0; particle indices ip are 1-BASED, bead
 * indices ib 0-BASED, walker indices 0-based (walkers are new: the reference has one).
 * Every function returns PIGS_OK (0) or a negative pigs_status; nothing calls exit/stop.
 * A context is bound to one device and one stream; calls on one context must be
 * serialized by the caller (one host thread per GPU); there is no global mutable state.
 */
#ifndef PIGS_HIP_H
#define PIGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIGS_ABI_VERSION 1
#define PIGS_MAXDIM 3

typedef enum pigs_status {
    PIGS_OK              = 0,
    PIGS_ERR_ARG         = -1,  /* bad argument / out-of-range index               */
    PIGS_ERR_HIP         = -2,  /* HIP runtime error (pigs_last_error has the text) */
    PIGS_ERR_NO_DEVICE   = -3,  /* no gfx950 device visible                        */
    PIGS_ERR_UNSUPPORTED = -4,  /* e.g. v_table=F with force terms (reference Q3)  */
    PIGS_ERR_COMM        = -5   /* RCCL error                                      */
} pigs_status;

/* The module globals the reference hot path reads implicitly
 * (global_mod.f90:5-12: dim,Np,Nb,Nmax,dr,rcut2,wf_table,v_table,Lbox,LboxHalf;
 *  system_mod.f90:8-9: Rm,a_ho) plus dt, which the reference passes per call. */
typedef struct pigs_params {
    int32_t dim, Np, Nb, Nmax;
    int32_t trap, wf_table, v_table, reserved;
    double  dr, rcut2, dt, Rm;
    double  Lbox[PIGS_MAXDIM];
    double  a_ho[PIGS_MAXDIM];
} pigs_params;

typedef struct pigs_ctx pigs_ctx;

/* ---- lifetime ------------------------------------------------------------------ */
/* Replaces the module-global set-up of vpi.f90:76-153 for the hot path: uploads both
 * tables (built host-side by the caller exactly as vpi_mod.f90:84-145 builds them, or
 * by pigs_build_tables) and reserves HBM for n_walkers resident worldlines. */
/* wf_table = 0 (the reference's default, vpi_mod.f90:59): the trial function is evaluated analytically (McMillan
 * u(r) = -0.5 (Rm/r)^5, system_mod.f90:38-66) and LogWF may be NULL.  v_table = 1 is mandatory (quirk Q3). */
int pigs_ctx_create(const pigs_params *p, const double *VTable, const double *LogWF,
                    int32_t n_walkers, int32_t device_id, pigs_ctx **out);
int pigs_ctx_destroy(pigs_ctx *ctx);
const char *pigs_last_error(void);
int pigs_abi_version(void);
int pigs_device_count(int32_t *n);
/* Blocks until all work queued on the context's stream has finished. */
int pigs_sync(pigs_ctx *ctx);
/* The context's hipStream_t (as void*) so a host can order its own work against it. */
int pigs_stream(pigs_ctx *ctx, void **hip_stream);

/* Tuning knobs.
 *   "k1_variant": which Delta-S kernel pigs_delta_action_* runs.
 *        0  auto (default).  The arithmetic follows from the SYSTEM, never from the size of a launch (an item's bits
 *           must not depend on what else travels with it): periodic systems with Np <= 256 use the short arithmetic
 *           -- every per-pair term to ~1 ulp, cutoff membership unchanged -- in the persistent
 *           LDS-table kernel (12) for launches of >= 16 items per CU and in its plain-grid twin (13, identical bits)
 *           below that; periodic systems beyond 256 particles use 7; trapped systems use 2
 *        1  plain statement of the reference's arithmetic          2  same terms bit for bit, short exact division
 *        7  short arithmetic on the global table                   8  7 + partner loads issued up front
 *        12 persistent LDS-table kernel, branch-free short arithmetic, look-ahead loads
 *        13 the per-item arithmetic of 12 on a plain grid (global table image)
 *        14 validation: the terms of 2 added in the reference's jp order -- Delta S, DeltaPot, DeltaF2 and DeltaLogPsi
 *           equal the reference's bit for bit (BASELINE config 2's "pair-action kernel vs CPU bit-compare")
 *      1, 2 and 14 sum bit-identical terms (1 / 2 in lane-strided order); 7-13 agree with them to ~1e-15 per term.
 *      (3-6, 9-11 were A/B forms of round 1 and are gone.)
 *      The environment variable PIGS_K1_VARIANT presets this key at pigs_ctx_create (test hook).
 *   "sweep_threads": workgroup size of the device-resident sampler (>= 512: the one-workgroup-per-CU form, 256: three
 *      workgroups per CU).  "sweep_split": 1 runs the diagonal bisection moves of a periodic system in the stage-machine
 *      kernel (pigs_diag.hip, three launches per MC step) instead of the one-launch kernel; it is chosen automatically
 *      for Nlev > 4.
 *   "cm_split": the TranslateChain moves of a periodic system (Np <= 256) by H cooperating workgroups per walker
 *      (pigs_cm.hip: bead ranges on H CUs, Delta S exchanged and added in bead order -- the trajectory does not depend
 *      on H, bit for bit).  -1 (default): H = min(4, CUs / walkers), at least 1 (one workgroup per walker exchanges
 *      nothing and is still the faster TranslateChain); 0: inside the sweep kernel; 1..4: at most that many.  H > 1 only
 *      while the context is the process's only one on its device; a cooperating workgroup that waits in vain (several
 *      PROCESSES crowding one chip: set 1 there) gives up after seconds and the next pigs_sync returns PIGS_ERR_HIP.
 *   "cm_exclusive": 1 = the caller vouches that the process's other contexts on this device are idle while this one
 *      samples, so H > 1 stays allowed although it is not the only live context (bench.py's extra legs).
 *   "cm_shared": 1 = several contexts of this process sample on this device AT ONCE (walker shards on one GPU): their
 *      TranslateChain kernels are chained through one event per device, never two at a time, so that H > 1 stays safe
 *      next to the other contexts' sweep kernels -- set "cm_split" so that H x walkers + the others' walkers <= CUs
 *      (two contexts of 64 walkers on 256 CUs: 3).  The shards then run staggered: one's TranslateChain on the CUs the
 *      other's bisection phase leaves idle.
 *   "cm_fault": TEST ONLY -- forces the time-out of that exchange once. */
int pigs_set_tuning(pigs_ctx *ctx, const char *key, int32_t value);
/* Device self-test: the kernels' short exact division / sqrt forms against IEEE `/` and sqrt()
 * on blocks*256*iters random operands; bad[0..3] = mismatch counts (sqrt, n/r, r/dr, n/dr). */
int pigs_selftest_fastmath(pigs_ctx *ctx, int32_t blocks, int32_t iters, uint64_t bad[4]);
/* Device self-test: the log() inside the device-resident sampler's Box-Muller transform (csrc/pigs_log_host.h: glibc's
 * algorithm and constants, the reference's `log` of random_mod.f90:213) against THIS host's libm, bit for bit, on n
 * arguments of the sampler's domain (stream uniforms, polar radii, random mantissas, the near-one branch).
 * *mismatches must come back 0 for the sampler's worldlines to be bit-identical to the reference's; *first_bad (may be
 * NULL) = one differing argument. */
int pigs_selftest_log(pigs_ctx *ctx, int64_t n, uint64_t seed, uint64_t *mismatches, double *first_bad);
/* Measurement aid: `reps` plain streaming reads of the context's resident worldlines (the bytes a full-chain Delta-S
 * stage reads) by a kernel that does nothing else; *bytes per pass, *seconds per pass (HIP events on the context's
 * stream).  bench.py quotes it next to K1's roofline as the rate this chip's memory system delivers to a reader. */
int pigs_selftest_stream_read(pigs_ctx *ctx, int32_t reps, double *bytes, double *seconds);

/* Host-side table fill: JastrowTable / PotentialTable (vpi_mod.f90:84-145) over
 * LogPsi / Potential (system_mod.f90:38-66,136-182), including dr = rmax/real(Nmax-1)
 * and the ghost cells.  Arrays hold Nmax+2 doubles. */
int pigs_build_tables(int32_t Nmax, double Rm, double rmax, double *VTable, double *LogWF,
                      double *dr_out);
/* Same with a choice of pair potential.  The reference selects its potential by editing
 * system_mod.f90 and recompiling: kind 0 = Aziz-II HFD-B(HE) (the active code,
 * system_mod.f90:136-182), 1 = Lennard-Jones V0*(1/r^6-1)/r^6 with V0 = 22.0228 (the commented
 * block system_mod.f90:70-83; BASELINE config 2), 2 = dipolar 1/r^3 (BASELINE config 5). */
#define PIGS_POT_AZIZ2 0
#define PIGS_POT_LJ 1
#define PIGS_POT_DIPOLAR 2
int pigs_build_tables_kind(int32_t kind, int32_t Nmax, double Rm, double rmax, double *VTable,
                           double *LogWF, double *dr_out);

/* ---- worldline residency (replaces the host array Path, vpi.f90:134) ------------ */
int pigs_path_upload(pigs_ctx *ctx, int32_t walker, const double *Path);
int pigs_path_download(pigs_ctx *ctx, int32_t walker, double *Path);
/* all walkers back to back: Paths(dim,Np,0:2*Nb,n_walkers) */
int pigs_path_upload_all(pigs_ctx *ctx, const double *Paths);
int pigs_path_download_all(pigs_ctx *ctx, double *Paths);

/* ---- K1: batched Delta S  (replaces the 24 `call UpdateAction` sites of vpi_mod.f90,
 * i.e. UpdateAction/UpdatePot/UpdateWf, vpi_mod.f90:2491-2841, + GreenFunction opt 0,
 * global_mod.f90:19-72) --------------------------------------------------------------
 * Item i moves bead ib[i] of particle ip[i] of walker walker[i] from xold(:,i) to
 * xnew(:,i); DeltaS[i] has exactly the semantics of UpdateAction's output.  Row ip of
 * the resident slice is never read (the reference's aliasing contract).
 * xnew/xold are (dim,n_items) column-major.  Host-pointer form: synchronous. */
int pigs_delta_action_batch(pigs_ctx *ctx, int64_t n_items,
                            const int32_t *walker, const int32_t *ip, const int32_t *ib,
                            const double *xnew, const double *xold, double *DeltaS);
/* Same with DEVICE pointers, asynchronous on the context's stream (inputs resident in HBM). */
int pigs_delta_action_batch_dev(pigs_ctx *ctx, int64_t n_items,
                                const int32_t *d_walker, const int32_t *d_ip, const int32_t *d_ib,
                                const double *d_xnew, const double *d_xold, double *d_DeltaS);
/* Latency-critical form for a host-driven sampler: the library owns PINNED, device-mapped
 * staging arrays that the host fills and the kernels read in place over PCIe (no memcpy calls);
 * DeltaS is written by the kernel straight into pinned host memory.  pigs_stage_reserve returns
 * (and, when growing, preserves the first `keep` items of) the item arrays;
 * pigs_delta_action_staged evaluates items [0,n) and returns when DeltaS[0,n) is valid. */
int pigs_stage_reserve(pigs_ctx *ctx, int64_t capacity, int64_t keep, int32_t **walker, int32_t **ip,
                       int32_t **ib, double **xnew, double **xold, double **DeltaS);
int pigs_delta_action_staged(pigs_ctx *ctx, int64_t n_items);
/* Same for commits: arrays (walker, ip, ib, x); pigs_commit_staged is asynchronous -- the arrays
 * may be rewritten only after the next synchronous call on this context has returned. */
int pigs_commit_reserve(pigs_ctx *ctx, int64_t capacity, int64_t keep, int32_t **walker, int32_t **ip,
                        int32_t **ib, double **x);
int pigs_commit_staged(pigs_ctx *ctx, int64_t n);

/* Test hook: the components UpdateAction combines (DeltaPot, DeltaF2, DeltaLogPsi), 3 per item. */
int pigs_delta_action_parts(pigs_ctx *ctx, int64_t n_items,
                            const int32_t *walker, const int32_t *ip, const int32_t *ib,
                            const double *xnew, const double *xold, double *parts);

/* A commit list is a sequence of assignments in the caller's program order: if a bead appears more than once, the last
 * value wins. */
/* ---- K5: commit (replaces `Path(k,ip,ib)=xnew(k)` on accept / OldChain restore, e.g.
 * vpi_mod.f90:370-374,948,987-995) -------------------------------------------------- */
int pigs_commit_beads(pigs_ctx *ctx, int64_t n, const int32_t *walker, const int32_t *ip,
                      const int32_t *ib, const double *x /* (dim,n) */);
/* Swap accept branch, vpi_mod.f90:2454-2464: exchange beads Nb..2Nb of particles iw, ik. */
int pigs_swap_tails(pigs_ctx *ctx, int32_t walker, int32_t iw, int32_t ik);

/* ---- K6: device-resident sampler (reference vpi.f90:297-439 with the movers TranslateChain and
 * MoveHeadBisection, MoveTailBisection, Bisection (sampling='bis') or MoveHead, MoveTail, Staging
 * (sampling='sta') of vpi_mod.f90).
 * One launch advances EVERY resident walker by one MC step with no host round trip: random
 * numbers (each walker's own MT19937 stream, identical to the reference's for its seed),
 * proposals, Delta S, Metropolis and commit all run on the GPU.  With CWorm > 0 the worm sector is
 * sampled too (OpenChain, CloseChain, TranslateHalfChain, MoveHead/TailHalfChain, StagingHalfChain,
 * Swap, OBDM histogram: vpi_mod.f90:383-476,1376-2487, sample_mod.f90:477-526); with CWorm = 0 the
 * reference's never-accepted open attempt (quirk Q11) is drawn so that the stream stays aligned. */
typedef struct pigs_sweep_params {
    int32_t Nlev, Nstag, CMFreq, Lstag;   /* namelist samp: bisection level, repetitions, CM period, Lstag */
    double  delta_cm;                     /* effective CM step (vpi.f90:93/123 scaling already applied)      */
    /* worm sector (namelist obdm); CWorm = 0 keeps every walker in the diagonal sector */
    double  CWorm, density, rbin;         /* rbin = rcut/real(Nbin) (vpi.f90:128), for the OBDM histogram     */
    int32_t swapping, Nobdm, Nbin, Npw;
    int32_t sampling, reserved;           /* diagonal movers: 0 = 'bis' (bisection), 1 = 'sta' (staging) */
} pigs_sweep_params;
int pigs_sampler_init(pigs_ctx *ctx, const pigs_sweep_params *sp);
/* seed walker's stream as the reference's sgrnd(seed) does */
int pigs_sampler_seed(pigs_ctx *ctx, int32_t walker, int32_t seed);
/* continue from a generator state in the reference's block form (mti, mt(0:623)) */
int pigs_sampler_set_rng(pigs_ctx *ctx, int32_t walker, int32_t mti, const int32_t mt[624]);
/* the walker's generator state, again in the reference's block form (what mtsavef writes) */
int pigs_sampler_get_rng(pigs_ctx *ctx, int32_t walker, int32_t *mti, int32_t mt[624]);
/* one MC step (istep is the 1-based step number: CM moves when mod(istep,CMFreq)==0); asynchronous */
int pigs_sampler_step(pigs_ctx *ctx, int32_t istep);
/* accepted-move counters per walker since pigs_sampler_init: acc[4*w+{0,1,2,3}] = CM, head, tail, bisection */
int pigs_sampler_counters(pigs_ctx *ctx, int64_t *acc);
/* accepted/attempted counters per walker since pigs_sampler_init, 16 per walker:
 * 0 cm 1 head 2 tail 3 bisection 4 try_open 5 acc_open 6 try_close 7 acc_close 8 cm_half 9 head_half
 * 10 tail_half 11 staging_half 12 try_swap 13 acc_swap 14 try_cm 15 try_stag */
int pigs_sampler_counters16(pigs_ctx *ctx, int64_t *cnt);
/* worm state of every walker: isopen[w], iworm[w] (1-based), xend(dim,2,w) */
int pigs_sampler_get_worm(pigs_ctx *ctx, int32_t *isopen, int32_t *iworm, double *xend);
int pigs_sampler_set_worm(pigs_ctx *ctx, const int32_t *isopen, const int32_t *iworm, const double *xend);
/* events of the LAST step, pigs_sampler_event_ints() = max(64, 4 + 2*(1+Nobdm)) ints per walker: [0] n, [1] isopen after
 * the step, then (code,arg) pairs in order: 1 open accepted (arg iworm) 2 close accepted 3 swap accepted (arg partner) */
int pigs_sampler_event_ints(pigs_ctx *ctx, int32_t *n);
int pigs_sampler_events(pigs_ctx *ctx, int32_t *events);
/* OBDM histogram nrho(0:Npw,Nbin,w) accumulated on the device since walker w's last reset; reset == NULL
 * keeps everything, otherwise walker w's histogram is zeroed after the copy where reset[w] != 0 (the
 * reference zeroes nrho only in blocks that normalise it, vpi.f90:520-532) */
int pigs_sampler_nrho(pigs_ctx *ctx, double *nrho, const int32_t *reset);
/* slice ib of every walker in the reference layout R(dim,Np,n_walkers) (for host-side g(r), S(k)) */
int pigs_slice_download(pigs_ctx *ctx, int32_t ib, double *R);

/* ---- K2/K3: estimator-side sums --------------------------------------------------- */
/* PotentialEnergy (sample_mod.f90:13-150) on one resident slice (test hook). */
int pigs_potential_energy_slice(pigs_ctx *ctx, int32_t walker, int32_t ib, int32_t want_F2,
                                double *Pot, double *F2);
/* ThermEnergy (sample_mod.f90:323-388) for walkers[0..n): E, Ec, Ep per walker.
 * walkers == NULL means walkers 0..n-1. */
int pigs_therm_energy_batch(pigs_ctx *ctx, int32_t n, const int32_t *walkers,
                            double *E, double *Ec, double *Ep);
/* ---- K4: LocalEnergy (sample_mod.f90:154-319) on slice ib (0 or 2*Nb in vpi.f90:443-444). */
int pigs_local_energy_batch(pigs_ctx *ctx, int32_t n, const int32_t *walkers, int32_t ib,
                            double *E, double *Kin, double *Pot);

/* ---- K7: structural estimators of slice ib (vpi.f90:466-469 uses ib = Nb): PairCorrelation and
 * StructureFactor (sample_mod.f90:392-473) for walkers[0..n).  gr(Nbin,n) receives this slice's
 * histogram increments (+2 per pair inside the cutoff, exact), Sk(dim,Nk,n) the per-k increments.
 * The caller accumulates and normalises as the reference does.  PBC only. */
int pigs_structure_batch(pigs_ctx *ctx, int32_t n, const int32_t *walkers, int32_t ib, int32_t Nbin,
                         double rbin, int32_t Nk, double *gr, double *Sk);
/* Every diagonal-sector estimator of one MC step (what vpi.f90:443-469 evaluates after a diagonal step) for n walkers in
 * ONE call: LocalEnergy at slices 0 and 2Nb (sample_mod.f90:154-319), ThermEnergy (sample_mod.f90:323-388), and -- when
 * gr and Sk are given (PBC runs) -- g(r) and S(k) at slice Nb (sample_mod.f90:392-473).  walkers may be NULL (0..n-1).
 * en[9*i + 0..2] = E, Kin, Pot at slice 0; [3..5] the same at slice 2Nb; [6..8] = E, Ec, Ep of ThermEnergy.
 * gr: n x Nbin, Sk: n x Nk x dim, laid out as pigs_structure_batch does.  Same results as the separate entry points
 * (the same kernels), one synchronisation instead of four. */
int pigs_diagonal_estimators(pigs_ctx *ctx, int32_t n, const int32_t *walkers, int32_t Nbin, double rbin, int32_t Nk,
                             double *en, double *gr, double *Sk);
/* The same overlapped with the sampler.  _begin snapshots the resident worldlines (a device-to-device copy ordered on the
 * context's stream: after every step queued so far, before whatever the caller queues next) and starts the estimator
 * kernels on a second stream of the context, on half of the chip; _end waits for them and returns the results in the
 * layout of pigs_diagonal_estimators (gr / Sk only if `structure` was non-zero).  Between the two the caller may queue the
 * next pigs_sampler_step: at 128 walkers per GPU the sampler leaves half of the CUs idle, where a step's estimators
 * (vpi.f90:443-469) then run for free.  Same kernels on a bit-identical copy: same results.  One pending batch per context. */
int pigs_diagonal_estimators_begin(pigs_ctx *ctx, int32_t n, const int32_t *walkers, int32_t Nbin, double rbin, int32_t Nk,
                                   int32_t structure);
int pigs_diagonal_estimators_end(pigs_ctx *ctx, double *en, double *gr, double *Sk);

/* ---- multi-GPU: block-estimator reduction (new) ------------------------ */
/* RCCL communicator over `nranks` contexts.  Single-process form (one host thread per
 * GPU, the Fortran host: pigs_vpi's &gpu n_gpus = G): pigs_comm_init_all.  Multi-process form: rank 0 obtains an id
 * with pigs_comm_unique_id, distributes the 128 bytes out of band, every rank calls
 * pigs_comm_init_rank.  pigs_comm_init_all on contexts that share a device (a one-GPU rehearsal) sets up an
 * in-process reduction instead of RCCL: same semantics (ranks added in rank order), no xGMI. */
int pigs_comm_unique_id(char id[128]);
int pigs_comm_init_rank(pigs_ctx *ctx, int32_t nranks, int32_t rank, const char id[128]);
int pigs_comm_init_all(pigs_ctx **ctxs, int32_t nranks);
/* Sum vec[0..n) (host memory, fp64) over all ranks, in place. */
int pigs_estimators_allreduce(pigs_ctx *ctx, double *vec, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* PIGS_HIP_H */
