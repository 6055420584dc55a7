// pigs_capi.hip -- the C ABI of include/pigs_hip.h over the gfx950 kernels.
//
// A context owns one device, one stream, the resident worldlines of its walkers (SoA
// layout of pigs_device.h), both tables, and grow-only scratch buffers; nothing is
// allocated inside the *_dev / launch paths (graph-capture safe).  There is NO CPU
// fallback anywhere in this library: without a HIP device every call fails loudly.
#include "../../include/pigs_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <unordered_set>
#include <atomic>
#include <vector>
#include <thread>
#include <algorithm>

#include "pigs_comm.h"
#include "pigs_kernels.h"

using namespace pigs;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(PIGS_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                             \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

// pinned, device-mapped host arrays of the staged (low-latency) entry points
struct PinBuf {
    void *h = nullptr, *d = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t want, size_t keep_bytes)
    {
        if (want <= bytes) return hipSuccess;
        void *nh = nullptr, *nd = nullptr;
        hipError_t e = hipHostMalloc(&nh, want, hipHostMallocMapped);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(&nd, nh, 0);
        if (e != hipSuccess) { (void)hipHostFree(nh); return e; }
        if (h && keep_bytes) memcpy(nh, h, keep_bytes < bytes ? keep_bytes : bytes);
        if (h) (void)hipHostFree(h);
        h = nh; d = nd; bytes = want;
        return hipSuccess;
    }
    void release() { if (h) (void)hipHostFree(h); h = d = nullptr; bytes = 0; }
};

// several contexts of one process meeting in host memory (pigs_comm_init_all on duplicate devices: rehearsal only)
struct HostGroup {
    int n = 0, arrived = 0, generation = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::vector<double>> slot;
    std::vector<double> sum;
};

struct pigs_ctx {
    pigs_params hp;
    DevParams   P;
    int         device    = 0;
    int         n_walkers = 0;
    hipStream_t stream    = nullptr;
    double *d_paths = nullptr, *d_VT = nullptr, *d_WF = nullptr;
    std::shared_ptr<HostGroup> hgroup;   // rehearsal form of the estimator reduction (several contexts on one GPU)
    int hrank = 0;
    double *d_VTimg = nullptr;           // [0, VT(0)] VT(0..Nmax+1) [0 0 0 0]: PipeTab image for the sampler (pigs_k1_device.h)
    size_t  path_doubles = 0;        // resident doubles per walker (padded SoA)
    size_t  raw_doubles  = 0;        // dim*Np*(2Nb+1): reference layout per walker
    DevBuf<int32_t> d_walker, d_ip, d_ib, d_slotw, d_slotb;
    DevBuf<double>  d_xnew, d_xold, d_out, d_parts, d_stage, d_slices, d_res;
    pigs_comm  *comm = nullptr;
    int         k1_variant = K1_AUTO;
    PinBuf      st_w, st_ip, st_ib, st_xn, st_xo, st_out;      // staged items
    PinBuf      cs_w, cs_ip, cs_ib, cs_x;                      // staged commits
    int64_t     st_cap = 0, cs_cap = 0;
    // device-resident sampler
    uint32_t   *d_rng = nullptr;
    unsigned long long *d_counters = nullptr;
    double     *d_worm = nullptr, *d_nrho = nullptr, *d_dklog = nullptr;
    int        *d_evlog = nullptr;
    size_t      nrho_doubles = 0;
    SweepParams sweep{};
    int         cm_freq = 1;
    int         sweep_threads = 512;
    bool        sweep_split = false;    // diagonal moves of periodic 'bis' systems in pigs_diag.hip's kernel (see pigs_sampler_step)
    int         n_cu = 256;
    // TranslateChain by several workgroups per walker (pigs_cm.hip): -1 as many as fit (default), 0 off, H >= 1 at most H
    int         cm_split = -1;
    bool        cm_exclusive = false;       // the caller vouches that no other context's kernels run on this device meanwhile
    bool        cm_shared = false;          // several contexts of this process sample on this device at once: see g_cm_gate
    unsigned long long *d_xch = nullptr;    // exchange buffer of the cooperating workgroups
    int        *h_cm_err = nullptr;         // (pinned, device-visible) set by a workgroup whose partner never answered
    unsigned int cm_seq = 1;                // sequence tags of the exchange: advanced by every launch
    bool        counted = false;            // in g_live_ctx
    bool        sampler_ready = false;
    // asynchronous estimators (pigs_diagonal_estimators_begin / _end): a snapshot of the worldlines and a second stream
    hipStream_t stream2 = nullptr;
    hipEvent_t  ev_snap = nullptr;
    double     *d_shadow = nullptr;
    DevBuf<int32_t> a_slotw, a_slotb;
    DevBuf<double>  a_slices, a_res;
    PinBuf      a_host;                     // results land here (pinned: the copy is truly asynchronous)
    std::vector<int32_t> a_sw, a_sb;        // slot lists: must outlive the asynchronous upload
    struct { bool on = false, launched = false; int n = 0, Nbin = 0, Nk = 0; double rbin = 0.0; bool structure = false;
             size_t ng = 0, nk = 0, nres = 0; } a_pend;
    hipEvent_t  ev_gate = nullptr;          // recorded behind the TranslateChain kernel of the next step (see launch_pending_estimators)
};

// live contexts per device of this process: the TranslateChain helpers (pigs_cm.hip) assume that the walkers of ONE
// context have the chip to themselves
static std::atomic<int> g_live_ctx[64];
static std::atomic<int> g_ctx_serial[64];   // contexts ever created per device (stream priority: see pigs_ctx_create)

// Several sampling contexts on ONE device (walker shards of the front end with `same_device`, tuning key "cm_shared"): the
// TranslateChain kernels of all of them are chained through one event per device, so that never two of them run at once --
// their workgroups wait for each other and must all be resident; next to the OTHER contexts' sweep kernels (one CU per
// walker, finite) they are, as long as H x walkers of the launching context + the walkers of the others <= CUs (the caller
// sets "cm_split" accordingly: 2 contexts x 64 walkers on 256 CUs -> H = 3).  The effect is a staggered schedule: one
// shard's TranslateChain on the CUs the other shard's bisection phase leaves idle.
static std::mutex g_cm_mutex[64];
static hipEvent_t g_cm_gate[64];

static int check_ctx(pigs_ctx *c)
{
    if (!c) return fail(PIGS_ERR_ARG, "null context");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(PIGS_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
    return PIGS_OK;
}

// A commit list is a SEQUENCE of assignments Path(:,ip,ib) = x (the caller's program order): when a bead appears more than
// once the last value must win, as it does in the reference's sequential code.  The kernel writes all entries in
// parallel, so earlier duplicates are marked here (walker = -1: the kernel skips them).  Found by the sharded front end
// test: a worm's bead Nb is re-selected (xend(:,1) / xend(:,2)) between half-chain moves and could be queued twice
// before one flush.
static void mark_superseded(int64_t n, int32_t *w, const int32_t *ip, const int32_t *ib, int M, int Np)
{
    if (n < 2) return;
    std::unordered_set<uint64_t> seen;
    seen.reserve((size_t)n * 2);
    for (int64_t i = n - 1; i >= 0; --i) {
        const uint64_t key = ((uint64_t)(uint32_t)w[i] * (uint64_t)M + (uint64_t)ib[i]) * (uint64_t)(Np + 1) + (uint64_t)ip[i];
        if (!seen.insert(key).second) w[i] = -1;
    }
}

extern "C" {

const char *pigs_last_error(void) { return g_err; }
int pigs_abi_version(void) { return PIGS_ABI_VERSION; }

int pigs_device_count(int32_t *n)
{
    if (!n) return fail(PIGS_ERR_ARG, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(PIGS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return PIGS_OK;
}

int pigs_ctx_create(const pigs_params *p, const double *VTable, const double *LogWF,
                    int32_t n_walkers, int32_t device_id, pigs_ctx **out)
{
    if (!p || !out) return fail(PIGS_ERR_ARG, "null pointer");
    *out = nullptr;
    if (p->dim < 1 || p->dim > PIGS_MAXDIM) return fail(PIGS_ERR_ARG, "dim=%d not in 1..3", p->dim);
    if (p->Np < 2 || p->Nb < 1 || p->Nmax < 4 || n_walkers < 1)
        return fail(PIGS_ERR_ARG, "bad sizes Np=%d Nb=%d Nmax=%d n_walkers=%d", p->Np, p->Nb, p->Nmax, n_walkers);
    if (!(p->dr > 0.0) || !(p->dt > 0.0)) return fail(PIGS_ERR_ARG, "dr and dt must be positive");
    // Reference quirk Q3: Force() is an empty stub, so the analytic (table-less) branch is
    // unusable for the force terms; and the kernels consume tables only.
    if (!p->v_table || !VTable) return fail(PIGS_ERR_UNSUPPORTED, "v_table=T with a VTable is mandatory (reference Force() is a stub)");
    // wf_table = F (the reference's default): the trial function is evaluated analytically (McMillan, pigs_device.h
    // log_psi) and LogWF may be null
    if (p->wf_table && !LogWF) return fail(PIGS_ERR_ARG, "wf_table=T needs a LogWF table");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(PIGS_ERR_NO_DEVICE, "no HIP device visible (%s); libpigs_hip has no CPU fallback",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(PIGS_ERR_ARG, "device_id=%d of %d", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));

    pigs_ctx *c = new (std::nothrow) pigs_ctx();
    if (!c) return fail(PIGS_ERR_ARG, "out of host memory");
    c->hp = *p;
    c->device = device_id;
    if (const char *ev = getenv("PIGS_K1_VARIANT")) {           // test / tuning hook: same as pigs_set_tuning("k1_variant")
        const int v = atoi(ev);
        if (k1_variant_valid(v)) c->k1_variant = v;
    }
    c->n_walkers = n_walkers;
    DevParams &P = c->P;
    memset(&P, 0, sizeof P);
    P.dim = p->dim; P.Np = p->Np; P.Nb = p->Nb; P.M = 2 * p->Nb + 1; P.Nmax = p->Nmax;
    P.NpPad = (p->Np + 7) & ~7;
    P.trap = p->trap; P.wf_table = p->wf_table; P.v_table = p->v_table; P.nW = n_walkers;
    P.dr = p->dr; P.rcut2 = p->rcut2; P.dt = p->dt; P.Rm = p->Rm;
    P.rdr = 1.0 / p->dr;                               // correctly rounded reciprocal (div_by)
    P.hrdr = 0.5 * P.rdr;
    for (int k = 0; k < 3; ++k) {
        P.Lbox[k]     = k < p->dim ? p->Lbox[k] : 1.0;
        P.LboxHalf[k] = 0.5 * P.Lbox[k];                 // vpi.f90:118
        P.rLbox[k]    = 1.0 / P.Lbox[k];
        P.a_ho[k]     = k < p->dim ? p->a_ho[k] : 1.0;
    }
    c->path_doubles = slice_doubles(P.dim, P.NpPad) * P.M;
    c->raw_doubles  = (size_t)P.dim * P.Np * P.M;

    int rc = PIGS_OK;
    do {
        {
            // Contexts on one device alternate between the normal and the high stream priority: the runtime maps streams onto
            // a few hardware queues PER PRIORITY (GPU_MAX_HW_QUEUES = 4 by default, shared with every other stream of the
            // process), and two sampling shards whose streams land on one hardware queue run one after the other instead of
            // side by side (round 3: bench.py's two-shard leg at 77 instead of 38.5 ms per MC step).  Different priorities
            // are different queues for certain; with CUs to spare the priority itself decides nothing.
            int least = 0, greatest = 0;
            const int serial = g_ctx_serial[device_id & 63].fetch_add(1);
            if ((serial & 1) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
                if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
            } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        }
        const size_t tb = (size_t)(p->Nmax + 2) * sizeof(double);
        if (hipMalloc((void **)&c->d_VT, tb) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        if (hipMalloc((void **)&c->d_WF, tb) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        if (hipMalloc((void **)&c->d_paths, c->path_doubles * n_walkers * sizeof(double)) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        if (hipMemcpy(c->d_VT, VTable, tb, hipMemcpyHostToDevice) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        if ((LogWF && p->wf_table ? hipMemcpy(c->d_WF, LogWF, tb, hipMemcpyHostToDevice) : hipMemset(c->d_WF, 0, tb)) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        {
            std::vector<double> img((size_t)p->Nmax + 2 + 6, 0.0);
            img[1] = VTable[0];
            memcpy(&img[2], VTable, tb);
            if (hipMalloc((void **)&c->d_VTimg, img.size() * sizeof(double)) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
            if (hipMemcpy(c->d_VTimg, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
        }
        if (hipMemset(c->d_paths, 0, c->path_doubles * n_walkers * sizeof(double)) != hipSuccess) { rc = PIGS_ERR_HIP; break; }
    } while (0);
    if (rc != PIGS_OK) {
        fail(rc, "context allocation failed: %s", hipGetErrorString(hipGetLastError()));
        pigs_ctx_destroy(c);
        return rc;
    }
    {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device_id) == hipSuccess && pr.multiProcessorCount > 0) c->n_cu = pr.multiProcessorCount;
    }
    c->counted = true;
    g_live_ctx[device_id & 63].fetch_add(1);
    *out = c;
    return PIGS_OK;
}

int pigs_ctx_destroy(pigs_ctx *c)
{
    if (!c) return PIGS_OK;
    if (c->counted) { g_live_ctx[c->device & 63].fetch_sub(1); c->counted = false; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { pigs_comm_destroy(c->comm); c->comm = nullptr; }
    c->d_walker.release(); c->d_ip.release(); c->d_ib.release(); c->d_slotw.release(); c->d_slotb.release();
    c->d_xnew.release(); c->d_xold.release(); c->d_out.release(); c->d_parts.release();
    c->d_stage.release(); c->d_slices.release(); c->d_res.release();
    c->st_w.release(); c->st_ip.release(); c->st_ib.release(); c->st_xn.release(); c->st_xo.release(); c->st_out.release();
    c->cs_w.release(); c->cs_ip.release(); c->cs_ib.release(); c->cs_x.release();
    if (c->d_rng) (void)hipFree(c->d_rng);
    if (c->d_worm) (void)hipFree(c->d_worm);
    if (c->d_nrho) (void)hipFree(c->d_nrho);
    if (c->d_dklog) (void)hipFree(c->d_dklog);
    if (c->d_evlog) (void)hipFree(c->d_evlog);
    if (c->d_xch) (void)hipFree(c->d_xch);
    if (c->h_cm_err) (void)hipHostFree(c->h_cm_err);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_paths) (void)hipFree(c->d_paths);
    if (c->d_VT) (void)hipFree(c->d_VT);
    if (c->d_VTimg) (void)hipFree(c->d_VTimg);
    if (c->d_WF) (void)hipFree(c->d_WF);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->ev_snap) (void)hipEventDestroy(c->ev_snap);
    if (c->ev_gate) (void)hipEventDestroy(c->ev_gate);
    if (c->d_shadow) (void)hipFree(c->d_shadow);
    c->a_host.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PIGS_OK;
}

// after a synchronisation: did a cooperating workgroup of the TranslateChain kernel give up waiting (pigs_cm.hip)?
static int check_cm(pigs_ctx *c)
{
    if (c->h_cm_err && *(volatile int *)c->h_cm_err)
        return fail(PIGS_ERR_HIP, "TranslateChain kernel: a cooperating workgroup timed out; the worldlines of this context are invalid");
    return PIGS_OK;
}

// every entry point that waits for the stream and then hands state of the walkers to the caller (worldlines, counters,
// events, generator, estimators) ends here: after a time-out in pigs_cm.hip none of it may be returned as if it were valid
static int sync_checked(pigs_ctx *c)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    return check_cm(c);
}
#define SYNC_CHECKED(c)                          \
    do {                                         \
        const int rc_ = sync_checked(c);         \
        if (rc_) return rc_;                     \
    } while (0)

int pigs_sync(pigs_ctx *c)
{
    int rc = check_ctx(c); if (rc) return rc;
    return sync_checked(c);
}

int pigs_stream(pigs_ctx *c, void **s)
{
    if (!c || !s) return fail(PIGS_ERR_ARG, "null pointer");
    *s = (void *)c->stream;
    return PIGS_OK;
}

int pigs_set_tuning(pigs_ctx *c, const char *key, int32_t value)
{
    if (!c || !key) return fail(PIGS_ERR_ARG, "null pointer");
    if (!strcmp(key, "k1_variant")) {
        if (!k1_variant_valid(value)) return fail(PIGS_ERR_ARG, "k1_variant=%d", value);
        c->k1_variant = value;
        return PIGS_OK;
    }
    if (!strcmp(key, "sweep_split")) {          // 1: stage-machine kernel (pigs_diag.hip) for the diagonal bisection moves; 0 (default): one kernel
        c->sweep_split = value != 0;
        return PIGS_OK;
    }
    if (!strcmp(key, "cm_split")) {             // TranslateChain by H workgroups per walker (pigs_cm.hip): -1 auto, 0 off, H = 1..4
        if (value < -1 || value > 4) return fail(PIGS_ERR_ARG, "cm_split=%d", value);
        c->cm_split = value;
        return PIGS_OK;
    }
    if (!strcmp(key, "cm_fault")) {             // TEST ONLY: force the time-out path of the TranslateChain exchange (pigs_cm.hip)
        c->sweep.cm_fault = (c->sweep.cm_fault & ~1) | (value != 0 ? 1 : 0);
        return PIGS_OK;
    }
    if (!strcmp(key, "cm_shared")) {            // 1: several contexts of this process sample on this device at once (see g_cm_gate);
        c->cm_shared = value != 0;              // set "cm_split" so that H x walkers + the other contexts' walkers <= CUs
        return PIGS_OK;
    }
    if (!strcmp(key, "cm_exclusive")) {         // 1: other contexts of this process on the device are idle while this one samples
        c->cm_exclusive = value != 0;           // (bench.py's extra legs next to its main context): cooperating workgroups allowed
        return PIGS_OK;
    }
    if (!strcmp(key, "sweep_threads")) {
        if (value != 256 && value != 512 && value != 768 && value != 1024) return fail(PIGS_ERR_ARG, "sweep_threads=%d", value);
        c->sweep_threads = value;
        return PIGS_OK;
    }
    return fail(PIGS_ERR_ARG, "unknown tuning key '%s'", key);
}

int pigs_selftest_fastmath(pigs_ctx *c, int32_t blocks, int32_t iters, uint64_t bad[4])
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!bad || blocks < 1 || iters < 1) return fail(PIGS_ERR_ARG, "bad arguments");
    unsigned long long *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), c->stream));
    HIPCHK(launch_selftest_fastmath(c->P, 0x1234567ull, blocks, iters, d, c->stream));
    unsigned long long h[4];
    HIPCHK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    HIPCHK(hipFree(d));
    for (int i = 0; i < 4; ++i) bad[i] = h[i];
    return PIGS_OK;
}

// Device log of the sampler's Gaussians (pigs_log_host.h) against THIS host's libm, bit for bit, on n arguments of the
// sampler's domain (selftest_log_arg): chunks of 2^24 results come back over PCIe and are compared by host threads
// while the device computes the next chunk.
int pigs_selftest_log(pigs_ctx *c, int64_t n, uint64_t seed, uint64_t *mismatches, double *first_bad)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!mismatches || n < 1) return fail(PIGS_ERR_ARG, "bad arguments");
    const size_t chunk = (size_t)1 << 24;
    double *d[2] = {nullptr, nullptr}, *h[2] = {nullptr, nullptr};
    uint64_t bad = 0; double firstx = 0.0; bool have = false;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc((void **)&d[k], chunk * sizeof(double));
        if (e == hipSuccess) e = hipHostMalloc((void **)&h[k], chunk * sizeof(double), hipHostMallocDefault);
    }
    auto compare = [&](const double *res, uint64_t first, size_t m) {
        const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        std::vector<uint64_t> nb(nt, 0); std::vector<double> fx(nt, 0.0);
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (size_t i = t; i < m; i += nt) {
                    const double x = selftest_log_arg(first + i, seed), want = std::log(x), got = res[i];
                    if (memcmp(&want, &got, 8) != 0 && !(want != want && got != got)) { if (!nb[t]) fx[t] = x; ++nb[t]; }
                }
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < nt; ++t) { if (nb[t] && !have) { firstx = fx[t]; have = true; } bad += nb[t]; }
    };
    uint64_t done = 0; int cur = 0; size_t prev_m = 0; uint64_t prev_first = 0;
    while (e == hipSuccess && done < (uint64_t)n) {
        const size_t m = (size_t)std::min<uint64_t>(chunk, (uint64_t)n - done);
        e = launch_selftest_log(done, m, seed, d[cur], c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h[cur], d[cur], m * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (prev_m) compare(h[cur ^ 1], prev_first, prev_m);            // overlaps the chunk in flight
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        prev_m = m; prev_first = done; done += m; cur ^= 1;
    }
    if (e == hipSuccess && prev_m) compare(h[cur ^ 1], prev_first, prev_m);
    for (int k = 0; k < 2; ++k) { if (d[k]) (void)hipFree(d[k]); if (h[k]) (void)hipHostFree(h[k]); }
    if (e != hipSuccess) return fail(PIGS_ERR_HIP, "pigs_selftest_log: %s", hipGetErrorString(e));
    *mismatches = bad;
    if (first_bad) *first_bad = firstx;
    return PIGS_OK;
}

int pigs_selftest_stream_read(pigs_ctx *c, int32_t reps, double *bytes, double *seconds)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!bytes || !seconds || reps < 1) return fail(PIGS_ERR_ARG, "bad arguments");
    const size_t nd = c->path_doubles * (size_t)c->n_walkers;
    double *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t e = hipMalloc((void **)&sink, sizeof(double));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    for (int r = 0; r < 3 && e == hipSuccess; ++r) e = launch_stream_read(c->d_paths, nd, c->n_cu, sink, c->stream);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps && e == hipSuccess; ++r) e = launch_stream_read(c->d_paths, nd, c->n_cu, sink, c->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);                    // nothing leaks on an error return
    if (e1) (void)hipEventDestroy(e1);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(PIGS_ERR_HIP, "pigs_selftest_stream_read: %s", hipGetErrorString(e));
    *bytes = (double)(nd * sizeof(double));
    *seconds = 1e-3 * (double)ms / reps;
    return PIGS_OK;
}

// ---- residency -------------------------------------------------------------------------
static int upload_range(pigs_ctx *c, int w0, int nw, const double *raw)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!raw) return fail(PIGS_ERR_ARG, "null Path");
    if (w0 < 0 || nw < 1 || w0 + nw > c->n_walkers) return fail(PIGS_ERR_ARG, "walker range %d+%d of %d", w0, nw, c->n_walkers);
    // stage in chunks of <= 64 walkers so the scratch stays small
    const int chunk = 64;
    for (int a = 0; a < nw; a += chunk) {
        const int m = nw - a < chunk ? nw - a : chunk;
        HIPCHK(c->d_stage.reserve(c->raw_doubles * m));
        HIPCHK(hipMemcpyAsync(c->d_stage.p, raw + c->raw_doubles * a, c->raw_doubles * m * sizeof(double),
                              hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_pack(c->P, c->d_paths, c->d_stage.p, w0 + a, m, c->stream));
        SYNC_CHECKED(c);
    }
    return PIGS_OK;
}

static int download_range(pigs_ctx *c, int w0, int nw, double *raw)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!raw) return fail(PIGS_ERR_ARG, "null Path");
    if (w0 < 0 || nw < 1 || w0 + nw > c->n_walkers) return fail(PIGS_ERR_ARG, "walker range %d+%d of %d", w0, nw, c->n_walkers);
    const int chunk = 64;
    for (int a = 0; a < nw; a += chunk) {
        const int m = nw - a < chunk ? nw - a : chunk;
        HIPCHK(c->d_stage.reserve(c->raw_doubles * m));
        HIPCHK(launch_unpack(c->P, c->d_paths, c->d_stage.p, w0 + a, m, c->stream));
        HIPCHK(hipMemcpyAsync(raw + c->raw_doubles * a, c->d_stage.p, c->raw_doubles * m * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
        SYNC_CHECKED(c);
    }
    return PIGS_OK;
}

int pigs_path_upload(pigs_ctx *c, int32_t walker, const double *Path) { return upload_range(c, walker, 1, Path); }
int pigs_path_download(pigs_ctx *c, int32_t walker, double *Path) { return download_range(c, walker, 1, Path); }
int pigs_path_upload_all(pigs_ctx *c, const double *Paths) { return c ? upload_range(c, 0, c->n_walkers, Paths) : fail(PIGS_ERR_ARG, "null context"); }
int pigs_path_download_all(pigs_ctx *c, double *Paths) { return c ? download_range(c, 0, c->n_walkers, Paths) : fail(PIGS_ERR_ARG, "null context"); }

// ---- K1 ----------------------------------------------------------------------------------
static int check_items(pigs_ctx *c, int64_t n, const int32_t *walker, const int32_t *ip, const int32_t *ib)
{
    if (n < 0 || n > 0x7fffffff) return fail(PIGS_ERR_ARG, "n_items=%lld out of range", (long long)n);
    if (n && (!walker || !ip || !ib)) return fail(PIGS_ERR_ARG, "null index array");
    for (int64_t i = 0; i < n; ++i) {
        if (walker[i] < 0 || walker[i] >= c->n_walkers || ip[i] < 1 || ip[i] > c->P.Np || ib[i] < 0 || ib[i] >= c->P.M)
            return fail(PIGS_ERR_ARG, "item %lld: walker=%d ip=%d ib=%d out of range", (long long)i, walker[i], ip[i], ib[i]);
    }
    return PIGS_OK;
}

static int delta_action_host(pigs_ctx *c, int64_t n, const int32_t *walker, const int32_t *ip,
                             const int32_t *ib, const double *xnew, const double *xold,
                             double *DeltaS, double *parts)
{
    int rc = check_ctx(c); if (rc) return rc;
    rc = check_items(c, n, walker, ip, ib); if (rc) return rc;
    if (n == 0) return PIGS_OK;
    if (!xnew || !xold || (!DeltaS && !parts)) return fail(PIGS_ERR_ARG, "null pointer");
    const size_t nd = (size_t)n * c->P.dim;
    HIPCHK(c->d_walker.reserve(n)); HIPCHK(c->d_ip.reserve(n)); HIPCHK(c->d_ib.reserve(n));
    HIPCHK(c->d_xnew.reserve(nd)); HIPCHK(c->d_xold.reserve(nd)); HIPCHK(c->d_out.reserve(n));
    if (parts) HIPCHK(c->d_parts.reserve((size_t)n * 3));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_walker.p, walker, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_ip.p, ip, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_ib.p, ib, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_xnew.p, xnew, nd * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_xold.p, xold, nd * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(launch_delta_action(c->P, c->k1_variant, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, (int)n, c->d_walker.p, c->d_ip.p,
                               c->d_ib.p, c->d_xnew.p, c->d_xold.p, c->d_out.p,
                               parts ? c->d_parts.p : nullptr, s));
    if (DeltaS) HIPCHK(hipMemcpyAsync(DeltaS, c->d_out.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (parts) HIPCHK(hipMemcpyAsync(parts, c->d_parts.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_delta_action_batch(pigs_ctx *c, int64_t n, const int32_t *walker, const int32_t *ip,
                            const int32_t *ib, const double *xnew, const double *xold, double *DeltaS)
{
    if (!DeltaS && n) return fail(PIGS_ERR_ARG, "null DeltaS");
    return delta_action_host(c, n, walker, ip, ib, xnew, xold, DeltaS, nullptr);
}

int pigs_delta_action_parts(pigs_ctx *c, int64_t n, const int32_t *walker, const int32_t *ip,
                            const int32_t *ib, const double *xnew, const double *xold, double *parts)
{
    if (!parts && n) return fail(PIGS_ERR_ARG, "null parts");
    return delta_action_host(c, n, walker, ip, ib, xnew, xold, nullptr, parts);
}

int pigs_delta_action_batch_dev(pigs_ctx *c, int64_t n, const int32_t *d_walker, const int32_t *d_ip,
                                const int32_t *d_ib, const double *d_xnew, const double *d_xold,
                                double *d_DeltaS)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || n > 0x7fffffff) return fail(PIGS_ERR_ARG, "n_items=%lld out of range", (long long)n);
    if (n == 0) return PIGS_OK;
    if (!d_walker || !d_ip || !d_ib || !d_xnew || !d_xold || !d_DeltaS) return fail(PIGS_ERR_ARG, "null device pointer");
    // indices are range-checked on the device (out-of-range items produce NaN, never a fault)
    HIPCHK(launch_delta_action(c->P, c->k1_variant, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, (int)n, d_walker, d_ip, d_ib,
                               d_xnew, d_xold, d_DeltaS, nullptr, c->stream));
    return PIGS_OK;
}

// ---- staged (pinned, zero-copy) forms --------------------------------------------------------
int pigs_stage_reserve(pigs_ctx *c, int64_t cap, int64_t keep, int32_t **walker, int32_t **ip, int32_t **ib,
                       double **xnew, double **xold, double **DeltaS)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (cap < 1 || cap > 0x7fffffff || keep < 0 || !walker || !ip || !ib || !xnew || !xold || !DeltaS)
        return fail(PIGS_ERR_ARG, "bad stage_reserve arguments");
    if (cap > c->st_cap) {
        SYNC_CHECKED(c);
        const size_t d = c->P.dim, k = (size_t)(keep < c->st_cap ? keep : c->st_cap);
        HIPCHK(c->st_w.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->st_ip.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->st_ib.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->st_xn.reserve(cap * d * sizeof(double), k * d * sizeof(double)));
        HIPCHK(c->st_xo.reserve(cap * d * sizeof(double), k * d * sizeof(double)));
        HIPCHK(c->st_out.reserve(cap * sizeof(double), k * sizeof(double)));
        c->st_cap = cap;
    }
    *walker = (int32_t *)c->st_w.h; *ip = (int32_t *)c->st_ip.h; *ib = (int32_t *)c->st_ib.h;
    *xnew = (double *)c->st_xn.h; *xold = (double *)c->st_xo.h; *DeltaS = (double *)c->st_out.h;
    return PIGS_OK;
}

int pigs_delta_action_staged(pigs_ctx *c, int64_t n)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || n > c->st_cap) return fail(PIGS_ERR_ARG, "n_items=%lld exceeds the staged capacity %lld", (long long)n, (long long)c->st_cap);
    if (n == 0) return PIGS_OK;
    // indices are range-checked on the device (bad item -> NaN)
    HIPCHK(launch_delta_action(c->P, c->k1_variant, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, (int)n,
                               (const int32_t *)c->st_w.d, (const int32_t *)c->st_ip.d, (const int32_t *)c->st_ib.d,
                               (const double *)c->st_xn.d, (const double *)c->st_xo.d, (double *)c->st_out.d,
                               nullptr, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_commit_reserve(pigs_ctx *c, int64_t cap, int64_t keep, int32_t **walker, int32_t **ip, int32_t **ib, double **x)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (cap < 1 || cap > 0x7fffffff || keep < 0 || !walker || !ip || !ib || !x) return fail(PIGS_ERR_ARG, "bad commit_reserve arguments");
    if (cap > c->cs_cap) {
        SYNC_CHECKED(c);
        const size_t d = c->P.dim, k = (size_t)(keep < c->cs_cap ? keep : c->cs_cap);
        HIPCHK(c->cs_w.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->cs_ip.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->cs_ib.reserve(cap * sizeof(int32_t), k * sizeof(int32_t)));
        HIPCHK(c->cs_x.reserve(cap * d * sizeof(double), k * d * sizeof(double)));
        c->cs_cap = cap;
    }
    *walker = (int32_t *)c->cs_w.h; *ip = (int32_t *)c->cs_ip.h; *ib = (int32_t *)c->cs_ib.h; *x = (double *)c->cs_x.h;
    return PIGS_OK;
}

int pigs_commit_staged(pigs_ctx *c, int64_t n)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || n > c->cs_cap) return fail(PIGS_ERR_ARG, "n=%lld exceeds the staged capacity %lld", (long long)n, (long long)c->cs_cap);
    if (n == 0) return PIGS_OK;
    const int32_t *w = (const int32_t *)c->cs_w.h, *ip = (const int32_t *)c->cs_ip.h, *ib = (const int32_t *)c->cs_ib.h;
    for (int64_t i = 0; i < n; ++i)
        if (w[i] < 0 || w[i] >= c->n_walkers || ip[i] < 1 || ip[i] > c->P.Np || ib[i] < 0 || ib[i] >= c->P.M)
            return fail(PIGS_ERR_ARG, "commit %lld: walker=%d ip=%d ib=%d out of range", (long long)i, w[i], ip[i], ib[i]);
    mark_superseded(n, (int32_t *)c->cs_w.h, ip, ib, c->P.M, c->P.Np);
    HIPCHK(launch_commit_beads(c->P, c->d_paths, n, (const int32_t *)c->cs_w.d, (const int32_t *)c->cs_ip.d,
                               (const int32_t *)c->cs_ib.d, (const double *)c->cs_x.d, c->stream));
    return PIGS_OK;
}

// ---- K5 ----------------------------------------------------------------------------------
int pigs_commit_beads(pigs_ctx *c, int64_t n, const int32_t *walker, const int32_t *ip,
                      const int32_t *ib, const double *x)
{
    int rc = check_ctx(c); if (rc) return rc;
    rc = check_items(c, n, walker, ip, ib); if (rc) return rc;
    if (n == 0) return PIGS_OK;
    if (!x) return fail(PIGS_ERR_ARG, "null x");
    const size_t nd = (size_t)n * c->P.dim;
    HIPCHK(c->d_walker.reserve(n)); HIPCHK(c->d_ip.reserve(n)); HIPCHK(c->d_ib.reserve(n));
    HIPCHK(c->d_xnew.reserve(nd));
    hipStream_t s = c->stream;
    std::vector<int32_t> wl(walker, walker + n);
    mark_superseded(n, wl.data(), ip, ib, c->P.M, c->P.Np);
    HIPCHK(hipMemcpyAsync(c->d_walker.p, wl.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_ip.p, ip, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_ib.p, ib, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_xnew.p, x, nd * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(launch_commit_beads(c->P, c->d_paths, n, c->d_walker.p, c->d_ip.p, c->d_ib.p, c->d_xnew.p, s));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_swap_tails(pigs_ctx *c, int32_t walker, int32_t iw, int32_t ik)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (walker < 0 || walker >= c->n_walkers || iw < 1 || iw > c->P.Np || ik < 1 || ik > c->P.Np)
        return fail(PIGS_ERR_ARG, "swap_tails(walker=%d, iw=%d, ik=%d) out of range", walker, iw, ik);
    if (iw == ik) return PIGS_OK;
    HIPCHK(launch_swap_tails(c->P, c->d_paths, walker, iw, ik, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

// ---- K6: device-resident sampler ----------------------------------------------------------------
namespace {
// reference random_mod.f90:5-31 (seeding) and the conversion of a block-form state (mti, mt) to
// the sliding form the kernel keeps: the first `mti` steps of the block twist are applied, so slot
// k < mti already holds element k+624 and the next output is slot mti.
void mt_seed_words(uint32_t seed, uint32_t *w)
{
    w[0] = seed;
    for (int i = 1; i < 624; ++i) w[i] = 69069u * w[i - 1];
}
void mt_twist_prefix(int n, uint32_t *w)
{
    for (int i = 0; i < n; ++i) {
        const uint32_t y = (w[i] & 0x80000000u) | (w[(i + 1) % 624] & 0x7fffffffu);
        w[i] = w[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}
// st: [624 sliding][624 block form][pos] from the reference's block form (mti, words)
void mt_block_to_device(int mti, const uint32_t *words, uint32_t *st)
{
    uint32_t *sl = st, *blk = st + 624;
    memcpy(blk, words, 624 * sizeof(uint32_t));
    if (mti >= 624) { mt_twist_prefix(624, blk); mti = 0; }        // block exhausted: next block, nothing consumed
    memcpy(sl, blk, 624 * sizeof(uint32_t));
    mt_twist_prefix(mti, sl);                                        // consumed slots already slid on
    st[1248] = (uint32_t)mti;
}
} // namespace

int pigs_sampler_init(pigs_ctx *c, const pigs_sweep_params *sp)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!sp) return fail(PIGS_ERR_ARG, "null sweep params");
    const bool sta = sp->sampling == 1;
    if (sp->sampling != 0 && sp->sampling != 1) return fail(PIGS_ERR_ARG, "sampling must be 0 ('bis') or 1 ('sta')");
    const bool worm0 = sp->CWorm > 0.0;
    // Nlev: any 2^Nlev <= 2 Nb up to 7 levels for periodic systems (pigs_diag.hip beyond 4), 4 for trapped ones;
    // Lstag <= Nb is a requirement of the worm's half-chain moves only (vpi_mod.f90:1376-1817): with CWorm = 0 any
    // Lstag <= 2 Nb works, as in the reference
    // (head / tail moves bisect 2^nl beads with nl = 2 .. max(Nlev, 2): vpi_mod.f90:1023)
    if ((!sta && (sp->Nlev < 1 || sp->Nlev > (c->P.trap ? 4 : 7) || (1 << (sp->Nlev > 2 ? sp->Nlev : 2)) > 2 * c->P.Nb)) || sp->Nstag < 0 || sp->CMFreq < 1 ||
        sp->Lstag < 2 || sp->Lstag > (worm0 ? c->P.Nb : 2 * c->P.Nb))
        return fail(PIGS_ERR_ARG, "sweep params out of range (Nlev=%d Nstag=%d CMFreq=%d Lstag=%d)", sp->Nlev, sp->Nstag, sp->CMFreq, sp->Lstag);
    const bool worm = sp->CWorm > 0.0;
    if (worm && (sp->Nobdm < 0 || sp->Nbin < 1 || sp->Npw < 0 || !(sp->rbin > 0.0) || !(sp->density > 0.0)))
        return fail(PIGS_ERR_ARG, "worm parameters out of range");
    SweepParams &k = c->sweep;
    memset(&k, 0, sizeof k);
    k.Nlev = sta ? 1 : sp->Nlev; k.Nstag = sp->Nstag; k.Lstag = sp->Lstag; k.staging = sta;
    k.delta_cm = sp->delta_cm; k.open_attempt = 1; k.do_cm = 1;
    k.worm = worm; k.swapping = sp->swapping != 0; k.Nobdm = worm ? sp->Nobdm : 0;
    k.Nbin = worm ? sp->Nbin : 1; k.Npw = worm ? sp->Npw : 0; k.rbin = worm ? sp->rbin : 1.0;
    k.log_cworm_density = worm ? std::log(sp->CWorm * sp->density) : 0.0;      // host libm, as the reference
    c->cm_freq = sp->CMFreq;
    // one workgroup per walker: the 8-wave form (periodic: table image in LDS) while every walker gets a CU of its own,
    // the 4-wave form (three workgroups per CU) beyond that (measured: scripts/sampler_bench.py)
    {
        int dev = 0, ncu = 256;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            ncu = pr.multiProcessorCount;
        c->n_cu = ncu;
        c->sweep_threads = sweep_form(c->P, c->sweep, c->n_walkers > ncu ? 256 : 1024);
    }
    if (sweep_lds_bytes(c->P, c->sweep, c->sweep_threads) > 160 * 1024) c->sweep_threads = sweep_form(c->P, c->sweep, 256);
    if (sweep_lds_bytes(c->P, c->sweep, c->sweep_threads) > 160 * 1024) return fail(PIGS_ERR_UNSUPPORTED, "worldline too long for the sampler's LDS staging");
    const size_t W = c->n_walkers;
    if (!c->d_rng) HIPCHK(hipMalloc((void **)&c->d_rng, W * kRngWords * sizeof(uint32_t)));
    if (!c->d_counters) HIPCHK(hipMalloc((void **)&c->d_counters, W * kCounters * sizeof(unsigned long long)));
    if (!c->d_worm) HIPCHK(hipMalloc((void **)&c->d_worm, W * kWormDoubles * sizeof(double)));
    // a step logs at most one open / close event and one swap per OBDM iteration
    k.ev_ints = kEvInts > 4 + 2 * (1 + k.Nobdm) ? kEvInts : 4 + 2 * (1 + k.Nobdm);
    if (c->d_evlog) { HIPCHK(hipFree(c->d_evlog)); c->d_evlog = nullptr; }
    HIPCHK(hipMalloc((void **)&c->d_evlog, W * k.ev_ints * sizeof(int)));
    if (c->d_nrho) { HIPCHK(hipFree(c->d_nrho)); c->d_nrho = nullptr; }
    c->nrho_doubles = W * (size_t)k.Nbin * (k.Npw + 1);
    HIPCHK(hipMalloc((void **)&c->d_nrho, c->nrho_doubles * sizeof(double)));
    if (c->d_dklog) { HIPCHK(hipFree(c->d_dklog)); c->d_dklog = nullptr; }
    // 0.5d0*real(dim)*log(2.d0*pi*real(Ls)*dt) of vpi_mod.f90:1873, tabulated with the host libm
    std::vector<double> dk(sp->Lstag + 2, 0.0);
    const double pi = std::acos(-1.0);
    for (int Ls = 1; Ls <= sp->Lstag + 1; ++Ls)
        dk[Ls] = 0.5 * (double)(float)c->P.dim * std::log(2.0 * pi * (double)(float)Ls * c->P.dt);
    HIPCHK(hipMalloc((void **)&c->d_dklog, dk.size() * sizeof(double)));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_dklog, dk.data(), dk.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->d_counters, 0, W * kCounters * sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(c->d_worm, 0, W * kWormDoubles * sizeof(double), s));
    HIPCHK(hipMemsetAsync(c->d_evlog, 0, W * k.ev_ints * sizeof(int), s));
    HIPCHK(hipMemsetAsync(c->d_nrho, 0, c->nrho_doubles * sizeof(double), s));
    std::vector<uint32_t> st(W * kRngWords);
    for (size_t w = 0; w < W; ++w) {
        uint32_t seedw[624];
        mt_seed_words(4357u, seedw);
        mt_block_to_device(624, seedw, &st[w * kRngWords]);
    }
    HIPCHK(hipMemcpyAsync(c->d_rng, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    SYNC_CHECKED(c);
    c->sampler_ready = true;
    return PIGS_OK;
}

int pigs_sampler_set_rng(pigs_ctx *c, int32_t walker, int32_t mti, const int32_t mt[624])
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready) return fail(PIGS_ERR_ARG, "pigs_sampler_init first");
    if (walker < 0 || walker >= c->n_walkers || !mt || mti < 0 || mti > 624) return fail(PIGS_ERR_ARG, "bad rng state");
    uint32_t st[kRngWords];
    mt_block_to_device(mti, (const uint32_t *)mt, st);
    HIPCHK(hipMemcpyAsync(c->d_rng + (size_t)walker * kRngWords, st, sizeof st, hipMemcpyHostToDevice, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_sampler_get_rng(pigs_ctx *c, int32_t walker, int32_t *mti, int32_t mt[624])
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready) return fail(PIGS_ERR_ARG, "pigs_sampler_init first");
    if (walker < 0 || walker >= c->n_walkers || !mt || !mti) return fail(PIGS_ERR_ARG, "bad rng request");
    uint32_t st[kRngWords];
    HIPCHK(hipMemcpyAsync(st, c->d_rng + (size_t)walker * kRngWords, sizeof st, hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    memcpy(mt, st + 624, 624 * sizeof(uint32_t));                    // block form: what mtsavef would hold
    *mti = (int32_t)st[1248];
    return PIGS_OK;
}

int pigs_sampler_seed(pigs_ctx *c, int32_t walker, int32_t seed)
{
    uint32_t w[624];
    mt_seed_words((uint32_t)seed, w);
    return pigs_sampler_set_rng(c, walker, 624, (const int32_t *)w);
}

static int launch_pending_estimators(pigs_ctx *c, hipEvent_t gate);

int pigs_sampler_step(pigs_ctx *c, int32_t istep)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready) return fail(PIGS_ERR_ARG, "pigs_sampler_init first");
    rc = check_cm(c); if (rc) return rc;                      // an earlier step's TranslateChain kernel gave up: stop here
    SweepParams sp = c->sweep;
    sp.do_cm = (istep % c->cm_freq) == 0;
    // Two forms of the diagonal bisection moves of a periodic system: inside the one-launch kernel (pigs_sampler.hip;
    // Nlev <= 4), or the stage machine of pigs_diag.hip between two launches of that kernel for the open / close attempt
    // and the worm moves (any Nlev with 2^Nlev <= 2 Nb, Nlev <= 7).  Measured at N=256, 161 beads, 128 walkers: 48.1 ms
    // vs 56.5 ms per MC step (profiles/r02_k6_stage_machine.txt), so the stage machine runs only where the other form
    // cannot, or on request (pigs_set_tuning "sweep_split" = 1).
    //
    // TranslateChain -- the one arithmetic-bound stage -- goes to its own kernel (pigs_cm.hip): H cooperating workgroups
    // per walker while the chip has H >= 2 CUs per walker, one otherwise (bit-identical trajectory whatever H): open /
    // close attempt, that kernel, the rest.
    int H = 0;
    if (c->cm_split != 0 && sp.do_cm) {
        H = cm_helpers(c->P, sp, c->n_cu);                    // what the chip holds (0: the kernel does not apply)
        if (c->cm_split > 0 && H > c->cm_split) H = c->cm_split;
        // cooperating workgroups wait for each other: only while this context has the chip to itself.  One workgroup per
        // walker (H = 1) exchanges nothing and is still the faster TranslateChain (sixteen waves on the LDS table image:
        // 46.8 -> 44.8 ms per MC step at 256 walkers, 95 -> 74 ms at 512, where the sweep kernel runs its 4-wave form)
        if (H > 1 && g_live_ctx[c->device & 63].load() != 1 && !c->cm_exclusive && !c->cm_shared) H = 1;
        // a lowered H means longer bead ranges per workgroup: 321 beads fit two workgroups per walker but not one (rows, LDS)
        // -- then TranslateChain stays inside the sweep kernel (round 3: the BASELINE-config-5 leg of bench.py next to a
        // second live context, and the sharded front end on one GPU, failed here with "invalid argument")
        if (H >= 1 && !cm_fits(c->P, H)) H = 0;
    }
    bool cm_done = false;
    if (H >= 1) {
        if (!c->d_xch) {
            const size_t nb = cm_exchange_words(c->P) * sizeof(unsigned long long);
            HIPCHK(hipMalloc((void **)&c->d_xch, nb));
            HIPCHK(hipMemsetAsync(c->d_xch, 0, nb, c->stream));
            HIPCHK(hipHostMalloc((void **)&c->h_cm_err, sizeof(int), hipHostMallocMapped));
            *c->h_cm_err = 0;
        }
        sp.parts = 1;
        HIPCHK(launch_sweep(c->P, sp, c->sweep_threads, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                            c->d_worm, c->d_evlog, c->d_nrho, c->d_dklog, c->stream));
        int *d_err = nullptr;
        HIPCHK(hipHostGetDevicePointer((void **)&d_err, c->h_cm_err, 0));
        hipError_t e = hipSuccess;
        if (c->cm_shared && H > 1) {
            std::lock_guard<std::mutex> lk(g_cm_mutex[c->device & 63]);
            hipEvent_t &gate = g_cm_gate[c->device & 63];
            if (!gate) e = hipEventCreateWithFlags(&gate, hipEventDisableTiming);
            else       e = hipStreamWaitEvent(c->stream, gate, 0);           // the device's previous TranslateChain kernel, whoever launched it
            if (e == hipSuccess) e = launch_cm(c->P, sp, H, c->cm_seq, c->d_paths, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                                               c->d_worm, c->d_xch, d_err, c->stream);
            if (e == hipSuccess) e = hipEventRecord(gate, c->stream);
        } else {
            e = launch_cm(c->P, sp, H, c->cm_seq, c->d_paths, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                          c->d_worm, c->d_xch, d_err, c->stream);
        }
        HIPCHK(e);
        c->cm_seq += (unsigned int)c->P.Np + 1;
        sp.do_cm = 0;
        cm_done = true;                                       // (the open / close attempt ran as well)
        // estimators of the previous step waiting for their turn (pigs_diagonal_estimators_begin): the TranslateChain
        // kernel fills the chip, the sweep kernel that follows leaves the CUs beyond one per walker idle -- they start there
        if (c->a_pend.on && !c->a_pend.launched) {
            HIPCHK(hipEventRecord(c->ev_gate, c->stream));
            rc = launch_pending_estimators(c, c->ev_gate); if (rc) return rc;
        }
    }
    if (c->a_pend.on && !c->a_pend.launched) {                // a step without the TranslateChain kernel: beside the whole step
        rc = launch_pending_estimators(c, c->ev_snap); if (rc) return rc;
    }
    const bool need_split = !c->P.trap && !sp.staging && sp.Nlev > 4;
    const bool split = (c->sweep_split || need_split) && diag_supported(c->P, sp);
    if (need_split && !split) return fail(PIGS_ERR_UNSUPPORTED, "Nlev=%d needs the stage-machine kernel, which does not fit this worldline", sp.Nlev);
    if (split) {
        if (!cm_done) {
            sp.parts = 1;
            HIPCHK(launch_sweep(c->P, sp, c->sweep_threads, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                                c->d_worm, c->d_evlog, c->d_nrho, c->d_dklog, c->stream));
        }
        HIPCHK(launch_diag(c->P, sp, 512, c->d_paths, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters, c->d_worm, c->stream));
        if (sp.worm) {
            sp.parts = 4;
            HIPCHK(launch_sweep(c->P, sp, c->sweep_threads, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                                c->d_worm, c->d_evlog, c->d_nrho, c->d_dklog, c->stream));
        }
    } else {
        sp.parts = cm_done ? 6 : 7;
        HIPCHK(launch_sweep(c->P, sp, c->sweep_threads, c->d_paths, c->d_VT, c->d_VTimg, c->d_WF, c->d_rng, c->d_counters,
                            c->d_worm, c->d_evlog, c->d_nrho, c->d_dklog, c->stream));
    }
    return PIGS_OK;
}

int pigs_sampler_counters16(pigs_ctx *c, int64_t *cnt)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !cnt) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null output");
    HIPCHK(hipMemcpyAsync(cnt, c->d_counters, (size_t)c->n_walkers * kCounters * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_sampler_counters(pigs_ctx *c, int64_t *acc)
{
    if (!c || !acc) return fail(PIGS_ERR_ARG, "null pointer");
    std::vector<int64_t> all((size_t)c->n_walkers * kCounters);
    int rc = pigs_sampler_counters16(c, all.data()); if (rc) return rc;
    for (int w = 0; w < c->n_walkers; ++w)
        for (int q = 0; q < 4; ++q) acc[4 * w + q] = all[(size_t)w * kCounters + q];
    return PIGS_OK;
}

int pigs_sampler_get_worm(pigs_ctx *c, int32_t *isopen, int32_t *iworm, double *xend)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !isopen || !iworm || !xend) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null output");
    std::vector<double> h((size_t)c->n_walkers * kWormDoubles);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_worm, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    const int d = c->P.dim;
    for (int w = 0; w < c->n_walkers; ++w) {
        isopen[w] = (int32_t)h[(size_t)w * kWormDoubles];
        iworm[w]  = (int32_t)h[(size_t)w * kWormDoubles + 1];
        for (int t = 0; t < 2 * d; ++t) xend[(size_t)w * 2 * d + t] = h[(size_t)w * kWormDoubles + 2 + t];
    }
    return PIGS_OK;
}

int pigs_sampler_set_worm(pigs_ctx *c, const int32_t *isopen, const int32_t *iworm, const double *xend)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !isopen || !iworm || !xend) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null input");
    std::vector<double> h((size_t)c->n_walkers * kWormDoubles, 0.0);
    const int d = c->P.dim;
    for (int w = 0; w < c->n_walkers; ++w) {
        if (isopen[w] && (iworm[w] < 1 || iworm[w] > c->P.Np)) return fail(PIGS_ERR_ARG, "walker %d: iworm=%d", w, iworm[w]);
        h[(size_t)w * kWormDoubles]     = isopen[w] ? 1.0 : 0.0;
        h[(size_t)w * kWormDoubles + 1] = (double)iworm[w];
        for (int t = 0; t < 2 * d; ++t) h[(size_t)w * kWormDoubles + 2 + t] = xend[(size_t)w * 2 * d + t];
    }
    HIPCHK(hipMemcpyAsync(c->d_worm, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_sampler_event_ints(pigs_ctx *c, int32_t *n)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !n) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null output");
    *n = c->sweep.ev_ints;
    return PIGS_OK;
}

int pigs_sampler_events(pigs_ctx *c, int32_t *events)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !events) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null output");
    HIPCHK(hipMemcpyAsync(events, c->d_evlog, (size_t)c->n_walkers * c->sweep.ev_ints * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_sampler_nrho(pigs_ctx *c, double *nrho, const int32_t *reset)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->sampler_ready || !nrho) return fail(PIGS_ERR_ARG, "pigs_sampler_init first / null output");
    HIPCHK(hipMemcpyAsync(nrho, c->d_nrho, c->nrho_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (reset) {
        // zero the histograms of the flagged walkers, one memset per run of consecutive walkers
        const size_t per = c->nrho_doubles / (size_t)c->n_walkers;
        for (int w = 0; w < c->n_walkers;) {
            if (!reset[w]) { ++w; continue; }
            int e = w;
            while (e < c->n_walkers && reset[e]) ++e;
            HIPCHK(hipMemsetAsync(c->d_nrho + (size_t)w * per, 0, (size_t)(e - w) * per * sizeof(double), c->stream));
            w = e;
        }
    }
    SYNC_CHECKED(c);
    return PIGS_OK;
}

int pigs_slice_download(pigs_ctx *c, int32_t ib, double *R)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!R || ib < 0 || ib >= c->P.M) return fail(PIGS_ERR_ARG, "bad slice request");
    const size_t n = (size_t)c->P.dim * c->P.Np * c->n_walkers;
    HIPCHK(c->d_stage.reserve(n));
    HIPCHK(launch_slice_gather(c->P, c->d_paths, ib, c->d_stage.p, c->stream));
    HIPCHK(hipMemcpyAsync(R, c->d_stage.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

// ---- K2/K3 -------------------------------------------------------------------------------
int pigs_potential_energy_slice(pigs_ctx *c, int32_t walker, int32_t ib, int32_t want_F2,
                                double *Pot, double *F2)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!Pot) return fail(PIGS_ERR_ARG, "null Pot");
    if (walker < 0 || walker >= c->n_walkers || ib < 0 || ib >= c->P.M) return fail(PIGS_ERR_ARG, "walker=%d ib=%d out of range", walker, ib);
    HIPCHK(c->d_slotw.reserve(1)); HIPCHK(c->d_slotb.reserve(1)); HIPCHK(c->d_slices.reserve(3));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slotw.p, &walker, sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_slotb.p, &ib, sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(launch_slice_energy(c->P, c->d_paths, c->d_VT, c->d_VTimg, 1, c->d_slotw.p, c->d_slotb.p, want_F2 ? 2 : 0, 0, c->d_slices.p, s));
    double h[3];
    HIPCHK(hipMemcpyAsync(h, c->d_slices.p, sizeof h, hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    *Pot = h[0];
    if (F2) *F2 = want_F2 ? h[1] : 0.0;
    return PIGS_OK;
}

int pigs_therm_energy_batch(pigs_ctx *c, int32_t n, const int32_t *walkers, double *E, double *Ec, double *Ep)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || (n > c->n_walkers && !walkers)) return fail(PIGS_ERR_ARG, "n=%d walkers", n);
    if (n == 0) return PIGS_OK;
    if (!E || !Ec || !Ep) return fail(PIGS_ERR_ARG, "null output");
    const int ns = 2 * c->P.Nb;                       // slices 0..2Nb-1 (Q8)
    std::vector<int32_t> sw((size_t)n * ns), sb((size_t)n * ns);
    for (int i = 0; i < n; ++i) {
        const int w = walkers ? walkers[i] : i;
        if (w < 0 || w >= c->n_walkers) return fail(PIGS_ERR_ARG, "walker %d out of range", w);
        for (int b = 0; b < ns; ++b) { sw[(size_t)i * ns + b] = w; sb[(size_t)i * ns + b] = b; }
    }
    const size_t nslot = (size_t)n * ns;
    HIPCHK(c->d_slotw.reserve(nslot)); HIPCHK(c->d_slotb.reserve(nslot));
    HIPCHK(c->d_slices.reserve(nslot * 3)); HIPCHK(c->d_res.reserve((size_t)n * 3));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slotw.p, sw.data(), nslot * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_slotb.p, sb.data(), nslot * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(launch_slice_energy(c->P, c->d_paths, c->d_VT, c->d_VTimg, (int)nslot, c->d_slotw.p, c->d_slotb.p, 1, 1, c->d_slices.p, s));
    HIPCHK(launch_therm_combine(c->P, n, c->d_slices.p, c->d_res.p, c->d_res.p + n, c->d_res.p + 2 * (size_t)n, s));
    HIPCHK(hipMemcpyAsync(E, c->d_res.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(Ec, c->d_res.p + n, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(Ep, c->d_res.p + 2 * (size_t)n, n * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);   // sw/sb must outlive the async copies
    return PIGS_OK;
}

// ---- K4 ----------------------------------------------------------------------------------
int pigs_local_energy_batch(pigs_ctx *c, int32_t n, const int32_t *walkers, int32_t ib,
                            double *E, double *Kin, double *Pot)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0) return fail(PIGS_ERR_ARG, "n=%d", n);
    if (n == 0) return PIGS_OK;
    if (!E || !Kin || !Pot) return fail(PIGS_ERR_ARG, "null output");
    if (ib < 0 || ib >= c->P.M) return fail(PIGS_ERR_ARG, "ib=%d out of range", ib);
    std::vector<int32_t> sw(n);
    for (int i = 0; i < n; ++i) {
        sw[i] = walkers ? walkers[i] : i;
        if (sw[i] < 0 || sw[i] >= c->n_walkers) return fail(PIGS_ERR_ARG, "walker %d out of range", sw[i]);
    }
    HIPCHK(c->d_slotw.reserve(n)); HIPCHK(c->d_res.reserve((size_t)n * 3));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slotw.p, sw.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(launch_local_energy(c->P, c->d_paths, c->d_VT, c->d_WF, n, c->d_slotw.p, ib, c->d_res.p, s));
    std::vector<double> h((size_t)n * 3);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_res.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    for (int i = 0; i < n; ++i) { E[i] = h[3 * i]; Kin[i] = h[3 * i + 1]; Pot[i] = h[3 * i + 2]; }
    return PIGS_OK;
}

// ---- K7 ----------------------------------------------------------------------------------
int pigs_structure_batch(pigs_ctx *c, int32_t n, const int32_t *walkers, int32_t ib, int32_t Nbin,
                         double rbin, int32_t Nk, double *gr, double *Sk)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || Nbin < 1 || Nk < 0 || !(rbin > 0.0) || ib < 0 || ib >= c->P.M) return fail(PIGS_ERR_ARG, "bad structure request");
    if (c->P.trap) return fail(PIGS_ERR_UNSUPPORTED, "structural estimators are defined for PBC runs only (vpi.f90:466)");
    if (n == 0) return PIGS_OK;
    if (!gr || !Sk) return fail(PIGS_ERR_ARG, "null output");
    std::vector<int32_t> sw(n);
    for (int i = 0; i < n; ++i) {
        sw[i] = walkers ? walkers[i] : i;
        if (sw[i] < 0 || sw[i] >= c->n_walkers) return fail(PIGS_ERR_ARG, "walker %d out of range", sw[i]);
    }
    const size_t ng = (size_t)n * Nbin, ns = (size_t)n * Nk * c->P.dim;
    HIPCHK(c->d_slotw.reserve(n)); HIPCHK(c->d_res.reserve(ng + ns));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slotw.p, sw.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(launch_structure(c->P, c->d_paths, n, c->d_slotw.p, ib, Nbin, rbin, Nk, c->d_res.p, c->d_res.p + ng, s));
    HIPCHK(hipMemcpyAsync(gr, c->d_res.p, ng * sizeof(double), hipMemcpyDeviceToHost, s));
    if (ns) HIPCHK(hipMemcpyAsync(Sk, c->d_res.p + ng, ns * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

// ---- all diagonal-sector estimators of one MC step in one call ------------------------------
// What vpi.f90:443-469 evaluates after a diagonal step -- LocalEnergy at slices 0 and 2Nb (K4), ThermEnergy (K2/K3),
// g(r) and S(k) at slice Nb (K7) -- for n walkers: one upload of the slot lists, five launches, ONE result copy, one
// synchronisation (the separate entry points cost four round trips: ~1.6 ms per step of 128 walkers in round 2's bench).
// en[9*i ..] = E,Kin,Pot (slice 0), E,Kin,Pot (slice 2Nb), E,Ec,Ep (ThermEnergy) of walker i; gr (n x Nbin) and
// Sk (n x Nk x dim) may both be NULL (trapped systems: vpi.f90:466 computes them for PBC runs only).
int pigs_diagonal_estimators(pigs_ctx *c, int32_t n, const int32_t *walkers, int32_t Nbin, double rbin, int32_t Nk,
                             double *en, double *gr, double *Sk)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || !en) return fail(PIGS_ERR_ARG, "n=%d / null output", n);
    if (n == 0) return PIGS_OK;
    const bool structure = gr || Sk;
    if (structure && (!gr || !Sk || Nbin < 1 || Nk < 0 || !(rbin > 0.0))) return fail(PIGS_ERR_ARG, "bad structure request");
    if (structure && c->P.trap) return fail(PIGS_ERR_UNSUPPORTED, "structural estimators are defined for PBC runs only (vpi.f90:466)");
    const int ns = 2 * c->P.Nb;                       // ThermEnergy: slices 0..2Nb-1 (Q8)
    const size_t nslot = (size_t)n * ns;
    // slot lists: [0, nslot) the ThermEnergy slices, [nslot, nslot + n) the walkers themselves (K4, K7)
    std::vector<int32_t> sw(nslot + n), sb(nslot);
    for (int i = 0; i < n; ++i) {
        const int w = walkers ? walkers[i] : i;
        if (w < 0 || w >= c->n_walkers) return fail(PIGS_ERR_ARG, "walker %d out of range", w);
        for (int b = 0; b < ns; ++b) { sw[(size_t)i * ns + b] = w; sb[(size_t)i * ns + b] = b; }
        sw[nslot + i] = w;
    }
    const size_t ng = structure ? (size_t)n * Nbin : 0, nk = structure ? (size_t)n * Nk * c->P.dim : 0;
    // result block: [LocalEnergy slice 0: 3n][LocalEnergy slice 2Nb: 3n][E n][Ec n][Ep n][gr][Sk]
    const size_t nres = (size_t)9 * n + ng + nk;
    HIPCHK(c->d_slotw.reserve(nslot + n)); HIPCHK(c->d_slotb.reserve(nslot));
    HIPCHK(c->d_slices.reserve(nslot * 3)); HIPCHK(c->d_res.reserve(nres));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slotw.p, sw.data(), sw.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_slotb.p, sb.data(), sb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    double *r = c->d_res.p;
    const int32_t *dw = c->d_slotw.p + nslot;
    HIPCHK(launch_local_energy(c->P, c->d_paths, c->d_VT, c->d_WF, n, dw, 0, r, s));
    HIPCHK(launch_local_energy(c->P, c->d_paths, c->d_VT, c->d_WF, n, dw, 2 * c->P.Nb, r + 3 * (size_t)n, s));
    HIPCHK(launch_slice_energy(c->P, c->d_paths, c->d_VT, c->d_VTimg, (int)nslot, c->d_slotw.p, c->d_slotb.p, 1, 1, c->d_slices.p, s));
    HIPCHK(launch_therm_combine(c->P, n, c->d_slices.p, r + 6 * (size_t)n, r + 7 * (size_t)n, r + 8 * (size_t)n, s));
    if (structure)
        HIPCHK(launch_structure(c->P, c->d_paths, n, dw, c->P.Nb, Nbin, rbin, Nk, r + 9 * (size_t)n, r + 9 * (size_t)n + ng, s));
    std::vector<double> h(nres);
    HIPCHK(hipMemcpyAsync(h.data(), r, nres * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    for (int i = 0; i < n; ++i) {
        for (int q = 0; q < 3; ++q) { en[9 * i + q] = h[3 * i + q]; en[9 * i + 3 + q] = h[3 * (size_t)n + 3 * i + q]; }
        en[9 * i + 6] = h[6 * (size_t)n + i]; en[9 * i + 7] = h[7 * (size_t)n + i]; en[9 * i + 8] = h[8 * (size_t)n + i];
    }
    if (structure) {
        memcpy(gr, h.data() + 9 * (size_t)n, ng * sizeof(double));
        if (nk) memcpy(Sk, h.data() + 9 * (size_t)n + ng, nk * sizeof(double));
    }
    return PIGS_OK;
}

// ---- the same, overlapped with the sampler ------------------------------------------------------
// At BASELINE's 128 walkers per GPU the device-resident sampler keeps 128 of the 256 CUs busy for 30 of a step's 38 ms,
// and the estimators of a step (3.6 ms of kernels and copies on the whole chip) then wait for nothing but the step's
// worldline.  _begin snapshots the worldlines (one device-to-device copy on the context's stream: 127 MB, ~50 us) and
// queues the estimator kernels on a SECOND stream of the context, on half the chip; the caller goes on -- typically
// with the next pigs_sampler_step -- and collects the results with _end.  Same kernels on the same bits as
// pigs_diagonal_estimators, hence the same results.  One batch can be pending per context.
// the kernels of the pending batch, on the second stream, once `gate` (an event of the first stream) has passed
static int launch_pending_estimators(pigs_ctx *c, hipEvent_t gate)
{
    if (!c->a_pend.on || c->a_pend.launched) return PIGS_OK;
    c->a_pend.launched = true;
    const int n = c->a_pend.n;
    if (n == 0) return PIGS_OK;
    const int ns = 2 * c->P.Nb;
    const size_t nslot = (size_t)n * ns, ng = c->a_pend.ng, nres = c->a_pend.nres;
    hipStream_t s = c->stream2;
    HIPCHK(hipStreamWaitEvent(s, gate, 0));
    HIPCHK(hipMemcpyAsync(c->a_slotw.p, c->a_sw.data(), c->a_sw.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->a_slotb.p, c->a_sb.data(), c->a_sb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    double *r = c->a_res.p;
    const int32_t *dw = c->a_slotw.p + nslot;
    const int half = c->n_cu / 2 > 0 ? c->n_cu / 2 : 1;
    HIPCHK(launch_local_energy(c->P, c->d_shadow, c->d_VT, c->d_WF, n, dw, 0, r, s));
    HIPCHK(launch_local_energy(c->P, c->d_shadow, c->d_VT, c->d_WF, n, dw, 2 * c->P.Nb, r + 3 * (size_t)n, s));
    HIPCHK(launch_slice_energy(c->P, c->d_shadow, c->d_VT, c->d_VTimg, (int)nslot, c->a_slotw.p, c->a_slotb.p, 1, 1, c->a_slices.p, s, half));
    HIPCHK(launch_therm_combine(c->P, n, c->a_slices.p, r + 6 * (size_t)n, r + 7 * (size_t)n, r + 8 * (size_t)n, s));
    if (c->a_pend.structure)
        HIPCHK(launch_structure(c->P, c->d_shadow, n, dw, c->P.Nb, c->a_pend.Nbin, c->a_pend.rbin, c->a_pend.Nk, r + 9 * (size_t)n,
                                r + 9 * (size_t)n + ng, s));
    HIPCHK(hipMemcpyAsync(c->a_host.h, r, nres * sizeof(double), hipMemcpyDeviceToHost, s));
    return PIGS_OK;
}

int pigs_diagonal_estimators_begin(pigs_ctx *c, int32_t n, const int32_t *walkers, int32_t Nbin, double rbin, int32_t Nk,
                                   int32_t structure)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0) return fail(PIGS_ERR_ARG, "n=%d", n);
    if (c->a_pend.on) return fail(PIGS_ERR_ARG, "an estimator batch is pending: pigs_diagonal_estimators_end first");
    rc = check_cm(c); if (rc) return rc;
    const bool st = structure != 0;
    if (st && (Nbin < 1 || Nk < 0 || !(rbin > 0.0))) return fail(PIGS_ERR_ARG, "bad structure request");
    if (st && c->P.trap) return fail(PIGS_ERR_UNSUPPORTED, "structural estimators are defined for PBC runs only (vpi.f90:466)");
    c->a_pend.n = n; c->a_pend.Nbin = Nbin; c->a_pend.Nk = Nk; c->a_pend.rbin = rbin; c->a_pend.structure = st;
    c->a_pend.launched = false;
    if (n == 0) { c->a_pend.on = true; c->a_pend.nres = 0; return PIGS_OK; }
    const int ns = 2 * c->P.Nb;
    const size_t nslot = (size_t)n * ns;
    c->a_sw.resize(nslot + n); c->a_sb.resize(nslot);
    for (int i = 0; i < n; ++i) {
        const int w = walkers ? walkers[i] : i;
        if (w < 0 || w >= c->n_walkers) return fail(PIGS_ERR_ARG, "walker %d out of range", w);
        for (int b = 0; b < ns; ++b) { c->a_sw[(size_t)i * ns + b] = w; c->a_sb[(size_t)i * ns + b] = b; }
        c->a_sw[nslot + i] = w;
    }
    const size_t ng = st ? (size_t)n * Nbin : 0, nk = st ? (size_t)n * Nk * c->P.dim : 0;
    const size_t nres = (size_t)9 * n + ng + nk;
    c->a_pend.ng = ng; c->a_pend.nk = nk; c->a_pend.nres = nres;
    const size_t nd = c->path_doubles * (size_t)c->n_walkers;
    if (!c->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_snap, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming));
        HIPCHK(hipMalloc((void **)&c->d_shadow, nd * sizeof(double)));
    }
    HIPCHK(c->a_slotw.reserve(nslot + n)); HIPCHK(c->a_slotb.reserve(nslot));
    HIPCHK(c->a_slices.reserve(nslot * 3)); HIPCHK(c->a_res.reserve(nres));
    HIPCHK(c->a_host.reserve(nres * sizeof(double), 0));
    // the snapshot is ordered on the context's stream: after everything queued so far, before whatever comes next.  The
    // kernels themselves are launched by the next pigs_sampler_step behind its TranslateChain kernel (which fills the
    // chip: estimators started beside it would only take CUs away from it) or, failing that, by _end.
    HIPCHK(hipMemcpyAsync(c->d_shadow, c->d_paths, nd * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_snap, c->stream));
    c->a_pend.on = true;
    return PIGS_OK;
}

int pigs_diagonal_estimators_end(pigs_ctx *c, double *en, double *gr, double *Sk)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (!c->a_pend.on) return fail(PIGS_ERR_ARG, "no estimator batch is pending: pigs_diagonal_estimators_begin first");
    c->a_pend.on = false;
    const int n = c->a_pend.n;
    if (n == 0) return PIGS_OK;
    if (!en || (c->a_pend.structure && (!gr || !Sk))) return fail(PIGS_ERR_ARG, "null output");
    c->a_pend.on = true;                                  // (launch_pending_estimators looks at it)
    rc = launch_pending_estimators(c, c->ev_snap);        // no sampler step came in between: start them now
    c->a_pend.on = false;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream2));
    rc = check_cm(c); if (rc) return rc;
    const double *h = reinterpret_cast<const double *>(c->a_host.h);
    for (int i = 0; i < n; ++i) {
        for (int q = 0; q < 3; ++q) { en[9 * i + q] = h[3 * i + q]; en[9 * i + 3 + q] = h[3 * (size_t)n + 3 * i + q]; }
        en[9 * i + 6] = h[6 * (size_t)n + i]; en[9 * i + 7] = h[7 * (size_t)n + i]; en[9 * i + 8] = h[8 * (size_t)n + i];
    }
    if (c->a_pend.structure) {
        memcpy(gr, h + 9 * (size_t)n, c->a_pend.ng * sizeof(double));
        if (c->a_pend.nk) memcpy(Sk, h + 9 * (size_t)n + c->a_pend.ng, c->a_pend.nk * sizeof(double));
    }
    return PIGS_OK;
}

// ---- multi-GPU ---------------------------------------------------------------------------
int pigs_comm_unique_id(char id[128])
{
    if (!id) return fail(PIGS_ERR_ARG, "null id");
    const char *err = pigs_comm_get_unique_id(id);
    return err ? fail(PIGS_ERR_COMM, "%s", err) : PIGS_OK;
}

int pigs_comm_init_rank(pigs_ctx *c, int32_t nranks, int32_t rank, const char id[128])
{
    int rc = check_ctx(c); if (rc) return rc;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) return fail(PIGS_ERR_ARG, "bad rank %d/%d", rank, nranks);
    if (c->comm) { pigs_comm_destroy(c->comm); c->comm = nullptr; }
    const char *err = pigs_comm_create_rank(&c->comm, nranks, rank, id);
    return err ? fail(PIGS_ERR_COMM, "%s", err) : PIGS_OK;
}

int pigs_comm_init_all(pigs_ctx **ctxs, int32_t nranks)
{
    if (!ctxs || nranks < 1) return fail(PIGS_ERR_ARG, "bad arguments");
    std::vector<int> devs(nranks);
    std::vector<pigs_comm *> comms(nranks, nullptr);
    bool distinct = true;
    for (int i = 0; i < nranks; ++i) {
        if (!ctxs[i]) return fail(PIGS_ERR_ARG, "null context %d", i);
        devs[i] = ctxs[i]->device;
        for (int k = 0; k < i; ++k) distinct = distinct && devs[k] != devs[i];
    }
    for (int i = 0; i < nranks; ++i) {
        if (ctxs[i]->comm) { pigs_comm_destroy(ctxs[i]->comm); ctxs[i]->comm = nullptr; }
        ctxs[i]->hgroup.reset();
    }
    if (!distinct) {
        // REHEARSAL form (several contexts on one GPU, e.g. a one-GPU test box: RCCL wants one device per rank): the
        // vectors meet in host memory, every rank adds the ranks' vectors in rank order (deterministic).  Not a
        // multi-GPU path: the product form is the RCCL communicator below.
        auto g = std::make_shared<HostGroup>();
        g->n = nranks;
        g->slot.resize(nranks);
        // (several contexts share one chip here: one workgroup per walker in pigs_cm.hip -- cooperating workgroups assume
        // that the walkers of ONE context have the chip to themselves -- except for TWO shards: their TranslateChain kernels
        // are chained device-wide (g_cm_gate) and take the CUs the other shard's sweep kernel leaves, H x own walkers +
        // the other's walkers <= CUs.  The shards then run staggered -- 38.5 instead of 39.9 ms per MC step of 2 x 64
        // walkers at N=256, scripts/k6_stagger.py.  More than two shards would share the process's four hardware queues.)
        for (int i = 0; i < nranks; ++i) {
            ctxs[i]->hgroup = g; ctxs[i]->hrank = i; ctxs[i]->cm_split = 1;
            if (nranks == 2 && ctxs[i]->n_walkers > 0) {
                int H = (ctxs[i]->n_cu - ctxs[1 - i]->n_walkers) / ctxs[i]->n_walkers;
                H = H > 4 ? 4 : H;
                if (H >= 2) { ctxs[i]->cm_split = H; ctxs[i]->cm_shared = true; }
            }
        }
        return PIGS_OK;
    }
    const char *err = pigs_comm_create_all(comms.data(), nranks, devs.data());
    if (err) return fail(PIGS_ERR_COMM, "%s", err);
    for (int i = 0; i < nranks; ++i) ctxs[i]->comm = comms[i];
    return PIGS_OK;
}

int pigs_estimators_allreduce(pigs_ctx *c, double *vec, int32_t n)
{
    int rc = check_ctx(c); if (rc) return rc;
    if (n < 0 || (n && !vec)) return fail(PIGS_ERR_ARG, "bad vector");
    if (n == 0) return PIGS_OK;
    if (c->hgroup) {
        HostGroup &g = *c->hgroup;
        std::unique_lock<std::mutex> lk(g.m);
        g.slot[c->hrank].assign(vec, vec + n);
        const int gen = g.generation;
        if (++g.arrived == g.n) {
            g.sum.assign(n, 0.0);
            for (int r = 0; r < g.n; ++r) {
                if ((int)g.slot[r].size() != n) { g.arrived = 0; ++g.generation; g.cv.notify_all(); return fail(PIGS_ERR_ARG, "ranks disagree on the vector length"); }
                for (int k = 0; k < n; ++k) g.sum[k] += g.slot[r][k];
            }
            g.arrived = 0;
            ++g.generation;
            g.cv.notify_all();
        } else {
            g.cv.wait(lk, [&] { return g.generation != gen; });
        }
        if ((int)g.sum.size() != n) return fail(PIGS_ERR_ARG, "ranks disagree on the vector length");
        memcpy(vec, g.sum.data(), (size_t)n * sizeof(double));
        return PIGS_OK;
    }
    if (!c->comm) return fail(PIGS_ERR_COMM, "no communicator: call pigs_comm_init_rank / pigs_comm_init_all first");
    HIPCHK(c->d_res.reserve(n));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_res.p, vec, n * sizeof(double), hipMemcpyHostToDevice, s));
    const char *err = pigs_comm_allreduce_sum_f64(c->comm, c->d_res.p, n, s);
    if (err) return fail(PIGS_ERR_COMM, "%s", err);
    HIPCHK(hipMemcpyAsync(vec, c->d_res.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    SYNC_CHECKED(c);
    return PIGS_OK;
}

} // extern "C"
