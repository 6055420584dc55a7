// pigs_sampler_device.h -- device-side pieces shared by the device-resident sampler's kernels (pigs_sampler.hip:
// open / close attempt, staging-type and worm movers; pigs_diag.hip: the stage machine of the diagonal bisection
// moves): the walker's random stream in LDS, the Metropolis question, nearest-image helpers.
#pragma once

#include "pigs_device.h"
#include "pigs_kernels.h"
#include "pigs_log_host.h"

namespace pigs {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// element j+624 from elements j, j+1, j+397
__device__ __forceinline__ uint32_t mt_next(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// y / (2^32-1), correctly rounded: short exact division by a constant (pigs_device.h div_by)
__device__ __forceinline__ double mt_real(uint32_t y)
{
    constexpr double d = 4294967295.0, rd = 1.0 / 4294967295.0;       // rd = RN(1/d)
    return div_by((double)mt_temper(y), d, rd);
}

// ---- the walker's random stream --------------------------------------------------------------------
// Absolute stream index a = 0, 1, ... counted from the start of the block the launch begins in.  LDS holds
//   W[a mod 1248]   the raw MT19937 word a, for every a in [block start of pos, frontier)
//   Gc[k mod 512]   the polar Box-Muller candidate of the word pair (k, k+1) for k in [pos, gdone):
//                   u1*sqrt(-2 ln q / q) if q = u1^2+u2^2 <= 1, NaN if the reference would reject the pair
// pos = next word to hand out, frontier = words generated so far, gdone = candidates computed so far
// (ctl[0], ctl[11], ctl[12]).  Word a+624 depends only on words a, a+1, a+397, so the frontier is pushed 64 words
// at a time by ONE wave (rng_produce) -- normally the last wave of the workgroup, during the Delta-S phase of
// a stage, while wave 0 is not consuming -- and what a consumer needs is almost always there already: a
// uniform is one LDS read + tempering, G Gaussians are one gather of 64 candidates + a ballot.  Exactly the
// reference's stream (random_mod.f90:35-115, 195-219): same words, same pairing, same rejections.
constexpr int kRing = 2048, kGRing = 512, kLook = 384;    // kRing: a power of two >= 2 * MT_N (ring indices are masks)

struct Rng {
    uint32_t *W;
    double   *Gc;
    double   *Lc;      // Lc[k mod 512] = log(uniform of word k): the Metropolis question of the replicated stages asks
                       // a >= log(u) instead of exp(a) >= u (no exp on the critical path)
    int      *ctl;
};

__device__ __forceinline__ int ring_w(int a) { return a & (kRing - 1); }

// first index of the block that holds the state a checkpoint would save: the reference regenerates a block only
// when a word beyond it is requested, so pos on a block boundary still belongs to the block before it
__device__ __forceinline__ int block_start(int pos) { return pos > 0 && pos % MT_N == 0 ? pos - MT_N : pos - pos % MT_N; }

// one wave: 64 more words if the ring has room and fewer than `want` words are ahead of pos, then up to 64 more
// candidates.  Wave-uniform control flow; call from all 64 lanes of one wave only.
__device__ __forceinline__ void rng_produce(const Rng &R, int want, int lane)
{
    int pos = R.ctl[0], f = R.ctl[11], gd = R.ctl[12];
    __builtin_amdgcn_wave_barrier();
    if (f < pos + want && f + kWave <= block_start(pos) + kRing) {
        const int a = f + lane;
        const uint32_t y = mt_next(R.W[ring_w(a - MT_N)], R.W[ring_w(a - MT_N + 1)], R.W[ring_w(a - MT_N + MT_M)]);
        __builtin_amdgcn_wave_barrier();
        R.W[ring_w(a)] = y;
        f += kWave;
        __builtin_amdgcn_wave_barrier();
    }
    if (gd < pos) gd = pos;                                           // uniforms taken since: those pairs are dead
    const int k = gd + lane;
    int gmax = f - 1 < pos + kGRing ? f - 1 : pos + kGRing;           // pair (k, k+1) needs word k+1; ring capacity
    if (k < gmax) {
        const double uk = mt_real(R.W[ring_w(k)]);
        const double u1 = 2.0 * uk - 1.0;
        const double u2 = 2.0 * mt_real(R.W[ring_w(k + 1)]) - 1.0;
        const double q  = u1 * u1 + u2 * u2;
        double g = __builtin_nan("");
        // log_host: the host libm's log to the bit (pigs_log_host.h) -- the Gaussians are the reference's, bit for bit
        if (q <= 1.0) g = u1 * sqrt_exact((-2.0 * log_host(q)) / q);
        R.Gc[k % kGRing] = g;
        R.Lc[k % kGRing] = log_host(uk);
    }
    gd = gd + kWave < gmax ? gd + kWave : (gd > gmax ? gd : gmax);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { R.ctl[11] = f; R.ctl[12] = gd; }
    __builtin_amdgcn_wave_barrier();
}

// keep the look-ahead topped up (a wave that has nothing better to do in this phase)
__device__ __forceinline__ void rng_background(const Rng &R, int lane)
{
    for (int it = 0; it < 4; ++it) {                                  // bounded: at most 256 words per call
        const int pos = R.ctl[0];
        if (R.ctl[11] >= pos + kLook && R.ctl[12] >= pos + kLook - 1) break;
        rng_produce(R, kLook, lane);
    }
}

// one uniform in [0,1]; call from ONE lane
__device__ __forceinline__ double take1(const Rng &R)
{
    const int pos = R.ctl[0];
    int f = R.ctl[11];
    for (; f <= pos; ++f)                                             // look-ahead exhausted: one word at a time
        R.W[ring_w(f)] = mt_next(R.W[ring_w(f - MT_N)], R.W[ring_w(f - MT_N + 1)], R.W[ring_w(f - MT_N + MT_M)]);
    R.ctl[11] = f;
    R.ctl[0] = pos + 1;
    return mt_real(R.W[ring_w(pos)]);
}

// G unit Gaussians into gbuf[0..G), by all 64 lanes of ONE wave: the first G accepted pairs among the candidates
// at pos, pos+2, pos+4, ... in stream order = G sequential rangauss calls
__device__ __forceinline__ void wave_gaussians(const Rng &R, int G, double *gbuf, int lane)
{
    int done = 0;
    __builtin_amdgcn_wave_barrier();
    while (done < G) {
        const int pos = R.ctl[0];
        while (R.ctl[12] < pos + 2 * kWave - 1) rng_produce(R, 2 * kWave, lane);     // wave-uniform
        const double g = R.Gc[(pos + 2 * lane) % kGRing];
        const bool acc = g == g;
        const unsigned long long m = __ballot(acc);
        const int rank  = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        const int need  = G - done;
        const int total = __builtin_popcountll(m);
        int used = kWave;
        if (total >= need) {
            const unsigned long long sel = __ballot(acc && rank == need - 1);
            used = __builtin_ctzll(sel) + 1;                          // pairs consumed by `need` calls
        }
        if (acc && rank < need) gbuf[done + rank] = g;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) R.ctl[0] = pos + 2 * used;
        __builtin_amdgcn_wave_barrier();
        done += total < need ? total : need;
    }
}

__device__ __forceinline__ double wrap_coord(const DevParams &P, bool trap, int k, double x)
{
    if (trap) return x;
    if (x >  P.LboxHalf[k]) x = x - P.Lbox[k];
    if (x < -P.LboxHalf[k]) x = x + P.Lbox[k];
    return x;
}

// xo + wrap(a - xo)  /  xo - wrap(xo - a): nearest images of an anchor bead (vpi_mod.f90:925-937)
__device__ __forceinline__ double image_prev(const DevParams &P, bool trap, int k, double a, double xo)
{
    double x = a - xo;
    if (!trap) {
        if (x < -P.LboxHalf[k]) x = x + P.Lbox[k];
        if (x >  P.LboxHalf[k]) x = x - P.Lbox[k];
    }
    return xo + x;
}

__device__ __forceinline__ double image_next(const DevParams &P, bool trap, int k, double a, double xo)
{
    double x = xo - a;
    if (!trap) {
        if (x < -P.LboxHalf[k]) x = x + P.Lbox[k];
        if (x >  P.LboxHalf[k]) x = x - P.Lbox[k];
    }
    return xo - x;
}

// Metropolis question on exp(a) (vpi_mod.f90:356-364); call from ONE lane
__device__ __forceinline__ bool metropolis(const Rng &R, double a)
{
    if (a >= -0x1p-54) return true;             // exp(a) rounds to >= 1: no uniform is drawn (as the reference)
    const int pos = R.ctl[0];
    if (a == a && R.ctl[12] > pos) {            // log(u) of the next uniform is tabulated (rng_produce): a >= log(u)
        const double lu = R.Lc[pos % kGRing];   // is the same question without exp(); the two forms differ only when
        R.ctl[0] = pos + 1;                     // exp(a) and u agree to the last bit
        return a >= lu;
    }
    const double e = exp(a);                    // look-ahead exhausted, or NaN (draws a uniform, rejects)
    if (e >= 1.0) return true;
    return e >= take1(R);
}

// The same question asked as a >= log(u), with log(u) of every stream word tabulated next to the Gaussian
// candidates by the producer wave (no exp on the stage's critical path).  exp(a) >= 1 (a >= -2^-54: it rounds to 1)
// accepts without a uniform, as the reference; the two forms differ only when exp(a) and u agree to the last bit.
// Call from all lanes of ONE wave (wave-uniform a).
__device__ __forceinline__ bool metropolis_log(const Rng &R, double a, int lane)
{
    if (a >= -0x1p-54) return true;
    if (!(a == a)) { bool r = false; if (lane == 0) r = metropolis(R, a); return __shfl((int)r, 0, kWave) != 0; }   // NaN: the plain form
    const int pos = R.ctl[0];
    while (R.ctl[12] <= pos) rng_produce(R, 2 * kWave, lane);              // wave-uniform; candidate pos carries log(u_pos)
    const double lu = R.Lc[pos % kGRing];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) R.ctl[0] = pos + 1;
    __builtin_amdgcn_wave_barrier();
    return a >= lu;
}


// cells shared by the swap partner weights (Np) and the task totals: the split evaluation of the staging-type
// stages ((bead, pass) x 8; with Np <= 256 only a lone end bead is split)
__host__ __device__ inline int sweep_tot_doubles(const DevParams &P, const SweepParams &sp)
{
    const int npass = (P.Np + kWave - 1) / kWave;
    const int ntot  = P.Np <= 256 ? 1 : (sp.Lstag > 8 ? sp.Lstag : 8);
    int n = P.Np;
    if (ntot * npass * 8 > n) n = ntot * npass * 8;
    if (n < 64) n = 64;                                               // eight tasks of the lone end bead (new | old distance x four passes)
    return (n + 1) & ~1;
}

} // namespace pigs
