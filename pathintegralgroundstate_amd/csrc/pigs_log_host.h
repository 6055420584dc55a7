// pigs_log_host.h -- log(x) with the bits of the HOST's libm (glibc 2.35 x86-64, the FMA build every CPU of this
// pool resolves `log` to), for the device-resident sampler's Box-Muller radius and log(u) table.
//
// The reference draws its Gaussians as u1*sqrt(-2 log(w)/w) (random_mod.f90:195-219) with libm's log, which is
// accurate to 0.52 ulp but not correctly rounded: a device log that is merely as accurate differs from it in the last
// bit for ~1 argument in 10, the proposals then differ by an ulp, and the mixed estimator (a second difference of a
// table interpolant) amplifies that to 6e-10 -- outside the 1e-10 contract.  So this is the same
// algorithm (ARM optimized-routines log.c as shipped in glibc >= 2.28: x = 2^k z, z in [0x1.6p-1, 0x1.6p0),
// r = z/c - 1 from a 128-entry table of (1/c, log c), log1p(r) by a degree-5 polynomial; a degree-11 polynomial in
// r = x - 1 with a split square for 1-2^-4 <= x < 1+0x1.09p-4), performing THE SAME floating-point operations in the
// same order, with a fused multiply-add exactly where the x86-64 FMA build has one (read off its instruction
// sequence: the build contracts a*b+c wherever the source has that shape).  Constants: pigs_glibc_log_table.h
// (generated from the image's libm by scripts/gen_glibc_log_table.py).
//
// Checked bit for bit: on the CPU by tests/test_host_logic.py (this header compiled by g++ -mfma against libm's log,
// 4e8 arguments of the sampler's domain, 0 mismatches) and on the GPU by pigs_selftest_log against the GPU box's own
// libm.  The translation unit must be compiled with -ffp-contract=off (the build does): every fusion is written out.
#pragma once

#include <stdint.h>

#ifndef __HIPCC__
#define __device__
#define PIGS_LOG_FN static inline
#else
#define PIGS_LOG_FN __device__ __forceinline__
#endif

#include "pigs_glibc_log_table.h"

namespace pigs {

PIGS_LOG_FN double log_host(double x)
{
    using namespace glibc_log;
    uint64_t ix;
    __builtin_memcpy(&ix, &x, 8);
    // 1 - 2^-4 <= x < 1 + 0x1.09p-4: the table form would lose log(c)+r to cancellation
    if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r  = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double p1 = __builtin_fma(r2, kB[3], __builtin_fma(r, kB[2], kB[1]));
        const double p2 = __builtin_fma(r2, kB[6], __builtin_fma(r, kB[5], kB[4]));
        double       p3 = __builtin_fma(r2, kB[9], __builtin_fma(r, kB[8], kB[7]));
        p3 = __builtin_fma(r3, kB[10], p3);
        double y = __builtin_fma(p3, r3, p2);
        y = __builtin_fma(y, r3, p1);
        // r = rhi + rlo with rhi*rhi exact; hi + lo = r - r^2/2 to twice the precision
        const double t   = __builtin_fma(r, 0x1p27, r);
        const double rhi = __builtin_fma(-0x1p27, r, t);
        const double rlo = r - rhi;
        const double s   = rhi * rhi;
        const double hi  = __builtin_fma(s, kB[0], r);
        double lo = __builtin_fma(s, kB[0], r - hi);
        lo = __builtin_fma(kB[0] * rlo, r + rhi, lo);
        y = __builtin_fma(y, r3, lo);
        return hi + y;
    }
    const uint32_t top = (uint32_t)(ix >> 48);
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {                         // zero, subnormal, negative, inf, NaN
        if (ix * 2 == 0) return -__builtin_inf();
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __builtin_nan("");
        const double xs = x * 0x1p52;                                 // subnormal: normalise
        __builtin_memcpy(&ix, &xs, 8);
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int      i   = (int)((tmp >> 45) & 127);
    const int      k   = (int)((int64_t)tmp >> 52);
    const uint64_t iz  = ix - (tmp & 0xfff0000000000000ull);
    double z;
    __builtin_memcpy(&z, &iz, 8);
    const double invc = kTab[2 * i], logc = kTab[2 * i + 1];
    const double r  = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w  = __builtin_fma(kd, kLn2hi, logc);
    const double hi = r + w;
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, kLn2lo, lo);
    const double r2 = r * r;
    const double q  = __builtin_fma(__builtin_fma(r, kA[4], kA[3]), r2, __builtin_fma(r, kA[2], kA[1]));
    lo = __builtin_fma(r2, kA[0], lo);
    const double y = __builtin_fma(r * r2, q, lo);
    return y + hi;
}

} // namespace pigs
