// TEST INFRASTRUCTURE.  csrc/pigs_log_host.h compiled for the CPU (g++ -mfma -ffp-contract=off) against the libm this
// machine resolves `log` to: counts arguments whose results differ in any bit.  Arguments: the device sampler's domain
// -- uniforms k/(2^32-1), polar radii q = u1^2+u2^2 <= 1 of MT-like pairs -- plus random doubles over (0, 4) and a dense
// sweep of the near-one branch.      usage: log_host_check N_MILLION [seed]   prints "<tested> <mismatches> <first bad hex>"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pigs_log_host.h"

static inline uint64_t rnd(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(int argc, char **argv)
{
    const long long nmil = argc > 1 ? atoll(argv[1]) : 10;
    uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 0x9E3779B97F4A7C15ull;
    unsigned long long tested = 0, bad = 0;
    double first = 0.0;
    #pragma omp parallel reduction(+:tested,bad)
    {
        uint64_t s = seed;
        #ifdef _OPENMP
        s ^= 0xD1B54A32D192ED03ull * (uint64_t)(omp_get_thread_num() + 1);
        const long long n = nmil * 1000000ll / omp_get_num_threads();
        #else
        const long long n = nmil * 1000000ll;
        #endif
        for (long long it = 0; it < n; ++it) {
            double x;
            const uint64_t a = rnd(s), b = rnd(s);
            switch (it & 3) {
            case 0: x = (double)(uint32_t)a / 4294967295.0; break;                              // a uniform of the stream
            case 1: { const double u1 = 2.0 * ((double)(uint32_t)a / 4294967295.0) - 1.0,       // a polar radius
                                   u2 = 2.0 * ((double)(uint32_t)b / 4294967295.0) - 1.0;
                      x = u1 * u1 + u2 * u2; if (x > 1.0) x = x - 1.0; break; }
            case 2: { uint64_t ix = (a >> 12) | ((uint64_t)(0x3ff - (b % 70)) << 52);              // random mantissa, 2^-69 .. 2
                      memcpy(&x, &ix, 8); break; }
            default: x = 0.9375 + (double)(a >> 11) * (1.0 / 9007199254740992.0) * 0.13; break; // the near-one branch
            }
            const double want = std::log(x), got = pigs::log_host(x);
            ++tested;
            if (memcmp(&want, &got, 8) != 0 && !(want != want && got != got)) {
                if (!bad) first = x;
                ++bad;
            }
        }
    }
    // edge values
    const double edge[] = {0.0, 1.0, 0x1p-1074, 0x1p-1022, 0.9375, 1.0 + 0x1.09p-4, 2.0, 1e300, 0.5, 4294967294.0 / 4294967295.0,
                           1.0 / 4294967295.0, __builtin_inf()};
    for (double x : edge) {
        const double want = std::log(x), got = pigs::log_host(x);
        ++tested;
        if (memcmp(&want, &got, 8) != 0) { if (!bad) first = x; ++bad; }
    }
    printf("%llu %llu %a\n", tested, bad, first);
    return bad != 0;
}
