// lat2.hip -- dependent-load latency by working-set size (L2 4 MB per XCD, Infinity Cache 256 MB, HBM), one wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
__global__ void k_chase(const int *chain, int n, int steps, unsigned long long *out)
{
    int j = threadIdx.x == 0 ? 0 : 0;
    for (int i = 0; i < n / 32 && i < (1 << 22); ++i) j = chain[j];       // (partial) warm-up walk
    j = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < steps; ++i) j = chain[j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; out[1] = j; }
}
int main()
{
    unsigned long long *out; hipMalloc(&out, 64);
    for (size_t mb : {1, 2, 8, 32, 128, 1024}) {
        const size_t n = mb * 1024 * 1024 / 4, stride = 32;                // one int per 128-byte line
        const size_t lines = n / stride;
        std::vector<int> perm(lines); for (size_t i = 0; i < lines; ++i) perm[i] = (int)i;
        std::mt19937 g(1); std::shuffle(perm.begin() + 1, perm.end(), g);
        std::vector<int> h(n, 0);
        for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * stride] = perm[(i + 1) % lines] * (int)stride;
        int *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        // the warm-up walks the whole cycle when it is short, a part of it otherwise
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, (int)(lines * 32 > (1u << 30) ? (1u << 30) : lines * 32), 2000, out);
        hipDeviceSynchronize();
        unsigned long long r[2]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
        printf("working set %5zu MB: %llu ticks per dependent load\n", mb, r[0]);
        hipFree(d);
    }
    return 0;
}
