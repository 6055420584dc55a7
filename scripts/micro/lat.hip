// lat.hip -- latency microbenchmarks on one CU (development aid for the K6 stage machine): what one wave pays for the
// primitives the stage chain is built from.  hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define T0() unsigned long long t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__global__ void k_lat(unsigned long long *out, const int *chain, double *gd, int nw_active)
{
    __shared__ int lds[4096];
    __shared__ double ldd[1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (i * 67 + 64) & 4095;
    for (int i = tid; i < 1024; i += blockDim.x) ldd[i] = 1.0 + i * 1e-3;
    __syncthreads();
    const int N = 256;
    // 0: empty timing
    if (wid == 0) { T0(); T1(); if (tid == 0) out[0] = t1 - t0; }
    // 1: dependent LDS reads (pointer chase), one wave
    if (wid == 0) {
        int j = lane;
        T0();
        for (int i = 0; i < N; ++i) j = lds[j];
        T1();
        if (tid == 0) out[1] = (t1 - t0) / N;
        if (j == -1) out[20] = j;
    }
    __syncthreads();
    // 2: dependent DP FMA chain
    if (wid == 0) {
        double x = ldd[lane], y = ldd[lane + 64];
        T0();
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
        asm volatile("" :: "v"(x));
        T1();
        if (tid == 0) out[2] = (t1 - t0) * 100 / N;
        if (x == -1.0) gd[0] = x;
    }
    __syncthreads();
    // 3: independent DP FMAs (4 chains)
    if (wid == 0) {
        double x0 = ldd[lane], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = ldd[lane + 64];
        T0();
#pragma unroll 8
        for (int i = 0; i < N; ++i) { x0 = __builtin_fma(x0, y, 1e-9); x1 = __builtin_fma(x1, y, 1e-9); x2 = __builtin_fma(x2, y, 1e-9); x3 = __builtin_fma(x3, y, 1e-9); }
        asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        T1();
        if (tid == 0) out[3] = (t1 - t0) * 100 / (4 * N);
        if (x0 + x1 + x2 + x3 == -1.0) gd[0] = x0;
    }
    __syncthreads();
    // 4: dependent int VALU chain
    if (wid == 0) {
        int x = lane;
        T0();
#pragma unroll 16
        for (int i = 0; i < N; ++i) { x = x * 3 + 1; asm volatile("" : "+v"(x)); }
        T1();
        if (tid == 0) out[4] = (t1 - t0) * 100 / (2 * N);
        if (x == -1) out[20] = x;
    }
    __syncthreads();
    // 5: barrier round trip, all waves
    {
        T0();
        for (int i = 0; i < N; ++i) __syncthreads();
        T1();
        if (tid == 0) out[5] = (t1 - t0) / N;
    }
    // 6: dependent global loads (L2 hits after first pass), one wave
    if (wid == 0) {
        int j = lane;
        for (int i = 0; i < 64; ++i) j = chain[j];       // warm
        j = lane;
        T0();
        for (int i = 0; i < 64; ++i) j = chain[j];
        T1();
        if (tid == 0) out[6] = (t1 - t0) / 64;
        if (j == -1) out[20] = j;
    }
    __syncthreads();
    // 7: readlane / readfirstlane dependent chain
    if (wid == 0) {
        int x = lane;
        T0();
#pragma unroll 16
        for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_readlane(x, 5) + lane; asm volatile("" : "+v"(x)); }
        T1();
        if (tid == 0) out[7] = (t1 - t0) * 100 / N;
        if (x == -1) out[20] = x;
    }
    __syncthreads();
    // 8: s_memtime back to back
    if (wid == 0) {
        T0();
        unsigned long long s = 0;
        for (int i = 0; i < 64; ++i) { s += __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        T1();
        if (tid == 0) out[8] = (t1 - t0) / 64;
        if (s == 1) out[20] = s;
    }
    __syncthreads();
    // 9: LDS pointer chase on wave 0 while the other nw_active-1 waves hammer LDS with gathers
    {
        int j = lane;
        unsigned long long dt = 0;
        if (wid == 0) {
            T0();
            for (int i = 0; i < N; ++i) j = lds[j];
            T1();
            dt = t1 - t0;
        } else if (wid < nw_active) {
            double s = 0;
            for (int i = 0; i < 4 * N; ++i) s += ldd[(lane * 17 + i * 5) & 1023];
            if (s == -1.0) gd[0] = s;
        }
        if (tid == 0) out[9] = dt / N;
        if (j == -1) out[20] = j;
    }
    __syncthreads();
    // 10: dependent DP chain on wave 0 while the other waves run DP too (same SIMD sharing: waves 0,4 share SIMD 0)
    {
        double x = ldd[lane], y = ldd[lane + 64];
        unsigned long long dt = 0;
        if (wid == 0) {
            T0();
#pragma unroll 16
            for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
            asm volatile("" :: "v"(x));
            T1();
            dt = t1 - t0;
        } else if (wid < nw_active) {
#pragma unroll 16
            for (int i = 0; i < 4 * N; ++i) x = __builtin_fma(x, y, 1e-9);
        }
        if (tid == 0) out[10] = dt * 100 / N;
        if (x == -1.0) gd[0] = x;
    }
    __syncthreads();
    // 11: LDS write then read by another wave through a barrier (producer/consumer hop), measured on wave 1
    {
        unsigned long long dt = 0;
        T0();
        for (int i = 0; i < 64; ++i) {
            if (wid == 0 && lane == 0) lds[0] = i;
            __syncthreads();
            const int v = lds[0];
            if (v == -1) out[20] = v;
            __syncthreads();
        }
        T1();
        dt = t1 - t0;
        if (tid == 64) out[11] = dt / 64;
    }
    // 12: exp + log DP (library) per call, dependent
    if (wid == 0) {
        double x = ldd[lane] * 0.5;
        T0();
        for (int i = 0; i < 32; ++i) x = log(exp(x) + 1.0);
        asm volatile("" :: "v"(x));
        T1();
        if (tid == 0) out[12] = (t1 - t0) / 32;
        if (x == -1.0) gd[0] = x;
    }
    __syncthreads();
    // 13: DP divide + sqrt dependent
    if (wid == 0) {
        double x = ldd[lane];
        T0();
        for (int i = 0; i < 32; ++i) x = sqrt(1.0 / x + 2.0);
        asm volatile("" :: "v"(x));
        T1();
        if (tid == 0) out[13] = (t1 - t0) / 32;
        if (x == -1.0) gd[0] = x;
    }
    __syncthreads();
    // 14: global store + s_waitcnt vmcnt(0) (store acknowledged), one wave
    if (wid == 0) {
        T0();
        for (int i = 0; i < 32; ++i) { gd[64 + lane] = (double)i; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        T1();
        if (tid == 0) out[14] = (t1 - t0) / 32;
    }
}

int main()
{
    const int n = 1 << 16;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (i * 1031 + 4099) & (n - 1);
    int *chain; unsigned long long *out; double *gd;
    hipMalloc(&chain, n * sizeof(int)); hipMalloc(&out, 64 * 8); hipMalloc(&gd, 4096);
    hipMemcpy(chain, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
    const char *names[] = {"empty stamp pair", "LDS dependent read", "DP FMA dependent x100", "DP FMA independent x100", "int VALU dependent x100 (per op)",
                           "__syncthreads", "global load dependent (L2)", "readlane+add x100", "s_memtime + wait", "LDS dep. read, others gather",
                           "DP FMA dep. x100, others busy", "write->barrier->read->barrier", "exp+log", "div+sqrt", "global store + vmcnt(0)"};
    for (int nw : {1, 4, 8}) {
        hipMemset(out, 0, 64 * 8);
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64 * nw), 0, 0, out, chain, gd, nw);
        hipDeviceSynchronize();
        unsigned long long r[64];
        hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
        printf("---- %d waves in the workgroup (s_memtime ticks)\n", nw);
        for (int i = 0; i < 15; ++i) printf("  %-36s %llu\n", names[i], r[i]);
    }
    // clock rate of s_memtime: time a long kernel
    return 0;
}
