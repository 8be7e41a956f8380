// Development (round 5): how long does the GPU take to START the workgroups of a launch shaped like k_march's (510 workgroups of 256
// threads, two per CU)?  Every workgroup stamps the device-wide 100-MHz clock (wall_clock64) when it starts, then stays for `hold` µs
// so that the launch is resident at once.  Variants: dynamic LDS (36 KB, as the march's private copy of `volumes`), VGPR
// footprint (an inline-asm touch of v220: the march has 221), private scratch (the march's generic step spills 416 B per lane).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/bench_dispatch.hip -o tools/micro/bench_dispatch && tools/micro/bench_dispatch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <bool BIGV, int SCRATCH>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int hold_ticks, int *sink) {
    extern __shared__ int lds[];
    const unsigned long long t0 = wall_clock64();
    if (BIGV) asm volatile("v_mov_b32 v220, 0" ::: "v220");
    int acc = 0;
    if (SCRATCH > 0) {
        volatile int priv[SCRATCH / 4 > 0 ? SCRATCH / 4 : 1];
        for (int i = 0; i < SCRATCH / 4; ++i) priv[i] = i + (int)threadIdx.x;
        for (int i = 0; i < SCRATCH / 4; ++i) acc += priv[(i * 7) % (SCRATCH / 4)];
    }
    if (threadIdx.x == 0) { out[blockIdx.x] = t0; lds[0] = acc; }
    while (wall_clock64() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(32);
    if (acc == 0x7fffffff) sink[0] = lds[0];
}

template <bool BIGV, int SCRATCH>
static void run(const char *name, int blocks, size_t lds, int hold_us) {
    unsigned long long *d;
    int *sink;
    hipMalloc(&d, blocks * sizeof(unsigned long long));
    hipMalloc(&sink, 64);
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k<BIGV, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<unsigned long long> h(blocks);
    double p50 = 0, p90 = 0, p100 = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipLaunchKernelGGL((k<BIGV, SCRATCH>), dim3(blocks), dim3(256), lds, 0, d, hold_us * 100, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        if (r >= 3) { p50 += (h[blocks / 2] - h[0]) / 100.0; p90 += (h[blocks * 9 / 10] - h[0]) / 100.0; p100 += (h[blocks - 1] - h[0]) / 100.0; }
    }
    printf("%-46s %4d workgroups: the median one starts %5.1f µs after the first, the 90th percentile %5.1f, the last %5.1f\n", name, blocks, p50 / reps, p90 / reps, p100 / reps);
    hipFree(d); hipFree(sink);
}

int main() {
    for (int blocks : {510, 255, 128}) {
        run<false, 0>("plain (no LDS, few VGPRs)", blocks, 0, 60);
        run<false, 0>("36 KB of LDS", blocks, 36 * 1024, 60);
        run<true, 0>("221 VGPRs", blocks, 0, 60);
        run<true, 0>("221 VGPRs + 36 KB of LDS", blocks, 36 * 1024, 60);
        run<true, 416>("221 VGPRs + 36 KB of LDS + 416 B of scratch", blocks, 36 * 1024, 60);
        run<false, 416>("416 B of scratch only", blocks, 0, 60);
    }
    return 0;
}
