// How long does hipMalloc take by size, and the first kernel write to that memory?  (round 6: the first rt_segmentize of a C5 handle
// spends 180 ms before its kernels.)   hipcc --offload-arch=gfx950 -O2 -o bench_malloc bench_malloc.hip && ./bench_malloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t mb : {64, 512, 1024, 4096}) {
            double *p = nullptr;
            double t0 = now();
            if (hipMalloc((void **)&p, mb << 20) != hipSuccess) { printf("hipMalloc %zu MB failed\n", mb); continue; }
            double t1 = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, p, (mb << 20) / 8);
            hipDeviceSynchronize();
            double t2 = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, p, (mb << 20) / 8);
            hipDeviceSynchronize();
            double t3 = now();
            hipFree(p);
            double t4 = now();
            printf("%5zu MB: hipMalloc %.2f ms, first kernel over it %.2f ms, second %.2f ms, hipFree %.2f ms\n", mb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
        }
    return 0;
}
