// Development microbenchmark (round 5): what does the WRITE side of k_materialise_lin cost on its own?
// Store-only kernels over the result layout — five f64 arrays + one i32 array, a workgroup of 256 threads writes one contiguous
// span of S records per array with 16-B stores (8 B for the i32 array), exactly the pattern of the record kernel's linear phase —
// against the same bytes as one interleaved 48-B record stream (AoS) and as one plain array.
//   hipcc --offload-arch=gfx950 -O3 -o bench_fill bench_fill.hip && ./bench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double __attribute__((ext_vector_type(2))) d2;
typedef int __attribute__((ext_vector_type(2))) i2;
typedef int __attribute__((ext_vector_type(4))) i4;

struct Arr { double *a[5]; int *e; };

// mode 0: blockIdx -> span in order; 1: XCD-contiguous (an XCD takes an eighth of the spans); 2: scattered (multiplicative hash)
__device__ __forceinline__ long span_of(long b, long nspans, int mode) {
    if (mode == 1) { const long per = (nspans + 7) / 8; return (b & 7) * per + (b >> 3); }
    if (mode == 2) return (long)(((unsigned long long)b * 2654435761ull) % (unsigned long long)nspans);
    return b;
}

template <int NA, bool WITH_E>
__global__ __launch_bounds__(256) void k_soa(Arr A, long n, int S, long nspans, int mode, int iters_per_wave_contig) {
    const long sp = span_of(blockIdx.x, nspans, mode);
    if (sp >= nspans) return;
    const long base = sp * S;  // first record of the span (S even)
    const int P = S / 2;
    const int kw = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (iters_per_wave_contig) {  // wave kw takes a contiguous quarter (as k_materialise_lin)
        const int per = (((P + 3) >> 2) + 7) & ~7;
        const int m0 = kw * per, m1 = P < m0 + per ? P : m0 + per;
        for (int mb = m0; mb < m1; mb += 64) {
            const int m = mb + lane;
            if (m < m1) {
                const long o = base + 2 * m;
                if (o + 1 < n) {
                    d2 v; v.x = (double)o; v.y = 1.0;
#pragma unroll
                    for (int a = 0; a < NA; ++a) *(d2 *)&A.a[a][o] = v;
                    if (WITH_E) { i2 c; c.x = (int)o; c.y = 1; *(i2 *)&A.e[o] = c; }
                }
            }
        }
    } else {  // the four waves side by side
        for (int m = threadIdx.x; m < P; m += 256) {
            const long o = base + 2 * m;
            if (o + 1 < n) {
                d2 v; v.x = (double)o; v.y = 1.0;
#pragma unroll
                for (int a = 0; a < NA; ++a) *(d2 *)&A.a[a][o] = v;
                if (WITH_E) { i2 c; c.x = (int)o; c.y = 1; *(i2 *)&A.e[o] = c; }
            }
        }
    }
}

// one interleaved stream of 48-B records: the workgroup's span is S * 48 contiguous bytes
__global__ __launch_bounds__(256) void k_aos(i4 *dst, long n, int S, long nspans, int mode) {
    const long sp = span_of(blockIdx.x, nspans, mode);
    if (sp >= nspans) return;
    const long base16 = sp * S * 3;  // in 16-B units
    const int n16 = S * 3;
    for (int m = threadIdx.x; m < n16; m += 256) {
        const long o = base16 + m;
        if (o < n * 3) { i4 v; v.x = (int)o; v.y = 1; v.z = 2; v.w = 3; dst[o] = v; }
    }
}

int main(int argc, char **argv) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long n : {9322163L, 114447177L}) {
        for (int S : {1136, 1760}) {
            const long nspans = (n + S - 1) / S;
            Arr A; for (int a = 0; a < 5; ++a) CK(hipMalloc(&A.a[a], (n + 4096) * 8)); CK(hipMalloc(&A.e, (n + 4096) * 4));
            i4 *aos; CK(hipMalloc(&aos, (n + 4096) * 48));
            auto timeit = [&](const char *name, auto launch, double bytes) {
                float best = 1e9;
                for (int rep = 0; rep < 4; ++rep) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
                printf("n=%9ld S=%4d %-64s %8.3f ms  %7.1f GB/s\n", n, S, name, best, bytes / best / 1e6);
            };
            const unsigned nb = (unsigned)(8 * ((nspans + 7) / 8));
            for (int mode = 0; mode < 3; ++mode) {
                char nm[128];
                snprintf(nm, sizeof nm, "SoA 5 x f64 + i32, 16-B stores, wave-contiguous, order %d", mode);
                timeit(nm, [&] { hipLaunchKernelGGL((k_soa<5, true>), dim3(nb), dim3(256), 0, 0, A, n, S, nspans, mode, 1); }, n * 44.0);
                snprintf(nm, sizeof nm, "SoA 5 x f64 + i32, 16-B stores, waves side by side, order %d", mode);
                timeit(nm, [&] { hipLaunchKernelGGL((k_soa<5, true>), dim3(nb), dim3(256), 0, 0, A, n, S, nspans, mode, 0); }, n * 44.0);
                snprintf(nm, sizeof nm, "AoS 48-B records, one stream, order %d", mode);
                timeit(nm, [&] { hipLaunchKernelGGL(k_aos, dim3(nb), dim3(256), 0, 0, aos, n, S, nspans, mode); }, n * 48.0);
            }
            timeit("SoA 1 x f64 only (one of the arrays)", [&] { hipLaunchKernelGGL((k_soa<1, false>), dim3(nb), dim3(256), 0, 0, A, n, S, nspans, 0, 1); }, n * 8.0);
            timeit("SoA 2 x f64", [&] { hipLaunchKernelGGL((k_soa<2, false>), dim3(nb), dim3(256), 0, 0, A, n, S, nspans, 0, 1); }, n * 16.0);
            timeit("hipMemsetAsync of the AoS buffer (48 B x n)", [&] { CK(hipMemsetAsync(aos, 1, n * 48, 0)); }, n * 48.0);
            for (int a = 0; a < 5; ++a) CK(hipFree(A.a[a])); CK(hipFree(A.e)); CK(hipFree(aos));
        }
    }
    return 0;
}
