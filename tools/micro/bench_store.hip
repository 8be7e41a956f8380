// Development microbenchmark: what store patterns does MI355X HBM like?  (hipcc --offload-arch=gfx950 -O3)
// All variants write the same number of bytes (N doubles per array, NA arrays).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// V0/V1: every wave writes K consecutive 64-element runs, shifted by `shift` elements
__global__ void k_stream(double *dst, long n, int K, int shift, int nt) {
    const long wave = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) {
        const long i = (wave * K + k) * 64 + lane + shift;
        if (i < n) { if (nt) __builtin_nontemporal_store(1.0, &dst[i]); else dst[i] = 1.0; }
    }
}
// V2: "tracks": wave w owns 16 consecutive tracks; for each track, runs of <=64 rows in memory order, NA arrays per run
template <int NA>
__global__ void k_tracks(double *const *dsts, const long *off, const int *cnt, long ntracks, int nt) {
    const long wave = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int t = 0; t < 16; ++t) {
        const long u = wave * 16 + t;
        if (u >= ntracks) return;
        const long o = off[u]; const int c = cnt[u];
        for (int r0 = 0; r0 < c; r0 += 64)
            if (r0 + lane < c)
#pragma unroll
                for (int a = 0; a < NA; ++a) { if (nt) __builtin_nontemporal_store(1.0, &dsts[a][o + r0 + lane]); else dsts[a][o + r0 + lane] = 1.0; }
    }
}
// V3: block-of-rows order (like k_compact2): for r0 in blocks of R rows: for each of the 16 tracks: run of <=R rows
template <int NA, int R>
__global__ void k_blocks(double *const *dsts, const long *off, const int *cnt, long ntracks, int nt) {
    const long wave = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int rl = lane % R, sub = lane / R;  // 64/R tracks per instruction
    int cmax = 0;
    for (int t = 0; t < 16; ++t) { const long u = wave * 16 + t; if (u < ntracks && cnt[u] > cmax) cmax = cnt[u]; }
    for (int r0 = 0; r0 < cmax; r0 += R)
        for (int t = sub; t < 16; t += 64 / R) {
            const long u = wave * 16 + t;
            if (u >= ntracks) continue;
            if (r0 + rl < cnt[u])
#pragma unroll
                for (int a = 0; a < NA; ++a) { if (nt) __builtin_nontemporal_store(1.0, &dsts[a][off[u] + r0 + rl]); else dsts[a][off[u] + r0 + rl] = 1.0; }
        }
}
// V4: one workgroup per (block of R rows, 16 tracks): like k_compact3 (different workgroups complete each other's lines)
template <int NA, int R>
__global__ void k_chunks(double *const *dsts, const long *off, const int *cnt, long ntracks, int nblk, int nt) {
    extern __shared__ double smem[];
    if (nt == 99) smem[threadIdx.x] = 1.0;  // keep the allocation alive
    const long wave = ((long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6));
    const long grp = wave / nblk; const int b = (int)(wave % nblk);   // consecutive waves: consecutive blocks of one group
    const int lane = threadIdx.x & 63;
    const int rl = lane % R, sub = lane / R;
    const int r0 = b * R;
    for (int t = sub; t < 16; t += 64 / R) {
        const long u = grp * 16 + t;
        if (u >= ntracks) continue;
        if (r0 + rl < cnt[u])
#pragma unroll
            for (int a = 0; a < NA; ++a) { if (nt) __builtin_nontemporal_store(1.0, &dsts[a][off[u] + r0 + rl]); else dsts[a][off[u] + r0 + rl] = 1.0; }
    }
}

// V5: workgroup = block j of 4 adjacent 16-track groups (what k_compact3 does); `scramble` permutes the workgroup order
template <int NA, int R>
__global__ void k_chunks_wg(double *const *dsts, const long *off, const int *cnt, long ntracks, int nblk, int nt, long nwg, long mult) {
    const long wg = mult ? (long)(((unsigned long long)blockIdx.x * (unsigned long long)mult) % (unsigned long long)nwg) : blockIdx.x;
    const long g64 = wg / nblk; const int b = (int)(wg % nblk);
    const long grp = g64 * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int rl = lane % R, sub = lane / R;
    const int r0 = b * R;
    for (int t = sub; t < 16; t += 64 / R) {
        const long u = grp * 16 + t;
        if (u >= ntracks) continue;
        if (r0 + rl < cnt[u])
#pragma unroll
            for (int a = 0; a < NA; ++a) { if (nt) __builtin_nontemporal_store(1.0, &dsts[a][off[u] + r0 + rl]); else dsts[a][off[u] + r0 + rl] = 1.0; }
    }
}

int main() {
    const long ntracks = 130456; const int NA = 6;
    std::mt19937 rng(1); std::vector<int> cnt(ntracks); std::vector<long> off(ntracks + 1, 0);
    for (long u = 0; u < ntracks; ++u) { cnt[u] = 20 + (int)(rng() % 104); off[u + 1] = off[u] + cnt[u]; }
    const long n = off[ntracks];
    printf("tracks %ld, records %ld, bytes per array %.1f MB, %d arrays\n", ntracks, n, n * 8 / 1e6, NA);
    std::vector<double *> h(NA); double **d_ptrs; long *d_off; int *d_cnt;
    for (int a = 0; a < NA; ++a) CK(hipMalloc(&h[a], (n + 1024) * 8));
    CK(hipMalloc(&d_ptrs, NA * sizeof(double *))); CK(hipMemcpy(d_ptrs, h.data(), NA * sizeof(double *), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_off, (ntracks + 1) * 8)); CK(hipMemcpy(d_off, off.data(), (ntracks + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cnt, ntracks * 4)); CK(hipMemcpy(d_cnt, cnt.data(), ntracks * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, double bytes) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-58s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    const double by1 = n * 8.0, by6 = n * 8.0 * NA;
    for (int nt = 0; nt < 2; ++nt) {
        printf("--- %s stores\n", nt ? "nontemporal" : "plain");
        for (int K : {1, 8, 32}) for (int shift : {0, 3}) {
            char nm[128]; snprintf(nm, sizeof nm, "stream 1 array, K=%d runs/wave, shift=%d", K, shift);
            const long waves = (n / 64 + K - 1) / K;
            timeit(nm, [&] { hipLaunchKernelGGL(k_stream, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, h[0], n, K, shift, nt); }, by1);
        }
        const long wv = (ntracks + 15) / 16;
        timeit("tracks order, 6 arrays (memory order per wave)", [&] { hipLaunchKernelGGL((k_tracks<6>), dim3((unsigned)((wv + 3) / 4)), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, nt); }, by6);
        timeit("tracks order, 1 array", [&] { hipLaunchKernelGGL((k_tracks<1>), dim3((unsigned)((wv + 3) / 4)), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, nt); }, by1);
        timeit("row blocks of 64, 6 arrays (k_compact2-like, fused)", [&] { hipLaunchKernelGGL((k_blocks<6, 64>), dim3((unsigned)((wv + 3) / 4)), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, nt); }, by6);
        timeit("row blocks of 64, 1 array (k_compact2-like)", [&] { hipLaunchKernelGGL((k_blocks<1, 64>), dim3((unsigned)((wv + 3) / 4)), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, nt); }, by1);
        timeit("row blocks of 32, 6 arrays", [&] { hipLaunchKernelGGL((k_blocks<6, 32>), dim3((unsigned)((wv + 3) / 4)), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, nt); }, by6);
        {
            const long nwg = ((ntracks + 63) / 64) * 4;
            timeit("WG = block j of 64 tracks, WG order (w, j)", [&] { hipLaunchKernelGGL((k_chunks_wg<6, 32>), dim3((unsigned)nwg), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, 4, nt, nwg, 0L); }, by6);
            timeit("WG = block j of 64 tracks, WG order scrambled", [&] { hipLaunchKernelGGL((k_chunks_wg<6, 32>), dim3((unsigned)nwg), dim3(256), 0, 0, d_ptrs, d_off, d_cnt, ntracks, 4, nt, nwg, 2654435761L % nwg | 1); }, by6);
        }
        for (int lds : {0, 46}) {
            char nm[128]; snprintf(nm, sizeof nm, "chunks of 32 rows, 6 arrays, %d KB LDS per 256-thread WG", lds);
            timeit(nm, [&] { hipLaunchKernelGGL((k_chunks<6, 32>), dim3((unsigned)((wv * 4 + 3) / 4)), dim3(256), lds * 1024, 0, d_ptrs, d_off, d_cnt, ntracks, 4, nt); }, by6);
        }
    }
    // reference: device-to-device copy of the same bytes
    timeit("hipMemcpy D2D of one array (read+write)", [&] { CK(hipMemcpyAsync(h[1], h[0], n * 8, hipMemcpyDeviceToDevice, 0)); }, 2 * by1);
    timeit("hipMemset of one array", [&] { CK(hipMemsetAsync(h[0], 0, n * 8, 0)); }, by1);
    return 0;
}
