// Development microbenchmark: what does one dependent per-lane fetch of a 128-B record cost on MI355X?
// (hipcc --offload-arch=gfx950 -O3 -o bench_gather bench_gather.hip)
// The march loads one 128-B walk record per lane and iteration, the address coming out of the previous one.
//   A: every lane loads its own record with 8 global_load_dwordx4 (what k_march does)
//   B: eight lanes load one record (one 128-B line per 8 lanes and instruction), transposed through LDS
//   C: as A but only 4 / 2 / 1 loads (64 / 32 / 16 B of the record)
// `share` consecutive lanes follow the same chain (parallel tracks cross the same cells).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct __attribute__((aligned(128))) Rec { int next; int pad[31]; };

template <int NLOAD>
__global__ void k_own(const Rec *rec, int steps, int share, int nrec, unsigned long long *cyc, int *sink) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int idx = (int)((wave * 977 + (lane / share) * 131) % nrec);
    int acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const uint4 *p = reinterpret_cast<const uint4 *>(rec + idx);
        uint4 v[NLOAD];
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) v[j] = p[j];
#pragma unroll
        for (int j = 1; j < NLOAD; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        acc ^= v[0].y;
        idx = v[0].x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[wave] = t1 - t0;
    if (acc == 0x12345678) sink[0] = acc;
}

// B: cooperative.  Tile: record r of the wave at 256-B row r/2, 16-B column ((r&1)*8 + piece) ^ ((r/2) & 15).
__global__ void k_coop(const Rec *rec, int steps, int share, int nrec, unsigned long long *cyc, int *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    unsigned char *tile = smem + wib * 8192;
    const long wave = (long)blockIdx.x * (blockDim.x / 64) + wib;
    int idx = (int)((wave * 977 + (lane / share) * 131) % nrec);
    int acc = 0;
    const int piece = lane & 7, sub = lane >> 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * j + sub;                       // the record this lane helps to load
            const int a = __shfl(idx, r, 64);
            v[j] = reinterpret_cast<const uint4 *>(rec + a)[piece];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * j + sub;
            const int col = (((r & 1) * 8 + piece) ^ ((r >> 1) & 15));
            *reinterpret_cast<uint4 *>(tile + (r >> 1) * 256 + col * 16) = v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint4 w[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int col = (((lane & 1) * 8 + p) ^ ((lane >> 1) & 15));
            w[p] = *reinterpret_cast<const uint4 *>(tile + (lane >> 1) * 256 + col * 16);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 1; p < 8; ++p) acc ^= w[p].x ^ w[p].y ^ w[p].z ^ w[p].w;
        acc ^= w[0].y;
        idx = w[0].x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[wave] = t1 - t0;
    if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int nrec = argc > 1 ? atoi(argv[1]) : 11730;   // pincell: 3 * 3910 records = 1.5 MB
    const int steps = 200;
    std::vector<Rec> h(nrec);
    std::mt19937 rng(7);
    // successor: a nearby record (mesh locality: +-40 records), as neighbouring cells have nearby ids
    for (int i = 0; i < nrec; ++i) {
        int n = i + (int)(rng() % 81) - 40;
        h[i].next = ((n % nrec) + nrec) % nrec;
        for (int k = 0; k < 31; ++k) h[i].pad[k] = (int)rng();
    }
    Rec *d; unsigned long long *dc; int *ds;
    CK(hipMalloc(&d, sizeof(Rec) * nrec));
    CK(hipMemcpy(d, h.data(), sizeof(Rec) * nrec, hipMemcpyHostToDevice));
    const int max_waves = 256 * 12;
    CK(hipMalloc(&dc, sizeof(unsigned long long) * max_waves));
    CK(hipMalloc(&ds, 4));
    auto report = [&](const char *name, int waves) {
        std::vector<unsigned long long> c(waves);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
        double s = 0, mx = 0;
        for (auto v : c) { s += (double)v; if ((double)v > mx) mx = (double)v; }
        printf("  %-28s %7.0f cycles/step (mean), %7.0f (slowest wave)\n", name, s / waves / steps, mx / steps);
    };
    for (int share : {1, 4, 16})
        for (int wgs : {1, 256, 768}) {
            const int waves = wgs * 4;
            printf("share=%d  workgroups=%d (x4 waves)\n", share, wgs);
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k_own<8>, dim3(wgs), dim3(256), 0, 0, d, steps, share, nrec, dc, ds);
                if (rep) report("A: own record, 8 loads", waves);
                hipLaunchKernelGGL(k_own<4>, dim3(wgs), dim3(256), 0, 0, d, steps, share, nrec, dc, ds);
                if (rep) report("C: own record, 4 loads", waves);
                hipLaunchKernelGGL(k_own<2>, dim3(wgs), dim3(256), 0, 0, d, steps, share, nrec, dc, ds);
                if (rep) report("C: own record, 2 loads", waves);
                hipLaunchKernelGGL(k_own<1>, dim3(wgs), dim3(256), 0, 0, d, steps, share, nrec, dc, ds);
                if (rep) report("C: own record, 1 load", waves);
                hipLaunchKernelGGL(k_coop, dim3(wgs), dim3(256), 4 * 8192, 0, d, steps, share, nrec, dc, ds);
                if (rep) report("B: 8 lanes per record + LDS", waves);
            }
        }
    return 0;
}
