// rt_cheap.hip — k_cheap: the cheap loop of the march and nothing else (gfx950; the lean plan, rt_internal.hpp DLean).
//
// One lane = one track (_segmentize_track!, src/track.jl:106-178), as in k_march — but this kernel holds ONLY the decision-only
// cheap step (rt_device.hpp topo_geo / topo_certified / topo_commit: which cell the reference emits next, through which edge) and
// its fill_volumes tally.  The exact step the loop almost never takes (walk_step, find_element + intersections: 200 registers,
// scratch) lives in another kernel: k_first (k_march<PHASE 1>) made every track's first record and left the lane's state in memory;
// a lane whose cheap step refuses here — or that has no certified successor record, or whose iteration bound reaches the cap —
// writes its state back, queues its march slot for k_serve (k_march<PHASE 2>) and LEAVES the loop for good.  The registers that
// frees are waves: four per SIMD instead of two.  The iteration is k_march's, statement for statement (same functions, same
// operands: the staged words are the same bits; the tally adds in another order, compared at 1e-10 like every fused tally).
#include "rt_internal.hpp"

namespace rt {

struct CheapArgsLayout {
    DMesh m; DTracks t; DParams prm; int32_t *counts; int32_t *status; DOut out; DStage stg; unsigned long long *ctl; DLean ln;
};
__device__ __forceinline__ const RT_K DStage *cheap_stage_args() {
    return (const RT_K DStage *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CheapArgsLayout, stg));
}
__device__ __forceinline__ unsigned long long *cheap_ctl() {
    return *(unsigned long long *const RT_K *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CheapArgsLayout, ctl));
}

#ifndef RT_CHEAP_OCC
#define RT_CHEAP_OCC 4
#endif

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, RT_CHEAP_OCC) void k_cheap(DMesh m, DTracks t, DParams prm, int32_t *__restrict__ counts,
                                                                    int32_t *__restrict__ status, DOut out, DStage stg,
                                                                    unsigned long long *__restrict__ ctl, DLean ln) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cheap_smem[];
    double *hist = reinterpret_cast<double *>(cheap_smem);  // [n_cells]: the workgroup's share of fill_volumes
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    lds_i32 *chunk_lds = (lds_i32 *)(cheap_smem + (size_t)m.n_cells * sizeof(double)) + wib * kMaxChunks;
    {
        const RT_K DStage *sk = cheap_stage_args();
        if (sk->cursor != stg.cursor || sk->element != stg.element || sk->pool_chunks != stg.pool_chunks || cheap_ctl() != ctl) {
            if (threadIdx.x == 0) stg.cursor[1] = 2;  // (guards CheapArgsLayout against drift, as k_march does)
            return;
        }
    }
    for (int c = lane; c < kMaxChunks; c += 64) chunk_lds[c] = -1;
    for (int c = threadIdx.x; c < m.n_cells; c += 64 * WAVES) hist[c] = 0.0;
    __syncthreads();
    const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wib;
    const int64_t slot = wave_id * 64 + lane;
    const bool act = slot < t.n;
    const int64_t sc = act ? slot : 0;
    const int32_t f0 = act ? ln.fl[sc] : kLnFinal;
    const bool mine = act && (f0 & (int32_t)kFlCheap) && !(f0 & (kLnFinal | kLnExact));  // the lane marches here
    uint32_t fl = mine ? (uint32_t)f0 & (kFlCheap | kFlUsed) : 0u;
    const double tA = t.As[sc], tB = t.Bs[sc], tC = t.Cs[sc];
    const double cs_u = t.Dxs[sc], sn_u = t.Dys[sc];
    const double w = t.w_slot[sc];
    TopoState ts;
    ts.pred = mine ? ln.pred[sc] : -1; ts.last = ln.last[sc]; ts.sp = ln.sp[sc]; ts.sn = ln.sn[sc]; ts.apos = (f0 & kLnApos) != 0;
    double ttP = ln.ttP[sc], ttN = ln.ttN[sc], ttp = ln.ttp[sc], dprev = ln.dprev[sc];
    int32_t i = ln.i[sc], it = ln.it[sc], last_word = ln.word[sc];
    const int32_t cap = (int32_t)(prm.iter_cap < 0x7fffffff ? prm.iter_cap : 0x7fffffff);
    const int kk = prm.k > 2 ? (prm.k < rt::kExtrasNever - 1 ? prm.k : rt::kExtrasNever - 1) : 2;
    TopoTrack tt = topo_track(m.walk_ok, m.d_vertex, prm.topo_tiny_max, prm.topo_rmax, prm.topo_end_err, prm.tiny_step, cs_u, sn_u);
    asm volatile("" : "+v"(tt.dv), "+v"(tt.c1), "+v"(tt.c2));
    const double nab = sqrt(tA * tA + tB * tB);
    const double wq = w / nab;
    const double tc1 = prm.tally_c1 * nab, tc2 = prm.tally_c2 * (nab * nab);
    int32_t n_exact_tally = 0;
    const RT_G TopoRec *trec_v = m.trec;
    asm volatile("" : "+v"(trec_v));
    // The wave's chunk j: cached in LDS once a lane has asked for it (a look-up of what k_first allocated, or an allocation)
    auto get_chunk = [&](const int j) -> int32_t {
        bool pending = true;
        int32_t minec = -1;
        for (;;) {
            const unsigned long long mask = __ballot(pending);
            if (!mask) break;
            const int L = __ffsll((long long)mask) - 1;
            const int jL = __shfl(j, L);
            int32_t c = chunk_lds[jL];
            if (c == -1) {
                if (lane == L) { c = lean_chunk(cheap_stage_args(), wave_id, jL); chunk_lds[jL] = c; }
                c = __shfl(c, L);
            }
            if (pending && j == jL) { minec = c; pending = false; }
        }
        return minec;
    };
    RT_G int32_t *row_el = ln.dump + lane;  // (the loop's store is unconditional: a lane that does not march here, or has no chunk, stores to the dump row)
    if (__ballot(mine)) {
        int32_t my_chunk = -1;
        if (mine) my_chunk = get_chunk((i - 1) >> kChunkLog2);  // the chunk of the lane's last record (every lane has one: k_first's)
        if (my_chunk >= 0) row_el = cheap_stage_args()->element + stage_slot(my_chunk, 0, lane);
    }
    {
        const RT_G TopoRec *R = trec_v + (ts.pred >= 0 ? ts.pred : 0);
        uint64_t c_hdr = R->hdr;
        double c_x2 = R->x2, c_y2 = R->y2;
        uint32_t c_c01 = R->c01, c_c23 = R->c23;
        for (;;) {
            const bool cheap = (fl & kFlCheap) != 0;
            if (!__ballot(cheap)) break;
            const TopoGeo g = topo_geo(ts, c_hdr, c_x2, c_y2, tA, tB, tC);
            const int32_t np = topo_next(g);
            const RT_G TopoRec *Rn = trec_v + (np >= 0 ? np : 0);
            const uint64_t n_hdr = Rn->hdr;
            const double n_x2 = Rn->x2, n_y2 = Rn->y2;
            const uint32_t n_c01 = Rn->c01, n_c23 = Rn->c23;
            asm volatile("" ::: "memory");  // the loads above stay above the store below
            int32_t kub;
            const bool ok = topo_certified(tt, ts, g, c_hdr, c_c01, c_c23, kk, kub);
            const bool over = it + kub > cap;  // (`it` is an upper bound of the reference's iterations after cheap steps)
            const bool commit = cheap && ok && !over;
            const double sp0 = ts.sp, sn0 = ts.sn;
            topo_advance(ts, g);
            bool inexact = false;
            {
                // fill_volumes (src/trackgenerator.jl:382) for this record, as k_march tallies it: the chord from the vertices' signed
                // distances and their positions along the line, used where its error bound allows (rt_mesh_prep.hpp)
                const double t2 = __builtin_fma(tB, c_x2, -(tA * c_y2));
                ttP = g.p2 ? t2 : ttP; ttN = g.p2 ? ttN : t2;
                const double den = ts.sp - ts.sn;
                double rc = __builtin_amdgcn_rcp(den);
                rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                const double tx = (ts.sp * ttN - ts.sn * ttP) * rc;
                const double ch = fabs(tx - ttp), dmin = fmin(fabs(den), dprev);
                inexact = !(ch >= tc1 && ch * dmin >= tc2);
                atomicAdd(&hist[g.cell], (commit && !inexact) ? wq * ch : 0.0);
                ttp = tx; dprev = fabs(den);
                n_exact_tally += (commit && inexact) ? 1 : 0;
            }
            if (commit) {
                ++i;
                it += kub;
                const int r = topo_commit(tt, ts, g);
                fl |= kFlUsed;
                if (r == kTopoEnd) fl = (fl & ~kFlCheap) | kFlDone;             // on the border, within tiny_step: src/track.jl:130-132
                else if (ts.pred < 0) fl = (fl & ~kFlCheap) | kFlMat;           // no certified successor: the exact step goes on
                else if (i >= kMaxIter) fl = (fl & ~kFlCheap) | kFlDone;         // MAX_ITER, src/track.jl:104,119
            } else if (cheap) {
                fl = (fl & ~kFlCheap) | (ok ? kFlRestart : kFlMat);  // refused: the exact step decides this record (k_serve)
                // per-call statistic (rt_last_stats): which certificate term refused — cold
                TopoState ts0 = ts;
                ts0.sp = sp0; ts0.sn = sn0;
                const uint32_t bad = ok ? 0u : topo_refusal_terms(tt, ts0, g, c_hdr, c_c01, c_c23, kk);
                unsigned long long *cb = cheap_ctl();
                const int first = __ffsll((long long)__ballot(1)) - 1;
                for (int b = 0; b < 9; ++b) {
                    const unsigned long long mb = __ballot((bad >> b) & 1u);
                    if (mb && lane == first) atomicAdd(cb + kCtlRefusal + b, (unsigned long long)__popcll(mb));
                }
            }
            // stage record i - 1 (a lane that decided nothing stores its last word again: same address, same bits)
            const int rw = (i - 1) & (kChunkRows - 1);
            if (__builtin_expect(commit && rw == 0, 0)) {
                const int32_t my_chunk = get_chunk((i - 1) >> kChunkLog2);
                if (my_chunk >= 0) row_el = cheap_stage_args()->element + stage_slot(my_chunk, 0, lane);
            }
            last_word = commit ? (g.code + 1) | (inexact ? kWordExactTally : 0) : last_word;
            row_el[rw * 16] = last_word;
            c_hdr = n_hdr; c_x2 = n_x2; c_y2 = n_y2; c_c01 = n_c01; c_c23 = n_c23;
            // ---- a lane that cannot go on here LEAVES now: its state to memory, its slot to the queue (k_serve may be running beside
            // this kernel and picks it up at once), and from now on its store of every iteration goes to a dump row — k_serve stages
            // into the lane's own column.  Cold: 535 of 9.3 M records at the headline configuration.
            const bool left = cheap && (fl & (kFlCheap | kFlDone)) == 0;
            const unsigned long long qm = __ballot(left);
            if (__builtin_expect(qm != 0, 0)) {
                asm volatile("" ::: "memory");
                if (left) {
                    ln.i[slot] = i; ln.it[slot] = it; ln.word[slot] = last_word; ln.last[slot] = ts.last;
                    ln.fl[slot] = (int32_t)(fl & (kFlUsed | kFlMat | kFlRestart));
                }
                __threadfence();  // (the lane's words and state first, then — release — the queue entry)
                if (left) {
                    const int L = __ffsll((long long)qm) - 1;
                    int32_t b0 = 0;
                    if (lane == L) b0 = atomicAdd((int32_t *)&ln.qctl[0], (int32_t)__popcll(qm));
                    const int32_t idx = __shfl(b0, L) + (int32_t)__popcll(qm & ((1ull << lane) - 1ull));
                    __hip_atomic_store((int32_t *)&ln.queue[idx], (int32_t)slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    row_el = ln.dump + lane;
                }
            }
        }
    }
    // ---- what the lanes whose track ended here leave
    const bool fin = mine && (fl & kFlDone) != 0;
    if (__ballot(mine)) {
        const int32_t u = t.perm[sc];
        if (fin) { counts[u] = i; status[u] = RT_TRACK_OK; t.cnt_slot[slot] = i; }
        auto wave_sum = [&](const int32_t v) -> unsigned long long {
            unsigned long long r = 0;
            for (int b = 0; b < 14; ++b) r += (unsigned long long)__popcll(__ballot((v >> b) & 1)) << b;
            return r;
        };
        const bool first = lane == 0;
        const unsigned long long ne = wave_sum(n_exact_tally);
        RT_G int32_t *acc = cheap_stage_args()->tile_acc;
        const int32_t i_end = fin ? i : 0;
        if (acc) {
            const int32_t tile = (int32_t)(u >> 10);
            const int32_t t0 = __builtin_amdgcn_readfirstlane(tile);
            RT_G int32_t *line = acc + (size_t)t0 * kTileAccStride;
            if (__ballot(act && tile != t0) == 0) {
                const unsigned long long ws = wave_sum(i_end);
                if (first && ws) atomicAdd((int32_t *)line, (int32_t)ws);
            } else if (i_end) {
                atomicAdd((int32_t *)(acc + (size_t)tile * kTileAccStride), (int32_t)i_end);
            }
            if (first && ne) atomicAdd((int32_t *)(line + 2), (int32_t)ne);
        } else if (first && ne) {
            atomicAdd(cheap_ctl() + kCtlExactTally, ne);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < m.n_cells; c += 64 * WAVES) {
        const double v = hist[c];
        if (v != 0.0) unsafeAtomicAdd((double *)&out.volumes[c], v);
    }
    // the workgroup is over: k_serve ends when every workgroup is and the queue is empty (release: the pushes above come first)
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add((int32_t *)&ln.qctl[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace rt

namespace rtx {

int launch_cheap(int waves, unsigned blocks, size_t smem, hipStream_t s, const rt::DMesh &m, const rt::DTracks &t, const rt::DParams &prm,
                 int32_t *counts, int32_t *status, const rt::DOut &out, const rt::DStage &stg, unsigned long long *ctl, const rt::DLean &ln) {
    auto go = [&]<int WAVES>() -> int {
        if (smem > 48 * 1024)
            RT_HIP(hipFuncSetAttribute((const void *)rt::k_cheap<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((rt::k_cheap<WAVES>), dim3(blocks), dim3(64 * WAVES), smem, s, m, t, prm, counts, status, out, stg, ctl, ln);
        return RT_SUCCESS;
    };
    if (waves == 4) return go.template operator()<4>();
    if (waves == 8) return go.template operator()<8>();
    if (waves == 16) return go.template operator()<16>();
    set_error("k_cheap: no instantiation for %d waves per workgroup", waves);
    return RT_ERR_INVALID;
}

}  // namespace rtx
