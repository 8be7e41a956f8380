// rt_march.hip — the march kernels of librt_segmentize.so (gfx950): one lane marches one track (_segmentize_track!,
// src/track.jl:106-178), k_seed / k_resolve for batches marched in pieces.  Launched through rtlaunch:: (rt_internal.hpp).
#include "rt_internal.hpp"

namespace rt {

// (development, -DRT_STOP_AFTER=N: every track stops after N records — a timing probe of the march's start; results are void)
#ifdef RT_STOP_AFTER
constexpr int kIterLimit = RT_STOP_AFTER;
#else
constexpr int kIterLimit = kMaxIter;
#endif

template <bool WIDEK>
__global__ __launch_bounds__(64) void k_seed(DMesh m, DTracks t, DParams prm, DSplit sp) {
    const int32_t cv = blockIdx.x;
    const int32_t k = sp.vw_k[cv];
    if (k == 0) return;
    const int32_t w = sp.vw_wave[cv];
    const int lane = threadIdx.x;
    const int64_t u = (int64_t)w * 64 + lane;
    if (u >= t.n) return;
    const int64_t pi = (int64_t)cv * 64 + lane;
    const double frac = (double)k / (double)sp.w_P[w];
    const double cs = t.cs[u], sn = t.sn[u];
    const double mx = t.px[u] + (frac * t.ell[u]) * cs, my = t.py[u] + (frac * t.ell[u]) * sn;
    int32_t el = -1;
    GenericOut go;
    go.eq = -1;
    if (!inboundary(m, mx, my, prm.tiny_step)) {
        const DGeo g = load_geo(m.geo);
        const int rc = generic_step<WIDEK>(g, mx, my, prm.k, -1, t.phi[u], t.A[u], t.B[u], t.C[u], go);
        // Any genuine segment of the track near M will do: whether the march really produces it is
        // checked bit for bit by the piece that arrives there (k_march), not assumed here.
        if (rc == 0 && go.eq >= 0 && go.ell >= m.l_min) el = go.element;
    }
    sp.s_el[pi] = el;
    if (el >= 0) {
        sp.s_eq[pi] = go.eq;
        sp.s_px[pi] = go.px; sp.s_py[pi] = go.py; sp.s_qx[pi] = go.qx; sp.s_qy[pi] = go.qy;
        sp.s_ell[pi] = go.ell;
    }
}

// One lane per track: follow the chain of matched pieces, keep exactly those, and finish the
// per-track results (count, status, the Σℓ check of src/track.jl:171, failure summary).
__global__ __launch_bounds__(256) void k_resolve(DTracks t, DParams prm, DSplit sp, int32_t *__restrict__ counts,
                                                 int32_t *__restrict__ status,
                                                 unsigned long long *__restrict__ fail_info) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= t.n) return;
    const int32_t w = (int32_t)(u >> 6), lane = (int32_t)(u & 63);
    const int32_t P = sp.w_P[w], base = sp.w_base[w];
    for (int k = 0; k < P; ++k) sp.p_valid[(int64_t)(base + k) * 64 + lane] = 0;
    int32_t total = 0, st = RT_TRACK_OK;
    int64_t iters = 0;
    double sum = 0.0;
    int k = 0;
#ifdef RT_STATS
    {   // development statistics of the split plan: control-block words 22.. (tracks, live seeds, pieces, records marched)
        unsigned long long alive = 0, cnt_all = 0;
        for (int kk2 = 0; kk2 < P; ++kk2) {
            if (kk2 > 0 && sp.s_el[(int64_t)(base + kk2) * 64 + lane] >= 0) ++alive;
            cnt_all += (unsigned long long)sp.p_count[(int64_t)(base + kk2) * 64 + lane];
        }
        atomicAdd(&fail_info[22], 1ull);
        atomicAdd(&fail_info[23], alive);
        atomicAdd(&fail_info[24], (unsigned long long)(P - 1));
        atomicAdd(&fail_info[25], cnt_all);
    }
#endif
    for (int guard = 0; guard < P; ++guard) {
#ifdef RT_STATS
        atomicAdd(&fail_info[26], 1ull);  // pieces kept
#endif
        const int64_t pi = (int64_t)(base + k) * 64 + lane;
        const int32_t c = sp.p_count[pi], fl = sp.p_flags[pi];
        iters += (int64_t)sp.p_rel[pi];  // (written by the march: the piece's iteration count; overwritten just below)
        sp.p_rel[pi] = total;
        sp.p_valid[pi] = c;
        total += c;
        sum += sp.p_sum[pi];
        if (st == RT_TRACK_OK) st = (fl >> 8) & 255;
        if (!(fl & 1) || st != RT_TRACK_OK) break;
        k = fl >> 16;  // the piece whose seed this one met
    }
    if (st == RT_TRACK_OK && !isapprox_s(t.ell[u], sum, prm.rtol)) st = RT_TRACK_LENGTH_MISMATCH;
    if (sum_check_is_marginal(t.ell[u], sum, prm.rtol, total)) atomicAdd(&fail_info[kCtlNearRtol], 1ull);
    {   // records of overrun pieces were marched (and, with fused volumes, accumulated) but are not kept
        int32_t all = 0;
        for (int kk2 = 0; kk2 < P; ++kk2) all += sp.p_count[(int64_t)(base + kk2) * 64 + lane];
        if (all != total) atomicAdd(&fail_info[7], (unsigned long long)(all - total));
    }
    // MAX_ITER counts the segments of a whole track (src/track.jl:104,119): the reference stops after 10000 of them and
    // then fails its Σℓ check.  Pieces count on their own, so a track that reaches the limit is flagged and the host
    // marches the batch again without splitting (practically never: est > MAX_ITER/2 already marches whole).
    if (total >= kMaxIter) atomicAdd(&fail_info[21], 1ull);  // (word 21 of the control block)
    // likewise the library's own guard on the reference's unbounded `continue` paths (RT_TRACK_ITER_CAP) counts the
    // iterations of a whole track: a track whose pieces together exceed it, or one of whose pieces ran into it, is marched
    // again whole, so that status and records are what the unsplit march gives
    // (piece boundaries shift the count by one or two iterations each: anything near the limit goes to the whole march)
    if (iters + 4 * P >= prm.iter_cap || st == RT_TRACK_ITER_CAP) atomicAdd(&fail_info[21], 1ull);
    counts[u] = total;
    status[u] = st;
    if (st != RT_TRACK_OK) {
        atomicAdd(&fail_info[0], 1ull);
        atomicMin(&fail_info[1], (unsigned long long)(u + 1));
    }
}

// One lane marches one track (_segmentize_track!, src/track.jl:106-178).  kStage: single pass,
// records go to the wave-interleaved staging pool (then k_compact3).  kCount / kFill: the
// two-pass variant (count, scan, re-march writing at the CSR offsets).  All modes set counts[] /
// status[] identically.  WAVES = 1: one wave per workgroup.  WAVES = 4 (kStage only): four
// consecutive waves share one workgroup and an LDS-private copy of `volumes`, so fill_volumes
// (src/trackgenerator.jl:371-386) is fused into the march as ds_add_f64 + one coalesced flush.
// SPLIT (kStage): the lanes march pieces of tracks (see DSplit above); with fused volumes the records of a
// piece that overran its stop seed are counted by k_resolve and the host recomputes the volumes (rare).
#ifdef RT_TIMING
// development only: in-kernel cycle stamps (s_memtime), tied to a value so the compiler keeps the order;
// RT_TIMING=2 also drains the memory queue before every stamp
__device__ __forceinline__ unsigned long long rt_tick(double dep) {
    unsigned long long t;
#if RT_TIMING == 2
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
#else
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
#endif
    return t;
}
#endif
// k_march's staging pointers are needed once per 32 iterations (chunk hand-out, row addresses) and on rare
// records: they are read from the kernel-argument segment with scalar loads where they are used instead of
// living in 19 SGPRs across the whole loop (which the kernel was spilling to VGPR lanes and reloading on
// its hot path).  The struct mirrors k_march's parameter list.
struct MarchArgsLayout {
    DMesh m; DTracks t; DParams prm; int32_t *counts; int32_t *status; const int64_t *offsets; DOut out; DStage stg;
    unsigned long long *fail_info; DSplit sp; DLean ln;
};
__device__ __forceinline__ const RT_K DStage *march_stage_args() {
    const RT_K char *ka = (const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr();
    return (const RT_K DStage *)(ka + offsetof(MarchArgsLayout, stg));
}

// the call's control block / parameters, read from the argument segment in cold branches (not held across the loop)
__device__ __forceinline__ unsigned long long *march_ctl() {
    return *(unsigned long long *const RT_K *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, fail_info));
}
__device__ __forceinline__ const RT_K DParams *march_prm_args() {
    return (const RT_K DParams *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, prm));
}

// WIDEK: k > kMaxK (the knn fallback of find_element serves its node list in batches); a separate instantiation, so that
// the march of the usual k keeps its register budget.
// TOPO (whole tracks, staged): the walk step split into a DECISION that needs no point at all (rt_device.hpp, topo_geo /
// topo_certified: which cell the reference emits next, through which edges — from the signed distances of the cell's
// vertices to the track line) and the ARITHMETIC of the record (exit point on the predicted edge with the reference's
// formula, ℓ), which no longer feeds the next iteration: a lane's dependent chain per record is one 32-B record fetch and
// a dozen instructions, and the next record's fetch is in flight while the certificates and the record are evaluated.
// The exact step (walk_step / generic) runs only for the lanes whose cheap step refused.
// PHASE (whole tracks with cheap steps; the lean plan, rt_internal.hpp DLean): 0 the march in one kernel; 1 k_first — start band and
// every track's first record, then the lane's state goes to memory; 2 k_serve — persistent waves claim queued march slots and
// march those tracks from their stored state to the end.
template <int MODE, int WAVES, bool SPLIT, bool WIDEK = false, bool TOPO = false, int PHASE = 0>
#ifndef RT_TOPO_OCC
#define RT_TOPO_OCC 0
#endif
__global__ __launch_bounds__(64 * WAVES, (SPLIT && MODE == kStage && WAVES > 1) ? 3 : (TOPO ? RT_TOPO_OCC : 0)) void k_march(DMesh m, DTracks t, DParams prm, int32_t *__restrict__ counts,
                                                      int32_t *__restrict__ status,
                                                      const int64_t *__restrict__ offsets, DOut out, DStage stg,
                                                      unsigned long long *__restrict__ fail_info, DSplit sp, DLean ln) {
    static_assert(PHASE == 0 || (TOPO && WAVES > 1), "the lean plan: whole tracks, cheap steps, fused fill_volumes");
    // The split plan's tables are used at the start and the end of a piece and when a record of the target's cell comes
    // up — never in the steady march: they are read from the argument segment where they are used (as `stg` is), so
    // that their 17 pointers do not occupy scalar registers across the loop.
    const RT_K DSplit *spk = (const RT_K DSplit *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, sp));
    (void)sp;
    static_assert(!TOPO || (MODE == kStage && !SPLIT), "cheap steps: staged whole tracks only");
    // (TOPO: the march DECIDES and stages codes; exit points, lengths and Σℓ are k_materialise's.  fill_volumes stays here, in the
    //  LDS-private copy: its sum is compared at 1e-10, not bit for bit, so a cheap record's length comes from the vertices'
    //  signed distances and positions along the line — one reciprocal — instead of the record's two divisions and square root.)
    constexpr bool FUSE = WAVES > 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char march_smem[];
    double *hist = reinterpret_cast<double *>(march_smem);  // [n_cells] when FUSE
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    // (an LDS-address-space pointer: through a generic one these become FLAT accesses that drain vmcnt)
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    lds_i32 *chunk_lds = (lds_i32 *)(march_smem + (FUSE ? (size_t)m.n_cells * sizeof(double) : 0)) + wib * kMaxChunks;
    if (MODE == kStage) {
        // the argument-segment view of `stg` must be the argument itself (guards MarchArgsLayout against drift:
        // a mismatch voids the attempt the way a pool overflow does, and the host reports it)
        const RT_K DStage *sk = march_stage_args();
        if (sk->cursor != stg.cursor || sk->qx != stg.qx || sk->element != stg.element || sk->pool_chunks != stg.pool_chunks ||
            *(unsigned long long *const RT_K *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, fail_info)) != fail_info) {
            if (threadIdx.x == 0) stg.cursor[1] = 2;
            return;
        }
        for (int c = lane; c < kMaxChunks; c += 64) chunk_lds[c] = -1;
        if (FUSE)
            for (int c = threadIdx.x; c < m.n_cells; c += 64 * WAVES) hist[c] = 0.0;
        // Records in completion order (DStage::cq): the host launches the record kernel BESIDE this one only when the grid's last
        // workgroups have started — workgroups are dealt to the XCDs in turn and every XCD starts its share in order, so the last
        // eight starting means that every workgroup has its slots (or, a batch of several residency rounds, that the last round has
        // begun): the record kernel's waiting workgroups can then never keep a march workgroup off the chip.
        if (TOPO && PHASE == 0 && FUSE && sk->cq && threadIdx.x == 0 && blockIdx.x + 8 >= gridDim.x)
            __hip_atomic_store(sk->cq_started + (gridDim.x - 1 - blockIdx.x), (unsigned long long)sk->cq_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
    int32_t end_cnt = 0, end_u = -1;  // (completion order: what the lane's track left — read behind the march, where the workgroup takes its span)
    for (;;) {  // (PHASE 2: one pass per claim of queued slots; else a single pass)
    int64_t wave_id = (int64_t)blockIdx.x * WAVES + wib;  // indexes the wave's chunk table (ctab)
    int64_t slot = wave_id * 64 + lane;
    if (PHASE == 2) {
        // ---- claim up to 64 queued march slots (DLean::queue); none left and every k_cheap workgroup over: the wave ends
        int32_t q_base = 0, q_cnt = 0;
        if (lane == 0) {
            RT_G int32_t *qc = ln.qctl;
            unsigned spins = 0;
            for (;;) {
                const int32_t head = __hip_atomic_load((int32_t *)&qc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int32_t tail = __hip_atomic_load((int32_t *)&qc[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (head < tail) {
                    const int32_t nq = tail - head < 64 ? tail - head : 64;
                    if (atomicCAS((int32_t *)&qc[1], head, head + nq) == head) { q_base = head; q_cnt = nq; break; }
                    continue;
                }
                if (__hip_atomic_load((int32_t *)&qc[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= ln.n_cheap_wgs) {
                    // (every push of k_cheap happens before its workgroup's count: one more look at the tail decides)
                    tail = __hip_atomic_load((int32_t *)&qc[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (__hip_atomic_load((int32_t *)&qc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tail) { q_cnt = -1; break; }
                    continue;
                }
                __builtin_amdgcn_s_sleep(16);
                // an exit every wave reaches: k_cheap is on its stream before this kernel is launched, so this never fires — if it
                // does (≈25 ms without work), the attempt is void and the host marches again with the one-kernel plan
                if (++spins > 60000u) { qc[3] = 1; q_cnt = -1; break; }
            }
        }
        q_base = __builtin_amdgcn_readfirstlane(q_base); q_cnt = __builtin_amdgcn_readfirstlane(q_cnt);
        if (q_cnt < 0) break;
        slot = t.n;
        if (lane < q_cnt) {
            int32_t e;
            unsigned spins = 0;
            while ((e = __hip_atomic_load((int32_t *)&ln.queue[q_base + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 4000000u) { ln.qctl[3] = 1; break; }
            }
            if (e >= 0) { slot = e; ln.queue[q_base + lane] = -1; }  // (consumed: the queue is all -1 again when the call ends)
        }
        wave_id = slot >> 6;
    }
    const int lane_o = PHASE == 2 ? (int)(slot & 63) : lane;  // the lane's column in its wave's staging chunks
    if (PHASE == 1) {
        // k_first leaves the wave's chunk table clean (-1: no chunk yet; lean_chunk)
        RT_G int32_t *row = stg.ctab + wave_id * kMaxChunks;
        if (wave_id * 64 < t.n)
            for (int c = lane; c < kMaxChunks; c += 64) row[c] = -1;
    }
    int32_t pk = 0, pP = 1, pw = 0;  // SPLIT: piece index, pieces per track, wave of tracks
    if (SPLIT) {
        const int64_t vidx = (int64_t)blockIdx.x * WAVES + wib;  // position in the dispatch order
        if (vidx < spk->n_vwaves) {
            wave_id = spk->vorder[vidx];
            pw = spk->vw_wave[wave_id];
            pk = spk->vw_k[wave_id];
            pP = spk->w_P[pw];
            slot = (int64_t)pw * 64 + lane;
        } else {
            slot = t.n;  // padding wave of the last workgroup
        }
    }
    if (slot < t.n) {
    const int32_t u = SPLIT ? (int32_t)slot : t.perm[slot];
    // SPLIT: the seed this piece starts from (k >= 1) and the next live seed, at which it stops.  Only the target's cell
    // and piece index live in registers across the march; its p and q are read when a record of that cell comes up.
    bool seed_pending = false, piece_dead = false, matched = false;
    int32_t tgt_el = -1, tgt_pj = 0;  // tgt_pj: index of the target piece's seed (canonical virtual wave * 64 + lane)
    if (SPLIT) {
        const int64_t pi = wave_id * 64 + lane;
        if (pk > 0) {
            if (spk->s_el[pi] < 0) piece_dead = true;
            else seed_pending = true;
        }
        for (int kk2 = pk + 1; kk2 < pP; ++kk2) {
            const int64_t pj = (int64_t)(spk->w_base[pw] + kk2) * 64 + lane;
            const int32_t e = spk->s_el[pj];
            if (e >= 0) { tgt_el = e; tgt_pj = (int32_t)pj; break; }
        }
    }
    // What a lane starts from.  Whole tracks: by march slot (k_slot_arrays left copies in march order) — every load of the kernel's
    // first trip is issued at once, none waits for perm[slot]; pieces: by uid (the slot of a piece's wave IS its first uid + lane).
    const double tA = SPLIT ? t.A[u] : t.As[slot], tB = SPLIT ? t.B[u] : t.Bs[slot], tC = SPLIT ? t.C[u] : t.Cs[slot];
    const double phi = SPLIT ? t.phi[u] : t.Phis[slot];
    const double cs_u = SPLIT ? t.cs[u] : t.Dxs[slot], sn_u = SPLIT ? t.sn[u] : t.Dys[slot];
    const double px_u = SPLIT ? t.px[u] : t.Pxs[slot], py_u = SPLIT ? t.py[u] : t.Pys[slot];
    // advance_step (src/point.jl:43): x + step * Point2D(cos ϕ, sin ϕ)
    const double sx = prm.tiny_step * cs_u;
    const double sy = prm.tiny_step * sn_u;
    double xpx = px_u + sx, xpy = py_u + sy;  // src/track.jl:114
    int64_t base = 0;
    double w = 0.0;
    if (MODE == kFill) base = offsets[u];
    if (MODE == kFill || FUSE) w = out.delta_s[(SPLIT ? t.azim[u] : t.Azs[slot]) - 1];
    int32_t my_chunk = -1;
    RT_G double *row_qx = nullptr, *row_qy = nullptr;  // this lane's slots of row 0 of its current chunk
    RT_G int32_t *row_el = nullptr;
    int i = 0;
    int32_t it = 0;
    const int32_t cap = (int32_t)(prm.iter_cap < 0x7fffffff ? prm.iter_cap : 0x7fffffff);
    int32_t prev_element = -1;
    int32_t n_generic = 0;  // records of this lane made by the generic step (whole-track kernels)
    int st = RT_TRACK_OK;
    double sum_ell = 0.0;
    Walk wk;
    wk.T = -1; wk.pred = -1; wk.last = -1;
    wk.ax = wk.ay = wk.bx = wk.by = wk.cx = wk.cy = 0.0; wk.dT = 1.0;
    // node window of find_element(xp) then find_element(xp, k) as the walk records count it (extras field: 0..14, 15 = never)
    const int kk = prm.k > 2 ? (prm.k < rt::kExtrasNever - 1 ? prm.k : rt::kExtrasNever - 1) : 2;
    double lqx = 0.0, lqy = 0.0;  // exit point of the last emitted segment
    // The walk step's mesh constants, held in VGPRs: as SGPRs they share a tuple of the argument load that the
    // register allocator spills as a whole and reloads (8 v_readlane) several times per iteration.
    DMesh mh = m;
    asm volatile("" : "+v"(mh.d_vertex), "+v"(mh.l_min), "+v"(mh.wrec));
    NextRec nr;
    load_next(mh, -1, nr);
    // per-lane state of the cheap step
    TopoTrack tt = topo_track(TOPO && m.walk_ok, m.d_vertex, prm.topo_tiny_max, prm.topo_rmax, prm.topo_end_err, prm.tiny_step, cs_u, sn_u);
    // (wave-uniform constants of the cheap loop in vector registers: the scalar file is the short one — 73 scalar values of the
    //  kernel live in vector lanes, and every use of one inside the loop is a v_readlane)
    if (TOPO) asm volatile("" : "+v"(tt.dv), "+v"(tt.c1), "+v"(tt.c2));
    TopoState ts;
    ts.pred = -1; ts.last = 0; ts.sp = ts.sn = 0.0; ts.apos = false;
    // kFlCheap: the lane takes cheap steps; kFlUsed: it has taken some (`it` is then an upper bound of the reference's
    // iterations); kFlMat: the exact step's state has to be rebuilt from `ts.last`; kFlWait: nothing to do until the wave
    // has no cheap lane left (an uncertified last step, a finished track); kFlDone / kFlRestart: see below
    // (the constants: rt_internal.hpp — k_cheap shares them)
    uint32_t fl = 0;
    // positions along the track line, t(x, y) = B·x − A·y ((B, −A) is the line's direction; general_form normalises the whole
    // (A, B, C), src/intersection.jl:11-18, so t is scaled by ‖(A, B)‖), of the end points of the lane's entry edge (by side, as
    // ts.sp / ts.sn) and of its last exit point: the chord a cheap record adds to fill_volumes is |Δt| / ‖(A, B)‖ — the scale rides in wq
    double ttP = 0.0, ttN = 0.0, ttp = 0.0;
    const double nab = (TOPO && FUSE) ? sqrt(tA * tA + tB * tB) : 1.0;
    const double wq = (TOPO && FUSE) ? w / nab : 0.0;
    // what the approximate chord may be used for (rt_mesh_prep.hpp, tally_c1 / tally_c2; s and t are scaled by ‖(A, B)‖)
    const double tc1 = (TOPO && FUSE) ? prm.tally_c1 * nab : 0.0, tc2 = (TOPO && FUSE) ? prm.tally_c2 * (nab * nab) : 0.0;
    double dprev = 0.0;  // |s_p − s_q| of the crossing the lane's last exit point came from (∞: an exact step's point)
    int32_t n_exact_tally = 0;  // cheap records of this lane whose fill_volumes term is left to k_materialise
    auto topo_tally_enter = [&]() {
        const double ta = __builtin_fma(tB, wk.ax, -(tA * wk.ay)), tb = __builtin_fma(tB, wk.bx, -(tA * wk.by));
        ttP = ts.apos ? ta : tb; ttN = ts.apos ? tb : ta;
        ttp = __builtin_fma(tB, lqx, -(tA * lqy));
        dprev = INFINITY;  // (an exact step's exit point)
    };
    int32_t n_cheap_it = 0, n_cheap_ref = 0;  // wave-uniform: cheap iterations of this wave, and those in which a lane was refused
    int32_t last_word = 0;  // staging word of the lane's last record
    // (the cheap loop stores without a branch: should the pool run out before a lane's first chunk — the attempt is void
    //  then and the host re-runs it — its row pointers must still be addresses inside the pool)
    if (TOPO) row_el = stg.element + lane_o;
    const RT_G TopoRec *trec_v = m.trec;
    const RT_G EdgeABC *etab_v = m.etab;
    if (TOPO) asm volatile("" : "+v"(trec_v), "+v"(etab_v));
    // Start band (:125-129 with no segment yet): step by tiny_step until xp leaves the boundary
    // band.  Run as its own loop so that the 64 lanes of the wave, whose bands differ in length
    // (≈1/sin ϕ or 1/|cos ϕ| steps), reach their first locate together.
    while (PHASE != 2 && !(SPLIT && (seed_pending || piece_dead)) && st == RT_TRACK_OK && inboundary(m, xpx, xpy, prm.tiny_step)) {
        if (++it > cap) { st = RT_TRACK_ITER_CAP; break; }
        xpx = xpx + sx; xpy = xpy + sy;
    }
#ifdef RT_TIMING
    unsigned long long tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tn = 0, tD = 0, wits = 0, wgen = 0;
    const unsigned long long tstart = rt_tick(xpx);
#endif
    // First row of a new chunk for a lane: wave-aggregated allocation among the lanes that are here.
    // chunk_lds[j] caches what the wave already owns.
    auto alloc_chunk = [&](const int j) -> int32_t {
        if (PHASE == 2) return lean_chunk(march_stage_args(), wave_id, j);  // (the lanes of a k_serve wave come from different waves)
        bool pending = true;
        int32_t mine = -1;
        for (;;) {
            const unsigned long long mask = __ballot(pending);
            if (!mask) break;
            const int L = __ffsll((long long)mask) - 1;
            const int jL = __shfl(j, L);
            int32_t c = chunk_lds[jL];
            if (c == -1) {
                if (lane == L) {
                    const RT_K DStage *sk = march_stage_args();
                    RT_G int32_t *cursor = sk->cursor;
                    // a chunk the host reserved (DStage: regions by chunk index; the cursor starts behind them) needs no atomic —
                    // every wave wants its first chunk at the same moment, on its first record: 2,039 atomics on one word
                    if (PHASE == 1) c = lean_chunk(sk, wave_id, jL);  // (k_cheap / k_serve look the chunk up in ctab)
                    else if (!SPLIT && jL < sk->n_regions && wave_id < sk->reg_cap[jL]) c = sk->reg_base[jL] + (int32_t)wave_id;
                    else c = atomicAdd((int32_t *)&cursor[0], 1);
                    if (PHASE == 1) { }
                    else if (c >= sk->pool_chunks) { c = -2; cursor[1] = 1; }  // pool exhausted: host grows it and re-runs
                    else {
                        sk->ctab[wave_id * kMaxChunks + jL] = c;
                        sk->cowner[c] = (int32_t)(wave_id * kMaxChunks + jL);
                    }
                    chunk_lds[jL] = c;
                }
                c = __shfl(c, L);
            }
            if (pending && j == jL) { mine = c; pending = false; }
        }
        return mine;
    };
    if (PHASE == 2) {
        // ---- the state k_first / k_cheap stored for this slot (DLean)
        const int32_t f = ln.fl[slot];
        i = ln.i[slot]; it = ln.it[slot]; last_word = ln.word[slot];
        if (f & kLnExact) {
            // behind the first record no cheap step was possible: the exact step's own state
            lqx = ln.lqx[slot]; lqy = ln.lqy[slot];
            prev_element = ln.prev_el[slot];
            const int32_t wl = ln.wk_last[slot];
            if (wl >= 0) {
                const int32_t cell = (int32_t)((uint32_t)wl / 3u);
                walk_enter(m, load_tri(load_geo(m.geo), cell), wk, cell, wl - 3 * cell);
            } else { wk.T = prev_element; wk.pred = -1; }
            xpx = lqx + sx; xpy = lqy + sy;  // :165
            fl = 0;
        } else {
            // a lane that left k_cheap: its cheap step refused (kFlMat: the exact step's state is rebuilt from the last record's
            // code), or its iteration bound reached the cap (kFlRestart)
            fl = (uint32_t)f & (kFlUsed | kFlMat | kFlRestart);
            ts.last = ln.last[slot]; ts.pred = -1;
        }
        if (i > 0) {
            my_chunk = alloc_chunk((i - 1) >> kChunkLog2);  // (the chunk of the last staged record: a look-up)
            if (my_chunk >= 0) row_el = march_stage_args()->element + stage_slot(my_chunk, 0, lane_o);
        }
    }
    // A lane whose track creeps (see below) for more than kCreepLocal tiny steps leaves the march loop and
    // waits for the wave: once every lane is out, all 64 lanes test 64 consecutive creep positions of that
    // track at a time (cooperative creep), then the lane marches on.
    bool creep_escalate = false;
    int creep_run = 0;  // generic tiny steps in a row
    for (;;) {
    while (!(SPLIT && piece_dead) && st == RT_TRACK_OK && i < kIterLimit && !creep_escalate && !(PHASE == 1 && (fl & kFlCheap))) {  // :119
        if (TOPO && PHASE != 1) {  // (k_first: a lane without a record takes no cheap step)
            // ---- cheap steps: a wave-uniform inner loop that runs while some lane is in cheap mode and no lane is due
            //      for an exact step (lanes whose track has ended, or that wait with an uncertified last step, idle here)
            {
                // The decision-only march: an iteration decides record n — which cell the reference emits next, left through
                // which edge (topo_geo / topo_certified / topo_commit: two FMAs and a dozen compares on the 32-B record) — and
                // stages its code, 3·cell + exit edge; exit point, length, Σℓ and fill_volumes are functions of (track line,
                // edge, previous record) and are evaluated by k_materialise, in parallel over all records, not on this chain.
                // The loads of record n + 1 are issued as soon as record n's exit edge is known and waited for at the end of
                // the iteration; the one 4-B store follows them (gfx950 retires loads and stores through one in-order counter:
                // a load issued behind a store waits for that store's acknowledgement as well) and is unconditional — behind a
                // store inside a branch the compiler waits for everything: a lane that decided nothing stores its last word
                // again (same address, same bits).
                const RT_G TopoRec *R = trec_v + (ts.pred >= 0 ? ts.pred : 0);
                uint64_t c_hdr = R->hdr;
                double c_x2 = R->x2, c_y2 = R->y2;
                uint32_t c_c01 = R->c01, c_c23 = R->c23;
                for (;;) {
                    const bool cheap = (fl & kFlCheap) != 0;
                    if (!__ballot(cheap)) break;
                    if (__ballot((fl & (kFlCheap | kFlWait)) == 0)) break;
                    const TopoGeo g = topo_geo(ts, c_hdr, c_x2, c_y2, tA, tB, tC);
                    const int32_t np = topo_next(g);
#ifdef RT_STATS_DISTINCT
                    {   // development: how many distinct successor records / exit edges the wave's cheap lanes fetch in this iteration
                        auto distinct = [&](const int32_t key) -> int {
                            unsigned long long act = __ballot(cheap);
                            int nd = 0;
                            while (act) {
                                const int32_t v = __builtin_amdgcn_readlane(key, __ffsll((long long)act) - 1);
                                act &= ~__ballot(key == v);
                                ++nd;
                            }
                            return nd;
                        };
                        const int d1 = distinct(np), d2 = distinct(g.code), na = __popcll(__ballot(cheap));
                        if (lane == 0) {
                            atomicAdd(march_ctl() + 44 + (d1 < 8 ? d1 : 8), 1ull);       // 45..52: distinct successor records 1..8+
                            atomicAdd(march_ctl() + 53 + (d2 < 8 ? d2 : 8), 1ull);       // 54..61: distinct exit edges 1..8+
                            atomicAdd(march_ctl() + 62, (unsigned long long)na);         // cheap lanes
                            atomicAdd(march_ctl() + 63, 1ull);                           // wave-iterations
                        }
                    }
#endif
                    const RT_G TopoRec *Rn = trec_v + (np >= 0 ? np : 0);
                    const uint64_t n_hdr = Rn->hdr;
                    const double n_x2 = Rn->x2, n_y2 = Rn->y2;
                    const uint32_t n_c01 = Rn->c01, n_c23 = Rn->c23;
                    asm volatile("" ::: "memory");  // the loads above stay above the store below
                    int32_t kub;
                    const bool ok = topo_certified(tt, ts, g, c_hdr, c_c01, c_c23, kk, kub);
                    const bool over = it + kub > cap;  // (`it` is an upper bound of the reference's iterations after cheap steps)
                    ++n_cheap_it;
                    n_cheap_ref += __ballot(cheap && !ok) != 0 ? 1 : 0;
                    const bool commit = cheap && ok && !over;
                    const double sp0 = ts.sp, sn0 = ts.sn;  // (the refusal statistic's cold branch evaluates the terms on the entry edge again)
                    topo_advance(ts, g);  // (every lane: the state of one that does not commit is dead)
                    bool inexact = false;
                    if (FUSE) {
                        // fill_volumes (src/trackgenerator.jl:382) for this record: the line meets the exit edge — its end points P, N on
                        // either side, s_P − s_N >= the record's k2 — at t = (s_P·t_N − s_N·t_P) / (s_P − s_N)
                        const double t2 = __builtin_fma(tB, c_x2, -(tA * c_y2));
                        ttP = g.p2 ? t2 : ttP; ttN = g.p2 ? ttN : t2;
                        const double den = ts.sp - ts.sn;
                        double rc = __builtin_amdgcn_rcp(den);
                        rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                        rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                        const double tx = (ts.sp * ttN - ts.sn * ttP) * rc;
                        // (a chord that is short, or one of whose ends is a shallow crossing, is left to k_materialise: its relative
                        //  error bound, rt_mesh_prep.hpp, is 2e-11·(c1/chord + c2/(chord·min D_x)) — per record, hence for every sum)
                        const double ch = fabs(tx - ttp), dmin = fmin(fabs(den), dprev);
                        inexact = !(ch >= tc1 && ch * dmin >= tc2);
                        atomicAdd(&hist[g.cell], (commit && !inexact) ? wq * ch : 0.0);  // (LDS-private; a lane that decided nothing adds 0)
                        ttp = tx; dprev = fabs(den);  // (of a lane that does not commit: dead until topo_tally_enter)
                        n_exact_tally += (commit && inexact) ? 1 : 0;
                    }
                    if (commit) {
                        ++i;
                        it += kub;
                        const int r = topo_commit(tt, ts, g);
                        fl |= kFlUsed;
                        if (r == kTopoEnd) fl = (fl & ~kFlCheap) | kFlDone | kFlWait;  // on the border, within tiny_step: :130-132
                        else if (ts.pred < 0) fl = (fl & ~kFlCheap) | kFlMat | kFlWait;
                        else if (i >= kIterLimit) fl = (fl & ~kFlCheap) | kFlWait;
                    } else if (cheap) {
                        fl = (fl & ~kFlCheap) | (ok ? kFlRestart : kFlMat);  // refused: the exact step decides this record
                        // per-call statistic (rt_last_stats): which certificate term refused — a cold branch (every refusal
                        // costs its wave an exact step anyway); one atomic per term and wave
                        TopoState ts0 = ts;
                        ts0.sp = sp0; ts0.sn = sn0;
                        const uint32_t bad = ok ? 0u : topo_refusal_terms(tt, ts0, g, c_hdr, c_c01, c_c23, kk);
                        unsigned long long *ctl = march_ctl();
                        const int first = __ffsll((long long)__ballot(1)) - 1;
                        for (int b = 0; b < 9; ++b) {
                            const unsigned long long mb = __ballot((bad >> b) & 1u);
                            if (mb && lane == first) atomicAdd(ctl + kCtlRefusal + b, (unsigned long long)__popcll(mb));
                        }
                    }
                    // stage record i - 1 (every lane in here has one: cheap steps follow an exact step's record)
                    const int rw = (i - 1) & (kChunkRows - 1);
                    if (__builtin_expect(commit && rw == 0, 0)) {
                        my_chunk = alloc_chunk((i - 1) >> kChunkLog2);
                        // (pool exhausted: the attempt is void and the host re-runs it; the row pointer stays inside the pool)
                        if (my_chunk >= 0) row_el = march_stage_args()->element + stage_slot(my_chunk, 0, lane_o);
                    }
                    last_word = commit ? (g.code + 1) | (inexact ? kWordExactTally : 0) : last_word;
                    row_el[rw * 16] = last_word;
                    c_hdr = n_hdr; c_x2 = n_x2; c_y2 = n_y2; c_c01 = n_c01; c_c23 = n_c23;
                }
            }
            // A wave whose lanes are refused in more than one iteration out of eight (a mesh with many records that carry the
            // walk step's certificates but not the cheap step's: every refusal is an exact pass the other lanes wait for)
            // goes on with exact steps only, i.e. as the march without cheap steps.
            if (__builtin_expect(tt.on && n_cheap_ref >= 16 && 8 * n_cheap_ref > n_cheap_it, 0)) {
                if (PHASE == 2 || march_prm_args()->topo_force) {  // option "topo" = 2: every record that carries a cheap certificate uses it
                                                                    // (k_serve: its lanes ARE the refused ones — the rule is about waves of whole batches)
                    n_cheap_ref = 0; n_cheap_it = 0;
                } else {
                    tt.on = false;
                    if (fl & kFlCheap) fl = (fl & ~kFlCheap) | kFlMat;
                }
            }
            const bool any_cheap = __ballot((fl & kFlCheap) != 0) != 0;
            if ((fl & kFlDone) || i >= kIterLimit) break;
            if ((fl & kFlCheap) || ((fl & kFlWait) && any_cheap)) continue;  // (an uncertified last step waits until no lane is cheap)
            if (__builtin_expect((fl & kFlRestart) || ((fl & kFlUsed) && it >= cap), 0)) {
                // the bound reached the iteration cap: this track is marched again from its start with exact steps only
                asm volatile("" ::: "memory");
                // its records have already been added to the fused volumes: the host recomputes them from the records
                atomicAdd(march_ctl() + kCtlRestarts, 1ull);
                tt.on = false; fl = 0; n_generic = 0; n_exact_tally = 0;
                i = 0; it = 0; prev_element = -1; wk.T = -1; wk.pred = -1; creep_run = 0; my_chunk = -1; sum_ell = 0.0;
                xpx = (SPLIT ? t.px[u] : t.Pxs[slot]) + sx; xpy = (SPLIT ? t.py[u] : t.Pys[slot]) + sy;  // (read again: not kept in registers across the march)
                continue;
            }
        }
        if (++it > cap) { if (TOPO && (fl & kFlUsed)) continue; st = RT_TRACK_ITER_CAP; break; }
        if (TOPO && PHASE != 1 && __builtin_expect((fl & kFlMat) != 0, 0)) {
            asm volatile("" ::: "memory");
            fl &= ~kFlMat;
            const int32_t cell = (int32_t)((uint32_t)ts.last / 3u);
            walk_enter(m, load_tri(load_geo(m.geo), cell), wk, cell, ts.last - 3 * cell);
            {   // the exit point of the lane's last (cheap) record, as k_materialise evaluates it: the reference re-seeds from it (:165)
                const RT_G EdgeABC *e = m.etab + ts.last;
                edge_exit_point(tA, tB, tC, e->A, e->B, e->C, lqx, lqy);
            }
            xpx = lqx + sx; xpy = lqy + sy;
            prev_element = cell;
        }
#ifdef RT_TIMING
        const unsigned long long tA_ = rt_tick(xpx);
        unsigned long long tC_ = 0;
        ++wits;
        if (tD) tacc3 += tA_ - tD;
#endif
        double px, py, qx, qy, ell;
        int32_t element = -1;
        const bool from_seed = SPLIT && seed_pending;
        int res = kWalkEmit;
        if (from_seed) {
            // first segment of a seeded piece: the seed itself (k_seed), then march on from its exit point
            const int64_t pi = wave_id * 64 + lane;
            element = spk->s_el[pi];
            px = spk->s_px[pi]; py = spk->s_py[pi]; qx = spk->s_qx[pi]; qy = spk->s_qy[pi]; ell = spk->s_ell[pi];
            const int seq = spk->s_eq[pi];
            if (m.walk_ok && seq >= 0) walk_enter(m, load_tri(load_geo(m.geo), element), wk, element, seq);
            else { wk.T = element; wk.pred = -1; }
            seed_pending = false;
        } else {
        // The reference locates first and tests the boundary second (:122-125); the locate
        // result is unused on both boundary branches, so the order is swapped here.
        if (__builtin_expect(inboundary(m, xpx, xpy, prm.tiny_step), 0)) {  // :125
            if (i == 0) {
                xpx = xpx + sx; xpy = xpy + sy;
                continue;  // :126-129
            }
            break;  // :130-132
        }
        load_next(mh, wk.pred, nr);
#ifdef RT_TIMING
        const unsigned long long tB_ = rt_tick(RT_TIMING == 2 ? nr.e2C : xpx);
        tacc0 += tB_ - tA_;
#endif
        res = walk_step(mh, wk, nr, kk, phi, tA, tB, tC, xpx, xpy, lqx, lqy, qx, qy, ell);
#ifdef RT_TIMING
        tC_ = rt_tick(ell + (double)res);
        tacc1 += tC_ - tB_;
#endif
#ifdef RT_STATS
        if (MODE != kFill && !SPLIT) atomicAdd(&fail_info[2 + res], 1ull);
#endif
        if (res != kWalkGeneric) creep_run = 0;
        if (__builtin_expect(res == kWalkSkip, 0)) {  // :147-150
            xpx = xpx + sx; xpy = xpy + sy;
            // creep on while the reference would keep locating T: each pass stands for one more march
            // iteration that ends in the same `continue`
            while (it < cap && !inboundary(m, xpx, xpy, prm.tiny_step) && walk_still_skip(mh, wk, nr, xpx, xpy)) {
                ++it;
                xpx = xpx + sx; xpy = xpy + sy;
            }
            continue;
        }
        px = lqx; py = lqy; element = wk.T;  // valid when res == kWalkEmit
        if (TOPO && res == kWalkEmit) {
            // per-call statistic: records of exact walk steps in a call with cheap steps (which made the rest)
            const unsigned long long act = __ballot(1);
            if (lane == __ffsll((long long)act) - 1)
                atomicAdd(*(unsigned long long *const RT_K *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                                              offsetof(MarchArgsLayout, fail_info)) + 14,
                          (unsigned long long)__popcll(act));
        }
#ifdef RT_STATS
        if (MODE != kFill && !SPLIT) {
            const unsigned long long any_gen = __ballot(res == kWalkGeneric);
            if (lane == __ffsll((long long)__ballot(1)) - 1) {
                atomicAdd(&fail_info[5], 1ull);                       // wave iterations reaching here
                if (any_gen) atomicAdd(&fail_info[6], 1ull);          // ... with at least one generic lane
            }
        }
#endif
#ifdef RT_TIMING
        if (__ballot(res == kWalkGeneric)) ++wgen;
#endif
        if (__builtin_expect(res == kWalkGeneric, 0)) {
            const DGeo g = load_geo(m.geo);  // scalar loads, here only: the generic step's pointers and grid parameters
            Tri tri;
            element = find_element<WIDEK>(g, xpx, xpy, prm.k, tri);   // :122 and :138-139
            if (element < 0) { st = RT_TRACK_LOCATE_FAILED; break; }  // :140-143
            // Creep: a track that leaves a cell at a very small angle next to a vertex takes hundreds of tiny
            // steps here (BWR-like config 4: 229 in a row through a 7e-8 sliver), each a full locate by one
            // lane.  After kCreepLocal in a row the lane asks the wave for help (cooperative creep below).
            if (element == prev_element) {  // :147-150
                xpx = xpx + sx; xpy = xpy + sy;
                creep_escalate = ++creep_run >= kCreepLocal;
                continue;
            }
            int eq;
            if (!intersections(tri, phi, tA, tB, tC, px, py, qx, qy, eq)) {  // :153
                st = RT_TRACK_UNDEF_INTERSECTION;
                break;
            }
            if (isapprox_v2(px, py, qx, qy)) {  // :156-159
                xpx = xpx + sx; xpy = xpy + sy;
                creep_escalate = ++creep_run >= kCreepLocal;
                continue;
            }
            creep_run = 0;
            ell = norm2(px - qx, py - qy);  // Segment ctor, src/segment.jl:31-33
            if (MODE != kFill && !SPLIT) ++n_generic;  // (added to the call's statistic when the wave ends: 2,039 waves doing
                                                       //  this atomic at the same moment, on their first step, cost the march 5 µs)
            if (MODE != kFill && SPLIT) {
                // per-call statistic (rt_last_stats): records the generic step produced — the walk step made the rest
                // (the control block's address is read from the argument segment here, not held across the loop)
                const unsigned long long act = __ballot(1);
                if (lane == __ffsll((long long)act) - 1)
                    atomicAdd(*(unsigned long long *const RT_K *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                                                  offsetof(MarchArgsLayout, fail_info)) + 15,
                              (unsigned long long)__popcll(act));
            }
            if (m.walk_ok && eq >= 0) walk_enter(m, tri, wk, element, eq);
            else { wk.T = element; wk.pred = -1; wk.last = -1; }
        }
        }
        if (SPLIT && !from_seed && element == tgt_el) {  // (tgt_el = -1: no target)
            asm volatile("" ::: "memory");  // a real, rare branch: a record of the target's cell
            if (qx == spk->s_qx[tgt_pj] && qy == spk->s_qy[tgt_pj] && px == spk->s_px[tgt_pj] && py == spk->s_py[tgt_pj]) {
                matched = true;  // the next piece starts with exactly this segment: stop here
                break;
            }
        }
        if (MODE == kFill) {
            const int64_t o = base + i;
            out.px[o] = px; out.py[o] = py; out.qx[o] = qx; out.qy[o] = qy;
            out.ell[o] = ell;
            out.element[o] = element + 1;
            if (out.fused_volumes) unsafeAtomicAdd((double *)&out.volumes[element], w * ell);  // src/trackgenerator.jl:382
        } else if (MODE == kStage) {
            const int r = i & (kChunkRows - 1);
            if (__builtin_expect(r == 0, 0)) my_chunk = alloc_chunk(i >> kChunkLog2);
            if (TOPO) {
                // One word per record (see DStage): an exact walk step's record is, like a cheap step's, a function of the track
                // line, its exit edge and the previous record — its code; the generic step's record (every track's first one,
                // refusals) keeps its own end points in the side list.
                if (my_chunk >= 0) {
                    if (r == 0) row_el = march_stage_args()->element + stage_slot(my_chunk, 0, lane_o);
                    int32_t word = wk.last + 1;
                    if (__builtin_expect(res != kWalkEmit, 0)) {
                        const RT_K DStage *sk = march_stage_args();
                        int32_t idx = (int32_t)slot;  // a track's first record: its reserved entry (2,039 waves take their first
                                                      // step at the same moment: no atomic there)
                        if (i != 0) {
                            const unsigned long long mm = __ballot(1);
                            const int L = __ffsll((long long)mm) - 1;
                            int32_t b0 = 0;
                            if (lane == L) b0 = atomicAdd((int32_t *)&sk->cursor[2], (int32_t)__popcll(mm));
                            idx = __shfl(b0, L) + (int32_t)__popcll(mm & ((1ull << lane) - 1ull));
                        }
                        if (idx < sk->side_cap) {
                            sk->s_px[idx] = px; sk->s_py[idx] = py; sk->s_qx[idx] = qx; sk->s_qy[idx] = qy;
                            sk->s_el[idx] = element + 1;
                        } else {
                            sk->cursor[3] = 1;  // side list exhausted: the host grows it and re-runs
                        }
                        word = -(idx + 1);
                    }
                    row_el[r * 16] = word;
                    last_word = word;
                }
            } else if (my_chunk >= 0) {
                if (r == 0) {  // per-lane addresses of the chunk's row 0, kept in VGPRs (the staging pointers are
                               // SGPR tuples that do not survive the generic branch unspilled)
                    const int64_t o0 = stage_slot(my_chunk, 0, lane_o);
                    const RT_K DStage *sk = march_stage_args();
                    row_qx = sk->qx + o0; row_qy = sk->qy + o0; row_el = sk->element + o0;
                }
                // A walk-step record starts where the lane's previous record ended (p = previous q, bit for
                // bit) and ℓ = ‖p − q‖ is a function of the two: only q and the cell are staged (20 B instead
                // of 44) and k_compact3 rebuilds p and ℓ.  Records of the generic step / a seed keep their
                // own p and are marked by a negative element.
                const bool derived = res == kWalkEmit && !from_seed;
                row_qx[r * 16] = qx; row_qy[r * 16] = qy;
                row_el[r * 16] = derived ? element + 1 : -(element + 1);
                if (__builtin_expect(!derived, 0)) {
                    const int64_t o = stage_slot(my_chunk, r, lane_o);
                    const RT_K DStage *sk = march_stage_args();
                    sk->px[o] = px; sk->py[o] = py;
                }
            }
            if (FUSE) atomicAdd(&hist[element], w * ell);  // fill_volumes, src/trackgenerator.jl:382 (LDS-private)
        }
#ifdef RT_TIMING
        tD = rt_tick(ell);
        if (!from_seed && res == kWalkEmit) { tacc2 += tD - tC_; ++tn; }
#endif
        if (MODE != kFill) sum_ell += ell;
        lqx = qx; lqy = qy;
        xpx = qx + sx; xpy = qy + sy;  // :165
        prev_element = element;        // :166
        ++i;                           // :168
        if (TOPO) {
            fl = (fl & kFlUsed) | (topo_enter(mh, tt, wk, tA, tB, tC, ts) ? kFlCheap : 0u);
            if (FUSE && (fl & kFlCheap)) topo_tally_enter();
        }
    }
    // ---- cooperative creep: every lane of the wave is out of the march loop here
    unsigned long long need = __ballot(creep_escalate);
    if (!need) break;
    {
        const DGeo g = load_geo(m.geo);
        while (need) {
            const int L = __ffsll((long long)need) - 1;  // the lane whose track creeps
            need &= need - 1;
            const int32_t l_prev = __shfl(prev_element, L, 64);
            const double l_sx = __shfl(sx, L, 64), l_sy = __shfl(sy, L, 64);
            const double l_phi = __shfl(phi, L, 64), l_tA = __shfl(tA, L, 64), l_tB = __shfl(tB, L, 64), l_tC = __shfl(tC, L, 64);
            for (;;) {
                // lane j tests position j of the creep: xp advanced j times, exactly as the serial loop adds
                double cx = __shfl(xpx, L, 64), cy = __shfl(xpy, L, 64);
                const int32_t l_it = __shfl(it, L, 64);
                for (int a = 0; a < 63; ++a)
                    if (a < lane) { cx = cx + l_sx; cy = cy + l_sy; }
                const bool ok = !inboundary(m, cx, cy, prm.tiny_step) &&
                                generic_tiny_step<WIDEK>(g, cx, cy, prm.k, l_prev, l_phi, l_tA, l_tB, l_tC);
                const unsigned long long okm = __ballot(ok);
                int n_ok = okm == ~0ull ? 64 : __ffsll((long long)~okm) - 1;  // leading positions at which the reference steps on
                const int allowed = cap - l_it;                              // it < cap, one count per step
                const int n_adv = n_ok < allowed ? n_ok : (allowed > 0 ? allowed : 0);
                // the lane's new xp is position n_adv (not consumed: the march loop evaluates it), reached by
                // the same additions
                if (lane == L) {
                    for (int a = 0; a < n_adv; ++a) { xpx = xpx + sx; xpy = xpy + sy; }
                    it = l_it + n_adv;
                }
                if (n_adv < 64 || l_it + 64 >= cap) break;  // the creep is over (or the iteration cap is next)
            }
            if (lane == L) { creep_escalate = false; creep_run = 0; }
        }
    }
    }  // for (;;)
#ifdef RT_TIMING
    if (!SPLIT && MODE == kStage && lane == __ffsll((long long)__ballot(1)) - 1) {
        atomicAdd(&fail_info[8], tacc0); atomicAdd(&fail_info[9], tacc1); atomicAdd(&fail_info[10], tacc2);
        atomicAdd(&fail_info[11], tacc3); atomicAdd(&fail_info[12], tn); atomicAdd(&fail_info[13], rt_tick(xpx) - tstart);
        atomicAdd(&fail_info[14], 1ull);
        if (stg.dbg) {
            stg.dbg[4 * wave_id + 0] = rt_tick(xpx) - tstart; stg.dbg[4 * wave_id + 1] = wits;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            stg.dbg[4 * wave_id + 2] = wgen | ((unsigned long long)hwid << 16) | ((unsigned long long)(xcc & 15) << 48);
            stg.dbg[4 * wave_id + 3] = tn | ((unsigned long long)(tstart & 0xffffffffffffull) << 16);
        }
    }
#endif
    if (SPLIT) {
        const int64_t pi = wave_id * 64 + lane;
        const int32_t tgt_k = tgt_el >= 0 ? tgt_pj / 64 - spk->w_base[pw] : 0;  // piece index of the target within its wave
        spk->p_count[pi] = i;
        spk->p_rel[pi] = it;  // iterations of this piece (k_resolve sums them, then reuses the slot)
        spk->p_flags[pi] = (matched ? 1 : 0) | (st << 8) | (tgt_k << 16);
        spk->p_sum[pi] = sum_ell;
    } else if (MODE != kFill) {
        // :171 isapprox(track.ℓ, sum(ℓ.(segments)); rtol) — TOPO: Σℓ is k_materialise's, and so is this check
        if (!TOPO) {
            if (st == RT_TRACK_OK && !isapprox_s(t.ell[u], sum_ell, prm.rtol)) st = RT_TRACK_LENGTH_MISMATCH;
            if (sum_check_is_marginal(t.ell[u], sum_ell, prm.rtol, i)) atomicAdd(march_ctl() + kCtlNearRtol, 1ull);
        }
        // PHASE 1 (k_first): the lane marched with exact steps until a cheap step became possible — behind its first record, as a
        // rule — and goes on in k_cheap (its state: DLean); a lane whose track ended or failed before that ENDS here.  (The exact
        // state, kLnExact, is for a lane that is to go on in k_serve without a cheap step: not used by this kernel's loop.)
        bool fin = true;
        if (PHASE == 1 && st == RT_TRACK_OK && (fl & kFlCheap)) {
            fin = false;
            const bool cheap = (fl & kFlCheap) != 0;
            ln.i[slot] = i; ln.it[slot] = it; ln.word[slot] = last_word;
            ln.fl[slot] = (int32_t)(fl & (kFlCheap | kFlUsed)) | (ts.apos ? kLnApos : 0) | (cheap ? 0 : kLnExact);
            if (cheap) {
                ln.pred[slot] = ts.pred; ln.last[slot] = ts.last; ln.sp[slot] = ts.sp; ln.sn[slot] = ts.sn;
                ln.ttP[slot] = ttP; ln.ttN[slot] = ttN; ln.ttp[slot] = ttp; ln.dprev[slot] = dprev;
            } else {
                ln.lqx[slot] = lqx; ln.lqy[slot] = lqy; ln.prev_el[slot] = prev_element; ln.wk_last[slot] = wk.last;
            }
            const unsigned long long qm = __ballot(!cheap);
            if (!cheap) {  // (k_serve starts behind this kernel: its end publishes the state)
                const int L = __ffsll((long long)qm) - 1;
                int32_t b0 = 0;
                if (lane == L) b0 = atomicAdd((int32_t *)&ln.qctl[0], (int32_t)__popcll(qm));
                ln.queue[__shfl(b0, L) + (int32_t)__popcll(qm & ((1ull << lane) - 1ull))] = (int32_t)slot;
            }
        } else if (PHASE == 1) {
            ln.fl[slot] = kLnFinal;
        }
        if (fin) { counts[u] = i; status[u] = st; }
        if (TOPO && PHASE == 0) { end_cnt = i; end_u = u; }
        if (TOPO) { if (fin) t.cnt_slot[slot] = i; if (FUSE && PHASE != 2) t.w_slot[slot] = w; }  // (k_materialise reads its units' counts and weights in slot order; w: δs of the track's angle, loaded at the start — two dependent loads here were the tail of every wave, the last one's included)
        {
            // What the wave leaves for the call: its records into the sum of its tile of uids (two-phase calls: the scan then needs no
            // pass over the counts to form the tile sums), and the per-call statistics (rt_last_stats) — records the generic step
            // produced, cheap records whose fill_volumes term is k_materialise's — summed over the wave's lanes bit by bit with ballots
            // (values <= kMaxIter < 2^14; some lanes of the batch's last wave are not here).  All three go to the TILE's own 128-B
            // line (DStage::tile_acc): device-scope atomics of many waves on one line are served one after the other, ≈80 ns each —
            // 2,039 waves on the four lines of a dense array of tile sums cost the C3 march 40 µs, two statistics on the control block's
            // line 3 µs.
            auto wave_sum = [&](const int32_t v) -> unsigned long long {
                unsigned long long r = 0;
                for (int b = 0; b < 14; ++b) r += (unsigned long long)__popcll(__ballot((v >> b) & 1)) << b;
                return r;
            };
            const bool first = lane == __ffsll((long long)__ballot(1)) - 1;
            const unsigned long long ng = wave_sum(n_generic), ne = (TOPO && FUSE) ? wave_sum(n_exact_tally) : 0ull;
            RT_G int32_t *acc = TOPO ? march_stage_args()->tile_acc : nullptr;
            if (acc) {
                static_assert(kScanTile == 1024, "tile of the offsets' scan");
                const int32_t tile = (int32_t)(u >> 10);
                const int32_t t0 = __builtin_amdgcn_readfirstlane(tile);
                RT_G int32_t *line = acc + (size_t)t0 * kTileAccStride;
                // (a wave's lanes are 64 consecutive uids — one tile — except where the batch's partial last wave of uids was packed
                //  into the march order)
                const int32_t i_end = fin ? i : 0;  // (a lane that goes on in another kernel is counted where it ends)
                if (__ballot(tile != t0) == 0) {
                    const unsigned long long ws = wave_sum(i_end);
                    if (first && ws) atomicAdd((int32_t *)line, (int32_t)ws);
                } else if (i_end) {
                    atomicAdd((int32_t *)(acc + (size_t)tile * kTileAccStride), (int32_t)i_end);
                }
                if (first && ng) atomicAdd((int32_t *)(line + 1), (int32_t)ng);
                if (first && ne) atomicAdd((int32_t *)(line + 2), (int32_t)ne);
            } else {
                if (first && ng) atomicAdd(&fail_info[15], ng);
                if (first && ne) atomicAdd(march_ctl() + kCtlExactTally, ne);
            }
        }
        if (st != RT_TRACK_OK) {
            atomicAdd(&fail_info[0], 1ull);
            atomicMin(&fail_info[1], (unsigned long long)(u + 1));
        }
    }
    }  // slot < t.n
    if (PHASE != 2) break;
    }  // for (;;): k_serve's claims
    if (FUSE) {
        __syncthreads();
        if (TOPO && PHASE == 0) {
            const RT_K DStage *sk = march_stage_args();
            if (sk->cq) {
                // ---- records in completion order: the workgroup's tracks, in march-slot order, take one span of the result arrays from
                //      the cursor (north_star: "appended via a warp-aggregated atomic cursor" — here aggregated over the workgroup: one
                //      returning atomic per 64 * WAVES tracks, at the workgroup's end, not 2,039 at the same moment); every track's offset
                //      goes to off_slot (k_materialise_lin reads its units by slot) and to tab_off (the per-track table, uid order).
                lds_i32 *own = (lds_i32 *)(march_smem + (size_t)m.n_cells * sizeof(double)) + wib * kMaxChunks;  // (every wave is behind its march: the chunk cache is free)
                int32_t incl = end_cnt;
                for (int o = 1; o < 64; o <<= 1) {
                    const int32_t v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                if (lane == 63) own[0] = incl;
                __syncthreads();
                lds_i32 *w0 = (lds_i32 *)(march_smem + (size_t)m.n_cells * sizeof(double));
                if (threadIdx.x == 0) {
                    unsigned long long tot = 0;
                    for (int w2 = 0; w2 < WAVES; ++w2) tot += (unsigned long long)(uint32_t)w0[w2 * kMaxChunks];
                    const unsigned long long b0 = atomicAdd(march_ctl() + kCtlCq, tot);
                    w0[1] = (int32_t)(uint32_t)b0; w0[2] = (int32_t)(uint32_t)(b0 >> 32);
                }
                __syncthreads();
                int64_t off = (int64_t)(((unsigned long long)(uint32_t)w0[2] << 32) | (unsigned long long)(uint32_t)w0[1]);
                for (int w2 = 0; w2 < WAVES; ++w2)
                    if (w2 < wib) off += (int64_t)(uint32_t)w0[w2 * kMaxChunks];
                off += (int64_t)(incl - end_cnt);
                if (end_u >= 0) {
                    sk->tab_off[end_u] = off;
                    t.off_slot[((int64_t)blockIdx.x * WAVES + wib) * 64 + lane] = off;
                }
                // queue the workgroup ON ITS XCD: every wave's stores (words, side list, counts, status, offsets) drained — they are in
                // this XCD's L2, where the record workgroups of this XCD (and only they: k_materialise_lin<QUEUE>) read them —, then the
                // entry by an agent-scope store.  No release fence: a write-back of the XCD's L2 per ending workgroup cost the chains
                // that were still marching half their speed.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // ("compact_debug" 64, tests: the workgroup does NOT queue itself — the record kernel beside the march must give up
                //  by itself and the host fall back to CSR order)
                if (threadIdx.x == 0 && !(out.dbg & 64)) {
                    unsigned x;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
                    const int xcc = (int)(x & 7u);
                    unsigned long long *ctl = march_ctl();
                    const int32_t k = atomicAdd(reinterpret_cast<int32_t *>(ctl + kCtlCqXcd + xcc), 1);  // this XCD's tail
                    if (k < sk->cq_blocks)
                        __hip_atomic_store(sk->cq + (int64_t)xcc * sk->cq_blocks + k, ((unsigned long long)sk->cq_epoch << 32) | (unsigned long long)blockIdx.x,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tail and the entry first: once every workgroup is counted, every tail is final)
                    atomicAdd(ctl + kCtlCq + 1, 1ull);
                }
            }
        }
        for (int c = threadIdx.x; c < m.n_cells; c += 64 * WAVES) {
            const double v = hist[c];
            if (v != 0.0) unsafeAtomicAdd((double *)&out.volumes[c], v);
        }
    }
}

}  // namespace rt

// ------------------------------------------------------------------- launchers -------------
namespace rtx {

// The instantiations the library carries: whole tracks fused with fill_volumes in four- / six-wave workgroups (exact steps, and
// cheap steps = the two-phase march), pieces fused in four-wave workgroups, and the one-wave kernels for meshes whose LDS copy
// of `volumes` does not fit and for a wide k; builds with -DRT_EXPERIMENTAL add the two-pass march (count / fill).
int launch_march(int mode, int waves, bool split, bool widek, bool topo, unsigned blocks, size_t smem, hipStream_t s, const rt::DMesh &m,
                 const rt::DTracks &t, const rt::DParams &prm, int32_t *counts, int32_t *status, const int64_t *offsets, const rt::DOut &out,
                 const rt::DStage &stg, unsigned long long *fail_info, const rt::DSplit &sp, int phase, const rt::DLean *lean) {
    const rt::DLean ln = lean ? *lean : rt::DLean{};
    auto go = [&]<int MODE, int WAVES, bool SPLIT, bool WIDEK, bool TOPO, int PHASE = 0>() -> int {
        if (smem > 48 * 1024)
            RT_HIP(hipFuncSetAttribute((const void *)rt::k_march<MODE, WAVES, SPLIT, WIDEK, TOPO, PHASE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((rt::k_march<MODE, WAVES, SPLIT, WIDEK, TOPO, PHASE>), dim3(blocks), dim3(64 * WAVES), smem, s, m, t, prm, counts, status,
                           offsets, out, stg, fail_info, sp, ln);
        return RT_SUCCESS;
    };
    if (mode == rt::kStage && phase != 0) {  // the lean plan's k_first (1) / k_serve (2): whole tracks, cheap steps
        if (!lean || !topo || split || widek) { set_error("k_march: phase %d needs the two-phase whole-track march", phase); return RT_ERR_INVALID; }
        if (phase == 1 && waves == 4) return go.template operator()<rt::kStage, 4, false, false, true, 1>();
        if (phase == 1 && waves == 6) return go.template operator()<rt::kStage, 6, false, false, true, 1>();
        if (phase == 2 && waves == 4) return go.template operator()<rt::kStage, 4, false, false, true, 2>();
        if (phase == 2 && waves == 6) return go.template operator()<rt::kStage, 6, false, false, true, 2>();
    }
    if (mode == rt::kStage) {
        if (topo && !split && !widek && waves == 4) return go.template operator()<rt::kStage, 4, false, false, true>();
        if (topo && !split && !widek && waves == 6) return go.template operator()<rt::kStage, 6, false, false, true>();
        if (!topo && !split && !widek && waves == 4) return go.template operator()<rt::kStage, 4, false, false, false>();
        if (!topo && !split && !widek && waves == 6) return go.template operator()<rt::kStage, 6, false, false, false>();
        if (!topo && split && !widek && waves == 4) return go.template operator()<rt::kStage, 4, true, false, false>();
        if (!topo && waves == 1) {
            if (split) return widek ? go.template operator()<rt::kStage, 1, true, true, false>() : go.template operator()<rt::kStage, 1, true, false, false>();
            return widek ? go.template operator()<rt::kStage, 1, false, true, false>() : go.template operator()<rt::kStage, 1, false, false, false>();
        }
    }
#ifdef RT_EXPERIMENTAL
    if (mode == rt::kCount && waves == 1 && !split && !topo)
        return widek ? go.template operator()<rt::kCount, 1, false, true, false>() : go.template operator()<rt::kCount, 1, false, false, false>();
    if (mode == rt::kFill && waves == 1 && !split && !topo)
        return widek ? go.template operator()<rt::kFill, 1, false, true, false>() : go.template operator()<rt::kFill, 1, false, false, false>();
#endif
    set_error("k_march: no instantiation for mode %d, %d waves, split %d, wide k %d, cheap steps %d", mode, waves, (int)split, (int)widek, (int)topo);
    return RT_ERR_INVALID;
}

void launch_seed(bool widek, unsigned blocks, hipStream_t s, const rt::DMesh &m, const rt::DTracks &t, const rt::DParams &prm, const rt::DSplit &sp) {
    if (widek) hipLaunchKernelGGL(rt::k_seed<true>, dim3(blocks), dim3(64), 0, s, m, t, prm, sp);
    else hipLaunchKernelGGL(rt::k_seed<false>, dim3(blocks), dim3(64), 0, s, m, t, prm, sp);
}

void launch_resolve(unsigned blocks, hipStream_t s, const rt::DTracks &t, const rt::DParams &prm, const rt::DSplit &sp, int32_t *counts,
                    int32_t *status, unsigned long long *fail_info) {
    hipLaunchKernelGGL(rt::k_resolve, dim3(blocks), dim3(256), 0, s, t, prm, sp, counts, status, fail_info);
}

}  // namespace rtx
