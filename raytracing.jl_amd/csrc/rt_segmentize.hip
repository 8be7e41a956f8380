// rt_segmentize.hip — HIP kernels + C ABI of the MI355X-native segmentize! path (gfx950).
//
// Replaces, behind include/rt_segmentize.h, the reference's
//   segmentize!            src/trackgenerator.jl:357-369
//   _segmentize_track!     src/track.jl:106-178
//   find_element & co.     src/mesh.jl:91-176
//   intersections & co.    src/intersection.jl:11-159, src/segment.jl:31-44
//   fill_volumes           src/trackgenerator.jl:371-386
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (csrc/Makefile).
// There is no CPU fallback in this library: without a GPU every compute entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_segmentize.h"
#include "rt_device.hpp"
#include "rt_mesh_prep.hpp"

namespace rthost {
thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {  // shared with rt_host.cpp
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace rthost
using rthost::g_last_error;
using rthost::set_error;

namespace {

#define RT_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    // owns its allocation: a handle's buffers are released when the handle is deleted, whether or not free_tracks /
    // free_mesh list them (a forgotten member leaked 1 GB per C5 handle in round 3)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

// ------------------------------------------------------------------- kernels -------------
namespace rt {

struct DOut {
    RT_G double *px, *py, *qx, *qy, *ell;
    RT_G int32_t *element;
    RT_G double *volumes;  // accumulated δs·ℓ per cell (un-normalised)
    const RT_G double *delta_s;
    int32_t fused_volumes;  // 1: accumulate δs·ℓ with global f64 atomics inside the fill march
    int32_t dbg;            // development (option "compact_debug"): 1 the compaction stores nothing, 2 it loads nothing
    int64_t cap;            // records the six arrays can hold: the single-pass compaction does not write beyond (the host
                            // sizes them from an estimate, sees the true total afterwards, and compacts again if it was short)
};

// Staging of the single-pass march: a pool of chunks, each kChunkRows rows of 64 lanes, per
// output array.  Lane l of a wave writes its i-th segment to row i of the wave's chunk list,
// column l — lanes of a wave emit in near lockstep, so each store instruction writes whole
// 512-B rows instead of 64 scattered 8-B pieces.  Chunks are handed out from one atomic
// cursor, once per wave and chunk (wave-aggregated), and recorded in `ctab` / `cowner` for the compaction.
#ifndef RT_CHUNK_LOG2
#define RT_CHUNK_LOG2 5  // 32 rows per chunk measured best (8: -18 %, 16: -6 % vs 32 at C3)
#endif
constexpr int kChunkLog2 = RT_CHUNK_LOG2;
constexpr int kChunkRows = 1 << kChunkLog2;
constexpr int kMaxChunks = kMaxIter / kChunkRows + 1;  // per wave

constexpr int32_t kWordExactTally = 1 << 30;  // staged word of a cheap record whose fill_volumes term k_materialise adds (see DStage)
struct DStage {
    RT_G double *qx, *qy;   // exit point of every record
    RT_G double *px, *py;   // entry point, only for records whose element is staged negative (see k_march)
    RT_G int32_t *element;
    RT_G int32_t *ctab;     // [n_waves][kMaxChunks] chunk ids
    RT_G int32_t *cowner;   // [pool_chunks] wave * kMaxChunks + j of the chunk's owner
    RT_G int32_t *cursor;   // [0] chunks handed out, [1] overflow flag
    int32_t pool_chunks;
    int32_t static0;        // 1: chunk w is reserved as the first chunk of march wave w (whole-track march; cursor starts at n_waves)
    // k_march<TOPO> stages ONE word per record in `element`: 3·cell + exit edge + 1 (the record is a function of the track's
    // line, that edge and the previous record: k_materialise computes it), or -(index + 1) of an entry of the side list below
    // for a record that keeps its own end points (the generic step's: every track's first one, refusals).  Bit 30 of a positive
    // word: the march has NOT added the record to fill_volumes (a shallow crossing: its chord from the vertices' distances
    // would be too inexact) — k_materialise adds δs·ℓ from the record's own length.  Entries
    // [0, side_static) are reserved — entry `march slot` for the track's first record —, the rest is handed out from
    // cursor[2]; cursor[3] flags an overflow (the host grows the list and re-runs, as for the pool).
    RT_G double *s_px, *s_py, *s_qx, *s_qy;
    RT_G int32_t *s_el;     // cell + 1
    int32_t side_cap, side_static;
#ifdef RT_TIMING
    unsigned long long *dbg;  // [n_waves][4] development: cycles, wave iterations, generic iterations, emits of the first lane
#endif
};

// Slot of (row, lane) inside a chunk: quarter-major — the 16 lanes of a quarter-wave keep their 32 rows in
// one contiguous 4-KB block, so the compaction workgroup of that quarter reads whole lines that nobody
// else needs; a march store (64 lanes, one row) still writes four full 128-B lines.
__device__ __forceinline__ int64_t stage_slot(int32_t chunk, int row, int lane) {
    return (((int64_t)chunk * 4 + (lane >> 4)) * kChunkRows + row) * 16 + (lane & 15);
}

enum MarchMode { kCount = 0, kFill = 1, kStage = 2 };
// The control block of a call (device, copied to pinned host memory by the scan's last block): words 0..15 failure summary /
// statistics, 16 total segments, 18..19 pool cursor + overflow flag, 20 ticket of the scan's "last block" step, 21 tracks
// that reached MAX_ITER segments in split mode, 22..26 development statistics (RT_STATS), and
constexpr int kCtlWords = 64;
constexpr int kCtlRefusal = 32;   // 32..40: cheap-step refusals by certificate term (order of topo_certified)
constexpr int kCtlRestarts = 41;  // tracks marched again with exact steps after cheap steps (their fused volumes were counted twice)
constexpr int kCtlNearRtol = 42;  // tracks whose Σℓ check (src/track.jl:171) sits within summation-order noise of its threshold
constexpr int kCtlFinishTicket = 43;  // k_finish: its "last block" ticket
constexpr int kCtlDeferred = 27;      // k_finish: tracks whose exact Σℓ it could not form (their records lie beyond the arrays' capacity)
// generic tiny steps in a row a lane takes on its own before the wave helps (k_march; 2 and 4 measured +30 % at
// C3 — a lane that escalates waits for the rest of its wave — 8..32 equal)
constexpr int kCreepLocal = 16;

// ---- track splitting ("pieces") ------------------------------------------------------------
// The march of a track is a serial dependent chain; a batch lasts as long as its longest track.
// In split mode a track is cut into P pieces by arclength.  Piece k >= 1 starts from a SEED: the
// segment (cell, p, q) of the cell that contains the point M_k of the track, computed with the
// generic locate + intersections (k_seed).  Every piece marches like a track, but stops — before
// emitting — at the segment that equals the next live seed bit for bit (cell id, p and q): from
// there on the reference's state (xp = q + tiny·d, prev_element = cell) is exactly the state the
// next piece started from, so the concatenation of the pieces IS the reference's segment list.
// A piece that never meets the next seed simply marches on to the end of the track, and
// k_resolve drops the pieces it overran: a miss costs time, never correctness.
struct DSplit {
    const RT_G int32_t *vorder;   // [n_vwaves] dispatch order (longest pieces first) -> canonical virtual wave
    const RT_G int32_t *vw_wave;  // [n_vwaves] canonical virtual wave -> wave of 64 consecutive uids
    const RT_G int32_t *vw_k;     // [n_vwaves] piece index within the wave
    const RT_G int32_t *w_base;   // [n_waves] first canonical virtual wave of a wave
    const RT_G int32_t *w_P;      // [n_waves] pieces per track of the wave
    RT_G int32_t *s_el, *s_eq;    // seeds, per piece (canonical virtual wave * 64 + lane); s_el < 0: no seed
    RT_G double *s_px, *s_py, *s_qx, *s_qy, *s_ell;
    RT_G int32_t *p_count, *p_flags;  // per piece: segments emitted; bit0 matched the next seed, bits 8..15 status, bits 16.. target piece
    RT_G double *p_sum;               // per piece: sum of its segment lengths, in march order
    RT_G int32_t *p_valid, *p_rel;    // after k_resolve: records kept from the piece / their offset inside the track's run
    int32_t n_vwaves;
};

// The Σℓ check `isapprox(track.ℓ, sum(ℓ.(segments)); rtol)` (src/track.jl:171) is decided here (and in the CPU checker) with a
// left-to-right sum; Julia's `sum` reassociates (pairwise blocks, @simd lanes), so its Σℓ can differ by a few ulp·n.  A
// track whose |ℓ − Σℓ| lies within 64·ulp·n·max(ℓ, Σℓ) of the threshold rtol·max(ℓ, Σℓ) could get the other status there:
// such tracks are counted (rt_last_stats) so that a caller knows when this cannot be pinned.
__device__ __forceinline__ bool sum_check_is_marginal(double ell, double sum, double rtol, int n, double band = 64.0) {
    const double big = fabs(ell) > fabs(sum) ? fabs(ell) : fabs(sum);
    return fabs(fabs(ell - sum) - rtol * big) <= band * 1.1102230246251565e-16 * (double)(n > 1 ? n : 1) * big;
}

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
    if (P == 0) return;  // hybrid mode: this wave of tracks was marched whole by the non-split kernel
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
// ---- every track's first record, ahead of the march (k_first) -------------------------------------------------------
// A track's first record is the one step of the march that has no prediction: start band (src/track.jl:125-129), then the
// literal locate (src/mesh.jl:103-146) and intersections (src/intersection.jl:34-119) — a chain of ≈15 dependent gathers
// (bucket -> node range -> nearest node -> its cells' entries one after the other -> ...) that every lane of the march
// walked alone, two waves per SIMD, right after the compaction had flushed the caches: ≈60 of the march's 178 µs at C3.
// k_first does that step for all tracks before the march with EIGHT lanes per track: the bucket's node range is scanned
// eight nodes at a time, the nearest node's incident cells are tested eight at a time (first hit in stored order wins, as in
// the reference), the three edges are intersected on three lanes — five dependent round trips instead of fifteen, on 16 k
// waves instead of 2 k.  It only handles the plain case (nearest node found within the bucket's 3x3 block, one of its cells
// contains the point, a regular segment comes out); anything else leaves the slot "not done" and the march takes that
// track from its start as before.  Same device functions, same operation order: the record is bit-identical.
// The record goes to row 0 of the wave's reserved first chunk; what the march needs to go on (iteration count, exit point,
// ℓ, walk state) goes to a per-slot SoA.
struct DFirst {
    RT_G int32_t *it;    // [n_slots] 0: not done; else the iterations counted up to and including the first emit
    RT_G int32_t *T;     // cell of the record
    RT_G int32_t *pred;  // walk record predicted next (-1: none)
    RT_G double *v;      // [10][n_slots]: qx, qy, ell, ax, ay, bx, by, cx, cy, dT
    int64_t n_slots;
};

__global__ __launch_bounds__(256) void k_first(DMesh m, DTracks t, DParams prm, DStage stg, DFirst f) {
    const int lane = threadIdx.x & 63, sub = threadIdx.x & 7, gbase = lane & ~7;
    const int64_t slot_raw = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool have = slot_raw < t.n;
    const int64_t slot = have ? slot_raw : t.n - 1;  // (idle groups shadow the last track and write nothing)
    const int32_t u = t.perm[slot];
    const DGeo g = load_geo(m.geo);
    const double tA = t.A[u], tB = t.B[u], tC = t.C[u], phi = t.phi[u];
    const double sx = prm.tiny_step * t.cs[u], sy = prm.tiny_step * t.sn[u];  // advance_step, src/point.jl:43
    double xpx = t.px[u] + sx, xpy = t.py[u] + sy;                               // src/track.jl:114
    const int32_t cap = (int32_t)(prm.iter_cap < 0x7fffffff ? prm.iter_cap : 0x7fffffff);
    int32_t it = 0;
    bool ok = true;
    while (inboundary(m, xpx, xpy, prm.tiny_step)) {  // start band, :125-129
        if (++it > cap) { ok = false; break; }
        xpx = xpx + sx; xpy = xpy + sy;
    }
    if (++it > cap) ok = false;  // the iteration that emits
    // ---- nn(kdtree, xp): the bucket's 3x3 block, eight nodes at a time; (squared distance, id) is a total order
    int ix, iy;
    bucket_of(g, xpx, xpy, ix, iy);
    const int b = iy * g.gnx + ix;
    const int32_t s0 = g.c3start[b], s1 = g.c3start[b + 1];
    double best = __builtin_huge_val();
    int32_t best_id = 0x7fffffff;
    for (int32_t q = s0 + sub; q < s1; q += 8) {
        const int32_t id = g.c3node[q];
        const double dx = xpx - g.c3x[q], dy = xpy - g.c3y[q];
        const double d2 = dx * dx + dy * dy;
        if (node_before(d2, id, best, best_id)) { best = d2; best_id = id; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int32_t oi = __shfl_xor(best_id, o, 64);
        if (node_before(ob, oi, best, best_id)) { best = ob; best_id = oi; }
    }
    {   // as nearest_node: accepted only if nothing outside the block can be nearer (else the ring search: left to the march)
        const double lb = ring_bound(g, xpx, xpy, ix, iy, 1) - 1e-9 * g.gh;
        ok = ok && best_id != 0x7fffffff && (lb == __builtin_huge_val() || (lb > 0.0 && best < lb * lb));
    }
    const int32_t nn = best_id != 0x7fffffff ? best_id : 0;
    // ---- the cells of node_cells[nn] in stored order, first hit wins (src/mesh.jl:110-118), eight at a time
    const int32_t f0 = g.ncp[nn], f1 = g.ncp[nn + 1];
    Tri tri{};
    int32_t element = -1;
    for (int32_t base = f0; base < f1 && element < 0; base += 8) {
        const int32_t q = base + sub;
        Tri c{};
        int32_t cell = -1;
        bool hit = false;
        if (q < f1) {
            const RT_G FanEntry *e = g.fan + q;
            c.x1 = e->x1; c.y1 = e->y1; c.x2 = e->x2; c.y2 = e->y2; c.x3 = e->x3; c.y3 = e->y3;
            c.adj[0] = e->adj[0]; c.adj[1] = e->adj[1]; c.adj[2] = e->adj[2];
            cell = e->cell;
            hit = point_in_triangle(c, xpx, xpy);
        }
        const unsigned hits = (unsigned)((__ballot(hit) >> gbase) & 0xffull);
        if (hits) {
            const int src = gbase + __builtin_ctz(hits);
            tri.x1 = __shfl(c.x1, src, 64); tri.y1 = __shfl(c.y1, src, 64); tri.x2 = __shfl(c.x2, src, 64);
            tri.y2 = __shfl(c.y2, src, 64); tri.x3 = __shfl(c.x3, src, 64); tri.y3 = __shfl(c.y3, src, 64);
            tri.adj[0] = __shfl(c.adj[0], src, 64); tri.adj[1] = __shfl(c.adj[1], src, 64); tri.adj[2] = __shfl(c.adj[2], src, 64);
            element = __shfl(cell, src, 64);
        }
    }
    ok = ok && element >= 0;  // (not found among the nearest node's cells: the knn fallback, left to the march)
    // ---- intersections(mesh, element, track): one edge per lane, then the reference's case analysis on every lane
    double px = 0, py = 0, qx = 0, qy = 0;
    int eq = -1;
    {
        const int e3 = sub < 3 ? sub : 0;
        const double ax = e3 == 0 ? tri.x1 : (e3 == 1 ? tri.x2 : tri.x3), ay = e3 == 0 ? tri.y1 : (e3 == 1 ? tri.y2 : tri.y3);
        const double bx = e3 == 0 ? tri.x2 : (e3 == 1 ? tri.x3 : tri.x1), by = e3 == 0 ? tri.y2 : (e3 == 1 ? tri.y3 : tri.y1);
        double ex = 0, ey = 0;
        const int h = ok ? edge_hit(tA, tB, tC, ax, ay, bx, by, ex, ey) : 0;
        const int h0 = __shfl(h, gbase, 64), h1 = __shfl(h, gbase + 1, 64), h2 = __shfl(h, gbase + 2, 64);
        const double ex0 = __shfl(ex, gbase, 64), ey0 = __shfl(ey, gbase, 64), ex1 = __shfl(ex, gbase + 1, 64), ey1 = __shfl(ey, gbase + 1, 64);
        const double ex2 = __shfl(ex, gbase + 2, 64), ey2 = __shfl(ey, gbase + 2, 64);
        ok = ok && intersections_combine(h0, ex0, ey0, h1, ex1, ey1, h2, ex2, ey2, phi, px, py, qx, qy, eq);  // :153
    }
    ok = ok && !isapprox_v2(px, py, qx, qy);  // :156-159 (a vertex touch steps on: left to the march)
    const double ell = norm2(px - qx, py - qy);  // Segment ctor, src/segment.jl:31-33
    if (!have || sub != 0) return;
    if ((slot & 63) == 0) {  // the wave's reserved first chunk (chunk w for march wave w), as the march would record it
        const int64_t w = slot >> 6;
        stg.ctab[w * kMaxChunks] = (int32_t)w;
        stg.cowner[w] = (int32_t)(w * kMaxChunks);
    }
    if (!ok) { f.it[slot] = 0; return; }
    Walk wk;
    if (m.walk_ok && eq >= 0) walk_enter(m, tri, wk, element, eq);
    else { wk.T = element; wk.pred = -1; wk.ax = wk.ay = wk.bx = wk.by = wk.cx = wk.cy = 0.0; wk.dT = 1.0; }
    const int64_t o = stage_slot((int32_t)(slot >> 6), 0, (int)(slot & 63));
    stg.qx[o] = qx; stg.qy[o] = qy; stg.element[o] = -(element + 1);  // a record of the generic step keeps its own p
    stg.px[o] = px; stg.py[o] = py;
    f.it[slot] = it; f.T[slot] = element; f.pred[slot] = wk.pred;
    RT_G double *v = f.v + slot;
    const int64_t S = f.n_slots;
    v[0] = qx; v[S] = qy; v[2 * S] = ell; v[3 * S] = wk.ax; v[4 * S] = wk.ay; v[5 * S] = wk.bx; v[6 * S] = wk.by;
    v[7 * S] = wk.cx; v[8 * S] = wk.cy; v[9 * S] = wk.dT;
}

// k_march's staging pointers are needed once per 32 iterations (chunk hand-out, row addresses) and on rare
// records: they are read from the kernel-argument segment with scalar loads where they are used instead of
// living in 19 SGPRs across the whole loop (which the kernel was spilling to VGPR lanes and reloading on
// its hot path).  The struct mirrors k_march's parameter list.
struct MarchArgsLayout {
    DMesh m; DTracks t; DParams prm; int32_t *counts; int32_t *status; const int64_t *offsets; DOut out; DStage stg;
    unsigned long long *fail_info; DSplit sp; DFirst fst;
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
// LDSREC (experiment, option "lds_records"): the workgroup first copies ALL walk records of the mesh into LDS and the
// lanes fetch their next record from there instead of from L2 — only meshes of a few hundred cells fit (80 B per record,
// three per cell); see DESIGN.md §4 for what it measures.
// TOPO (whole tracks, staged): the walk step split into a DECISION that needs no point at all (rt_device.hpp, topo_geo /
// topo_certified: which cell the reference emits next, through which edges — from the signed distances of the cell's
// vertices to the track line) and the ARITHMETIC of the record (exit point on the predicted edge with the reference's
// formula, ℓ), which no longer feeds the next iteration: a lane's dependent chain per record is one 32-B record fetch and
// a dozen instructions, and the next record's fetch is in flight while the certificates and the record are evaluated.
// The exact step (walk_step / generic) runs only for the lanes whose cheap step refused.
template <int MODE, int WAVES, bool SPLIT, bool WIDEK = false, bool LDSREC = false, bool TOPO = false>
#ifndef RT_TOPO_OCC
#define RT_TOPO_OCC 0
#endif
__global__ __launch_bounds__(64 * WAVES, (SPLIT && MODE == kStage && WAVES > 1) ? 3 : (TOPO ? RT_TOPO_OCC : 0)) void k_march(DMesh m, DTracks t, DParams prm, int32_t *__restrict__ counts,
                                                      int32_t *__restrict__ status,
                                                      const int64_t *__restrict__ offsets, DOut out, DStage stg,
                                                      unsigned long long *__restrict__ fail_info, DSplit sp, DFirst fst) {
    // The split plan's tables are used at the start and the end of a piece and when a record of the target's cell comes
    // up — never in the steady march: they are read from the argument segment where they are used (as `stg` is), so
    // that their 17 pointers do not occupy scalar registers across the loop.
    const RT_K DSplit *spk = (const RT_K DSplit *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, sp));
    (void)sp; (void)fst;
    static_assert(!TOPO || (MODE == kStage && !SPLIT && !LDSREC), "cheap steps: staged whole tracks only");
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
        if (LDSREC) {
            typedef __attribute__((address_space(3))) double lds_f64w;
            lds_f64w *dst = (lds_f64w *)(march_smem + (((size_t)m.n_cells * sizeof(double) + (size_t)WAVES * kMaxChunks * sizeof(int32_t) + 15) & ~(size_t)15));
            const RT_G double *src = (const RT_G double *)m.wrec;  // (the header word travels as a bit pattern)
            for (int c = threadIdx.x; c < 3 * m.n_cells * 10; c += 64 * WAVES) dst[c] = src[c];
        }
        __syncthreads();
    }
    typedef __attribute__((address_space(3))) const WalkRec lds_rec_t;
    lds_rec_t *lrec = (lds_rec_t *)(march_smem + (((size_t)m.n_cells * sizeof(double) + (size_t)WAVES * kMaxChunks * sizeof(int32_t) + 15) & ~(size_t)15));
    int64_t wave_id = (int64_t)blockIdx.x * WAVES + wib;  // indexes the wave's chunk table (ctab)
    int64_t slot = wave_id * 64 + lane;
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
    const double tA = t.A[u], tB = t.B[u], tC = t.C[u];
    const double phi = t.phi[u];
    // advance_step (src/point.jl:43): x + step * Point2D(cos ϕ, sin ϕ)
    const double sx = prm.tiny_step * t.cs[u];
    const double sy = prm.tiny_step * t.sn[u];
    double xpx = t.px[u] + sx, xpy = t.py[u] + sy;  // src/track.jl:114
    int64_t base = 0;
    double w = 0.0;
    if (MODE == kFill) base = offsets[u];
    if (MODE == kFill || FUSE) w = out.delta_s[t.azim[u] - 1];
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
    wk.T = -1; wk.pred = -1;
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
    TopoTrack tt = topo_track(TOPO && m.walk_ok, m.d_vertex, prm.topo_tiny_max, prm.topo_rmax, prm.topo_end_err, prm.tiny_step, t.cs[u], t.sn[u]);
    TopoState ts;
    ts.pred = -1; ts.last = 0; ts.sa = ts.sb = 0.0;
    // kFlCheap: the lane takes cheap steps; kFlUsed: it has taken some (`it` is then an upper bound of the reference's
    // iterations); kFlMat: the exact step's state has to be rebuilt from `ts.last`; kFlWait: nothing to do until the wave
    // has no cheap lane left (an uncertified last step, a finished track); kFlDone / kFlRestart: see below
    constexpr uint32_t kFlCheap = 1, kFlUsed = 2, kFlMat = 4, kFlWait = 8, kFlDone = 16, kFlRestart = 32;
    uint32_t fl = 0;
    // positions along the track line, t(x, y) = B·x − A·y ((B, −A) is the line's direction; general_form normalises the whole
    // (A, B, C), src/intersection.jl:11-18, so t is scaled by ‖(A, B)‖), of the end points of the lane's entry edge (as ts.sa /
    // ts.sb) and of its last exit point: the chord a cheap record adds to fill_volumes is |Δt| / ‖(A, B)‖ — the scale rides in wq
    double tta = 0.0, ttb = 0.0, ttp = 0.0;
    const double nab = (TOPO && FUSE) ? sqrt(tA * tA + tB * tB) : 1.0;
    const double wq = (TOPO && FUSE) ? w / nab : 0.0;
    const double tau_s = (TOPO && FUSE) ? prm.tally_tau * nab : 0.0;  // (s is scaled by ‖(A, B)‖ as t is)
    bool pin = false;  // the lane's last exit point came from a shallow crossing: the next chord starts there
    auto topo_tally_enter = [&]() {
        tta = __builtin_fma(tB, wk.ax, -(tA * wk.ay)); ttb = __builtin_fma(tB, wk.bx, -(tA * wk.by));
        ttp = __builtin_fma(tB, lqx, -(tA * lqy));
        pin = false;  // (an exact step's exit point)
    };
    int32_t n_cheap_it = 0, n_cheap_ref = 0;  // wave-uniform: cheap iterations of this wave, and those in which a lane was refused
    int32_t last_word = 0;  // staging word of the lane's last record
    // (the cheap loop stores without a branch: should the pool run out before a lane's first chunk — the attempt is void
    //  then and the host re-runs it — its row pointers must still be addresses inside the pool)
    if (TOPO) row_el = stg.element + lane;
    const RT_G TopoRec *trec_v = m.trec;
    const RT_G EdgeABC *etab_v = m.etab;
    if (TOPO) asm volatile("" : "+v"(trec_v), "+v"(etab_v));
    // The track's first record may have been made by k_first (whole tracks, staged, reserved first chunks): the march then
    // starts behind it — iteration count, exit point, Σℓ, walk state and the staging row pointers as its own first
    // iteration would have left them.  The state is read through the argument segment (nothing of it lives across the loop).
    if (MODE == kStage && !SPLIT && !TOPO) {
        const RT_K DFirst *fk = (const RT_K DFirst *)((const RT_K char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MarchArgsLayout, fst));
        const RT_G int32_t *f_it = fk->it;
        if (f_it != nullptr) {
            const int32_t fit = f_it[slot];
            if (fit > 0) {
                const int64_t S = fk->n_slots;
                const RT_G double *v = fk->v + slot;
                it = fit; i = 1; n_generic = 1;
                lqx = v[0]; lqy = v[S];
                const double ell0 = v[2 * S];
                sum_ell = ell0;
                wk.ax = v[3 * S]; wk.ay = v[4 * S]; wk.bx = v[5 * S]; wk.by = v[6 * S]; wk.cx = v[7 * S]; wk.cy = v[8 * S]; wk.dT = v[9 * S];
                wk.T = fk->T[slot]; wk.pred = fk->pred[slot];
                prev_element = wk.T;
                xpx = lqx + sx; xpy = lqy + sy;  // :165
                my_chunk = (int32_t)wave_id;
                const int64_t o0 = stage_slot(my_chunk, 0, lane);
                const RT_K DStage *sk = march_stage_args();
                row_qx = sk->qx + o0; row_qy = sk->qy + o0; row_el = sk->element + o0;
                if (TOPO) last_word = -(wk.T + 1);
                if (FUSE) atomicAdd(&hist[wk.T], w * ell0);  // fill_volumes, src/trackgenerator.jl:382 (LDS-private)
                if (TOPO) fl = topo_enter(mh, tt, wk, tA, tB, tC, ts) ? kFlCheap : 0u;
            }
        }
    }
    // Start band (:125-129 with no segment yet): step by tiny_step until xp leaves the boundary
    // band.  Run as its own loop so that the 64 lanes of the wave, whose bands differ in length
    // (≈1/sin ϕ or 1/|cos ϕ| steps), reach their first locate together.
    // (i == 0: a lane that starts behind a first record of k_first is past its start band — its xp may lie in the END band)
    while (i == 0 && !(SPLIT && (seed_pending || piece_dead)) && st == RT_TRACK_OK && inboundary(m, xpx, xpy, prm.tiny_step)) {
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
                    // a wave's first chunk is chunk `wave_id` when the host reserved one per wave (the cursor then starts
                    // behind them): every wave allocates at the same moment, on its first record — 2,039 atomics on one word
                    if (!SPLIT && jL == 0 && sk->static0) c = (int32_t)wave_id;
                    else c = atomicAdd((int32_t *)&cursor[0], 1);
                    if (c >= sk->pool_chunks) { c = -2; cursor[1] = 1; }  // pool exhausted: host grows it and re-runs
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
    // A lane whose track creeps (see below) for more than kCreepLocal tiny steps leaves the march loop and
    // waits for the wave: once every lane is out, all 64 lanes test 64 consecutive creep positions of that
    // track at a time (cooperative creep), then the lane marches on.
    bool creep_escalate = false;
    int creep_run = 0;  // generic tiny steps in a row
    for (;;) {
    while (!(SPLIT && piece_dead) && st == RT_TRACK_OK && i < kMaxIter && !creep_escalate) {  // :119
        if (TOPO) {
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
                    bool inexact = false;
                    if (FUSE) {
                        // fill_volumes (src/trackgenerator.jl:382) for this record: the line meets the exit edge (p, q) — end points on
                        // opposite sides, |s_p − s_q| >= the record's k2 — at t = (s_p·t_q − s_q·t_p) / (s_p − s_q)
                        const bool same = rec_same(c_hdr);
                        const double t0 = same ? tta : ttb, t1 = same ? ttb : tta;
                        const double t2 = __builtin_fma(tB, c_x2, -(tA * c_y2));
                        const double sp = g.exit1 ? g.s1 : g.s2, sq = g.exit1 ? g.s2 : g.s0;
                        const double tp = g.exit1 ? t1 : t2, tq = g.exit1 ? t2 : t0;
                        const double den = sp - sq;
                        double rc = __builtin_amdgcn_rcp(den);
                        rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                        rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
                        const double tx = (sp * tq - sq * tp) * rc;
                        // (a chord one of whose ends is a shallow crossing is left to k_materialise: rt_mesh_prep.hpp, tally_tau)
                        const bool shallow = !(fabs(den) >= tau_s);
                        inexact = pin || shallow;
                        atomicAdd(&hist[g.cell], (commit && !inexact) ? wq * fabs(tx - ttp) : 0.0);  // (LDS-private; a lane that decided nothing adds 0)
                        if (commit) { ttp = tx; tta = tp; ttb = tq; pin = shallow; }
                    }
                    if (commit) {
                        ++i;
                        it += kub;
                        const int r = topo_commit(tt, ts, g);
                        fl |= kFlUsed;
                        if (r == kTopoEnd) fl = (fl & ~kFlCheap) | kFlDone | kFlWait;  // on the border, within tiny_step: :130-132
                        else if (ts.pred < 0) fl = (fl & ~kFlCheap) | kFlMat | kFlWait;
                        else if (i >= kMaxIter) fl = (fl & ~kFlCheap) | kFlWait;
                    } else if (cheap) {
                        fl = (fl & ~kFlCheap) | (ok ? kFlRestart : kFlMat);  // refused: the exact step decides this record
                        // per-call statistic (rt_last_stats): which certificate term refused — a cold branch (every refusal
                        // costs its wave an exact step anyway); one atomic per term and wave
                        const uint32_t bad = ok ? 0u : topo_refusal_terms(tt, ts, g, c_hdr, c_c01, c_c23, kk);
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
                        if (my_chunk >= 0) row_el = march_stage_args()->element + stage_slot(my_chunk, 0, lane);
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
                if (march_prm_args()->topo_force) {  // option "topo" = 2: every record that carries a cheap certificate uses it
                    n_cheap_ref = 0; n_cheap_it = 0;
                } else {
                    tt.on = false;
                    if (fl & kFlCheap) fl = (fl & ~kFlCheap) | kFlMat;
                }
            }
            const bool any_cheap = __ballot((fl & kFlCheap) != 0) != 0;
            if ((fl & kFlDone) || i >= kMaxIter) break;
            if ((fl & kFlCheap) || ((fl & kFlWait) && any_cheap)) continue;  // (an uncertified last step waits until no lane is cheap)
            if (__builtin_expect((fl & kFlRestart) || ((fl & kFlUsed) && it >= cap), 0)) {
                // the bound reached the iteration cap: this track is marched again from its start with exact steps only
                asm volatile("" ::: "memory");
                // its records have already been added to the fused volumes: the host recomputes them from the records
                atomicAdd(march_ctl() + kCtlRestarts, 1ull);
                tt.on = false; fl = 0; n_generic = 0;
                i = 0; it = 0; prev_element = -1; wk.T = -1; wk.pred = -1; creep_run = 0; my_chunk = -1; sum_ell = 0.0;
                xpx = t.px[u] + sx; xpy = t.py[u] + sy;
                continue;
            }
        }
        if (++it > cap) { if (TOPO && (fl & kFlUsed)) continue; st = RT_TRACK_ITER_CAP; break; }
        if (TOPO && __builtin_expect((fl & kFlMat) != 0, 0)) {
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
        if (LDSREC) {
            lds_rec_t *R = lrec + (wk.pred >= 0 ? wk.pred : 0);
            nr.hdr = R->hdr; nr.dT = R->dT; nr.x2 = R->x2; nr.y2 = R->y2;
            nr.e1A = R->e1A; nr.e1B = R->e1B; nr.e1C = R->e1C; nr.e2A = R->e2A; nr.e2B = R->e2B; nr.e2C = R->e2C;
        } else {
            load_next(mh, wk.pred, nr);
        }
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
            else { wk.T = element; wk.pred = -1; }
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
                    if (r == 0) row_el = march_stage_args()->element + stage_slot(my_chunk, 0, lane);
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
                    const int64_t o0 = stage_slot(my_chunk, 0, lane);
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
                    const int64_t o = stage_slot(my_chunk, r, lane);
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
        counts[u] = i;
        status[u] = st;
        if (TOPO) t.cnt_slot[slot] = i;  // (k_materialise reads its units' counts in slot order)
        {
            // per-call statistic (rt_last_stats): records the generic step produced, summed over the wave's active lanes
            // bit by bit with ballots (n_generic <= kMaxIter < 2^14)
            unsigned long long ng = 0;
            for (int b = 0; b < 14; ++b) ng += (unsigned long long)__popcll(__ballot((n_generic >> b) & 1)) << b;
            if (lane == __ffsll((long long)__ballot(1)) - 1 && ng) atomicAdd(&fail_info[15], ng);
        }
        if (st != RT_TRACK_OK) {
            atomicAdd(&fail_info[0], 1ull);
            atomicMin(&fail_info[1], (unsigned long long)(u + 1));
        }
    }
    }  // slot < t.n
    if (FUSE) {
        __syncthreads();
        for (int c = threadIdx.x; c < m.n_cells; c += 64 * WAVES) {
            const double v = hist[c];
            if (v != 0.0) unsafeAtomicAdd((double *)&out.volumes[c], v);
        }
    }
}

// Lean staging -> compact CSR records, all six arrays in one pass.  One 4-wave workgroup per
// (march wave, quarter of its 64 consecutive tracks): wave k moves chunk 4 s + k of the quarter's 16
// tracks (rows 32 (4 s + k) ..), s = 0, 1, ... — almost always s = 0 only, so the workgroup writes the
// 16 tracks' whole contiguous span of every output array and the partial cache lines at the ends of a
// 32-row run are completed by a sibling wave a moment later (run ends shared between workgroups on
// different XCDs, hence different L2s, cost 30 % of the store rate).  Each wave reads its quarter's
// 4-KB blocks of (qx, qy, ±cell) once, transposes them in private LDS tiles, derives p (tile column
// shifted by one row; slot 0 = last row of the previous chunk; staged p for marked rows, element < 0:
// first record of a track / piece, generic step) and ℓ = ‖p − q‖ with the march's own expression (Segment
// ctor, src/segment.jl:31-33), so the records are bit-identical to fully staged ones, and writes every
// track's 32 rows as one run per output array.  20 B read + 44 B written per segment instead of 44 + 44.
// All loads are issued before the first store: gfx950 retires both through one in-order vmcnt queue.
constexpr int kC3Pitch = kChunkRows + 4;  // doubles per track in a tile: slot 0 = carry, slots 1..32 = rows
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_compact3(DTracks t, const int32_t *__restrict__ counts,
                                                  const int64_t *__restrict__ offsets, DStage stg, DOut out, DSplit sp,
                                                  const int32_t *__restrict__ corder) {
    static_assert(kChunkRows == 32, "k_compact3 moves 32-row chunks");
    __shared__ double tiles_x[4][16 * kC3Pitch];  // 36.9 KB per workgroup: four workgroups per CU
    __shared__ double tiles_y[4][16 * kC3Pitch];
    if (stg.cursor[1] != 0) return;  // pool overflow: this attempt is void
    // corder (large batches): workgroups take the march waves in the order of their output addresses — a batch that takes
    // several rounds of workgroups anyway then writes the 44-B records front to back instead of scattered over gigabytes
    const int64_t w = corder ? corder[blockIdx.x >> 2] : (blockIdx.x >> 2);  // SPLIT: canonical virtual wave (one piece of 64 consecutive tracks)
    const int q = blockIdx.x & 3;       // quarter: tracks 16 q .. 16 q + 15 of the wave
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = lane & 15, rr = lane >> 4;     // load mapping: track tl, rows rr, rr+4, ...
    const int rowL = lane & 31, sub = lane >> 5;  // store mapping: row rowL of tracks sub, sub+2, ...
    // LDS-address-space pointers: through generic pointers the tile accesses become FLAT instructions, which
    // take the vector-memory path (and its in-order counter) beside the global loads and stores
    typedef __attribute__((address_space(3))) volatile double lds_f64;
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    lds_f64 *tx = (lds_f64 *)tiles_x[k], *ty = (lds_f64 *)tiles_y[k];
    lds_i32 *te = (lds_i32 *)tiles_x[k];  // the x tile is reused for the cell ids
    const int64_t slot = (SPLIT ? (int64_t)sp.vw_wave[w] : w) * 64 + 16 * q + tl;  // lanes 0..15: their track's count / offset
    int32_t cnt = 0;
    int64_t off = 0;
    if (slot < t.n) {
        if (SPLIT) {
            const int64_t pi = w * 64 + 16 * q + tl;
            cnt = sp.p_valid[pi];  // 0 for a piece that was overrun
            off = offsets[slot] + sp.p_rel[pi];
        } else {
            const int32_t u = t.perm[slot];
            cnt = counts[u];
            off = offsets[u];
        }
    }
    int32_t gmax = cnt;
    for (int o = 8; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(gmax, o, 64);
        gmax = v > gmax ? v : gmax;
    }
    gmax = __shfl(gmax, 0, 64);
    const RT_G int32_t *ctab = stg.ctab + w * kMaxChunks;
    const int lane_q = 16 * q + tl;  // this lane's column of the march wave (load mapping)
    for (int j = k; (j << kChunkLog2) < gmax; j += 4) {
        const int r0 = j << kChunkLog2;
        const int32_t c = ctab[j];
        double vx[8], vy[8];
        int32_t ve[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t sidx = stage_slot(c, i * 4 + rr, lane_q);
            if (out.dbg & 2) { vx[i] = (double)sidx; vy[i] = 1.0; ve[i] = 1; continue; }
            vx[i] = __builtin_nontemporal_load(&stg.qx[sidx]);
            vy[i] = __builtin_nontemporal_load(&stg.qy[sidx]);
            ve[i] = __builtin_nontemporal_load(&stg.element[sidx]);
        }
        double hx = 0.0, hy = 0.0;  // lanes 0..15: q of the row before this chunk's first
        if (j > 0 && lane < 16) {
            const int64_t sidx = stage_slot(ctab[j - 1], kChunkRows - 1, lane_q);
            hx = stg.qx[sidx]; hy = stg.qy[sidx];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = i * 4 + rr;
            tx[tl * kC3Pitch + 1 + rl] = vx[i];
            ty[tl * kC3Pitch + 1 + rl] = vy[i];
        }
        if (lane < 16) { tx[tl * kC3Pitch] = hx; ty[tl * kC3Pitch] = hy; }
        __builtin_amdgcn_wave_barrier();
        // Pass 1 gathers the records (and fetches the staged p of marked rows) into registers, pass 2 only
        // stores: a load between the stores would have to wait for every store queued before it.
        double rpx[8], rpy[8], rqx[8], rqy[8];
        int32_t re[8];
        int64_t ro[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int tt = 2 * g + sub;
            const int32_t ct = __shfl(cnt, tt, 64);
            const int64_t ot = __shfl(off, tt, 64);
            const int row = r0 + rowL;
            ro[g] = (row < ct && ot + row < out.cap) ? ot + row : -1;
            rqx[g] = tx[tt * kC3Pitch + 1 + rowL]; rqy[g] = ty[tt * kC3Pitch + 1 + rowL];
            rpx[g] = tx[tt * kC3Pitch + rowL]; rpy[g] = ty[tt * kC3Pitch + rowL];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) te[tl * kC3Pitch + 1 + i * 4 + rr] = ve[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int tt = 2 * g + sub;
            re[g] = te[tt * kC3Pitch + 1 + rowL];
            if (ro[g] >= 0 && re[g] < 0) {  // this record keeps its own entry point
                const int64_t sidx = stage_slot(c, rowL, 16 * q + tt);
                rpx[g] = stg.px[sidx]; rpy[g] = stg.py[sidx];
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (ro[g] >= 0 && !((out.dbg & 1) && rpx[g] != -1.25)) {
                const int64_t o = ro[g];
                // plain stores: the partial lines at the ends of a run wait in L2 for the sibling wave's half
                // (nontemporal stores push them out half-written: +30 % compaction time)
                out.px[o] = rpx[g];
                out.py[o] = rpy[g];
                out.qx[o] = rqx[g];
                out.qy[o] = rqy[g];
                out.ell[o] = norm2(rpx[g] - rqx[g], rpy[g] - rqy[g]);
                out.element[o] = re[g] < 0 ? -re[g] : re[g];
            }
        }
        __builtin_amdgcn_wave_barrier();  // the tiles are rewritten if this wave has a further chunk
    }
}

// k_compact3's stores are runs of 32 rows (256 B) per track and array, two runs per store instruction; on batches whose
// records run to gigabytes (C5: 5 GB) they reach 3.75 TB/s while the loads alone run at 3.9 and a plain copy at 4.7-5.2.
// k_compact4 (whole tracks) writes the SAME records in memory order: a workgroup still owns (march wave, quarter) = 16 tracks
// and loads the same 4-KB staging blocks, four chunks = 128 rows per round, into track-major LDS tiles; after a barrier
// its 256 threads walk the concatenation of the 16 tracks' rows of the round — for tracks of up to 128 records that IS the
// workgroup's contiguous output span — so every store instruction writes 512 consecutive bytes and the four waves write
// 2 KB side by side.  Position -> (track, row) is a rank among the 16 wave-uniform prefix sums.  Same values, same
// expressions (p = previous q or the staged p of a marked row, ℓ = ‖p − q‖): bit-identical records.
constexpr int kC4Rows = 4 * kChunkRows;   // rows per round
constexpr int kC4Pitch = kC4Rows + 1;     // doubles per track in a tile: slot 0 = q of the row before the round, then the rows
__global__ __launch_bounds__(256) void k_compact4(DTracks t, const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets,
                                                  DStage stg, DOut out, const int32_t *__restrict__ corder) {
    static_assert(kChunkRows == 32, "k_compact4 moves 32-row chunks");
    __shared__ double tiles_x[16 * kC4Pitch];
    __shared__ double tiles_y[16 * kC4Pitch];
    __shared__ int32_t tiles_e[16 * kC4Rows];
    __shared__ int32_t s_cnt[16];
    __shared__ int64_t s_off[16];
    if (stg.cursor[1] != 0) return;  // pool overflow: this attempt is void
    typedef __attribute__((address_space(3))) volatile double lds_f64;
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    typedef __attribute__((address_space(3))) volatile int64_t lds_i64;
    lds_f64 *tx = (lds_f64 *)tiles_x, *ty = (lds_f64 *)tiles_y;
    lds_i32 *te = (lds_i32 *)tiles_e, *scnt = (lds_i32 *)s_cnt;
    lds_i64 *soff = (lds_i64 *)s_off;
    const int64_t w = corder ? corder[blockIdx.x >> 2] : (blockIdx.x >> 2);
    const int q = blockIdx.x & 3;
    const int tid = threadIdx.x, k = tid >> 6, lane = tid & 63, tl = lane & 15, rr = lane >> 4;
    if (tid < 16) {
        const int64_t slot = w * 64 + 16 * q + tid;
        int32_t cnt = 0;
        int64_t off = 0;
        if (slot < t.n) {
            const int32_t u = t.perm[slot];
            cnt = counts[u];
            off = offsets[u];
        }
        scnt[tid] = cnt; soff[tid] = off;
    }
    __syncthreads();
    int32_t cnt16[16];  // wave-uniform: scalar registers
    int32_t gmax = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        cnt16[j] = __builtin_amdgcn_readfirstlane(scnt[j]);
        gmax = cnt16[j] > gmax ? cnt16[j] : gmax;
    }
    const RT_G int32_t *ctab = stg.ctab + w * kMaxChunks;
    const int lane_q = 16 * q + tl;  // this lane's column of the march wave (load mapping)
    for (int r0 = 0; r0 < gmax; r0 += kC4Rows) {
        // ---- load: wave k moves chunk (r0 / 32) + k of the quarter into the tiles' rows 32 k ...
        const int j = (r0 >> kChunkLog2) + k;
        if ((j << kChunkLog2) < gmax) {
            const int32_t c = ctab[j];
            double vx[8], vy[8];
            int32_t ve[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t sidx = stage_slot(c, i * 4 + rr, lane_q);
                vx[i] = __builtin_nontemporal_load(&stg.qx[sidx]);
                vy[i] = __builtin_nontemporal_load(&stg.qy[sidx]);
                ve[i] = __builtin_nontemporal_load(&stg.element[sidx]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = 32 * k + i * 4 + rr;
                tx[tl * kC4Pitch + 1 + rl] = vx[i];
                ty[tl * kC4Pitch + 1 + rl] = vy[i];
                te[tl * kC4Rows + rl] = ve[i];
            }
        }
        if (r0 > 0 && tid < 16) {  // q of the row before this round's first
            const int64_t sidx = stage_slot(ctab[(r0 >> kChunkLog2) - 1], kChunkRows - 1, 16 * q + tid);
            tx[tid * kC4Pitch] = stg.qx[sidx]; ty[tid * kC4Pitch] = stg.qy[sidx];
        }
        __syncthreads();
        // ---- store: position p of the concatenated rows of this round -> (track, row); gather first, then only stores
        int32_t pre[17];  // wave-uniform prefix sums of the tracks' rows in this round
        pre[0] = 0;
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) {
            const int32_t left = cnt16[jt] - r0;
            pre[jt + 1] = pre[jt] + (left < 0 ? 0 : (left > kC4Rows ? kC4Rows : left));
        }
        double rpx[8], rpy[8], rqx[8], rqy[8];
        int32_t re[8];
        int64_t ro[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int32_t p = tid + 256 * e;
            ro[e] = -1;
            if (p < pre[16]) {
                int tt = 0;
#pragma unroll
                for (int jt = 1; jt < 16; ++jt) tt += p >= pre[jt] ? 1 : 0;
                int32_t base = 0;
#pragma unroll
                for (int jt = 1; jt < 16; ++jt) base = p >= pre[jt] ? pre[jt] : base;
                const int rl = p - base;            // row inside the round
                const int64_t o = soff[tt] + r0 + rl;
                ro[e] = o < out.cap ? o : -1;
                rqx[e] = tx[tt * kC4Pitch + 1 + rl]; rqy[e] = ty[tt * kC4Pitch + 1 + rl];
                rpx[e] = tx[tt * kC4Pitch + rl]; rpy[e] = ty[tt * kC4Pitch + rl];
                re[e] = te[tt * kC4Rows + rl];
                if (ro[e] >= 0 && re[e] < 0) {  // this record keeps its own entry point
                    const int row = r0 + rl;
                    const int64_t sidx = stage_slot(ctab[row >> kChunkLog2], row & (kChunkRows - 1), 16 * q + tt);
                    rpx[e] = stg.px[sidx]; rpy[e] = stg.py[sidx];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ro[e] >= 0) {
                const int64_t o = ro[e];
                out.px[o] = rpx[e];
                out.py[o] = rpy[e];
                out.qx[o] = rqx[e];
                out.qy[o] = rqy[e];
                out.ell[o] = norm2(rpx[e] - rqx[e], rpy[e] - rqy[e]);
                out.element[o] = re[e] < 0 ? -re[e] : re[e];
            }
        }
        __syncthreads();  // the tiles are rewritten by the next round
    }
}


// ---- codes -> records (the parallel half of the two-phase march) -----------------------------------------------------
// k_march<..., TOPO> decides; this kernel computes.  Per record the march left one word (DStage): 3·cell + exit edge + 1, or
// -(index + 1) of a side-list entry that holds the end points of a record of the generic step.  From the words, with
// k_compact3's data movement and shape (one 4-wave workgroup per unit = 16 tracks of a march wave, wave k takes the 32-row
// chunks k, k + 4, ...; transposing LDS tiles; every track's 32 rows stored as one run per array; every global load of a chunk
// before its first store — gfx950 retires both through one in-order counter, and the workgroups that follow hide the rest):
//   q = intersection(track.ABC, general_form of the exit edge)   src/intersection.jl:127-138 (edge_exit_point: walk_step's
//       expression; `etab` holds the host's general forms, evaluated with the reference's operations — bit-identical),
//   p = the previous record's q (bit-identical to the reference's own intersection with the shared edge: negating an edge's
//       general form negates numerator and denominator alike), or the side list's p,
//   ℓ = ‖p − q‖                                                  Segment ctor, src/segment.jl:31-33,
//   Σℓ per track and isapprox(track.ℓ, Σℓ; rtol)                 src/track.jl:171-175.  The partial sums of a track's chunks are
//       added in the order its waves finish, so the check is decided by MARGIN (any summation order is within n·2⁻⁵³·Σ of the
//       left-to-right sum of the reference's check); a track inside 96 such bands of the threshold is listed
//       and k_finish sums its ℓ again left to right.
// (fill_volumes stays with the march: a persistent variant of this kernel with an LDS copy of `volumes` per workgroup was built
//  and measured at twice the compaction's time — a wave's loads queue behind its own stores, chunk after chunk — DESIGN.md §4.)
// The gathers of the exit edges run in the LOAD mapping (the 16 lanes of a row are neighbouring tracks, which mostly cross the
// same edge: they share cache lines; in the store mapping every lane would fetch a line of its own).
// RECORDS: write the 44-B records.  ROWS: leave (ℓ, cell) of every staged row, slot-indexed like the rows, for rt_sweep.
struct DMat {
    const RT_G EdgeABC *etab;
    const RT_G int32_t *corder;   // large batches: march waves in the order of their output addresses (as k_compact3)
    int64_t n_units;              // 4 per march wave
    double rtol;
    int32_t tally;                // 1: Σℓ + status (the call's first pass over the codes); 0: records / rows only
    int32_t force_exact;          // tests: every track takes k_finish's left-to-right sum
    int32_t marg_cap;
    RT_G int32_t *marg;           // [0] count, [1 ...] march slots of the tracks k_finish has to sum exactly
    RT_G double *ell_rows;        // ROWS
    RT_G int32_t *cell_rows;
    RT_G double *vacc;            // fill_volumes' accumulator: the terms of the records the march flagged (kWordExactTally) are added here
    unsigned long long *ctl;      // the call's control block ([0] failed tracks, [1] first failing uid + 1)
};

#ifndef RT_MAT_OCC
#define RT_MAT_OCC 3  // waves per SIMD the kernel is compiled for
#endif
template <bool RECORDS, bool ROWS>
__global__ __launch_bounds__(256, RT_MAT_OCC) void k_materialise(DTracks t, const int32_t *__restrict__ counts, int32_t *__restrict__ status,
                                                     const int64_t *__restrict__ offsets, DStage stg, DOut out, DMat a) {
    static_assert(kChunkRows == 32, "k_materialise moves 32-row chunks");
    __shared__ double tiles_x[4][16 * kC3Pitch];  // per wave: the chunk's exit points (slot 0 of a track: the row before, i.e. the
    __shared__ double tiles_y[4][16 * kC3Pitch];  // first row's entry point), then its lengths (x tile) and cells (y tile)
    __shared__ double s_sum[16];                  // Σℓ of the unit's tracks
    __shared__ int64_t s_off[16];
    __shared__ int32_t s_cnt[16];
    if (stg.cursor[1] != 0 || stg.cursor[3] != 0) return;  // pool / side list overflow: this attempt is void
    const int kw = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = lane & 15, rr = lane >> 4;     // load mapping: track tl, rows rr, rr + 4, ...
    const int rowL = lane & 31, sub = lane >> 5;  // store mapping: row rowL of tracks sub, sub + 2, ...
    typedef __attribute__((address_space(3))) volatile double lds_f64;
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    typedef __attribute__((address_space(3))) volatile int64_t lds_i64;
    lds_f64 *X = (lds_f64 *)tiles_x[kw], *Y = (lds_f64 *)tiles_y[kw];
    lds_i32 *Yi = (lds_i32 *)tiles_y[kw];
    lds_i32 *scnt = (lds_i32 *)s_cnt;
    lds_i64 *soff = (lds_i64 *)s_off;
    const int64_t unit = blockIdx.x;
    const int64_t w = a.corder ? a.corder[unit >> 2] : (unit >> 2);
    const int q = (int)(unit & 3);
    const int64_t slot = w * 64 + 16 * q + tl;
    // every lane holds its load-mapping track's uid, count, offset and line (the 4 lanes of a track load the same words)
    // (everything a unit needs first is read in march-slot order, side by side: counts, offsets, lines, the wave's first chunk id)
    int32_t cnt = 0, u = 0;
    int64_t off = 0;
    double tA = 0.0, tB = 0.0, tC = 0.0;
    const bool have = slot < t.n;
    const RT_G int32_t *ctab = stg.ctab + w * kMaxChunks;
    const int32_t c_first = ctab[kw];
    if (have) {
        u = t.perm[slot];
        cnt = t.cnt_slot[slot];
        off = t.off_slot[slot];
        tA = t.As[slot]; tB = t.Bs[slot]; tC = t.Cs[slot];
    }
    if (threadIdx.x < 16) { scnt[tl] = cnt; soff[tl] = off; s_sum[tl] = 0.0; }
    int32_t gmax = cnt;
    for (int o = 8; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(gmax, o, 64);
        gmax = v > gmax ? v : gmax;
    }
    gmax = __shfl(gmax, 0, 64);
    __syncthreads();
    const int lane_q = 16 * q + tl;
    const int tb = tl * kC3Pitch;
    double acc = 0.0;  // Σℓ of this lane's rows of its load-mapping track
    for (int j = kw; (j << kChunkLog2) < gmax; j += 4) {
        const int r0 = j << kChunkLog2;
        const int32_t c = j == kw ? c_first : ctab[j];
        // ---- the chunk's words, in the load mapping (lane = track tl, rows 4 i + rr)
        int32_t ve[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ve[i] = __builtin_nontemporal_load(&stg.element[stage_slot(c, 4 * i + rr, lane_q)]);
        bool flagged = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!(r0 + 4 * i + rr < cnt)) ve[i] = 0;  // beyond the track's end: no record
            flagged = flagged || (ve[i] > 0 && (ve[i] & kWordExactTally) != 0);
        }
        const bool any_flagged = a.tally && __ballot(flagged) != 0;
        int32_t fmask = 0;  // rows of this lane whose fill_volumes term is added below
        if (__builtin_expect(__ballot(flagged) != 0, 0)) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (ve[i] > 0 && (ve[i] & kWordExactTally)) { fmask |= 1 << i; ve[i] &= ~kWordExactTally; }
        }
        // lanes 0..15 also hold the entry point of the chunk's first row: the exit point of the row before (another chunk's last
        // row) or, for a record that keeps its own end points, the side list's p
        double hx = 0.0, hy = 0.0;
        if (j > 0) {  // (uniform; a track's first chunk starts with a record of the generic step)
            int32_t hw = 0;
            if (lane < 16 && cnt > r0) hw = stg.element[stage_slot(ctab[j - 1], kChunkRows - 1, lane_q)];
            if (hw > 0) hw &= ~kWordExactTally;
            const RT_G EdgeABC *he = a.etab + (hw > 0 ? hw - 1 : 0);
            const double hA = he->A, hB = he->B, hC = he->C;
            edge_exit_point(tA, tB, tC, hA, hB, hC, hx, hy);
            if (__builtin_expect(hw < 0, 0)) { hx = stg.s_qx[-hw - 1]; hy = stg.s_qy[-hw - 1]; }
        }
        if (lane < 16 && ve[0] < 0) {  // the chunk's first row keeps its own entry point (every track's first record: chunk 0)
            const int32_t idx = -ve[0] - 1;
            hx = stg.s_px[idx]; hy = stg.s_py[idx];
        }
        // ---- exit points (four rows at a time: the gathers' registers)
        bool slow = false;  // a marked record that is not its chunk's first row: its entry point is fetched where it is needed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double eA[4], eB[4], eC[4];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const int i = 4 * h + i2;
                const RT_G EdgeABC *e = a.etab + (ve[i] > 0 ? ve[i] - 1 : 0);
                eA[i2] = e->A; eB[i2] = e->B; eC[i2] = e->C;
            }
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const int i = 4 * h + i2;
                double qx, qy;
                edge_exit_point(tA, tB, tC, eA[i2], eB[i2], eC[i2], qx, qy);  // src/intersection.jl:127-138
                if (__builtin_expect(ve[i] < 0, 0)) {  // a record of the generic step: its own q (and cell)
                    const int32_t idx = -ve[i] - 1;
                    qx = stg.s_qx[idx]; qy = stg.s_qy[idx];
                    if (i == 0 && rr == 0) ve[i] = 3 * (stg.s_el[idx] - 1) + 1;  // (its p sits in slot 0: from here on an ordinary word)
                    else slow = true;
                }
                X[tb + 1 + 4 * i + rr] = qx;
                Y[tb + 1 + 4 * i + rr] = qy;
            }
        }
        if (lane < 16) { X[tb] = hx; Y[tb] = hy; }
        const bool any_slow = __ballot(slow) != 0;
        __builtin_amdgcn_wave_barrier();
        // ---- ℓ = ‖p − q‖ (Segment ctor, src/segment.jl:31-33) in the load mapping; p, q to the output in the store mapping
        double dl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = 4 * i + rr;
            double px = X[tb + rl], py = Y[tb + rl];
            const double qx = X[tb + rl + 1], qy = Y[tb + rl + 1];
            if (__builtin_expect(any_slow, 0))
                if (ve[i] < 0) { px = stg.s_px[-ve[i] - 1]; py = stg.s_py[-ve[i] - 1]; }
            dl[i] = norm2(px - qx, py - qy);
            acc += ve[i] != 0 ? dl[i] : 0.0;
        }
        if (__builtin_expect(any_flagged, 0)) {
            // fill_volumes (src/trackgenerator.jl:382) for the records the march left out: δs[azim]·ℓ with the record's own length
            const double wt = have ? out.delta_s[t.azim[u] - 1] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((fmask >> i) & 1) unsafeAtomicAdd((double *)&a.vacc[(int32_t)((uint32_t)(ve[i] - 1) / 3u)], wt * dl[i]);
        }
        if (RECORDS) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int tt = 2 * g + sub;
                const int sb = tt * kC3Pitch + rowL;
                const int row = r0 + rowL;
                const int64_t o = soff[tt] + row;
                const double px = X[sb], qx = X[sb + 1], py = Y[sb], qy = Y[sb + 1];
                // plain stores: the partial lines at the ends of a run wait in L2 for the sibling wave's half
                if (row < scnt[tt] && o < out.cap && !((out.dbg & 1) && px != -1.25)) { out.px[o] = px; out.py[o] = py; out.qx[o] = qx; out.qy[o] = qy; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lengths and cells through the tiles
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int32_t wd = ve[i];
            if (__builtin_expect(any_slow, 0))
                if (wd < 0) wd = 3 * (stg.s_el[-wd - 1] - 1) + 1;
            const int32_t cell = (int32_t)((uint32_t)(wd > 0 ? wd - 1 : 0) / 3u) + 1;
            X[tb + 1 + 4 * i + rr] = dl[i];
            Yi[tb + 1 + 4 * i + rr] = cell;
            if (ROWS && wd != 0) {
                const int64_t sidx = stage_slot(c, 4 * i + rr, lane_q);
                a.ell_rows[sidx] = dl[i];
                a.cell_rows[sidx] = cell;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (RECORDS) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int tt = 2 * g + sub;
                const int sb = tt * kC3Pitch + 1 + rowL;
                const int row = r0 + rowL;
                const int64_t o = soff[tt] + row;
                const double ell = X[sb];
                const int32_t el = Yi[sb];
                if (row < scnt[tt] && o < out.cap && !((out.dbg & 1) && ell != -1.25)) { out.ell[o] = ell; out.element[o] = el; }
            }
            if (__builtin_expect(any_slow, 0)) {
                // the entry points of marked records that are not their chunk's first row, straight from the load mapping
                // (stores to the same addresses as above, later in program order: these stay)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (ve[i] < 0 && off + r0 + 4 * i + rr < out.cap) {
                        const int32_t idx = -ve[i] - 1;
                        out.px[off + r0 + 4 * i + rr] = stg.s_px[idx];
                        out.py[off + r0 + 4 * i + rr] = stg.s_py[idx];
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the tiles are rewritten if this wave has a further chunk
    }
    if (a.tally) {
        // Σℓ of the 16 tracks over this wave's rows: the four lanes of a track, then the four waves' parts in LDS
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 16 && acc != 0.0) atomicAdd(&s_sum[lane], acc);
        __syncthreads();
        if (kw == 0 && lane < 16 && have) {
            const double S = s_sum[lane];
            const double L = t.ell[u];
            // any-order sum against the left-to-right one: within cnt·2⁻⁵³·Σ; 96 bands hold the statistic's 64 (k_finish)
            if (a.force_exact || sum_check_is_marginal(L, S, a.rtol, cnt, 96.0)) {
                const int32_t e = atomicAdd((int32_t *)&a.marg[0], 1);
                if (e < a.marg_cap) a.marg[1 + e] = (int32_t)slot;  // (marg_cap = every march slot: cannot overflow)
            } else if (status[u] == RT_TRACK_OK && !isapprox_s(L, S, a.rtol)) {  // src/track.jl:171-175
                status[u] = RT_TRACK_LENGTH_MISMATCH;
                atomicAdd(&a.ctl[0], 1ull);
                atomicMin(&a.ctl[1], (unsigned long long)(u + 1));
            }
        }
    }
}

// After k_materialise: the tracks whose Σℓ check a sum in another order cannot decide are summed left to right — from the
// records, or from the ℓ rows when the call wrote no records — and checked as the reference does (src/track.jl:171-175); the
// statistic of rt_last_stats (tracks within 64 summation-order bands of the threshold) is counted here.  The block that
// finishes last — a ticket — copies the control block to the host and writes the call's sequence number behind it.
__global__ __launch_bounds__(256) void k_finish(DTracks t, const int32_t *__restrict__ counts, int32_t *__restrict__ status,
                                                const int64_t *__restrict__ offsets, const double *__restrict__ ell, int64_t cap,
                                                DStage stg, const double *__restrict__ ell_rows, double rtol, int32_t *__restrict__ marg,
                                                double *__restrict__ volumes, double *__restrict__ vacc, int32_t n_cells, double n_azim_2,
                                                unsigned long long *__restrict__ ctl, unsigned long long *__restrict__ host_copy,
                                                unsigned long long seq) {
    __shared__ int last_wg;
    const bool void_attempt = stg.cursor[1] != 0 || stg.cursor[3] != 0;
    // volumes ./= n_azim_2 (src/trackgenerator.jl:386): the march accumulated into `vacc` (k_materialise added the terms of the
    // records the march left to it), which is read, scaled into `volumes` and left ZERO for the next call's march
    if (volumes)
        for (int c = blockIdx.x * 256 + threadIdx.x; c < n_cells; c += gridDim.x * 256) {
            volumes[c] = vacc[c] / n_azim_2;
            vacc[c] = 0.0;
        }
    if (!void_attempt) {
        const int32_t nm = marg[0];
        for (int32_t e = blockIdx.x * 256 + threadIdx.x; e < nm; e += gridDim.x * 256) {
            const int32_t slot = marg[1 + e];
            if (slot < 0) continue;  // done by an earlier pass
            const int32_t u = t.perm[slot];
            const int32_t cnt = counts[u];
            const int64_t off = offsets[u];
            double S = 0.0;
            if (ell_rows) {
                const RT_G int32_t *ctab = stg.ctab + (int64_t)(slot >> 6) * kMaxChunks;
                for (int32_t r = 0; r < cnt; ++r) S += ell_rows[stage_slot(ctab[r >> kChunkLog2], r & (kChunkRows - 1), slot & 63)];
            } else if (ell && off + cnt <= cap) {
                for (int32_t r = 0; r < cnt; ++r) S += ell[off + r];
            } else {
                atomicAdd(&ctl[kCtlDeferred], 1ull);  // the host compacts again with larger arrays and calls this once more
                continue;
            }
            marg[1 + e] = -1 - slot;
            const double L = t.ell[u];
            if (sum_check_is_marginal(L, S, rtol, cnt)) atomicAdd(&ctl[kCtlNearRtol], 1ull);
            if (status[u] == RT_TRACK_OK && !isapprox_s(L, S, rtol)) {
                status[u] = RT_TRACK_LENGTH_MISMATCH;
                atomicAdd(&ctl[0], 1ull);
                atomicMin(&ctl[1], (unsigned long long)(u + 1));
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_wg = atomicAdd((unsigned int *)&ctl[kCtlFinishTicket], 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last_wg) return;
    __threadfence();
    if (threadIdx.x == 0 && __hip_atomic_load(&ctl[kCtlDeferred], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) marg[0] = 0;  // the list is consumed
    if (threadIdx.x == 0) ctl[kCtlFinishTicket] = 0;  // (a second pass of this call counts again)
    if (host_copy) {
        __syncthreads();
        if (threadIdx.x < kCtlWords) host_copy[threadIdx.x] = __hip_atomic_load(&ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&host_copy[kCtlWords], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- exclusive scan of per-track counts (int32) into CSR offsets (int64) ----------------
constexpr int kScanBlock = 256;
constexpr int kScanPer = 4;
constexpr int kScanTile = kScanBlock * kScanPer;

// Start of a call: the control block (failure summary, total, pool cursor, scan ticket) and `volumes` are reset
// by one small kernel instead of a host-to-device copy and a memset.
// The reset image of control-block word i: [1] first failing uid, an atomicMin target; [18] pool cursor (low word; chunks below
// first_chunk are reserved) + overflow flag; [19] side-list cursor (low word; entries below side_first are reserved) + overflow flag
__device__ __forceinline__ unsigned long long ctl_reset_word(int i, int32_t first_chunk, int32_t side_first) {
    return i == 1 ? ~0ull : (i == 18 ? (unsigned long long)(uint32_t)first_chunk : (i == 19 ? (unsigned long long)(uint32_t)side_first : 0ull));
}
__global__ void k_prologue(unsigned long long *__restrict__ ctl, double *__restrict__ volumes, int32_t n_cells, int32_t first_chunk, int32_t side_first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kCtlWords) ctl[i] = ctl_reset_word(i, first_chunk, side_first);
    if (volumes && i < n_cells) volumes[i] = 0.0;
}

// Pass 1 of the scan: the sum of every tile of kScanTile counts.  The block that finishes last (a ticket
// in the control block, no waiting) then scans the tile sums into exclusive tile offsets, writes the
// total, and — host_copy, optional — copies the 32-word control block to pinned host memory: the march (and
// k_resolve) are over when this kernel runs, so `total`, the failure summary and the pool cursor are final
// and the call needs no device-to-host copy after its last kernel.
// ctl_next (optional): the OTHER control block — calls alternate between two — is reset here for the next call (cursor behind
// `first_chunk_next` reserved chunks), so that a call needs no reset kernel in front of its march.
__global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(const int32_t *__restrict__ counts, int64_t n,
                                                               int64_t *__restrict__ tile_sums, int64_t n_tiles,
                                                               int64_t *__restrict__ total,
                                                               unsigned int *__restrict__ ticket,
                                                               const unsigned long long *__restrict__ ctl,
                                                               unsigned long long *__restrict__ host_copy,
                                                               unsigned long long *__restrict__ ctl_next, int32_t first_chunk_next,
                                                               int32_t side_first_next, unsigned long long seq) {
    __shared__ int64_t red[kScanBlock / 64];
    __shared__ int64_t carry;
    __shared__ int last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j)
        if (i0 + j < n) s += counts[i0 + j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t tot = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) tot += red[w];
        __hip_atomic_store(&tile_sums[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned int)(n_tiles - 1);
        carry = 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    for (int64_t base = 0; base < n_tiles; base += kScanBlock) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n_tiles ? __hip_atomic_load(&tile_sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int64_t incl = v;  // inclusive scan inside the wave, then across the block's waves
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        __syncthreads();  // red[] of the previous round has been read
        if (lane == 63) red[wv] = incl;
        __syncthreads();
        int64_t wave_off = 0;
        for (int w = 0; w < wv; ++w) wave_off += red[w];
        if (i < n_tiles) tile_sums[i] = carry + wave_off + incl - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry += wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
    if (host_copy) {
        __syncthreads();
        __threadfence();
        if (threadIdx.x < kCtlWords) host_copy[threadIdx.x] = __builtin_nontemporal_load(&ctl[threadIdx.x]);
        // the call's sequence number behind the copy, written once the copy is visible to the host: a stream-ordered call
        // (option "async") returns when it sees it, while the compaction is still running
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&host_copy[kCtlWords], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (ctl_next && threadIdx.x < kCtlWords) {
        const int i = threadIdx.x;
        ctl_next[i] = ctl_reset_word(i, first_chunk_next, side_first_next);
    }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_write(const int32_t *__restrict__ counts, int64_t n,
                                                           const int64_t *__restrict__ tile_offsets,
                                                           const int64_t *__restrict__ total,
                                                           int64_t *__restrict__ offsets,
                                                           double *__restrict__ volumes, int32_t n_cells,
                                                           double n_azim_2, double *__restrict__ vacc,
                                                           const int32_t *__restrict__ iperm, int64_t *__restrict__ off_slot) {
    // volumes ./= n_azim_2 (src/trackgenerator.jl:386) rides along when fill_volumes was fused into the march: the march
    // accumulated into `vacc`, which is read, scaled into `volumes` and left ZERO for the next call's march
    if (volumes)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += gridDim.x * blockDim.x) {
            volumes[c] = vacc[c] / n_azim_2;
            vacc[c] = 0.0;
        }
    __shared__ int64_t wsum[kScanBlock / 64];
    const int64_t i0 = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
    int64_t c[kScanPer];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        c[j] = (i0 + j < n) ? counts[i0 + j] : 0;
        s += c[j];
    }
    // inclusive scan of per-thread sums inside the wave, then across the block's waves
    int64_t incl = s;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wsum[w];
    int64_t run = tile_offsets[blockIdx.x] + wave_off + incl - s;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        if (i0 + j < n) {
            offsets[i0 + j] = run;
            if (iperm) off_slot[iperm[i0 + j]] = run;  // (the offsets in march-slot order, for k_materialise)
        }
        run += c[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = *total;
}

// fill_volumes (src/trackgenerator.jl:371-386) as its own pass over the compact records: each
// workgroup owns a contiguous range of tracks (hence a contiguous range of segments, read
// coalesced), accumulates δs[azim]·ℓ into an LDS-private copy of `volumes` with LDS atomics and
// flushes it with coalesced global atomics.  Random global f64 atomics from the march itself
// (64 lanes → 64 different lines) run ~17x below the coalesced rate and cost more than the march.
__global__ __launch_bounds__(1024) void k_volumes(const int64_t *__restrict__ offsets, int64_t n_tracks,
                                                  const int32_t *__restrict__ azim,
                                                  const double *__restrict__ delta_s,
                                                  const int32_t *__restrict__ element,
                                                  const double *__restrict__ ell, double *__restrict__ volumes,
                                                  int32_t n_cells, int32_t tpb, int32_t use_lds,
                                                  const int32_t *__restrict__ overflow, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (overflow && *overflow) return;  // staging pool overflowed: this attempt's records are void
    double *hist = reinterpret_cast<double *>(smem);
    int32_t *rel = reinterpret_cast<int32_t *>(smem + (use_lds ? (size_t)n_cells * sizeof(double) : 0));
    const int64_t u0 = (int64_t)blockIdx.x * tpb;
    const int64_t u1 = u0 + tpb < n_tracks ? u0 + tpb : n_tracks;
    if (u0 >= u1) return;
    const int nt = (int)(u1 - u0);
    const int64_t s0 = offsets[u0], s1 = offsets[u1] < cap ? offsets[u1] : cap;  // (records beyond the arrays' capacity: the host compacts again)
    if (use_lds)
        for (int c = threadIdx.x; c < n_cells; c += blockDim.x) hist[c] = 0.0;
    for (int j = threadIdx.x; j <= nt; j += blockDim.x) rel[j] = (int32_t)(offsets[u0 + j] - s0);
    __syncthreads();
    for (int64_t s = s0 + threadIdx.x; s < s1; s += blockDim.x) {
        const int32_t r = (int32_t)(s - s0);
        int lo = 0, hi = nt;  // largest j with rel[j] <= r
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rel[mid] <= r) lo = mid; else hi = mid;
        }
        const double w = delta_s[azim[u0 + lo] - 1];
        const double v = w * ell[s];
        const int32_t e = element[s] - 1;
        if (use_lds) atomicAdd(&hist[e], v);
        else unsafeAtomicAdd(&volumes[e], v);
    }
    if (use_lds) {
        __syncthreads();
        for (int c = threadIdx.x; c < n_cells; c += blockDim.x) {
            const double v = hist[c];
            if (v != 0.0) unsafeAtomicAdd(&volumes[c], v);
        }
    }
}

// Segment.τ (src/segment.jl:14,28: "storage for transport-related data (e.g., optical thickness)") for consumers that stay
// on the GPU: τ[s][g] = Σt[element[s]][g] · ℓ[s] over the device-resident records, G values per segment like the
// per-segment vector of the reference.  One thread per (segment, group) pair: ℓ and the cell id are read once per
// G consecutive lanes, the cross-section table is cache-resident, the writes are fully coalesced.
constexpr int kTauSegs = 2048;  // segments per workgroup
__global__ __launch_bounds__(256) void k_fill_tau(const double *__restrict__ ell, const int32_t *__restrict__ element,
                                                  const double *__restrict__ sigma_t, int64_t total, int32_t n_groups,
                                                  uint32_t inv_groups, double *__restrict__ tau) {
    // a workgroup owns kTauSegs consecutive segments: ℓ and the cell ids are read once, coalesced, into LDS; the
    // kTauSegs·G values are then produced in memory order (index / G by a multiply-high with the precomputed reciprocal)
    __shared__ double s_ell[kTauSegs];
    __shared__ int32_t s_el[kTauSegs];
    const int64_t s0 = (int64_t)blockIdx.x * kTauSegs;
    const int ns = (int)(total - s0 < kTauSegs ? total - s0 : kTauSegs);
    for (int j = threadIdx.x; j < ns; j += 256) {
        s_ell[j] = __builtin_nontemporal_load(&ell[s0 + j]);
        s_el[j] = __builtin_nontemporal_load(&element[s0 + j]) - 1;
    }
    __syncthreads();
    const uint32_t nv = (uint32_t)ns * (uint32_t)n_groups;
    double *out = tau + s0 * n_groups;
    for (uint32_t j = threadIdx.x; j < nv; j += 256) {
        const uint32_t sl = inv_groups ? __umulhi(j, inv_groups) : j;  // j / n_groups (exact while j < 2^32 / n_groups; 0: one group)
        const uint32_t g = j - sl * (uint32_t)n_groups;
        __builtin_nontemporal_store(sigma_t[(int64_t)s_el[sl] * n_groups + g] * s_ell[sl], &out[j]);
    }
}

// ---- transport sweep over the cyclic tracks ------------------------------------------------------
// The consumer the reference's Track/Segment layout exists for (README.md:127-135: "for track in tg.tracks_by_uid, for
// segment in track.segments: segment.ℓ, segment.element"; Segment.τ is its per-segment storage, src/segment.jl:14,28; the
// tracks form closed loops through next_track_fwd / next_track_bwd and dir_next_track_*, src/track.jl:42-77, walked as in
// demo/makie.jl:103-133): one method-of-characteristics sweep.  Every track is traversed forward (segments in march order)
// and backward (reversed); along a segment of length ℓ in cell e, for every energy group g,
//     τ = Σt[e][g]·ℓ,   Δ = (ψ − q[e][g]/Σt[e][g]) · (−expm1(−τ)),   ψ ← ψ − Δ,   φ[e][g] += w_track · Δ
// (ψ_out = ψ_in·e^{−τ} + (q/Σt)(1 − e^{−τ}) in its cancellation-free form); ψ starts from the track's incoming boundary
// flux and ends as its outgoing flux, which k_sweep_link hands to the linked track's entry for the next sweep (0 behind a
// Vacuum boundary).  One lane per track, the march's own lane mapping — so the STAGED variant reads the march's staging
// rows directly (20 B per segment, each row of a wave is four full 128-B lines; p = previous q, ℓ = ‖p − q‖ with the
// Segment constructor's expression, bit-identical to the compact records') and a device-resident consumer never needs the
// compaction; the other variant reads ℓ and the cell id of the compact CSR records.  The per-cell tallies are accumulated
// like fill_volumes: ds_add_f64 into an LDS-private copy of φ for GP groups at a time (the 160 KB of LDS hold 4 groups
// of the pincell mesh), flushed once per workgroup; meshes whose copy does not fit tally with global atomics.
struct DSweep {
    DStage stg;                       // STAGED: the march's staging rows
    const RT_G double *ell;           // compact records
    const RT_G int32_t *element;
    const RT_G int64_t *offsets;      // CSR offsets per uid
    const RT_G int32_t *counts;       // records per uid
    const RT_G int32_t *perm;         // march slot -> uid
    const RT_G int32_t *azim;         // default weight: delta_s[azim[u] - 1], as fill_volumes weighs a segment
    const RT_G double *delta_s;
    const RT_G double *w;             // explicit per-track weight (or null)
    const RT_G double *xs;            // [n_cells * G][2]: Σt, q / Σt
    const RT_G double *psi_in;        // [2][n][G] incoming boundary flux: forward (at track.p), backward (at track.q)
    RT_G double *psi_out;             // [2][n][G] outgoing flux at the other end
    RT_G double *phi;                 // [n_cells * G] tallies
    int64_t n;
    int32_t n_waves, n_cells, G, g0, ng, use_lds;
    int32_t debug;  // development: bit 0 skip the tallies
    RT_G double *ell_rows;  // STAGED: ℓ of every staged row, slot-indexed like the rows — written by the forward waves of a pass that
                            // derives ℓ from the exit points (when non-null), read by the ELLROWS passes instead of the exit points
};

// Software pipeline (the row addresses do not depend on data, unlike the march's): in iteration t the rows of step t + 2 and
// the cross sections of step t + 1 are in flight while step t is evaluated; every load is unconditional (clamped indices,
// results masked) so that no wait is forced by a branch, and the one rare load inside a branch — the staged entry point of a
// marked record — is issued BEFORE the iteration's prefetches: gfx950 returns loads in order, so waiting for it leaves the
// prefetches in flight.  The wave's chunk ids sit in registers (lane j holds chunk j) and are read with v_readlane.
template <bool STAGED, int GP, bool LDS, bool ELLROWS>
__global__ __launch_bounds__(1024) void k_sweep(DSweep a) {
    static_assert(STAGED || !ELLROWS, "ℓ rows belong to the staging rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char sweep_smem[];
    double *hist = reinterpret_cast<double *>(sweep_smem);  // [n_cells * GP] when LDS
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and known to be
    if (LDS) {
        for (int c = threadIdx.x; c < a.n_cells * GP; c += blockDim.x) hist[c] = 0.0;
        __syncthreads();
    }
    // a sweep wave = (march wave, direction).  The march waves are ordered longest first and the sweep is bound by
    // instruction issue, so the waves are dealt to the workgroups round-robin: wave k of workgroup b takes sweep wave
    // k * gridDim + b — every workgroup gets the same mix of long and short tracks and all finish together (contiguous
    // blocks of 16 sweep waves left the CU with the longest tracks working 1.6x longer than the average one).
    const int64_t sw = (int64_t)wib * gridDim.x + blockIdx.x;
    const int64_t mw = sw >> 1;
    const int dir = (int)(sw & 1);
    if (mw < a.n_waves) {
        const int64_t slot = mw * 64 + lane;
        const bool have = slot < a.n;
        const int32_t u = have ? a.perm[slot] : 0;
        const int32_t cnt = have ? a.counts[u] : 0;
        int32_t mc = cnt;
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t v = __shfl_xor(mc, o, 64);
            mc = v > mc ? v : mc;
        }
        const int maxcnt = __builtin_amdgcn_readfirstlane(mc);
        const double w = !have ? 0.0 : (a.w ? a.w[u] : a.delta_s[a.azim[u] - 1]);
        const int64_t off = (!STAGED && have) ? a.offsets[u] : 0;
        const int64_t pbase = ((int64_t)dir * a.n + u) * a.G + a.g0;
        const int ng = a.ng;
        double psi[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) psi[g] = (have && g < ng) ? a.psi_in[pbase + g] : 0.0;
        // step t visits row r(t): 0, 1, ... forward; maxcnt-1, ..., 0 backward (demo/makie.jl:103: "the segments are stored in
        // reverse order for backward tracks"), all lanes in lockstep — a lane is active while r(t) < its count.  Steps beyond
        // the end are clamped to the last one (prefetches only).
        auto row_of = [&](const int t) -> int {
            const int tc = t < maxcnt ? t : maxcnt - 1;
            return dir ? maxcnt - 1 - tc : tc;
        };
        // cross sections of GP groups of cell `e` (a padded group repeats the last real one; its result is never used)
        auto load_xs = [&](const int32_t e, double (&st)[GP], double (&qs)[GP]) {
            const RT_G double *x = a.xs + ((int64_t)e * a.G + a.g0) * 2;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const int gi = g < ng ? g : ng - 1;
                st[g] = x[2 * gi]; qs[g] = x[2 * gi + 1];
            }
        };
        ExpPoly poly = exp_poly();  // (in vector registers: see one_minus_exp_neg)
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(poly.c[i]));
        // one segment: attenuation and tally for the GP groups of this pass.  A lane beyond its track's end evaluates a segment
        // of length 0: τ = 0, 1 − e^{−0} = 0 exactly, Δ = ±0 — its ψ keeps its bits, and one select does for all groups.
        auto segment = [&](const int32_t e, const double ell_row, const bool act, const double (&st)[GP], const double (&qs)[GP]) {
            const double ell = act ? ell_row : 0.0;
            double wd[GP];
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const double tau = st[g] * ell;
                const double ex = one_minus_exp_neg(tau, poly);  // −expm1(−τ) to within an ulp (rt_device.hpp)
                const double d = (psi[g] - qs[g]) * ex;
                psi[g] = psi[g] - d;
                wd[g] = w * d;
            }
            // Neighbouring lanes are neighbouring parallel tracks: at the same row most of them are in the same cell, and
            // atomics of one wave instruction to one address are served one lane at a time (measured at C3: the tallies were
            // 0.21 of the sweep's 0.62 ms).  Lanes of an aligned pair, then quad, with equal cells are therefore summed first —
            // two DPP row shifts, no LDS traffic — and only the lanes left over add to the tally.  The sweep is bound by
            // instruction issue, so folding further costs more than the atomics it saves: over 2 / 4 / 8 / 16 lanes the
            // sweep took 0.440 / 0.438 / 0.466 / 0.494 ms (0.414 without any tally).
            bool mine = act;
            if (!(a.debug & 2)) {
                const int32_t key = act ? e : -1 - lane;  // (an inactive lane matches nobody)
                // lane l with (l mod 2n) == 0 takes over lane l + n (row_shl:n reads lane l + n of the 16-lane row)
                auto fold = [&]<int NSH>() {
                    // (bound_ctrl: a lane whose source lies outside its row reads 0 and no `old` value has to be moved in first;
                    //  the lanes that use what they read — `take`, `given` — never read across a row's end)
                    const int32_t key_up = __builtin_amdgcn_update_dpp(0, key, 0x100 + NSH, 0xf, 0xf, true);
                    const int32_t key_dn = __builtin_amdgcn_update_dpp(0, key, 0x110 + NSH, 0xf, 0xf, true);
                    const bool take = ((lane & (2 * NSH - 1)) == 0) && key_up == key;
                    const bool given = ((lane & (2 * NSH - 1)) == NSH) && key_dn == key;
#pragma unroll
                    for (int g = 0; g < GP; ++g) {
                        const uint64_t bits = __builtin_bit_cast(uint64_t, wd[g]);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(uint32_t)bits, 0x100 + NSH, 0xf, 0xf, true);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(uint32_t)(bits >> 32), 0x100 + NSH, 0xf, 0xf, true);
                        const double up = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
                        wd[g] = __builtin_fma(up, take ? 1.0 : 0.0, wd[g]);  // (one instruction; the values are finite)
                    }
                    mine = mine && !given;
                };
                fold.template operator()<1>(); fold.template operator()<2>();
            }
            if (mine && !(a.debug & 1)) {
#pragma unroll
                for (int g = 0; g < GP; ++g)
                    if (g < ng) {  // (uniform)
                        if (LDS) atomicAdd(&hist[e * GP + g], wd[g]);
                        else unsafeAtomicAdd((double *)&a.phi[(int64_t)e * a.G + a.g0 + g], wd[g]);
                    }
            }
        };
        if (maxcnt > 0) {
            if (STAGED) {
                // the wave's chunk ids: lane j holds chunks j, j + 64, ... (kMaxChunks = 313: five registers cover MAX_ITER rows)
                const RT_G int32_t *ctab = a.stg.ctab + mw * kMaxChunks;
                const int nchunks = (maxcnt + kChunkRows - 1) >> kChunkLog2;
                int32_t cv[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) cv[k] = (k * 64 + lane < nchunks) ? ctab[k * 64 + lane] : 0;
                // (v_readlane reads a lane whether or not it is active: call this in wave-uniform control flow only — inside a
                //  divergent branch the selected register of an inactive holder lane is stale)
                auto chunk_of = [&](const int r) -> int32_t {
                    const int j = r >> kChunkLog2;
                    const int32_t v = j < 64 ? cv[0] : (j < 128 ? cv[1] : (j < 192 ? cv[2] : (j < 256 ? cv[3] : cv[4])));
                    return __builtin_amdgcn_readlane(v, j & 63);
                };
                struct Row { double qx, qy; int32_t el; };
                // the chunk id of a row is looked up only when the row stream enters another 32-row chunk (two streams: the row
                // being evaluated and the one being prefetched); both lookups stay in scalar registers
                int cj0 = -1, cj2 = -1;
                int32_t cid0 = 0, cid2 = 0;
                auto slot_cached = [&](const int r, int &cj, int32_t &cid) -> int64_t {
                    const int j = r >> kChunkLog2;
                    if (j != cj) { cj = j; cid = chunk_of(r); }  // (uniform)
                    return stage_slot(cid, r & (kChunkRows - 1), lane);
                };
                auto slot_of = [&](const int r) -> int64_t { return stage_slot(chunk_of(r), r & (kChunkRows - 1), lane); };
                auto load_row = [&](const int r) -> Row {
                    const int64_t sl = slot_of(r);
                    return Row{a.stg.qx[sl], a.stg.qy[sl], a.stg.element[sl]};
                };
                auto cell_of = [&](const Row &R, const int r) -> int32_t { return r < cnt ? (R.el < 0 ? -R.el : R.el) - 1 : 0; };
                // One step: Ra holds row r(t), Rb row r(t + 1) and Rc — until this step's prefetch replaces it — row r(t − 1).  The
                // loop is unrolled three times with the roles rotated, so that no row register is moved from one stage of the
                // pipeline to the next; steps t >= maxcnt of the last round do nothing (act is false, their loads are clamped).
                // Measured at C3, 7 groups, same box: rotating by moves 0.373 ms, three steps per round 0.358, six (the cross
                // sections' two stages rotated as well; 32 scalar registers spilled) 0.366; one copy of the loop per direction
                // (forward and backward waves of a CU then run different code) 0.396.
                if constexpr (!ELLROWS) {
                    const int DIR = dir;
                    auto row_d = row_of;
                    Row R0 = load_row(row_d(0)), R1 = load_row(row_d(1)), R2{0.0, 0.0, 0};
                    double stA[GP], qsA[GP], stB[GP], qsB[GP];
                    load_xs(cell_of(R0, row_d(0)), stA, qsA);
                    auto step = [&](const int t, const Row &Ra, const Row &Rb, Row &Rc, const double (&st0)[GP], const double (&qs0)[GP],
                                    double (&st1)[GP], double (&qs1)[GP]) {
                        const int r = row_d(t);
                        const bool act = r < cnt && t < maxcnt;
                        // entry point: the previous record's exit point — forward the row before, backward the NEXT step's row — or,
                        // for marked records (cell < 0: first record of a track, records of the generic step), the staged one
                        double dx = (DIR ? Rb.qx : Rc.qx) - Ra.qx, dy = (DIR ? Rb.qy : Rc.qy) - Ra.qy;
                        const int64_t sl0 = slot_cached(r, cj0, cid0);  // (outside the branch: see chunk_of)
                        double px = 0.0, py = 0.0;
                        const bool marked = act && Ra.el < 0;
                        if (marked) { px = a.stg.px[sl0]; py = a.stg.py[sl0]; }
                        const int64_t sl2 = slot_cached(row_d(t + 2), cj2, cid2);
                        Rc = Row{a.stg.qx[sl2], a.stg.qy[sl2], a.stg.element[sl2]};
                        load_xs(cell_of(Rb, row_d(t + 1)), st1, qs1);
                        if (marked) { dx = px - Ra.qx; dy = py - Ra.qy; }
                        const double ell = norm2(dx, dy);  // Segment ctor, src/segment.jl:31-33 (as k_compact3)
                        if (a.ell_rows != nullptr && !DIR && act) a.ell_rows[sl0] = ell;  // (uniform && uniform && lane: for the ELLROWS passes)
                        segment(cell_of(Ra, r), ell, act, st0, qs0);
                    };
                    for (int t = 0; t < maxcnt; t += 3) {
                        step(t, R0, R1, R2, stA, qsA, stB, qsB);
                        step(t + 1, R1, R2, R0, stB, qsB, stA, qsA);
                        step(t + 2, R2, R0, R1, stA, qsA, stB, qsB);
#pragma unroll
                        for (int g = 0; g < GP; ++g) { stA[g] = stB[g]; qsA[g] = qsB[g]; }
                    }
                }
                if constexpr (ELLROWS) {
                    // the same pipeline over (ℓ, cell) rows — ℓ as an earlier pass over these staging rows left it: 12 B per row instead
                    // of 20, no square root, no entry point to pick
                    struct LRow { double ell; int32_t el; };
                    auto load_lrow = [&](const int64_t sl) -> LRow { return LRow{a.ell_rows[sl], a.stg.element[sl]}; };
                    auto lcell = [&](const LRow &R, const int r) -> int32_t { return r < cnt ? (R.el < 0 ? -R.el : R.el) - 1 : 0; };
                    LRow L0 = load_lrow(slot_of(row_of(0))), L1 = load_lrow(slot_of(row_of(1))), L2{0.0, 0};
                    double stA[GP], qsA[GP], stB[GP], qsB[GP];
                    load_xs(lcell(L0, row_of(0)), stA, qsA);
                    auto lstep = [&](const int t, const LRow &Ra, const LRow &Rb, LRow &Rc, const double (&st0)[GP], const double (&qs0)[GP],
                                     double (&st1)[GP], double (&qs1)[GP]) {
                        const int r = row_of(t);
                        const bool act = r < cnt && t < maxcnt;
                        Rc = load_lrow(slot_cached(row_of(t + 2), cj2, cid2));
                        load_xs(lcell(Rb, row_of(t + 1)), st1, qs1);
                        segment(lcell(Ra, r), Ra.ell, act, st0, qs0);
                    };
                    for (int t = 0; t < maxcnt; t += 3) {
                        lstep(t, L0, L1, L2, stA, qsA, stB, qsB);
                        lstep(t + 1, L1, L2, L0, stB, qsB, stA, qsA);
                        lstep(t + 2, L2, L0, L1, stA, qsA, stB, qsB);
#pragma unroll
                        for (int g = 0; g < GP; ++g) { stA[g] = stB[g]; qsA[g] = qsB[g]; }
                    }
                }
            } else {
                struct Rec { double ell; int32_t el; };
                auto load_rec = [&](const int r) -> Rec {
                    const int rc = r < cnt ? r : (cnt > 0 ? cnt - 1 : 0);  // (a lane's own records only; masked where r >= cnt)
                    if (cnt == 0) return Rec{0.0, 1};                      // (a track without records: offsets[u] may equal the total)
                    return Rec{a.ell[off + rc], a.element[off + rc]};
                };
                auto cell_of = [&](const Rec &R, const int r) -> int32_t { return r < cnt ? R.el - 1 : 0; };
                Rec R0 = load_rec(row_of(0)), R1 = load_rec(row_of(1));
                double st0[GP], qs0[GP];
                load_xs(cell_of(R0, row_of(0)), st0, qs0);
                for (int t = 0; t < maxcnt; ++t) {
                    const int r = row_of(t);
                    const Rec R2 = load_rec(row_of(t + 2));
                    double st1[GP], qs1[GP];
                    load_xs(cell_of(R1, row_of(t + 1)), st1, qs1);
                    segment(cell_of(R0, r), R0.ell, r < cnt, st0, qs0);
                    R0 = R1; R1 = R2;
#pragma unroll
                    for (int g = 0; g < GP; ++g) { st0[g] = st1[g]; qs0[g] = qs1[g]; }
                }
            }
        }
        if (have)
#pragma unroll
            for (int g = 0; g < GP; ++g)
                if (g < ng) a.psi_out[pbase + g] = psi[g];
    }
    if (LDS) {
        __syncthreads();
        for (int c = threadIdx.x; c < a.n_cells * GP; c += blockDim.x) {
            const double v = hist[c];
            const int cell = c / GP, g = c - cell * GP;
            if (v != 0.0 && g < a.ng) unsafeAtomicAdd((double *)&a.phi[(int64_t)cell * a.G + a.g0 + g], v);
        }
    }
}

// The boundary flux of the next sweep: entry (direction d', track v) receives the outgoing flux of the (direction, track)
// linked to it through next_track_fwd / next_track_bwd and dir_next_track_* (src/track.jl:42-77; the gather map is built on
// the host from rt_trace's link arrays), 0 behind a Vacuum boundary or where nothing is linked.
__global__ __launch_bounds__(256) void k_sweep_link(const int32_t *__restrict__ src_of, const double *__restrict__ psi_out,
                                                    double *__restrict__ psi_in, int64_t n2, int32_t G, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (entry slot, group)
    if (i >= n2 * G) return;
    const int64_t slot = i / G;
    const int32_t g = (int32_t)(i - slot * G);
    const int32_t sc = src_of[slot];  // source track * 2 + source direction, -1: none
    psi_in[i] = sc < 0 ? 0.0 : psi_out[((int64_t)(sc & 1) * n + (sc >> 1)) * G + g];
}

__global__ void k_scale_volumes(double *__restrict__ vol, int32_t n_cells, double n_azim_2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) vol[i] = vol[i] / n_azim_2;  // volumes ./= n_azim_2, src/trackgenerator.jl:386
}

}  // namespace rt

// ------------------------------------------------------------------- handles -------------
struct rt_mesh {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int32_t n_nodes = 0, n_cells = 0;
    DevBuf<double> x, y;
    DevBuf<int32_t> cn, ncp, ncd, gstart, gnode, c3start, c3node;
    DevBuf<double> c3x, c3y;
    DevBuf<rt::FanEntry> fan;
    DevBuf<rt::WalkRec> wrec;
    DevBuf<int32_t> adjr;
    DevBuf<rt::TopoRec> trec;   // cheap-step records and the cells' edge general forms
    DevBuf<rt::EdgeABC> etab;
    DevBuf<rt::DGeo> geo;
    rt::DMesh d{};
    rt_enqueue_hook enqueue_hook = nullptr;  // see rt_mesh_set_enqueue_hook
    void *enqueue_hook_user = nullptr;
    int64_t iter_cap = 4000000;
    bool walk_available = false;
    int volumes_mode = 2;  // 0: skip (measurement only), 1: fused global atomics in the fill march, 2: separate LDS-privatised pass
    int single_pass = 1;   // 1: staged single-pass march + compaction, 0: count / scan / fill (two marches)
    int split = -1;         // track splitting (see DSplit), read by rt_tracks_create: -1 auto (only batches that leave the chip
                            // underfilled), 0 off, > 0 pieces of about `split` expected segments
    int n_cus = 256;
    int lds_per_block = 64 * 1024;  // hipDeviceAttributeMaxSharedMemoryPerBlock
    int sweep_gp = 0, sweep_waves = 0;  // rt_sweep: groups per pass / waves per workgroup (0: automatic)
    int sweep_ell = 1;  // rt_sweep over staged rows: keep ℓ of every row from the first pass for the later ones (0: every pass derives it)
    int sweep_debug = 0, compact_debug = 0;
    int march_waves = 0;     // 4 / 6: waves per workgroup of the fused march (0: automatic)
    int compact_kernel = 0;  // 4: k_compact4 (memory-order stores) for whole-track batches; else k_compact3
    int first = 0;  // 1: every track's first record by k_first, eight lanes per track, ahead of the whole-track march.  Built,
                    // parity-green, and measured SLOWER (C3: the march 178 -> 163 µs, k_first itself 50 µs): off
    int topo = 1;          // 1: cheap steps (k_march<..., TOPO>) for whole-track batches when the mesh allows it; 2: forced — also on
                           // meshes where fewer than 90 % of the walkable records carry a cheap certificate, and a wave that is
                           // refused often does not hand back to exact steps (tests and fuzzing: every cheap certificate is exercised)
    int async_calls = 0;   // 1: rt_segmentize returns once total, status summary and offsets' scan are known to the host; the
                           // compaction may still be running on the stream (every entry point that touches results waits)
    int timing = 0;        // 1: record HIP events between the kernels of a call for rt_last_timing (≈4 µs of stream time each)
    bool topo_available = false;
    double topo_tiny_max = 0.0, topo_rmax = 0.0, topo_end_err = 0.0, tally_tau = 0.0;
    int64_t test_tally_tau = 0;  // tests only: overrides tally_tau (in 1e-12; < 0: ∞ — every cheap record tallied by k_materialise)
    int64_t n_records_topo = 0;
    int hybrid = 0;        // 1: batches that fill the chip march only their longest waves in pieces, beside the whole-track march of the rest
                           // (measured slower at every threshold on MI355X — the full batch is within 1.6x of its throughput floor — DESIGN.md §4)
    int lds_records = 0;   // experiment: 1 = eight-wave workgroups with all walk records in LDS (meshes that fit), 2 = the same shape from L2
    int hybrid_pct = 55;   // ... those whose expected segment count exceeds this percentage of the batch's longest
    int fuse_volumes = 1;  // 1: fill_volumes inside the single-pass march (LDS-private) when the mesh fits
    int compact = 1;       // 0: rt_segmentize stops after march + scan (offsets, status, volumes); the 44-B records are produced on
                           // demand (rt_fetch_segments*, rt_device_pointers, rt_fill_tau), and rt_sweep reads the staged rows directly
    int64_t pool_chunks_hint = 0;  // > 0: initial staging-pool size in chunks (tests force the overflow path)
    int64_t test_out_records = 0;   // tests only: capacity of the output arrays on a handle's first call (forces the re-compaction path)
    int test_volumes_fallback = 0;  // tests only: take the split mode's volumes recomputation path unconditionally
    int test_exact_sums = 0;        // tests only: every track's Σℓ check by k_finish's left-to-right sum (two-phase march)
    int64_t side_entries_hint = 0;  // tests only: capacity of the dynamic part of the side list on a handle's first call (forces its overflow path)
    int sort_mode = 2;     // march order: 0 uid order, 1 longest track first, 2 uid-contiguous waves, longest wave first
    double kappa = 0.0;    // expected segments per unit track length (sizes the staging pool)
    std::string prep_note;
    // diagnostics of the host preprocessing (rt_mesh_info)
    int64_t n_records = 0, n_records_walk = 0;
    int32_t n_cells_fragile = 0, n_cells_wild = 0, n_edges_nonmanifold = 0, extras_max = 0;
    double eps_min = 0.0, eps_max = 0.0, prep_ms = 0.0;
};

// A piece of a handle's input arena (one device allocation, filled by one host-to-device copy).
template <typename T>
struct DevView {
    T *p = nullptr;
    size_t cap = 0;  // (counted with the arena, not here)
    void release() { p = nullptr; }
};

struct rt_tracks {
    rt_mesh *mesh = nullptr;
    int64_t n = 0;
    DevBuf<unsigned char> in_arena;  // px | py | phi | cos ϕ | sin ϕ | A | B | C | ℓ | A, B, C in march order | azim_idx | march order | its inverse | compaction order
    DevView<double> px, py, phi, cs, sn, A, B, C, ell;
    DevView<int32_t> corder;  // march waves sorted by the uid of their first track (the compaction order of large batches)
    DevView<int32_t> azim, perm;  // perm: march order of all tracks
    DevView<double> As, Bs, Cs;   // the track lines in march order (k_materialise)
    DevView<int32_t> iperm;       // uid -> march slot
    DevBuf<int32_t> cnt_slot;     // record counts / CSR offsets in march-slot order (whole-track two-phase calls)
    DevBuf<int64_t> off_slot;
    DevBuf<int32_t> perm_whole;  // ... of those the hybrid plan marches whole
    rt::DTracks d{};
    // results
    bool segmentized = false;
    int64_t total = 0;
    DevBuf<int32_t> counts, status, element;
    DevBuf<int64_t> offsets, tile_sums;
    // one control block: words 0..15 failure summary / stats, 16 total segments, 18..19 pool cursor + overflow flag,
    // 20 ticket of the scan's "last block" step, 21 tracks that reached MAX_ITER segments in split mode
    DevBuf<unsigned long long> ctl;  // two blocks of kCtlWords: calls alternate, each call's scan resets the other block
    int ctl_idx = 0;                 // block of the next call
    bool ctl_clean[2] = {false, false};
    int64_t ctl_first_chunk[2] = {-1, -1};  // ... reset with this many reserved chunks (low word) and side-list entries (high word)
    DevBuf<double> vacc;             // fused fill_volumes accumulates here; k_scan_write scales it into `volumes` and zeroes it
    bool vacc_clean = false;
#ifdef RT_TIMING
    DevBuf<unsigned long long> dbg;
#endif
    unsigned long long *h_ctl = nullptr;  // pinned: [0..63] init image, [64..127] read-back
    unsigned long long *h_res_dev = nullptr;  // device address of the read-back half
    DevBuf<double> spx, spy, sqx, sqy, sell, volumes, delta_s;
    DevBuf<double> tau, sigma_t;  // rt_fill_tau
    int32_t tau_groups = 0;
    DevBuf<double> volumes_prev;  // the previous call's volumes: the two buffers alternate (see rt_device_pointers)
    // staging pool of the single-pass march
    DevBuf<double> gpx, gpy, gqx, gqy;
    DevBuf<int32_t> gelement, ctab, cowner;
    // two-phase march (k_march<TOPO> + k_materialise): the side list of records that keep their own end points, the
    // workgroups' shares of `volumes`, the list of tracks whose Σℓ check k_finish decides with a left-to-right sum
    DevBuf<double> side_px, side_py, side_qx, side_qy;
    DevBuf<int32_t> side_el, marg;
    int64_t side_cap = 0, side_needed_last = 0;
    bool marg_clean = false;
    DevBuf<int32_t> fst_i;   // k_first: it, T, pred per march slot
    DevBuf<double> fst_v;    // ... and its ten doubles
    int32_t last_first = 0;  // 1: the last call made the first records with k_first
    int64_t pool_chunks = 0, chunks_needed_last = 0, total_last = 0;
    // split mode (pieces of tracks)
    int32_t n_vwaves = 0;
    bool hybrid = false;   // the split plan covers only the longest waves; perm[0 .. n_whole) lists the tracks marched whole
    int64_t n_whole = 0;
    hipStream_t aux_stream = nullptr;  // hybrid: the pieces march beside the whole tracks
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf<int32_t> vorder, vw_wave, vw_k, w_base, w_P, s_el, s_eq, p_count, p_flags, p_valid, p_rel;
    DevBuf<double> s_px, s_py, s_qx, s_qy, s_ell, p_sum;
    double sum_ell = 0.0;
    int32_t azim_min = 1, azim_max = 0;  // range of azim_idx (checked against n_azim_2 by rt_segmentize)
    int64_t n_generic_records = 0;       // rt_last_stats
    bool force_unsplit = false;  // a track reached MAX_ITER segments in split mode: this track set marches whole from now on
    int32_t last_topo = 0;  // 1: the last call marched with cheap steps
    int64_t n_exact_walk_records = 0;  // ... and this many of its records came from exact walk steps
    int32_t last_march_waves = 0, last_split = 0, last_widek = 0;  // which instantiation of the march the last call launched
    std::vector<double> h_delta_s;  // what delta_s on the device currently holds
    void *pin[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // rt_fetch_segments_pinned
    size_t pin_cap = 0;                                                     // records
    int64_t *pin_off = nullptr;  // rt_fetch_pinned: offsets[n + 1] and status[n], page-locked like the records
    int32_t *pin_st = nullptr;
    hipEvent_t ev[8] = {};
    double ms[8] = {};
    // what the compaction of the last single-pass call needs (it may run later, on demand: option "compact" = 0)
    struct CompactPlan {
        rt::DStage stg{}, stg_pieces{};
        rt::DTracks d_whole{};
        rt::DSplit sp{};
        const int32_t *corder = nullptr;
        int64_t n_whole_waves = 0;
        bool split = false, split_all = false, staged = false;  // staged: the last call left staged rows (single-pass mode)
        bool codes = false;   // ... as one word per record (k_march<TOPO>): k_materialise turns them into records / (ℓ, cell) rows
        double rtol = 0.0;
    } cplan;
    bool compacted = false;  // the six record arrays hold the last call's records
    bool in_flight = false;  // option "async": the last rt_segmentize returned while its compaction was still on the stream
    unsigned long long call_seq = 0;  // sequence number the scan writes behind its host copy of the control block
    // rt_sweep: the gather map of the cyclic linking, per-track weights, cross sections, boundary fluxes, tallies
    DevBuf<int32_t> sw_src;
    DevBuf<double> sw_w, sw_xs, sw_psi_in, sw_psi_out, sw_phi;
    DevBuf<double> sw_ell;      // ℓ of every staged row (slot-indexed like the staging pool), left by the first staged pass after a call
    bool sw_ell_valid = false;  // ... of the last rt_segmentize
    DevBuf<int32_t> sw_cell;    // codes: cell + 1 of every staged row, beside sw_ell (k_materialise<.., ROWS>)
    bool sw_links = false, sw_has_w = false, sw_has_xs = false, sw_done = false;
    int32_t sw_groups = 0, sw_last_input = 0, sw_last_gp = 0, sw_last_passes = 0;
    int64_t refusals[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // cheap-step refusals of the last call by certificate term
    int64_t n_near_rtol = 0, n_restarts = 0;
    int64_t n_failed = 0, first_failed_uid = 0;
    int32_t first_failed_status = 0;
};

namespace {

inline size_t nw_all_early(size_t n) { return (n + 63) / 64; }

// Wait for a stream the way a latency-bound caller wants it: hipStreamSynchronize may sleep on an interrupt and
// wake well after the last kernel ended.  Poll for the first milliseconds, then sleep.
hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) return hipStreamSynchronize(s);
    }
}

// Stream-ordered calls (option "async"): wait until the scan's host copy of the control block carries this call's sequence number.
// Falls back to waiting for the stream when the number does not arrive (a failed launch never writes it).
hipError_t wait_seq(const unsigned long long *h_res, unsigned long long seq, hipStream_t s) {
    const volatile unsigned long long *flag = h_res + rt::kCtlWords;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
            return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? hipSuccess : hipErrorUnknown;
        }
    }
}

// Every entry point that reads what a call produced first waits for a call that is still on the stream (option "async").
int finish_call(rt_tracks *t) {
    if (t && t->in_flight) {
        RT_HIP(hipSetDevice(t->mesh->device));
        RT_HIP(wait_stream(t->mesh->stream));
        t->in_flight = false;
    }
    return RT_SUCCESS;
}

template <typename T>
int upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s) {
    RT_HIP(b.reserve(n > 0 ? n : 1));
    if (n) RT_HIP(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_SUCCESS;
}

int build_mesh(rt_mesh *m, const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes,
               int32_t n_cells, const int32_t *ncp_in, const int32_t *ncd_in, const double *bb) {
    // --- ids to 0-based
    std::vector<int32_t> cn(3 * (size_t)n_cells);
    for (size_t i = 0; i < cn.size(); ++i) {
        const int32_t v = cell_nodes[i] - 1;
        if (v < 0 || v >= n_nodes) { set_error("cell_nodes[%zu] = %d out of range", i, cell_nodes[i]); return RT_ERR_INVALID; }
        cn[i] = v;
    }
    const int32_t p0 = ncp_in[0];  // 0- or 1-based CSR offsets
    if (p0 != 0 && p0 != 1) { set_error("node_cells_ptrs must start at 0 or 1"); return RT_ERR_INVALID; }
    std::vector<int32_t> ncp(n_nodes + 1);
    for (int32_t i = 0; i <= n_nodes; ++i) {
        ncp[i] = ncp_in[i] - p0;
        if (ncp[i] < 0 || (i > 0 && ncp[i] < ncp[i - 1])) { set_error("node_cells_ptrs not monotone"); return RT_ERR_INVALID; }
    }
    const int32_t nnz = ncp[n_nodes];
    std::vector<int32_t> ncd(nnz > 0 ? nnz : 1);
    for (int32_t i = 0; i < nnz; ++i) {
        const int32_t v = ncd_in[i] - 1;
        if (v < 0 || v >= n_cells) { set_error("node_cells_data[%d] = %d out of range", i, ncd_in[i]); return RT_ERR_INVALID; }
        ncd[i] = v;
    }
    // --- node grid, per-cell walk records and certificate margins (rt_mesh_prep.hpp)
    const double W = bb[2] - bb[0], H = bb[3] - bb[1];
    if (!(W > 0) || !(H > 0) || !std::isfinite(W) || !std::isfinite(H)) {  // also: inboundary() relies on a finite box
        set_error("empty or non-finite bounding box");
        return RT_ERR_INVALID;
    }
    const auto t_prep0 = std::chrono::steady_clock::now();
    rtprep::Prep P = rtprep::prepare(x, y, n_nodes, cn.data(), n_cells, bb);
    m->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep0).count();
    const std::vector<int32_t> &gstart = P.gstart, &gnode = P.gnode;
    const double gh = P.gh, ginv = P.ginv;
    const int gnx = P.gnx, gny = P.gny;
    hipStream_t s = m->stream;
    int rc;
    if ((rc = upload(m->x, x, n_nodes, s))) return rc;
    if ((rc = upload(m->y, y, n_nodes, s))) return rc;
    if ((rc = upload(m->cn, cn.data(), cn.size(), s))) return rc;
    if ((rc = upload(m->ncp, ncp.data(), ncp.size(), s))) return rc;
    if ((rc = upload(m->ncd, ncd.data(), (size_t)nnz, s))) return rc;
    if ((rc = upload(m->gstart, gstart.data(), gstart.size(), s))) return rc;
    if ((rc = upload(m->gnode, gnode.data(), (size_t)n_nodes, s))) return rc;
    if ((rc = upload(m->c3start, P.c3start.data(), P.c3start.size(), s))) return rc;
    if ((rc = upload(m->c3node, P.c3node.data(), P.c3node.size(), s))) return rc;
    if ((rc = upload(m->c3x, P.c3x.data(), P.c3x.size(), s))) return rc;
    if ((rc = upload(m->c3y, P.c3y.data(), P.c3y.size(), s))) return rc;
    std::vector<rt::FanEntry> fan((size_t)std::max(nnz, 1));
    for (int32_t i = 0; i < nnz; ++i) {  // node -> cells with the cells' vertices, in the table's own order
        const int32_t c = ncd[i];
        rt::FanEntry &e = fan[i];
        e.x1 = x[cn[3 * c]]; e.y1 = y[cn[3 * c]]; e.x2 = x[cn[3 * c + 1]]; e.y2 = y[cn[3 * c + 1]]; e.x3 = x[cn[3 * c + 2]]; e.y3 = y[cn[3 * c + 2]];
        e.cell = c;
        for (int q = 0; q < 3; ++q) e.adj[q] = P.adjr[(size_t)3 * c + q];
    }
    if ((rc = upload(m->fan, fan.data(), fan.size(), s))) return rc;
    if ((rc = upload(m->wrec, reinterpret_cast<const rt::WalkRec *>(P.wrec.data()), P.wrec.size(), s))) return rc;
    if ((rc = upload(m->adjr, P.adjr.data(), P.adjr.size(), s))) return rc;
    if ((rc = upload(m->trec, reinterpret_cast<const rt::TopoRec *>(P.trec.data()), P.trec.size(), s))) return rc;
    if ((rc = upload(m->etab, reinterpret_cast<const rt::EdgeABC *>(P.etab.data()), P.etab.size(), s))) return rc;
    RT_HIP(hipStreamSynchronize(s));  // host vectors die at return
    m->n_nodes = n_nodes;
    m->n_cells = n_cells;
    rt::DMesh &d = m->d;
    using rt::as_global;
    rt::DGeo g{};
    g.x = as_global(m->x.p); g.y = as_global(m->y.p); g.cn = as_global(m->cn.p); g.ncp = as_global(m->ncp.p);
    g.ncd = as_global(m->ncd.p); g.gstart = as_global(m->gstart.p); g.gnode = as_global(m->gnode.p);
    g.c3start = as_global(m->c3start.p); g.c3node = as_global(m->c3node.p); g.c3x = as_global(m->c3x.p); g.c3y = as_global(m->c3y.p);
    g.fan = as_global((const rt::FanEntry *)m->fan.p);
    g.gx0 = bb[0]; g.gy0 = bb[1]; g.gh = gh; g.ginv = ginv; g.gnx = gnx; g.gny = gny;
    g.n_nodes = n_nodes;
    if ((rc = upload(m->geo, &g, 1, s))) return rc;
    RT_HIP(hipStreamSynchronize(s));
    d.geo = (const RT_K rt::DGeo *)m->geo.p;
    d.bx0 = bb[0]; d.by0 = bb[1]; d.bx1 = bb[2]; d.by1 = bb[3];
    d.n_cells = n_cells;
    d.wrec = as_global(m->wrec.p); d.adjr = as_global(m->adjr.p); d.d_vertex = P.d_vertex; d.l_min = P.l_min;
    d.walk_ok = P.walk_ok ? 1 : 0;
    d.trec = as_global((const rt::TopoRec *)m->trec.p); d.etab = as_global((const rt::EdgeABC *)m->etab.p);
    m->topo_available = P.walk_ok && P.topo_ok;
    m->topo_tiny_max = P.topo_tiny_max; m->topo_rmax = P.topo_rmax; m->topo_end_err = P.topo_end_err; m->tally_tau = P.tally_tau;
    m->n_records_topo = P.n_records_topo;
    m->walk_available = P.walk_ok;
    m->kappa = P.kappa;
    m->prep_note = P.note;
    m->n_records = P.n_records; m->n_records_walk = P.n_records_walk;
    m->n_cells_fragile = P.n_cells_fragile; m->n_cells_wild = P.n_cells_wild;
    m->n_edges_nonmanifold = P.n_edges_nonmanifold; m->extras_max = P.extras_max;
    m->eps_min = P.eps_min; m->eps_max = P.eps_max;
    return RT_SUCCESS;
}

void free_mesh(rt_mesh *m) {
    m->x.release(); m->y.release(); m->cn.release(); m->ncp.release(); m->ncd.release();
    m->gstart.release(); m->gnode.release(); m->c3start.release(); m->c3node.release(); m->c3x.release(); m->c3y.release(); m->fan.release(); m->wrec.release(); m->adjr.release(); m->trec.release(); m->etab.release(); m->geo.release();
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

void pin_release_to_cache(rt_tracks *t);  // defined with rt_fetch_segments_pinned

// The six record arrays of a handle, sized for `tot` records, as the kernels see them.
int reserve_records(rt_tracks *t, int64_t tot, rt::DOut &out) {
    using rt::as_global;
    const size_t cap = (size_t)(tot > 0 ? tot : 1);
    RT_HIP(t->spx.reserve(cap)); RT_HIP(t->spy.reserve(cap)); RT_HIP(t->sqx.reserve(cap));
    RT_HIP(t->sqy.reserve(cap)); RT_HIP(t->sell.reserve(cap)); RT_HIP(t->element.reserve(cap));
    out.px = as_global(t->spx.p); out.py = as_global(t->spy.p); out.qx = as_global(t->sqx.p);
    out.qy = as_global(t->sqy.p); out.ell = as_global(t->sell.p); out.element = as_global(t->element.p);
    out.cap = (int64_t)std::min({t->spx.cap, t->spy.cap, t->sqx.cap, t->sqy.cap, t->sell.cap, t->element.cap});
    return RT_SUCCESS;
}

// Codes -> records and / or (ℓ, cell) rows (k_materialise) for the plan of the last two-phase call.  tally: the call's first
// pass over the codes — Σℓ and status (k_finish completes them).
int launch_materialise(rt_tracks *t, const rt::DOut &out, hipStream_t s, bool records, bool rows, bool tally, unsigned long long *d_ctl) {
    using rt::as_global;
    rt_mesh *m = t->mesh;
    const rt_tracks::CompactPlan &c = t->cplan;
    if (t->n <= 0 || c.n_whole_waves <= 0) return RT_SUCCESS;
    rt::DMat a{};
    a.etab = m->d.etab; a.corder = as_global(c.corder);
    a.n_units = 4 * c.n_whole_waves; a.rtol = c.rtol; a.tally = tally ? 1 : 0;
    a.force_exact = m->test_exact_sums; a.ctl = d_ctl; a.vacc = as_global(t->vacc.p);
    if (tally) {
        RT_HIP(t->marg.reserve((size_t)c.n_whole_waves * 64 + 1));
        if (!t->marg_clean) { RT_HIP(hipMemsetAsync(t->marg.p, 0, sizeof(int32_t), s)); t->marg_clean = true; }
        a.marg = as_global(t->marg.p); a.marg_cap = (int32_t)std::min<int64_t>(c.n_whole_waves * 64, 0x7fffffff);
    }
    if (rows) {
        const size_t slots = (size_t)t->pool_chunks * rt::kChunkRows * 64;
        RT_HIP(t->sw_ell.reserve(slots > 0 ? slots : 1)); RT_HIP(t->sw_cell.reserve(slots > 0 ? slots : 1));
        a.ell_rows = as_global(t->sw_ell.p); a.cell_rows = as_global(t->sw_cell.p);
    }
    const unsigned blocks = (unsigned)a.n_units;
    if (records && rows)
        hipLaunchKernelGGL((rt::k_materialise<true, true>), dim3(blocks), dim3(256), 0, s, c.d_whole, (const int32_t *)t->counts.p, t->status.p,
                           (const int64_t *)t->offsets.p, c.stg, out, a);
    else if (records)
        hipLaunchKernelGGL((rt::k_materialise<true, false>), dim3(blocks), dim3(256), 0, s, c.d_whole, (const int32_t *)t->counts.p, t->status.p,
                           (const int64_t *)t->offsets.p, c.stg, out, a);
    else if (rows)
        hipLaunchKernelGGL((rt::k_materialise<false, true>), dim3(blocks), dim3(256), 0, s, c.d_whole, (const int32_t *)t->counts.p, t->status.p,
                           (const int64_t *)t->offsets.p, c.stg, out, a);
    else { set_error("k_materialise: nothing to write"); return RT_ERR_INVALID; }
    return RT_SUCCESS;
}

// k_finish behind a tallying k_materialise: exact Σℓ of the listed tracks; copies the control block to the host.
void launch_finish(rt_tracks *t, const rt::DOut &out, hipStream_t s, bool from_rows, bool scale_volumes, double n_azim_2,
                   unsigned long long *d_ctl, unsigned long long *h_res_dev, unsigned long long seq) {
    const rt_tracks::CompactPlan &c = t->cplan;
    const unsigned blocks = t->mesh->test_exact_sums ? 64u : 8u;
    hipLaunchKernelGGL(rt::k_finish, dim3(blocks), dim3(256), 0, s, c.d_whole, (const int32_t *)t->counts.p, t->status.p,
                       (const int64_t *)t->offsets.p, from_rows ? (const double *)nullptr : (const double *)t->sell.p, out.cap, c.stg,
                       from_rows ? (const double *)t->sw_ell.p : (const double *)nullptr, c.rtol, t->marg.p,
                       scale_volumes ? t->volumes.p : (double *)nullptr, t->vacc.p, t->mesh->n_cells, n_azim_2, d_ctl, h_res_dev, seq);
}

// Staged rows -> compact CSR records for the plan of the last single-pass call: k_compact3 over (q, ±cell) rows, or — codes —
// k_materialise without its tallies.
void launch_compaction(rt_tracks *t, const rt::DOut &out, hipStream_t s) {
    const rt_tracks::CompactPlan &c = t->cplan;
    if (c.codes) { (void)launch_materialise(t, out, s, true, false, false, nullptr); return; }
    // k_compact4 (stores in memory order) was built for batches whose records run to gigabytes and measured no faster:
    // C5 1.95 vs 1.84 ms, C4 0.270 vs 0.262, C3 0.149 vs 0.151 — it runs only on request (option "compact_kernel" = 4)
    const bool use4 = t->mesh->compact_kernel == 4;
    if (t->n > 0 && !c.split_all && c.n_whole_waves > 0) {
        if (use4)
            hipLaunchKernelGGL(rt::k_compact4, dim3(4u * (unsigned)c.n_whole_waves), dim3(256), 0, s, c.d_whole,
                               (const int32_t *)t->counts.p, (const int64_t *)t->offsets.p, c.stg, out, c.corder);
        else
            hipLaunchKernelGGL(rt::k_compact3<false>, dim3(4u * (unsigned)c.n_whole_waves), dim3(256), 0, s, c.d_whole,
                               (const int32_t *)t->counts.p, (const int64_t *)t->offsets.p, c.stg, out, c.sp, c.corder);
    }
    if (t->n > 0 && c.split)
        hipLaunchKernelGGL(rt::k_compact3<true>, dim3(4u * (unsigned)t->n_vwaves), dim3(256), 0, s, t->d,
                           (const int32_t *)t->counts.p, (const int64_t *)t->offsets.p, c.stg_pieces, out, c.sp, (const int32_t *)nullptr);
}

// Option "compact" = 0 leaves the records staged; whoever needs the 44-B records (fetch, device pointers, τ) gets them here.
int ensure_compacted(rt_tracks *t) {
    if (t->compacted) return RT_SUCCESS;
    if (!t->cplan.staged) { set_error("the last rt_segmentize left no staged records"); return RT_ERR_NOT_SEGMENTIZED; }
    rt::DOut out{};
    if (int rc = reserve_records(t, t->total, out)) return rc;
    out.delta_s = rt::as_global(t->delta_s.p);
    launch_compaction(t, out, t->mesh->stream);
    RT_HIP(hipStreamSynchronize(t->mesh->stream));
    RT_HIP(hipGetLastError());
    t->compacted = true;
    return RT_SUCCESS;
}

// rt_sweep over a two-phase call's staging: the (ℓ, cell) rows, written by the call itself ("compact" = 0) or here on first use.
int ensure_rows(rt_tracks *t) {
    if (t->sw_ell_valid) return RT_SUCCESS;
    rt::DOut out{};
    out.delta_s = rt::as_global(t->delta_s.p);
    if (int rc = launch_materialise(t, out, t->mesh->stream, false, true, false, nullptr)) return rc;
    RT_HIP(hipGetLastError());
    t->sw_ell_valid = true;
    return RT_SUCCESS;
}

void free_tracks(rt_tracks *t) {
    t->px.release(); t->py.release(); t->phi.release(); t->cs.release(); t->sn.release();
    t->A.release(); t->B.release(); t->C.release(); t->ell.release(); t->azim.release(); t->perm.release(); t->perm_whole.release(); t->corder.release();
    t->in_arena.release(); t->cnt_slot.release(); t->off_slot.release();
    t->counts.release(); t->status.release(); t->element.release(); t->offsets.release();
    t->tile_sums.release(); t->ctl.release(); t->vacc.release();
#ifdef RT_TIMING
    t->dbg.release();
#endif
    if (t->h_ctl) (void)hipHostFree(t->h_ctl);
    if (t->pin_off) (void)hipHostFree(t->pin_off);
    if (t->pin_st) (void)hipHostFree(t->pin_st);
    pin_release_to_cache(t);
    t->spx.release(); t->spy.release(); t->sqx.release(); t->sqy.release(); t->sell.release();
    t->volumes.release(); t->volumes_prev.release(); t->delta_s.release(); t->tau.release(); t->sigma_t.release();
    t->gpx.release(); t->gpy.release(); t->gqx.release(); t->gqy.release();
    t->gelement.release(); t->ctab.release(); t->cowner.release(); t->fst_i.release(); t->fst_v.release();
    t->sw_src.release(); t->sw_w.release(); t->sw_xs.release(); t->sw_psi_in.release(); t->sw_psi_out.release(); t->sw_phi.release();
    t->sw_ell.release(); t->sw_cell.release();
    t->side_px.release(); t->side_py.release(); t->side_qx.release(); t->side_qy.release(); t->side_el.release(); t->marg.release();
    t->vorder.release(); t->vw_wave.release(); t->vw_k.release(); t->w_base.release(); t->w_P.release();
    t->s_el.release(); t->s_eq.release(); t->p_count.release(); t->p_flags.release(); t->p_valid.release(); t->p_rel.release();
    t->s_px.release(); t->s_py.release(); t->s_qx.release(); t->s_qy.release(); t->s_ell.release(); t->p_sum.release();
    for (auto &e : t->ev)
        if (e) (void)hipEventDestroy(e);
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->aux_stream) (void)hipStreamDestroy(t->aux_stream);
    delete t;
}

}  // namespace

// Page-locked staging blocks for the upload of a track set: kept process-wide (pinning costs milliseconds), one per concurrent
// caller (rt_multi_create uploads its shards from several threads).
namespace {
struct StagingBlock { void *p = nullptr; size_t cap = 0; bool busy = false; };
std::vector<StagingBlock> g_staging;
std::mutex g_staging_mutex;
long g_staging_calls = 0;
// allocate: pin a new block when none fits (milliseconds per 10 MB: only worth it for a process that uploads track sets repeatedly)
int staging_acquire(size_t bytes, void **out, bool *allocate_if_missing) {
    std::lock_guard<std::mutex> lk(g_staging_mutex);
    const bool alloc = g_staging_calls++ > 0;  // a process's first track set goes up from the caller's pageable arrays
    if (allocate_if_missing) *allocate_if_missing = alloc;
    for (size_t i = 0; i < g_staging.size(); ++i)
        if (!g_staging[i].busy && g_staging[i].cap >= bytes) { g_staging[i].busy = true; *out = g_staging[i].p; return (int)i; }
    if (!alloc) return -1;
    for (size_t i = 0; i < g_staging.size(); ++i)
        if (!g_staging[i].busy) {  // too small: replace it
            if (g_staging[i].p) (void)hipHostFree(g_staging[i].p);
            g_staging[i] = StagingBlock{};
            const size_t cap = bytes + bytes / 8 + 4096;
            if (hipHostMalloc(&g_staging[i].p, cap, hipHostMallocDefault) != hipSuccess) { g_staging[i].p = nullptr; return -1; }
            g_staging[i].cap = cap; g_staging[i].busy = true; *out = g_staging[i].p;
            return (int)i;
        }
    StagingBlock b;
    const size_t cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&b.p, cap, hipHostMallocDefault) != hipSuccess) return -1;
    b.cap = cap; b.busy = true;
    g_staging.push_back(b);
    *out = b.p;
    return (int)g_staging.size() - 1;
}
void staging_release(int slot) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_staging_mutex);
    g_staging[(size_t)slot].busy = false;
}
// f(i0, i1) over [0, n) on a few host threads (results must not depend on the split); what a worker throws is rethrown here
template <typename F>
void par_ranges(size_t n, size_t grain, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)nt, (size_t)16, n / std::max<size_t>(grain, 1) + 1}));
    if (nt == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);
    size_t done = 0;
    try {
        for (unsigned k = 0; k + 1 < nt; ++k) {
            const size_t i0 = n * k / nt, i1 = n * (k + 1) / nt;
            std::exception_ptr *slot = &err[k];
            th.emplace_back([=, &f]() { try { f(i0, i1); } catch (...) { *slot = std::current_exception(); } });
            done = i1;
        }
    } catch (...) {  // no thread to be had: the caller's thread does the rest
    }
    try { f(done, n); } catch (...) { err[nt - 1] = std::current_exception(); }
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}
}  // namespace


// ------------------------------------------------------------------- C ABI ---------------
extern "C" {

int32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char *rt_last_error(void) { return g_last_error.c_str(); }

const char *rt_status_message(int32_t status) {
    switch (status) {
        case RT_TRACK_OK: return "";
        case RT_TRACK_LOCATE_FAILED:
            return "Try increasing `k`. If the problem persists, raise an issue, this might be a case that "
                   "hasn't been presented before.";
        case RT_TRACK_LENGTH_MISMATCH:
            return "Track with `uid` %d has a length that do not match the sum of its segments lengths with the "
                   "provided tolerance `rtol`. Check whether this is an actual error or increase `rtol`.";
        case RT_TRACK_UNDEF_INTERSECTION: return "UndefVarError: `x_int1` not defined";
        case RT_TRACK_ITER_CAP: return "segmentize!: iteration cap reached while stepping by `tiny_step` (no progress).";
        default: return "unknown track status";
    }
}

int32_t rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static rt_mesh *mesh_create_impl(int32_t device, const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes,
                                 int32_t n_cells, const int32_t *node_cells_ptrs, const int32_t *node_cells_data, const double *bb);
static rt_tracks *tracks_create_impl(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py, const double *phi,
                                     const double *cos_phi, const double *sin_phi, const double *A, const double *B, const double *C,
                                     const double *ell, const int32_t *azim_idx);
static int64_t segmentize_impl(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2);

// No C++ exception may cross the C ABI (a Julia ccall or a ctypes caller would end in std::terminate): the entry points
// that allocate host memory catch what the standard library throws and report it through rt_last_error.
rt_mesh *rt_mesh_create(int32_t device, const double *x, const double *y, int32_t n_nodes,
                        const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                        const int32_t *node_cells_data, const double *bb) {
    try {
        return mesh_create_impl(device, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb);
    } catch (const std::exception &e) {
        set_error("rt_mesh_create: %s", e.what());
        return nullptr;
    }
}
rt_tracks *rt_tracks_create(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py, const double *phi,
                            const double *cos_phi, const double *sin_phi, const double *A, const double *B, const double *C,
                            const double *ell, const int32_t *azim_idx) {
    try {
        return tracks_create_impl(mesh, n_tracks, px, py, phi, cos_phi, sin_phi, A, B, C, ell, azim_idx);
    } catch (const std::exception &e) {
        set_error("rt_tracks_create: %s", e.what());
        return nullptr;
    }
}
int64_t rt_segmentize(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
    try {
        return segmentize_impl(t, tiny_step, k, rtol, delta_s, n_azim_2);
    } catch (const std::exception &e) {
        set_error("rt_segmentize: %s", e.what());
        return RT_ERR_INVALID;
    }
}

static rt_mesh *mesh_create_impl(int32_t device, const double *x, const double *y, int32_t n_nodes,
                        const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                        const int32_t *node_cells_data, const double *bb) {
    if (!x || !y || !cell_nodes || !node_cells_ptrs || !node_cells_data || !bb || n_nodes <= 0 || n_cells <= 0) {
        set_error("rt_mesh_create: null pointer or empty mesh");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("rt_mesh_create: no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_error("rt_mesh_create: device %d out of range [0,%d)", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    rt_mesh *m = new rt_mesh();
    struct Guard { rt_mesh *p; ~Guard() { if (p) free_mesh(p); } } guard{m};  // released on success
    m->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->n_cus = cus;
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0) m->lds_per_block = lds;
    }
    if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        return nullptr;
    }
    m->stream = m->own_stream;
    if (build_mesh(m, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb) != RT_SUCCESS) return nullptr;
    // development knob: RT_OPTIONS="name=value,name=value" applies rt_set_option at creation
    if (const char *env = getenv("RT_OPTIONS")) {
        std::string e(env);
        size_t pos = 0;
        while (pos < e.size()) {
            size_t end = e.find(',', pos);
            if (end == std::string::npos) end = e.size();
            const std::string kv = e.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) (void)rt_set_option(m, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1));
            pos = end + 1;
        }
    }
    guard.p = nullptr;
    return m;
}

void rt_mesh_destroy(rt_mesh *mesh) {
    if (!mesh) return;
    (void)hipSetDevice(mesh->device);
    free_mesh(mesh);
}

int32_t rt_mesh_set_stream(rt_mesh *mesh, void *hip_stream) {
    if (!mesh) { set_error("null mesh"); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(mesh->device));
    RT_HIP(hipStreamSynchronize(mesh->stream));  // (a stream-ordered call, option "async", may still be on the old stream)
    mesh->stream = hip_stream ? (hipStream_t)hip_stream : mesh->own_stream;
    return RT_SUCCESS;
}
void *rt_mesh_get_stream(rt_mesh *mesh) { return mesh ? (void *)mesh->stream : nullptr; }

int32_t rt_mesh_set_enqueue_hook(rt_mesh *mesh, rt_enqueue_hook hook, void *user) {
    if (!mesh) { set_error("rt_mesh_set_enqueue_hook: null mesh"); return RT_ERR_INVALID; }
    mesh->enqueue_hook = hook;
    mesh->enqueue_hook_user = user;
    return RT_SUCCESS;
}

int32_t rt_set_option(rt_mesh *mesh, const char *name, int64_t value) {
    if (!mesh || !name) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!strcmp(name, "iter_cap")) { mesh->iter_cap = value > 0 ? value : 4000000; return RT_SUCCESS; }
    if (!strcmp(name, "volumes_mode")) { mesh->volumes_mode = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "single_pass")) { mesh->single_pass = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "split")) { mesh->split = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "fuse_volumes")) { mesh->fuse_volumes = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "compact")) { mesh->compact = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_gp")) { mesh->sweep_gp = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_ell")) { mesh->sweep_ell = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_waves")) { mesh->sweep_waves = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_debug")) { mesh->sweep_debug = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "compact_debug")) { mesh->compact_debug = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "compact_kernel")) { mesh->compact_kernel = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "march_waves")) { mesh->march_waves = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "first")) { mesh->first = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "lds_records")) { mesh->lds_records = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "topo")) { mesh->topo = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RT_SUCCESS; }
    if (!strcmp(name, "timing")) { mesh->timing = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "async")) { mesh->async_calls = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "hybrid")) { mesh->hybrid = value != 0; return RT_SUCCESS; }          // read by rt_tracks_create
    if (!strcmp(name, "hybrid_pct")) { mesh->hybrid_pct = (int)std::min<int64_t>(95, std::max<int64_t>(30, value)); return RT_SUCCESS; }
    if (!strcmp(name, "pool_chunks_hint")) { mesh->pool_chunks_hint = value; return RT_SUCCESS; }
    if (!strcmp(name, "test_out_records")) { mesh->test_out_records = value; return RT_SUCCESS; }
    if (!strcmp(name, "test_volumes_fallback")) { mesh->test_volumes_fallback = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "test_exact_sums")) { mesh->test_exact_sums = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "test_tally_tau")) { mesh->test_tally_tau = value; return RT_SUCCESS; }
    if (!strcmp(name, "side_entries_hint")) { mesh->side_entries_hint = value; return RT_SUCCESS; }
    if (!strcmp(name, "sort_mode")) { mesh->sort_mode = (int)value; return RT_SUCCESS; }  // read by rt_tracks_create
    if (!strcmp(name, "walk")) {  // 0: generic step only (literal emulation), 1: certified walk step + generic fallback
        mesh->d.walk_ok = (value != 0 && mesh->walk_available) ? 1 : 0;
        return RT_SUCCESS;
    }
    set_error("unknown option '%s'", name);
    return RT_ERR_INVALID;
}

static rt_tracks *tracks_create_impl(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py,
                            const double *phi, const double *cos_phi, const double *sin_phi, const double *A,
                            const double *B, const double *C, const double *ell, const int32_t *azim_idx) {
    if (!mesh || n_tracks < 0 || n_tracks > 0x7fffffff ||
        (n_tracks > 0 && (!px || !py || !phi || !cos_phi || !sin_phi || !A || !B || !C || !ell || !azim_idx))) {
        set_error("rt_tracks_create: null pointer or bad track count");
        return nullptr;
    }
    if (hipSetDevice(mesh->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    rt_tracks *t = new rt_tracks();
    struct Guard { rt_tracks *p; ~Guard() { if (p) free_tracks(p); } } guard{t};  // released on success
    t->mesh = mesh;
    t->n = n_tracks;
    hipStream_t s = mesh->stream;
    const size_t n = (size_t)n_tracks;
    // march order (default 2): waves of 64 CONSECUTIVE uids, longest wave first.  Neighbouring
    // tracks of one angle cross the same cells at the same time (shared walk records, coherent
    // branches) and have nearly equal lengths; sorting individual tracks by length measured 20 %
    // slower because it scatters the lanes of a wave over the whole mesh.
    // (one call is what the reference makes, src/trackgenerator.jl:357-369: the host's share of it — wave maxima, the fills,
    //  the copy into the staging block — runs on a few threads; the sorts are over waves, not tracks)
    std::vector<int32_t> perm(n);
    if (mesh->sort_mode == 1) {
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return ell[a] > ell[b]; });
    } else if (mesh->sort_mode == 2) {
        const size_t nw = (n + 63) / 64;
        std::vector<double> wmax(nw, 0.0);
        par_ranges(nw, 512, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                double mx = 0.0;
                for (size_t i = w * 64; i < std::min(n, w * 64 + 64); ++i) mx = std::max(mx, ell[i]);
                wmax[w] = mx;
            }
        });
        std::vector<int32_t> worder(nw);
        std::iota(worder.begin(), worder.end(), 0);
        std::stable_sort(worder.begin(), worder.end(), [&](int32_t a, int32_t b) { return wmax[a] > wmax[b]; });
        // the batch's last wave of uids may be partial: the slots behind it are packed (no padding), so its position shifts them
        std::vector<size_t> first(nw + 1, 0);
        for (size_t w = 0; w < nw; ++w) first[w + 1] = first[w] + std::min<size_t>(64, n - (size_t)worder[w] * 64);
        par_ranges(nw, 512, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                size_t k2 = first[w];
                for (size_t l = 0; l < 64 && (size_t)worder[w] * 64 + l < n; ++l) perm[k2++] = (int32_t)(worder[w] * 64 + l);
            }
        });
    } else {
        std::iota(perm.begin(), perm.end(), 0);
    }
    std::vector<int32_t> h_corder;
    if (nw_all_early(n) > 4096) {  // batches of many rounds: compaction in output order (measured -8 % at 16 k waves, +1.5 % at 2 k)
        const size_t nw = (n + 63) / 64;
        h_corder.resize(nw);
        std::iota(h_corder.begin(), h_corder.end(), 0);
        std::stable_sort(h_corder.begin(), h_corder.end(), [&](int32_t a, int32_t b) { return perm[(size_t)a * 64] < perm[(size_t)b * 64]; });
    }
    {
        // Σℓ in a fixed order (blocks of 4096 tracks, added in block order) whatever the number of threads; range of azim_idx
        const size_t nb = (n + 4095) / 4096;
        std::vector<double> bs(nb, 0.0);
        std::vector<int32_t> bmin(nb, 0x7fffffff), bmax(nb, (int32_t)0x80000000);
        par_ranges(nb, 16, [&](size_t b0, size_t b1) {
            for (size_t b = b0; b < b1; ++b) {
                double sl = 0.0;
                int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
                for (size_t i = b * 4096; i < std::min(n, b * 4096 + 4096); ++i) { sl += ell[i]; lo = std::min(lo, azim_idx[i]); hi = std::max(hi, azim_idx[i]); }
                bs[b] = sl; bmin[b] = lo; bmax[b] = hi;
            }
        });
        for (size_t b = 0; b < nb; ++b) { t->sum_ell += bs[b]; t->azim_min = std::min(t->azim_min, bmin[b]); t->azim_max = std::max(t->azim_max, bmax[b]); }
    }
    if (n > 0) {
        if (t->azim_min < 1) {  // δs[azim_idx] is read on the device (fill_volumes, src/trackgenerator.jl:379-382)
            set_error("rt_tracks_create: azim_idx must be 1-based (smallest value %d)", t->azim_min);
            return nullptr;
        }
    }
    // split plan: pieces per wave of 64 consecutive uids, canonical numbering, dispatch order
    std::vector<int32_t> h_vorder, h_vw_wave, h_vw_k, h_w_base, h_w_P, h_perm_whole;
    // Splitting pays when the batch has too few waves to fill the chip (the march is then bound by its
    // longest dependent chain: -36 % at 6.5 k tracks, -50 % at 420).  On a batch whose waves are all resident
    // anyway the split variant of the march plus its seed and resolve kernels costs more than shorter
    // chains win back (+17 % march time at 130 k tracks even with one piece per track).
    const size_t nw_all = (n + 63) / 64;
    const bool auto_split = mesh->split < 0 && nw_all < 1536;
    const int p_auto = auto_split ? (int)std::min<size_t>(16, (2048 + nw_all - 1) / std::max<size_t>(1, nw_all)) : 1;
    if ((mesh->split > 0 || (auto_split && p_auto > 1)) && n > 0) {
        const size_t nw = (n + 63) / 64;
        h_w_base.resize(nw); h_w_P.resize(nw);
        std::vector<double> piece_len;
        int32_t nv = 0;
        for (size_t w = 0; w < nw; ++w) {
            double lmax = 0.0;
            for (size_t l = 0; l < 64 && w * 64 + l < n; ++l) lmax = std::max(lmax, ell[w * 64 + l]);
            const double est = lmax * mesh->kappa;  // expected segments of the longest track of the wave
            int32_t P = mesh->split > 0 ? std::min(8, (int32_t)std::ceil(est / (double)mesh->split))
                                        : std::min(p_auto, (int32_t)(est / 12.0));  // auto: pieces of >= ~12 segments
            P = std::max(1, P);
            if (est > 0.5 * rt::kMaxIter) P = 1;  // MAX_ITER (src/track.jl:104) counts whole tracks
            h_w_base[w] = nv; h_w_P[w] = P;
            for (int32_t k = 0; k < P; ++k) { h_vw_wave.push_back((int32_t)w); h_vw_k.push_back(k); piece_len.push_back(lmax / P); }
            nv += P;
        }
        h_vorder.resize(nv);
        std::iota(h_vorder.begin(), h_vorder.end(), 0);
        std::stable_sort(h_vorder.begin(), h_vorder.end(), [&](int32_t a, int32_t b) { return piece_len[a] > piece_len[b]; });
        t->n_vwaves = nv;
    } else if (mesh->split < 0 && mesh->hybrid && mesh->sort_mode == 2 && !auto_split && n > 0) {
        // Hybrid plan for batches that fill the chip.  The march lasts as long as its longest track's chain while the
        // mean track is half as long: only the waves whose expected segment count exceeds hybrid_pct of the longest are
        // cut into pieces (marched by the split kernel on a second stream); all others keep the lean whole-track kernel.
        const size_t nw = nw_all;
        std::vector<double> west(nw, 0.0);
        double est_max = 0.0;
        for (size_t w = 0; w < nw; ++w) {
            double lmax = 0.0;
            for (size_t l = 0; l < 64 && w * 64 + l < n; ++l) lmax = std::max(lmax, ell[w * 64 + l]);
            west[w] = lmax * mesh->kappa;
            est_max = std::max(est_max, west[w]);
        }
        const double T = std::max(24.0, 0.01 * mesh->hybrid_pct * est_max);
        h_w_base.assign(nw, 0); h_w_P.assign(nw, 0);
        std::vector<double> piece_len;
        int32_t nv = 0;
        for (size_t w = 0; w < nw; ++w) {
            if (!(west[w] > T) || west[w] > 0.5 * rt::kMaxIter) continue;  // (MAX_ITER counts whole tracks, src/track.jl:104)
            const int32_t P = std::min(4, (int32_t)std::ceil(west[w] / T));
            if (P < 2) continue;
            h_w_base[w] = nv; h_w_P[w] = P;
            for (int32_t k = 0; k < P; ++k) { h_vw_wave.push_back((int32_t)w); h_vw_k.push_back(k); piece_len.push_back(west[w] / P); }
            nv += P;
        }
        if (nv > 0) {
            h_vorder.resize(nv);
            std::iota(h_vorder.begin(), h_vorder.end(), 0);
            std::stable_sort(h_vorder.begin(), h_vorder.end(), [&](int32_t a, int32_t b) { return piece_len[a] > piece_len[b]; });
            t->n_vwaves = nv;
            t->hybrid = true;
            // the whole-track march's order: the remaining waves, longest first (the order `perm` already has)
            for (size_t i = 0; i < n; ++i)
                if (h_w_P[(size_t)perm[i] / 64] == 0) h_perm_whole.push_back(perm[i]);
            t->n_whole = (int64_t)h_perm_whole.size();
        } else {
            h_w_base.clear(); h_w_P.clear();
        }
    }
    // One device allocation, one page-locked staging block (kept process-wide), one host-to-device copy: the eleven pageable
    // uploads into eleven allocations of round 3 were 31.7 ms of a C5 call whose kernels take 3.
    bool ok = true;
    {
        const size_t na = (n + 31) & ~(size_t)31;  // every array starts on a 256-B boundary
        const size_t ncord = (h_corder.size() + 63) & ~(size_t)63;
        const size_t bytes = 12 * na * sizeof(double) + 3 * na * sizeof(int32_t) + ncord * sizeof(int32_t) + 256;
        void *stage = nullptr;
        bool may_pin = false;
        const int slot = staging_acquire(bytes, &stage, &may_pin);
        struct Rel { int s; ~Rel() { staging_release(s); } } rel{slot};
        ok = t->in_arena.reserve(bytes) == hipSuccess;
        unsigned char *db = t->in_arena.p;
        const double *src8[9] = {px, py, phi, cos_phi, sin_phi, A, B, C, ell};
        DevView<double> *dst8[12] = {&t->px, &t->py, &t->phi, &t->cs, &t->sn, &t->A, &t->B, &t->C, &t->ell, &t->As, &t->Bs, &t->Cs};
        if (ok) {
            for (int a = 0; a < 12; ++a) dst8[a]->p = (double *)(db + (size_t)a * na * sizeof(double));
            t->azim.p = (int32_t *)(db + 12 * na * sizeof(double)); t->perm.p = t->azim.p + na; t->iperm.p = t->perm.p + na;
            t->corder.p = h_corder.empty() ? nullptr : t->iperm.p + na;
        }
        // the host image of the arena: the staging block, or — a process's first track set, before anything is pinned — only the
        // derived arrays in a pageable vector (the caller's arrays then go up from where they lie)
        std::vector<unsigned char> derived;
        unsigned char *hb = (unsigned char *)stage;
        const size_t derived_off = 9 * na * sizeof(double);
        if (ok && slot < 0) derived.resize(bytes - derived_off);
        if (ok) {
            unsigned char *hd = slot >= 0 ? hb + derived_off : derived.data();  // the derived arrays' part of the image
            double *h_As = (double *)hd, *h_Bs = h_As + na, *h_Cs = h_Bs + na;
            int32_t *h_az = (int32_t *)(hd + 3 * na * sizeof(double)), *h_pm = h_az + na, *h_ip = h_pm + na, *h_co = h_ip + na;
            par_ranges(n, 16384, [&](size_t i0, size_t i1) {
                if (slot >= 0)
                    for (int a = 0; a < 9; ++a) memcpy((double *)(hb + (size_t)a * na * sizeof(double)) + i0, src8[a] + i0, (i1 - i0) * sizeof(double));
                memcpy(h_az + i0, azim_idx + i0, (i1 - i0) * sizeof(int32_t));
                memcpy(h_pm + i0, perm.data() + i0, (i1 - i0) * sizeof(int32_t));
                for (size_t i = i0; i < i1; ++i) {  // (slot i holds track perm[i])
                    const int32_t u = perm[i];
                    h_As[i] = A[u]; h_Bs[i] = B[u]; h_Cs[i] = C[u];
                    h_ip[u] = (int32_t)i;
                }
            });
            if (!h_corder.empty()) memcpy(h_co, h_corder.data(), h_corder.size() * sizeof(int32_t));
            if (slot >= 0) {
                ok = hipMemcpyAsync(db, hb, bytes - 256, hipMemcpyHostToDevice, s) == hipSuccess;
            } else {
                for (int a = 0; a < 9 && ok && n > 0; ++a) ok = hipMemcpyAsync(dst8[a]->p, src8[a], n * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
                ok = ok && hipMemcpyAsync(db + derived_off, derived.data(), derived.size() - 256, hipMemcpyHostToDevice, s) == hipSuccess;
            }
            ok = ok && hipStreamSynchronize(s) == hipSuccess;
        }
        ok = ok && t->cnt_slot.reserve(na + 64) == hipSuccess && t->off_slot.reserve(na + 64) == hipSuccess;
    }
    if (ok && t->n_vwaves > 0) {
        const size_t np = (size_t)t->n_vwaves * 64;
        ok = upload(t->vorder, h_vorder.data(), h_vorder.size(), s) == 0 && upload(t->vw_wave, h_vw_wave.data(), h_vw_wave.size(), s) == 0 &&
             upload(t->vw_k, h_vw_k.data(), h_vw_k.size(), s) == 0 && upload(t->w_base, h_w_base.data(), h_w_base.size(), s) == 0 &&
             upload(t->w_P, h_w_P.data(), h_w_P.size(), s) == 0 && t->s_el.reserve(np) == hipSuccess && t->s_eq.reserve(np) == hipSuccess &&
             t->p_count.reserve(np) == hipSuccess && t->p_flags.reserve(np) == hipSuccess && t->p_valid.reserve(np) == hipSuccess &&
             t->p_rel.reserve(np) == hipSuccess && t->s_px.reserve(np) == hipSuccess && t->s_py.reserve(np) == hipSuccess &&
             t->s_qx.reserve(np) == hipSuccess && t->s_qy.reserve(np) == hipSuccess && t->s_ell.reserve(np) == hipSuccess &&
             t->p_sum.reserve(np) == hipSuccess;
    }
    for (auto &e : t->ev)
        if (ok && hipEventCreate(&e) != hipSuccess) ok = false;
    if (ok && t->hybrid)
        ok = upload(t->perm_whole, h_perm_whole.data(), h_perm_whole.size(), s) == 0 && hipStreamCreateWithFlags(&t->aux_stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) == hipSuccess;
    if (ok && hipStreamSynchronize(s) != hipSuccess) ok = false;
    if (!ok) {
        if (g_last_error.empty()) set_error("rt_tracks_create: upload failed");
        return nullptr;
    }
    rt::DTracks &d = t->d;
    using rt::as_global;
    d.px = as_global(t->px.p); d.py = as_global(t->py.p); d.phi = as_global(t->phi.p); d.cs = as_global(t->cs.p);
    d.sn = as_global(t->sn.p); d.A = as_global(t->A.p); d.B = as_global(t->B.p); d.C = as_global(t->C.p);
    d.ell = as_global(t->ell.p); d.azim = as_global(t->azim.p); d.perm = as_global(t->perm.p);
    d.As = as_global(t->As.p); d.Bs = as_global(t->Bs.p); d.Cs = as_global(t->Cs.p); d.iperm = as_global(t->iperm.p);
    d.cnt_slot = as_global(t->cnt_slot.p); d.off_slot = as_global(t->off_slot.p);
    d.n = n_tracks;
    guard.p = nullptr;
    return t;
}

void rt_tracks_destroy(rt_tracks *tracks) {
    if (!tracks) return;
    (void)hipSetDevice(tracks->mesh->device);
    (void)finish_call(tracks);
    free_tracks(tracks);
}

#ifdef RT_HOST_TIMING
static double g_ht[6];
static long g_hn;
static inline double ht_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif
static int64_t segmentize_impl(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s,
                      int32_t n_azim_2) {
#ifdef RT_HOST_TIMING
    const double ht0 = ht_now();
#endif
    if (!t || !delta_s || n_azim_2 <= 0) { set_error("rt_segmentize: bad arguments"); return RT_ERR_INVALID; }
    if (k < 0) {  // knn(kdtree, x, k, ...) rejects a negative k (src/mesh.jl:123); any k >= 0 is honoured
        set_error("rt_segmentize: k = %d (must be >= 0)", k);
        return RT_ERR_INVALID;
    }
    if (t->n > 0 && t->azim_max > n_azim_2) {
        set_error("rt_segmentize: track azim_idx reaches %d but delta_s has n_azim_2 = %d entries", t->azim_max, n_azim_2);
        return RT_ERR_INVALID;
    }
    rt_mesh *m = t->mesh;
    RT_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int64_t n = t->n;
    t->segmentized = false;
    t->tau_groups = 0;  // τ of the previous records is void
    for (double &v : t->ms) v = 0.0;

    rt::DParams prm;
    prm.tiny_step = tiny_step; prm.rtol = rtol; prm.k = k; prm.n_azim_2 = n_azim_2; prm.iter_cap = m->iter_cap;
    prm.topo_tiny_max = m->topo_tiny_max; prm.topo_rmax = m->topo_rmax; prm.topo_end_err = m->topo_end_err;
    prm.topo_force = m->topo == 2 ? 1 : 0; prm.pad_ = 0;
    prm.tally_tau = m->test_tally_tau != 0 ? (m->test_tally_tau < 0 ? (double)INFINITY : 1e-12 * (double)m->test_tally_tau) : m->tally_tau;

    const int64_t n_tiles = (n + rt::kScanTile - 1) / rt::kScanTile;
    const int64_t n_waves = (n + 63) / 64;
    RT_HIP(t->counts.reserve(n + 1));
    RT_HIP(t->status.reserve(n + 1));
    RT_HIP(t->offsets.reserve(n + 1));
    RT_HIP(t->tile_sums.reserve(n_tiles + 1));
    RT_HIP(t->ctl.reserve(2 * rt::kCtlWords));
    RT_HIP(t->vacc.reserve(m->n_cells));
    if (!t->h_ctl) {
        RT_HIP(hipHostMalloc((void **)&t->h_ctl, (2 * rt::kCtlWords + 8) * sizeof(unsigned long long), hipHostMallocDefault));
        for (int i = 0; i < 2 * rt::kCtlWords + 8; ++i) t->h_ctl[i] = 0;  // (h_res[kCtlWords]: the sequence number of the call it holds)
        t->h_ctl[1] = ~0ull;  // first failing uid: atomicMin target
    }
    // the call's control block: calls alternate between two, and the scan of a call resets the other one for the next call —
    // in the steady state no reset kernel runs in front of the march (its launch gap was 5 µs of every step)
    const int cb = t->ctl_idx;
    unsigned long long *const d_ctl = t->ctl.p + (size_t)cb * rt::kCtlWords;
    unsigned long long *const d_ctl_other = t->ctl.p + (size_t)(1 - cb) * rt::kCtlWords;
    const bool ctl_was_clean = t->ctl_clean[cb];
    const int64_t ctl_was_first = t->ctl_first_chunk[cb];
    const bool vacc_was_clean = t->vacc_clean;
    t->ctl_clean[0] = t->ctl_clean[1] = false;  // (set again when this call has succeeded)
    t->vacc_clean = false;
    unsigned long long *const d_fail = d_ctl;
    int64_t *const d_total = reinterpret_cast<int64_t *>(d_ctl + 16);
    int32_t *const d_cursor = reinterpret_cast<int32_t *>(d_ctl + 18);
    unsigned long long *const h_res = t->h_ctl + rt::kCtlWords;
    // the same pinned block as the device sees it (k_scan_tile_sums writes it); looked up once per handle
    if (!t->h_res_dev) RT_HIP(hipHostGetDevicePointer((void **)&t->h_res_dev, h_res, 0));
    unsigned long long *const h_res_dev = t->h_res_dev;
    std::swap(t->volumes, t->volumes_prev);  // a consumer may still be all-reducing the previous call's volumes
    RT_HIP(t->volumes.reserve(m->n_cells));
    if (t->h_delta_s.size() != (size_t)n_azim_2 || memcmp(t->h_delta_s.data(), delta_s, sizeof(double) * n_azim_2) != 0) {
        t->h_delta_s.assign(delta_s, delta_s + n_azim_2);
        if (int rc = upload(t->delta_s, t->h_delta_s.data(), (size_t)n_azim_2, s)) return rc;
    }

    rt::DOut out{};
    using rt::as_global;
    out.volumes = as_global(t->volumes.p);  // (single pass with fused fill_volumes: the accumulator `vacc`, see below)
    out.delta_s = as_global(t->delta_s.p);
    out.fused_volumes = (m->volumes_mode == 1 && !m->single_pass) ? 1 : 0;
    out.dbg = m->compact_debug;
    rt::DStage stg{};
    rt::DSplit sp{};
    // Track pieces (DSplit): every wave of a batch too small to fill the chip, or — hybrid plan — only the longest waves
    // of a full batch, beside the whole-track march of the rest.  The hybrid plan needs the fused-volumes kernels of the
    // usual k; a call that cannot use it (or any plan, once a track reached MAX_ITER segments) marches every track whole.
    const bool widek_ = k > rt::kMaxK;
    const size_t hist_bytes_ = (size_t)m->n_cells * sizeof(double);
    int fuse_waves_ = (3 * (hist_bytes_ + 4 * rt::kMaxChunks * sizeof(int32_t)) <= 158 * 1024 || (n + 63) / 64 > 3072) ? 4 : 6;
    if (m->march_waves == 4 || m->march_waves == 6) fuse_waves_ = m->march_waves;  // (experiments)
    const bool fuse_ = m->volumes_mode == 2 && m->fuse_volumes && 2 * (hist_bytes_ + fuse_waves_ * rt::kMaxChunks * sizeof(int32_t)) <= 158 * 1024 && !widek_;
    // Option "compact" = 0: stop after march + scan (a device-resident consumer, rt_sweep, reads the staged rows); the separate
    // volumes pass needs the compact records, so a call that cannot fuse fill_volumes compacts anyway.  Whole tracks only.
    const bool do_compact = m->compact || !fuse_ || !m->single_pass;
    const bool plan_ok = m->single_pass && t->n_vwaves > 0 && !t->force_unsplit && do_compact;
    const bool hybrid = plan_ok && t->hybrid && fuse_;
    const bool split = plan_ok && (!t->hybrid || hybrid);  // pieces are marched in this call
    // Cheap steps (k_march<..., TOPO>): whole-track batches on meshes with cheap-step records, the usual k, fill_volumes fused.
    const bool topo = m->single_pass && m->topo && m->topo_available && m->d.walk_ok && !split && !hybrid && !widek_ && n > 0 &&
                      fuse_ && tiny_step > 0 && tiny_step <= m->topo_tiny_max && m->lds_records == 0 &&
                      (m->topo == 2 || 10 * m->n_records_topo >= 9 * m->n_records_walk);
    t->last_topo = topo ? 1 : 0;
    if (split) {
        sp.vorder = as_global(t->vorder.p); sp.vw_wave = as_global(t->vw_wave.p); sp.vw_k = as_global(t->vw_k.p);
        sp.w_base = as_global(t->w_base.p); sp.w_P = as_global(t->w_P.p);
        sp.s_el = as_global(t->s_el.p); sp.s_eq = as_global(t->s_eq.p);
        sp.s_px = as_global(t->s_px.p); sp.s_py = as_global(t->s_py.p); sp.s_qx = as_global(t->s_qx.p);
        sp.s_qy = as_global(t->s_qy.p); sp.s_ell = as_global(t->s_ell.p);
        sp.p_count = as_global(t->p_count.p); sp.p_flags = as_global(t->p_flags.p); sp.p_sum = as_global(t->p_sum.p);
        sp.p_valid = as_global(t->p_valid.p); sp.p_rel = as_global(t->p_rel.p);
        sp.n_vwaves = t->n_vwaves;
    }
    const unsigned grid = (unsigned)n_waves;
    // compaction order of the whole-track waves (only when every track marches whole with the full march order)
    const int32_t *corder = (t->corder.p && m->single_pass && !t->n_vwaves) ? (const int32_t *)t->corder.p : (const int32_t *)nullptr;
    int64_t total = 0;
    unsigned long long fi[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float f = 0;

    // copy_out: k_scan_tile_sums' last block also writes the control block to the pinned host copy; scale: k_scan_write also
    // applies volumes ./= n_azim_2 (fused fill_volumes only: `volumes` is final once the march has ended)
    int32_t first_chunk_this_call = 0, side_first_this_call = 0;
    // reset_other: the scan's last block also resets the OTHER control block for the next call (single-pass calls)
    auto scan_counts = [&](bool copy_out, bool scale, bool reset_other, bool slot_order = false) -> int {
        if (n > 0) {
            hipLaunchKernelGGL(rt::k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, n,
                               t->tile_sums.p, n_tiles, d_total, reinterpret_cast<unsigned int *>(d_ctl + 20),
                               (const unsigned long long *)d_ctl, copy_out ? h_res_dev : (unsigned long long *)nullptr,
                               reset_other ? d_ctl_other : (unsigned long long *)nullptr, first_chunk_this_call, side_first_this_call,
                               ++t->call_seq);
            hipLaunchKernelGGL(rt::k_scan_write, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, n,
                               t->tile_sums.p, d_total, t->offsets.p, scale ? t->volumes.p : (double *)nullptr, m->n_cells,
                               (double)n_azim_2, t->vacc.p, slot_order ? (const int32_t *)t->iperm.p : (const int32_t *)nullptr, t->off_slot.p);
        } else {
            RT_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), s));
            RT_HIP(hipMemsetAsync(t->offsets.p, 0, sizeof(int64_t), s));
        }
        return RT_SUCCESS;
    };
    auto reserve_out = [&](int64_t tot) -> int { return reserve_records(t, tot, out); };
    // HIP events between the kernels (rt_last_timing) only on request: each costs ≈4 µs of stream time
    auto rec = [&](int i) -> int {
        if (m->timing) RT_HIP(hipEventRecord(t->ev[i], s));
        return RT_SUCCESS;
    };
    t->compacted = false;
    t->sw_ell_valid = false;
    t->cplan = rt_tracks::CompactPlan{};
    // fill_volumes as its own pass over the compact records + volumes ./= n_azim_2
    bool fused_volumes_this_call = false;
    bool volumes_pass = true;  // false: fill_volumes rode along with the march and the scan, no ev[6]
    auto launch_volumes = [&]() -> int {
        if (m->volumes_mode == 2 && n > 0 && !fused_volumes_this_call) {
            const int64_t want_blocks = 512;
            int32_t tpb = (int32_t)std::max<int64_t>(1, (n + want_blocks - 1) / want_blocks);
            tpb = std::min(tpb, 4096);
            const int64_t nb = (n + tpb - 1) / tpb;
            const size_t hist_bytes = (size_t)m->n_cells * sizeof(double);
            const size_t rel_bytes = ((size_t)tpb + 1) * sizeof(int32_t);
            const int use_lds = hist_bytes + rel_bytes <= 150 * 1024 ? 1 : 0;
            const size_t shmem = (use_lds ? hist_bytes : 0) + rel_bytes;
            if (shmem > 48 * 1024)
                RT_HIP(hipFuncSetAttribute((const void *)rt::k_volumes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL(rt::k_volumes, dim3((unsigned)nb), dim3(1024), shmem, s, (const int64_t *)t->offsets.p, n,
                               (const int32_t *)t->azim.p, (const double *)t->delta_s.p, (const int32_t *)t->element.p,
                               (const double *)t->sell.p, t->volumes.p, m->n_cells, tpb, use_lds,
                               m->single_pass ? (const int32_t *)(d_cursor + 1) : (const int32_t *)nullptr, out.cap);
        }
        if (!(fused_volumes_this_call && n > 0))  // the fused path scales inside k_scan_write
            hipLaunchKernelGGL(rt::k_scale_volumes, dim3((unsigned)((m->n_cells + 255) / 256)), dim3(256), 0, s, t->volumes.p,
                               m->n_cells, (double)n_azim_2);
        return RT_SUCCESS;
    };

    const bool widek = k > rt::kMaxK;  // find_element's knn fallback beyond the in-register list: separate kernel instantiations
    rt::DFirst fst{};  // (it == nullptr: the march makes every first record itself)
    const int64_t *march_offsets = nullptr;
    hipStream_t march_stream = s;          // (the hybrid path launches its pieces on the auxiliary stream)
    const rt::DTracks *march_tracks = &t->d;
    const rt::DStage *march_stage = &stg;
    auto march = [&]<int MODE, int WAVES, bool SPLIT, bool WIDEK, bool LDSREC = false, bool TOPO = false>(unsigned blocks, size_t smem) -> int {
        t->last_march_waves = WAVES; t->last_split = std::max(t->last_split, SPLIT ? 1 : 0); t->last_widek = WIDEK ? 1 : 0;
        if (smem > 48 * 1024)
            RT_HIP(hipFuncSetAttribute((const void *)rt::k_march<MODE, WAVES, SPLIT, WIDEK, LDSREC, TOPO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((rt::k_march<MODE, WAVES, SPLIT, WIDEK, LDSREC, TOPO>), dim3(blocks), dim3(64 * WAVES), smem, march_stream, m->d, *march_tracks, prm,
                           t->counts.p, t->status.p, march_offsets, out, *march_stage, d_fail, sp, fst);
        return RT_SUCCESS;
    };
    t->last_split = 0;
    if (!m->single_pass) { if (int rc = rec(0)) return rc; }  // single pass: the call is timed from ev[1], after the 2-µs prologue
    if (!m->single_pass) RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
    if (m->single_pass) {
        // ---- staged single-pass march; the pool is sized from the Cauchy–Crofton estimate
        //      (or from what the previous call needed) and grown + re-run on overflow
        // fill_volumes fused into the march when an LDS copy of `volumes` (+ 4 chunk tables) leaves room
        // for two workgroups per CU; larger meshes use the separate k_volumes pass.
        // The march fits three waves per SIMD (12 per CU): four-wave workgroups when three copies of the
        // LDS histogram fit in the CU's 160 KB, six-wave workgroups (two copies) for larger meshes
        // (six-wave workgroups measured -10 % march time on a batch that is resident at once, BWR-like C4, and
        //  +5 % on one that takes many rounds, C5 on one GPU).  A wide k marches with the one-wave kernels only:
        // fewer instantiations of a rare case.
        const size_t hist_bytes = hist_bytes_;
        const int fuse_waves = fuse_waves_;
        const size_t fuse_smem = hist_bytes + fuse_waves * rt::kMaxChunks * sizeof(int32_t);
        const bool fuse = fuse_;
        const bool split_all = split && !hybrid;  // every wave in pieces (small batches, or "split" > 0)
        const int64_t n_whole_waves = hybrid ? (t->n_whole + 63) / 64 : n_waves;
        RT_HIP(t->ctab.reserve((size_t)std::max<int64_t>(1, split_all ? t->n_vwaves : n_whole_waves + (hybrid ? t->n_vwaves : 0)) * rt::kMaxChunks));
        int64_t want = t->chunks_needed_last > 0
                           ? t->chunks_needed_last + t->chunks_needed_last / 16 + 16
                           : (int64_t)(1.3 * (m->kappa * t->sum_ell + (double)n) / (64.0 * rt::kChunkRows)) + 2 * ((split ? t->n_vwaves : 0) + (split_all ? 0 : n_whole_waves)) + 64;
        if (m->pool_chunks_hint > 0 && t->pool_chunks == 0) want = m->pool_chunks_hint;
        fused_volumes_this_call = fuse;
        // the side list of the two-phase march: one reserved entry per march slot (a track's first record) + the records the
        // generic step makes further on — estimated from the share of records without a walk certificate (or the last call's need)
        const int64_t side_static = n_whole_waves * 64;
        int64_t side_want = 0;
        if (topo) {
            const double unwalked = m->n_records > 0 ? 1.0 - (double)m->n_records_walk / (double)m->n_records : 1.0;
            const int64_t dyn = t->side_needed_last > 0 ? t->side_needed_last + t->side_needed_last / 8 + 1024
                                                        : (int64_t)(1.3 * unwalked * m->kappa * t->sum_ell) + n / 16 + 4096;
            side_want = side_static + ((m->side_entries_hint > 0 && t->side_cap == 0) ? m->side_entries_hint : dyn);
        }
        for (int attempt = 0;; ++attempt) {
            if (want > t->pool_chunks || (!topo && t->gqx.cap < (size_t)t->pool_chunks * rt::kChunkRows * 64)) {
                want = std::max(want, t->pool_chunks);
                const size_t slots = (size_t)want * rt::kChunkRows * 64;
                // (q, ±cell) rows with sparse p for the exact march; the two-phase march stages one 4-B word per record
                if (!topo) {
                    RT_HIP(t->gpx.reserve(slots)); RT_HIP(t->gpy.reserve(slots)); RT_HIP(t->gqx.reserve(slots)); RT_HIP(t->gqy.reserve(slots));
                }
                RT_HIP(t->gelement.reserve(slots));
                RT_HIP(t->cowner.reserve((size_t)want));
                t->pool_chunks = want;
            }
            if (topo && side_want > t->side_cap) {
                const size_t ne = (size_t)std::min<int64_t>(side_want, 0x7ffffff0);
                RT_HIP(t->side_px.reserve(ne)); RT_HIP(t->side_py.reserve(ne)); RT_HIP(t->side_qx.reserve(ne)); RT_HIP(t->side_qy.reserve(ne));
                RT_HIP(t->side_el.reserve(ne));
                t->side_cap = (int64_t)ne;
            }
            // The six output arrays are sized from the Cauchy–Crofton estimate of the record count (or from what the
            // previous call produced), not from the pool's slots: march -> scan -> compaction still run back to back
            // without a host sync — the compaction simply does not write beyond the capacity, and in the rare call
            // whose total exceeds it the host grows the arrays and compacts again (the staged rows are still there).
            const int64_t est_records = t->total_last > 0 ? t->total_last + t->total_last / 32 + 4096
                                                          : (int64_t)(1.08 * m->kappa * t->sum_ell) + 2 * n + 4096;
            if (do_compact)
                if (int rc = reserve_out(std::min<int64_t>(m->test_out_records > 0 && t->total_last == 0 ? m->test_out_records : est_records,
                                                           t->pool_chunks * rt::kChunkRows * 64))) return rc;
            stg.px = as_global(t->gpx.p); stg.py = as_global(t->gpy.p); stg.qx = as_global(t->gqx.p);
            stg.qy = as_global(t->gqy.p); stg.element = as_global(t->gelement.p);
            stg.ctab = as_global(t->ctab.p); stg.cowner = as_global(t->cowner.p); stg.cursor = as_global(d_cursor);
            stg.pool_chunks = (int32_t)std::min<int64_t>(t->pool_chunks, 0x7fffffff);
            stg.static0 = (!split && n_whole_waves < stg.pool_chunks) ? 1 : 0;
            if (topo) {
                stg.s_px = as_global(t->side_px.p); stg.s_py = as_global(t->side_py.p); stg.s_qx = as_global(t->side_qx.p);
                stg.s_qy = as_global(t->side_qy.p); stg.s_el = as_global(t->side_el.p);
                stg.side_cap = (int32_t)t->side_cap; stg.side_static = (int32_t)side_static;
            }
#ifdef RT_TIMING
            RT_HIP(t->dbg.reserve((size_t)std::max<int64_t>(1, n_waves) * 4));
            RT_HIP(hipMemsetAsync(t->dbg.p, 0, sizeof(unsigned long long) * 4 * std::max<int64_t>(1, n_waves), s));
            stg.dbg = t->dbg.p;
#endif
            // hybrid: the pieces (split kernel) use the chunk tables behind those of the whole-track waves
            rt::DStage stg_pieces = stg;
            if (hybrid) stg_pieces.ctab = stg.ctab + n_whole_waves * rt::kMaxChunks;
            rt::DTracks d_whole = t->d;
            if (hybrid) { d_whole.perm = as_global(t->perm_whole.p); d_whole.n = t->n_whole; }
            {
                rt_tracks::CompactPlan &c = t->cplan;
                c.stg = stg; c.stg_pieces = stg_pieces; c.d_whole = d_whole; c.sp = sp; c.corder = corder;
                c.n_whole_waves = n_whole_waves; c.split = split; c.split_all = split_all; c.staged = false;
                c.codes = topo; c.rtol = rtol;
            }
            // Everything one attempt puts on the stream(s), as one function.  (Capturing it once into a HIP graph and replaying it
            // was tried: the event-record nodes keep the ≈6-µs gaps between the kernels, and hipEventElapsedTime fails on
            // events that were only ever recorded inside a graph — DESIGN.md §4.)
            // every track's first record ahead of the march (k_first): whole tracks with their reserved first chunks, the usual k
            const bool use_first = m->first && !split && !hybrid && !topo && stg.static0 && !widek && n > 0 && m->lds_records == 0;
            fst = rt::DFirst{};
            if (use_first) {
                const size_t ns = (size_t)n_whole_waves * 64;
                RT_HIP(t->fst_i.reserve(3 * ns)); RT_HIP(t->fst_v.reserve(10 * ns));
                fst.it = as_global(t->fst_i.p); fst.T = as_global(t->fst_i.p + ns); fst.pred = as_global(t->fst_i.p + 2 * ns);
                fst.v = as_global(t->fst_v.p); fst.n_slots = (int64_t)ns;
            }
            t->last_first = use_first ? 1 : 0;
            // fused fill_volumes accumulates into `vacc` (zero between calls: k_scan_write leaves it so); otherwise the separate
            // pass adds into `volumes`, zeroed here.  The reset kernel runs only when the control block or the accumulator is
            // not known to be clean: a handle's first call, a re-run after a pool overflow, a changed number of reserved chunks.
            first_chunk_this_call = stg.static0 ? (int32_t)n_whole_waves : 0;
            side_first_this_call = topo ? (int32_t)side_static : 0;
            const int64_t reset_key = (int64_t)first_chunk_this_call | ((int64_t)side_first_this_call << 32);
            if (fuse && n > 0) out.volumes = as_global(t->vacc.p);
            const bool need_reset = attempt > 0 || !ctl_was_clean || ctl_was_first != reset_key || !(fuse && n > 0 && vacc_was_clean);
            auto enqueue_attempt = [&]() -> int {
                if (need_reset)
                    hipLaunchKernelGGL(rt::k_prologue, dim3((unsigned)((std::max(m->n_cells, rt::kCtlWords) + 255) / 256)), dim3(256), 0, s, d_ctl,
                                       (fuse && n > 0) ? t->vacc.p : t->volumes.p, m->n_cells, first_chunk_this_call, side_first_this_call);
                if (int rc = rec(1)) return rc;
                if (use_first)
                    hipLaunchKernelGGL(rt::k_first, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, m->d, t->d, prm, stg, fst);
                if (n > 0 && split) {
                    hipStream_t ps = s;  // the stream the pieces march on
                    if (hybrid) {
                        ps = t->aux_stream;
                        RT_HIP(hipEventRecord(t->ev_fork, s));
                        RT_HIP(hipStreamWaitEvent(ps, t->ev_fork, 0));
                    }
                    if (widek) hipLaunchKernelGGL(rt::k_seed<true>, dim3((unsigned)t->n_vwaves), dim3(64), 0, ps, m->d, t->d, prm, sp);
                    else hipLaunchKernelGGL(rt::k_seed<false>, dim3((unsigned)t->n_vwaves), dim3(64), 0, ps, m->d, t->d, prm, sp);
                    int rc;
                    march_stream = ps; march_stage = &stg_pieces; march_tracks = &t->d;
                    if (fuse && fuse_waves == 4) rc = march.template operator()<rt::kStage, 4, true, false>((unsigned)((t->n_vwaves + 3) / 4), fuse_smem);
                    else if (fuse) rc = march.template operator()<rt::kStage, 6, true, false>((unsigned)((t->n_vwaves + 5) / 6), fuse_smem);
                    else if (widek) rc = march.template operator()<rt::kStage, 1, true, true>((unsigned)t->n_vwaves, rt::kMaxChunks * sizeof(int32_t));
                    else rc = march.template operator()<rt::kStage, 1, true, false>((unsigned)t->n_vwaves, rt::kMaxChunks * sizeof(int32_t));
                    march_stream = s; march_stage = &stg;
                    if (rc) return rc;
                    hipLaunchKernelGGL(rt::k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ps, t->d, prm, sp, t->counts.p,
                                       t->status.p, d_fail);
                    if (hybrid) RT_HIP(hipEventRecord(t->ev_join, ps));
                }
                if (n > 0 && !split_all) {  // whole tracks: all of them, or those the hybrid plan leaves whole
                    int rc;
                    march_tracks = &d_whole;
                    // experiment: eight-wave workgroups (one per CU) with all walk records in LDS (1), or from L2 as usual (2: its control)
                    const size_t lds_base = ((hist_bytes + 8 * rt::kMaxChunks * sizeof(int32_t) + 15) & ~(size_t)15);
                    const size_t lds_smem = lds_base + (size_t)3 * m->n_cells * sizeof(rt::WalkRec);
                    if (topo && fuse_waves == 4)
                        rc = march.template operator()<rt::kStage, 4, false, false, false, true>((unsigned)((n_whole_waves + 3) / 4), fuse_smem);
                    else if (topo)
                        rc = march.template operator()<rt::kStage, 6, false, false, false, true>((unsigned)((n_whole_waves + 5) / 6), fuse_smem);
                    else if (fuse && m->lds_records == 1 && !hybrid && lds_smem <= 160 * 1024)
                        rc = march.template operator()<rt::kStage, 8, false, false, true>((unsigned)((n_whole_waves + 7) / 8), lds_smem);
                    else if (fuse && m->lds_records == 2 && !hybrid && lds_smem <= 160 * 1024)
                        rc = march.template operator()<rt::kStage, 8, false, false, false>((unsigned)((n_whole_waves + 7) / 8), lds_smem);
                    else if (fuse && fuse_waves == 4) rc = march.template operator()<rt::kStage, 4, false, false>((unsigned)((n_whole_waves + 3) / 4), fuse_smem);
                    else if (fuse) rc = march.template operator()<rt::kStage, 6, false, false>((unsigned)((n_whole_waves + 5) / 6), fuse_smem);
                    else if (widek) rc = march.template operator()<rt::kStage, 1, false, true>((unsigned)n_whole_waves, rt::kMaxChunks * sizeof(int32_t));
                    else rc = march.template operator()<rt::kStage, 1, false, false>((unsigned)n_whole_waves, rt::kMaxChunks * sizeof(int32_t));
                    march_tracks = &t->d;
                    if (rc) return rc;
                    if (hybrid) RT_HIP(hipStreamWaitEvent(s, t->ev_join, 0));
                }
                if (int rc = rec(2)) return rc;
                if (int rc = scan_counts(!topo, fuse && !topo, true, topo)) return rc;  // (two-phase: k_finish scales the volumes, behind k_materialise)
                if (int rc = rec(3)) return rc;  // every event record costs ≈4 µs of stream time: none is recorded twice
                if (topo) {
                    // codes -> records (or, "compact" = 0, (ℓ, cell) rows) + Σℓ / status; k_finish completes them and copies the control
                    // block to the host
                    if (int rc = launch_materialise(t, out, s, do_compact, !do_compact, true, d_ctl)) return rc;
                    launch_finish(t, out, s, !do_compact, fuse, (double)n_azim_2, d_ctl, h_res_dev, t->call_seq);
                } else if (do_compact) {
                    launch_compaction(t, out, s);
                }
                if (int rc = rec(5)) return rc;
                if (int rc = launch_volumes()) return rc;
                volumes_pass = !(fuse && n > 0);
                if (volumes_pass) { if (int rc = rec(6)) return rc; }

                return RT_SUCCESS;
            };
#ifdef RT_HOST_TIMING
            const double ht1 = ht_now();
#endif
            if (int rc_enq = enqueue_attempt()) return rc_enq;
#ifdef RT_HOST_TIMING
            const double ht2 = ht_now();
#endif
            int32_t cur[4] = {0, 0, 0, 0};  // pool cursor, pool overflow / argument mismatch, side-list cursor, side-list overflow
            if (n == 0) RT_HIP(hipMemcpyAsync(h_res, d_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            if (attempt == 0 && m->enqueue_hook) m->enqueue_hook(m->enqueue_hook_user);
            // option "async": back to the caller as soon as the scan's copy of the control block has arrived — total, failure
            // summary and pool cursor are final then, the compaction goes on behind the call (whole-track calls without events)
            const bool async_call = m->async_calls && !m->timing && n > 0 && !split && !hybrid;
            // two-phase calls: the control block's copy and the sequence number behind it are the LAST thing the call's last kernel
            // writes (k_finish's last block, after every other block of it has finished) — seeing the number in pinned memory is
            // seeing the call complete, a few microseconds before the stream reports it (hipStreamQuery); what is still to happen
            // on the stream is that kernel's retirement, which every later operation on the stream is ordered behind anyway
            const bool seq_done = topo && !m->timing && n > 0;
            if (async_call || seq_done) RT_HIP(wait_seq(h_res, t->call_seq, s));
            else RT_HIP(wait_stream(s));
            t->in_flight = async_call || seq_done;  // (accessors wait for the stream: immediate here)
#ifdef RT_HOST_TIMING
            {
                const double ht3 = ht_now();
                g_ht[0] += ht1 - ht0; g_ht[1] += ht2 - ht1; g_ht[2] += ht3 - ht2; ++g_hn;
                if (g_hn % 50 == 0) fprintf(stderr, "[rt host] per call: before enqueue %.1f us, enqueue %.1f us, wait %.1f us\n", g_ht[0] / g_hn, g_ht[1] / g_hn, g_ht[2] / g_hn);
            }
#endif
            memcpy(fi, h_res, sizeof(fi));
            memcpy(&total, h_res + 16, sizeof(total));
            memcpy(cur, h_res + 18, sizeof(cur));
            t->chunks_needed_last = cur[0];
            if (topo) t->side_needed_last = std::max<int64_t>(0, (int64_t)cur[2] - side_static);
            if (do_compact && !cur[1] && !cur[3] && total > out.cap) {
                // the estimate was short: grow the outputs and compact again (staging pool and offsets are still valid)
                if (int rc = reserve_out(total + total / 32 + 4096)) return rc;
                launch_compaction(t, out, s);
                if (topo && h_res[rt::kCtlDeferred] != 0) {
                    // tracks whose exact Σℓ k_finish could not form from the truncated records: once more, from the complete ones
                    launch_finish(t, out, s, false, false, (double)n_azim_2, d_ctl, h_res_dev, t->call_seq);
                    RT_HIP(hipStreamSynchronize(s));
                    memcpy(fi, h_res, sizeof(fi));
                }
                if (!fused_volumes_this_call && m->volumes_mode == 2) {  // the separate volumes pass read truncated records
                    RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
                    if (int rc = launch_volumes()) return rc;
                }
                RT_HIP(hipStreamSynchronize(s));
            }
            if (!cur[1] && split && fuse && (fi[7] != 0 || m->test_volumes_fallback)) {
                // some piece marched past the seed it should have stopped at: its surplus records were dropped by
                // k_resolve but had already been added to the fused volumes — recompute them from the kept records
                fused_volumes_this_call = false;
                RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
                if (int rc = launch_volumes()) return rc;
                RT_HIP(hipStreamSynchronize(s));
            }
            if (!cur[1] && !cur[3] && topo && fuse && h_res[rt::kCtlRestarts] != 0) {
                // a track whose iteration bound reached the cap was marched again with exact steps: its cheap records had
                // already been added to the fused volumes — recompute them from the records
                if (!do_compact) {
                    if (int rc = reserve_out(total)) return rc;
                    launch_compaction(t, out, s);
                }
                fused_volumes_this_call = false;
                RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
                if (int rc = launch_volumes()) return rc;
                RT_HIP(hipStreamSynchronize(s));
                t->compacted = true;
            }
            if (!cur[1] && split && h_res[21] != 0) {
                t->force_unsplit = true;
                return segmentize_impl(t, tiny_step, k, rtol, delta_s, n_azim_2);
            }
            if (!cur[1] && !cur[3]) {
                t->cplan.staged = true;
                if (do_compact) t->compacted = true;
                if (topo && !do_compact) t->sw_ell_valid = true;  // (k_materialise left the (ℓ, cell) rows)
                if (n > 0) {  // this call's scan has reset the other control block and (fused) left the accumulator zero
                    t->ctl_clean[1 - cb] = true; t->ctl_first_chunk[1 - cb] = reset_key;
                    t->vacc_clean = fuse;
                    t->ctl_idx = 1 - cb;
                }
                break;
            }
            t->marg_clean = false;  // (a void attempt may have left entries in the list of tracks to sum exactly)
            if (attempt >= 3) { set_error("staging pool / side list overflow persists (%d chunks, %d entries needed)", cur[0], cur[2]); return RT_ERR_HIP; }
            if (cur[1]) want = (int64_t)cur[0] + cur[0] / 8 + 64;  // the cursor kept counting: this is what the march needs
            if (cur[3]) side_want = (int64_t)cur[2] + cur[2] / 8 + 1024;
        }
    } else {
        RT_HIP(hipMemcpyAsync(d_ctl, t->h_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        if (int rc = rec(1)) return rc;
        if (n > 0) {
            if (int rc = widek ? march.template operator()<rt::kCount, 1, false, true>(grid, sizeof(int32_t))
                               : march.template operator()<rt::kCount, 1, false, false>(grid, sizeof(int32_t))) return rc;
        }
        if (int rc = rec(2)) return rc;
        if (int rc = scan_counts(false, false, false)) return rc;
        if (int rc = rec(3)) return rc;
        RT_HIP(hipMemcpyAsync(h_res, d_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        if (m->enqueue_hook) m->enqueue_hook(m->enqueue_hook_user);  // (the count march is the longer half of this mode)
        RT_HIP(hipStreamSynchronize(s));
        memcpy(fi, h_res, sizeof(fi));
        memcpy(&total, h_res + 16, sizeof(total));
        if (int rc = reserve_out(total)) return rc;
        if (int rc = rec(4)) return rc;
        if (n > 0) {
            march_offsets = t->offsets.p;
            if (int rc = widek ? march.template operator()<rt::kFill, 1, false, true>(grid, sizeof(int32_t))
                               : march.template operator()<rt::kFill, 1, false, false>(grid, sizeof(int32_t))) return rc;
        }
        if (int rc = rec(5)) return rc;
        if (int rc = launch_volumes()) return rc;
        if (int rc = rec(6)) return rc;
        RT_HIP(hipStreamSynchronize(s));
        t->compacted = true;
    }
    RT_HIP(hipGetLastError());
    if (m->timing) {
        RT_HIP(hipEventElapsedTime(&f, t->ev[m->single_pass ? 1 : 0], t->ev[volumes_pass ? 6 : 5])); t->ms[0] = f;   // whole call, device side
        RT_HIP(hipEventElapsedTime(&f, t->ev[1], t->ev[2])); t->ms[2] = f;   // march (staged, or count)
        RT_HIP(hipEventElapsedTime(&f, t->ev[2], t->ev[3])); t->ms[3] = f;   // offsets scan (+ volumes ./= n_azim_2 when fused)
        RT_HIP(hipEventElapsedTime(&f, t->ev[m->single_pass ? 3 : 4], t->ev[5])); t->ms[4] = f;   // compaction (or fill march)
        if (volumes_pass) { RT_HIP(hipEventElapsedTime(&f, t->ev[5], t->ev[6])); t->ms[5] = f; }   // volumes as its own pass
    }
#ifdef RT_TIMING
    if (const char *path = getenv("RT_TIMING_DUMP")) {
        if (t->dbg.p) {
            std::vector<unsigned long long> h((size_t)((n + 63) / 64) * 4);
            RT_HIP(hipMemcpy(h.data(), t->dbg.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
        }
    }
    fprintf(stderr, "[rt timing] per wave-iteration (lane-0 view, cycles): top+load %.0f | walk_step %.0f | emit %.0f | loop-back %.0f | iters/wave %.1f | loop cycles/wave %.0f\n",
            (double)fi[8] / fi[12], (double)fi[9] / fi[12], (double)fi[10] / fi[12], (double)fi[11] / fi[12], (double)fi[12] / fi[14], (double)fi[13] / fi[14]);
#endif
#ifdef RT_STATS_DISTINCT
    {
        fprintf(stderr, "[rt distinct] wave-iterations %llu, cheap lanes per iteration %.1f\n  distinct successor records 1..8+:", h_res[63], (double)h_res[62] / (double)std::max<unsigned long long>(1, h_res[63]));
        for (int b = 1; b <= 8; ++b) fprintf(stderr, " %.3f", (double)h_res[44 + b] / (double)std::max<unsigned long long>(1, h_res[63]));
        fprintf(stderr, "\n  distinct exit edges 1..8+:");
        for (int b = 1; b <= 8; ++b) fprintf(stderr, " %.3f", (double)h_res[53 + b] / (double)std::max<unsigned long long>(1, h_res[63]));
        fprintf(stderr, "\n");
    }
#endif
#ifdef RT_STATS
    if (split)
        fprintf(stderr, "[rt stats] split plan: %llu tracks, %llu of %llu seeds alive, %llu pieces kept, %llu records marched by pieces, %llu dropped\n",
                h_res[22], h_res[23], h_res[24], h_res[26], h_res[25], fi[7]);
    fprintf(stderr, "[rt stats] walk: generic=%llu skip=%llu emit=%llu | wave-iterations=%llu with-generic-lane=%llu | chunks=%lld pool=%lld\n",
            fi[2], fi[3], fi[4], fi[5], fi[6], (long long)t->chunks_needed_last, (long long)t->pool_chunks);
#endif
    if (hybrid) t->last_split = 2;
    t->total = total;
    t->total_last = total;
    t->n_generic_records = (int64_t)fi[15];
    t->n_exact_walk_records = topo ? (int64_t)fi[14] : 0;
    for (int b = 0; b < 9; ++b) t->refusals[b] = m->single_pass ? (int64_t)h_res[rt::kCtlRefusal + b] : 0;
    t->n_near_rtol = (int64_t)h_res[rt::kCtlNearRtol];
    t->n_restarts = m->single_pass ? (int64_t)h_res[rt::kCtlRestarts] : 0;
    t->n_failed = (int64_t)fi[0];
    t->first_failed_uid = fi[0] ? (int64_t)fi[1] : 0;
    t->first_failed_status = 0;
    if (fi[0]) {
        int32_t stt = 0;
        RT_HIP(hipMemcpy(&stt, t->status.p + (fi[1] - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
        t->first_failed_status = stt;
    }
    t->segmentized = true;
    return total;
}

int32_t rt_wait(rt_tracks *t) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    return finish_call(t);
}

int32_t rt_failed_tracks(rt_tracks *t, int64_t *n_failed, int64_t *first_uid, int32_t *first_status) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (n_failed) *n_failed = t->n_failed;
    if (first_uid) *first_uid = t->first_failed_uid;
    if (first_status) *first_status = t->first_failed_status;
    return RT_SUCCESS;
}

int32_t rt_fetch_offsets(rt_tracks *t, int64_t *seg_offsets, int32_t *status) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (seg_offsets) RT_HIP(hipMemcpy(seg_offsets, t->offsets.p, sizeof(int64_t) * (t->n + 1), hipMemcpyDeviceToHost));
    if (status && t->n) RT_HIP(hipMemcpy(status, t->status.p, sizeof(int32_t) * t->n, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_fetch_segments(rt_tracks *t, double *px, double *py, double *qx, double *qy, double *ell,
                          int32_t *element) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    const size_t nb = sizeof(double) * (size_t)t->total;
    if (t->total == 0) return RT_SUCCESS;
    if (px) RT_HIP(hipMemcpy(px, t->spx.p, nb, hipMemcpyDeviceToHost));
    if (py) RT_HIP(hipMemcpy(py, t->spy.p, nb, hipMemcpyDeviceToHost));
    if (qx) RT_HIP(hipMemcpy(qx, t->sqx.p, nb, hipMemcpyDeviceToHost));
    if (qy) RT_HIP(hipMemcpy(qy, t->sqy.p, nb, hipMemcpyDeviceToHost));
    if (ell) RT_HIP(hipMemcpy(ell, t->sell.p, nb, hipMemcpyDeviceToHost));
    if (element) RT_HIP(hipMemcpy(element, t->element.p, sizeof(int32_t) * (size_t)t->total, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

// Page-locked host buffers are expensive to create (≈45 ms for C3's 410 MB) and cheap to keep: one set is kept
// process-wide when a handle dies, so that a host that creates a fresh handle per call (the Julia shim) pays
// for pinning once.
namespace {
struct PinSet { void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap = 0; };
PinSet g_pin_cache;
std::mutex g_pin_mutex;
void pin_free(PinSet &ps) {
    for (void *&q : ps.p) { if (q) (void)hipHostFree(q); q = nullptr; }
    ps.cap = 0;
}
}  // namespace

namespace {
void pin_release_to_cache(rt_tracks *t) {
    if (!t->pin_cap) return;
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    PinSet mine;
    for (int a = 0; a < 6; ++a) { mine.p[a] = t->pin[a]; t->pin[a] = nullptr; }
    mine.cap = t->pin_cap; t->pin_cap = 0;
    if (mine.cap > g_pin_cache.cap) std::swap(mine, g_pin_cache);
    pin_free(mine);
}
}  // namespace

int32_t rt_fetch_segments_pinned(rt_tracks *t, void **host_ptrs) {
    if (!t || !host_ptrs) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    const size_t n = (size_t)t->total;
    if (n > t->pin_cap) {
        pin_release_to_cache(t);
        {
            std::lock_guard<std::mutex> lk(g_pin_mutex);
            if (g_pin_cache.cap >= n) {
                for (int a = 0; a < 6; ++a) { t->pin[a] = g_pin_cache.p[a]; g_pin_cache.p[a] = nullptr; }
                t->pin_cap = g_pin_cache.cap; g_pin_cache.cap = 0;
            }
        }
        if (n > t->pin_cap) {
            const size_t cap = n + n / 8 + 64;
            for (int a = 0; a < 6; ++a) RT_HIP(hipHostMalloc(&t->pin[a], cap * (a < 5 ? sizeof(double) : sizeof(int32_t)), hipHostMallocDefault));
            t->pin_cap = cap;
        }
    }
    hipStream_t s = t->mesh->stream;
    const void *src[6] = {t->spx.p, t->spy.p, t->sqx.p, t->sqy.p, t->sell.p, t->element.p};
    for (int a = 0; a < 6 && n > 0; ++a)
        RT_HIP(hipMemcpyAsync(t->pin[a], src[a], n * (a < 5 ? sizeof(double) : sizeof(int32_t)), hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    for (int a = 0; a < 6; ++a) host_ptrs[a] = t->pin[a];
    return RT_SUCCESS;
}

int32_t rt_fetch_pinned(rt_tracks *t, void **host_ptrs) {
    if (!t || !host_ptrs) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (!t->pin_off) RT_HIP(hipHostMalloc((void **)&t->pin_off, sizeof(int64_t) * (size_t)(t->n + 1), hipHostMallocDefault));
    if (!t->pin_st) RT_HIP(hipHostMalloc((void **)&t->pin_st, sizeof(int32_t) * (size_t)std::max<int64_t>(t->n, 1), hipHostMallocDefault));
    hipStream_t s = t->mesh->stream;
    // (queued in front of the records' copies: one synchronisation for all eight arrays)
    RT_HIP(hipMemcpyAsync(t->pin_off, t->offsets.p, sizeof(int64_t) * (size_t)(t->n + 1), hipMemcpyDeviceToHost, s));
    if (t->n) RT_HIP(hipMemcpyAsync(t->pin_st, t->status.p, sizeof(int32_t) * (size_t)t->n, hipMemcpyDeviceToHost, s));
    if (int32_t rc = rt_fetch_segments_pinned(t, host_ptrs + 2)) return rc;
    host_ptrs[0] = t->pin_off;
    host_ptrs[1] = t->pin_st;
    return RT_SUCCESS;
}

int32_t rt_fetch_volumes(rt_tracks *t, double *volumes) {
    if (!t || !volumes) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    RT_HIP(hipMemcpy(volumes, t->volumes.p, sizeof(double) * t->mesh->n_cells, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_fill_tau(rt_tracks *t, const double *sigma_t, int32_t n_groups, void **tau_dev, double *ms) {
    if (!t || !sigma_t || n_groups <= 0) { set_error("rt_fill_tau: bad arguments"); return RT_ERR_INVALID; }
    if (n_groups > 1024) { set_error("rt_fill_tau: at most 1024 groups"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    rt_mesh *m = t->mesh;
    RT_HIP(hipSetDevice(m->device));
    if (int rc = ensure_compacted(t)) return rc;
    hipStream_t s = m->stream;
    const size_t n = (size_t)t->total * (size_t)n_groups;
    t->tau_groups = 0;  // (set again once the kernel has been enqueued: a failure below leaves no τ to fetch)
    RT_HIP(t->tau.reserve(n > 0 ? n : 1));
    if (int rc = upload(t->sigma_t, sigma_t, (size_t)m->n_cells * n_groups, s)) return rc;
    RT_HIP(hipEventRecord(t->ev[0], s));
    if (n > 0) {
        const unsigned blocks = (unsigned)((t->total + rt::kTauSegs - 1) / rt::kTauSegs);
        const uint32_t inv = n_groups == 1 ? 0u : (uint32_t)(0x100000000ull / (uint64_t)n_groups) + 1u;  // ≥ 2^32 / G; 0 = one group
        hipLaunchKernelGGL(rt::k_fill_tau, dim3(blocks), dim3(256), 0, s, (const double *)t->sell.p, (const int32_t *)t->element.p,
                           (const double *)t->sigma_t.p, t->total, n_groups, inv, t->tau.p);
    }
    RT_HIP(hipEventRecord(t->ev[7], s));
    RT_HIP(hipStreamSynchronize(s));
    RT_HIP(hipGetLastError());
    t->tau_groups = n_groups;
    if (ms) { float f = 0; RT_HIP(hipEventElapsedTime(&f, t->ev[0], t->ev[7])); *ms = f; }
    if (tau_dev) *tau_dev = t->tau.p;
    return RT_SUCCESS;
}

int32_t rt_fetch_tau(rt_tracks *t, double *tau) {
    if (!t || !tau) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized || t->tau_groups <= 0) { set_error("rt_fill_tau has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    const size_t n = (size_t)t->total * (size_t)t->tau_groups;
    if (n) RT_HIP(hipMemcpy(tau, t->tau.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

// ---- rt_sweep -----------------------------------------------------------------------------------------------------
static int32_t sweep_set_links_impl(rt_tracks *t, const int64_t *next_fwd, const int64_t *next_bwd, const int8_t *dir_fwd,
                                    const int8_t *dir_bwd, const int8_t *bc_fwd, const int8_t *bc_bwd) {
    if (!t || (t->n > 0 && (!next_fwd || !next_bwd || !dir_fwd || !dir_bwd || !bc_fwd || !bc_bwd))) { set_error("rt_sweep_set_links: null argument"); return RT_ERR_INVALID; }
    const int64_t n = t->n;
    if (n >= (1ll << 30)) { set_error("rt_sweep_set_links: too many tracks"); return RT_ERR_INVALID; }
    // gather map: entry slot (direction d', track v) <- source (track u, direction d), written in the order a sequential
    // sweep hands fluxes on (uid ascending, forward before backward): the last writer wins where links are not one-to-one
    std::vector<int32_t> src((size_t)std::max<int64_t>(1, 2 * n), -1);
    for (int64_t u = 0; u < n; ++u)
        for (int d = 0; d < 2; ++d) {
            const int64_t v = (d == 0 ? next_fwd[u] : next_bwd[u]) - 1;  // 1-based uids, as trace! links them
            const int dn = d == 0 ? dir_fwd[u] : dir_bwd[u];             // 0 Forward, 1 Backward (src/track.jl:11-14)
            const int bc = d == 0 ? bc_fwd[u] : bc_bwd[u];               // 0 Vacuum (src/boundary.jl:12-16)
            if (v == -1) continue;  // uid 0: the linked track is not in this track set (a shard: its owner receives the flux)
            if (v < 0 || v >= n || (dn != 0 && dn != 1) || bc < 0 || bc > 2) {
                set_error("rt_sweep_set_links: track %lld has a bad link (next uid %lld, dir %d, bc %d)", (long long)(u + 1), (long long)(v + 1), dn, bc);
                return RT_ERR_INVALID;
            }
            src[(size_t)dn * n + v] = bc == 0 ? -1 : (int32_t)(u * 2 + d);
        }
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = upload(t->sw_src, src.data(), src.size(), t->mesh->stream)) return rc;
    RT_HIP(hipStreamSynchronize(t->mesh->stream));
    t->sw_links = true;
    return RT_SUCCESS;
}

static int32_t sweep_impl(rt_tracks *t, int32_t G, const double *sigma_t, const double *source, const double *track_weight,
                          const double *psi_in, int32_t input, double *ms) {
    if (!t || G <= 0 || G > 4096 || input < 0 || input > 2) { set_error("rt_sweep: bad arguments"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (!t->sw_links) { set_error("rt_sweep: rt_sweep_set_links has not run"); return RT_ERR_INVALID; }
    rt_mesh *m = t->mesh;
    RT_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int64_t n = t->n;
    const size_t npsi = (size_t)std::max<int64_t>(1, 2 * n * G), nphi = (size_t)m->n_cells * G;
    if (G != t->sw_groups) {  // a new group structure: no cross sections, zero boundary flux
        t->sw_has_xs = false; t->sw_done = false;
        RT_HIP(t->sw_psi_in.reserve(npsi)); RT_HIP(t->sw_psi_out.reserve(npsi)); RT_HIP(t->sw_phi.reserve(nphi));
        RT_HIP(hipMemsetAsync(t->sw_psi_in.p, 0, npsi * sizeof(double), s));
        t->sw_groups = G;
    }
    if (sigma_t) {
        std::vector<double> xs(2 * nphi);
        for (size_t i = 0; i < nphi; ++i) {
            const double st = sigma_t[i], q = source ? source[i] : 0.0;
            // τ = Σt·ℓ must be finite and >= 0: one_minus_exp_neg assembles 2^n from exponent bits for n <= 0 only, and a
            // non-finite contribution would spread through the tallies' lane folds
            if (!(st >= 0.0) || !std::isfinite(st) || !std::isfinite(q)) {
                set_error("rt_sweep: sigma_t[%zu] = %g, source = %g (cross sections must be finite and >= 0)", i, st, q);
                return RT_ERR_INVALID;
            }
            xs[2 * i] = st;
            xs[2 * i + 1] = st > 0.0 ? q / st : 0.0;  // (a void cell: no attenuation, no source term)
        }
        if (int rc = upload(t->sw_xs, xs.data(), xs.size(), s)) return rc;
        RT_HIP(hipStreamSynchronize(s));  // the host vector dies here
        t->sw_has_xs = true;
    } else if (source) { set_error("rt_sweep: source given without sigma_t"); return RT_ERR_INVALID; }
    if (!t->sw_has_xs) { set_error("rt_sweep: no cross sections yet (sigma_t is NULL)"); return RT_ERR_INVALID; }
    if (track_weight) {
        if (int rc = upload(t->sw_w, track_weight, (size_t)n, s)) return rc;
        t->sw_has_w = true;
    }
    if (psi_in && n > 0) RT_HIP(hipMemcpyAsync(t->sw_psi_in.p, psi_in, (size_t)(2 * n * G) * sizeof(double), hipMemcpyHostToDevice, s));
    // option "async": the sweep's kernels are queued and the call returns (no events, no wait) — what was handed over in host
    // arrays has to be on the device before that
    const bool async_sweep = m->async_calls && !m->timing;
    if (async_sweep && (track_weight || (psi_in && n > 0))) RT_HIP(hipStreamSynchronize(s));
    // which records: the march's staging rows (whole-track single-pass calls leave them behind) or the compact CSR arrays
    const bool staged_ok = t->cplan.staged && !t->cplan.split && t->cplan.n_whole_waves == (n + 63) / 64;
    if (input == 2 && !staged_ok) { set_error("rt_sweep: the last rt_segmentize left no whole-track staging rows (track pieces or two-pass mode)"); return RT_ERR_INVALID; }
    const bool staged = input == 2 || (input == 0 && staged_ok);
    if (!staged)
        if (int rc = ensure_compacted(t)) return rc;
    using rt::as_global;
    rt::DSweep a{};
    a.stg = t->cplan.stg;
    a.ell = as_global((const double *)t->sell.p); a.element = as_global((const int32_t *)t->element.p);
    a.offsets = as_global((const int64_t *)t->offsets.p); a.counts = as_global((const int32_t *)t->counts.p);
    a.perm = as_global((const int32_t *)t->perm.p); a.azim = as_global((const int32_t *)t->azim.p);
    a.delta_s = as_global((const double *)t->delta_s.p);
    a.w = t->sw_has_w ? as_global((const double *)t->sw_w.p) : nullptr;
    a.xs = as_global((const double *)t->sw_xs.p);
    a.psi_in = as_global((const double *)t->sw_psi_in.p); a.psi_out = as_global(t->sw_psi_out.p); a.phi = as_global(t->sw_phi.p);
    a.n = n; a.n_waves = (int32_t)((n + 63) / 64); a.n_cells = m->n_cells; a.G = G; a.debug = m->sweep_debug;
    // groups per pass: as many as an LDS-private copy of their tallies allows (up to 4); none fits: global atomics.  The last pass
    // takes what is left with the kernel compiled for that many groups (7 groups = 4 + 3: a padded fourth group was an eighth
    // of the sweep's arithmetic).
    const size_t lds_cap = (size_t)std::min(m->lds_per_block, 160 * 1024) - 1024;
    int gp = std::min(G, 4);
    if (m->sweep_gp >= 1 && m->sweep_gp <= 4) gp = std::min(gp, m->sweep_gp);
    while (gp > 1 && (size_t)m->n_cells * gp * sizeof(double) > lds_cap) --gp;
    a.use_lds = (size_t)m->n_cells * gp * sizeof(double) <= lds_cap ? 1 : 0;
    if (!a.use_lds) gp = std::min(G, 4);
    if (m->sweep_gp >= 8) a.use_lds = 0;  // experiment: tallies straight to HBM (measured 4x slower at C3: 2.1 ms against 0.48)
    if (!async_sweep) RT_HIP(hipEventRecord(t->ev[0], s));
    RT_HIP(hipMemsetAsync(t->sw_phi.p, 0, nphi * sizeof(double), s));
    int passes = 0;
    // Staged rows: the first pass after an rt_segmentize derives ℓ from the exit points and leaves it in `sw_ell`, slot-indexed
    // like the rows; every later pass — of this sweep and of all following sweeps over the same segmentation — reads (ℓ, cell)
    // rows instead (12 B instead of 20, no square root, no entry point).  Option "sweep_ell" = 0 switches this off.
    bool ell_rows = false;
    if (staged && t->cplan.codes) {
        // a two-phase call staged codes: the sweep reads (ℓ, cell) rows, which the call itself left ("compact" = 0) or which
        // k_materialise writes now, once per segmentation
        if (int rc = ensure_rows(t)) return rc;
        a.stg.element = as_global(t->sw_cell.p);
        ell_rows = true;
    } else if (staged && m->sweep_ell) {
        const size_t slots = (size_t)t->pool_chunks * rt::kChunkRows * 64;
        if (t->sw_ell.reserve(slots > 0 ? slots : 1) == hipSuccess) ell_rows = true;
        else (void)hipGetLastError();  // (no memory for it: every pass derives ℓ itself)
    }
    a.ell_rows = ell_rows ? as_global(t->sw_ell.p) : nullptr;
    auto launch = [&]<bool STAGED, int GP, bool LDS>(int g0) -> int {
        size_t smem = a.use_lds ? (size_t)m->n_cells * GP * sizeof(double) : 0;
        // (compact records: more than one eight-wave workgroup per CU thrashes its L1 — a pass of few groups asks for LDS it
        //  does not use, so that it still gets a CU to itself: 5 groups = 4 + 1 took 0.88 ms against 0.58 for 7 = 4 + 3)
        if (!STAGED && a.use_lds) smem = std::max(smem, std::min(lds_cap, (size_t)81 * 1024));
        // one workgroup per CU (its tallies fill the LDS): sixteen waves when the rows are the staging rows (every load
        // instruction reads four full lines), eight when they are the compact records (64 lanes, 64 lines: sixteen waves
        // thrash the CU's L1 — 1.04 against 0.62 ms at C3); two or more workgroups per CU: eight waves each
        int W = (smem > 79 * 1024 && STAGED) ? 16 : 8;
        if (m->sweep_waves == 4 || m->sweep_waves == 8 || m->sweep_waves == 16) W = m->sweep_waves;
        const unsigned blocks = (unsigned)((2 * (int64_t)a.n_waves + W - 1) / W);
        a.g0 = g0; a.ng = GP;
        if (STAGED && ell_rows && t->sw_ell_valid) {
            if constexpr (STAGED) {
                if (smem > 48 * 1024)
                    RT_HIP(hipFuncSetAttribute((const void *)rt::k_sweep<true, GP, LDS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                hipLaunchKernelGGL((rt::k_sweep<true, GP, LDS, true>), dim3(blocks), dim3(64 * W), smem, s, a);
            }
        } else {
            if (smem > 48 * 1024)
                RT_HIP(hipFuncSetAttribute((const void *)rt::k_sweep<STAGED, GP, LDS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((rt::k_sweep<STAGED, GP, LDS, false>), dim3(blocks), dim3(64 * W), smem, s, a);
            if (STAGED && ell_rows) t->sw_ell_valid = true;  // (the forward waves of this pass have written every row's ℓ)
        }
        ++passes;
        return RT_SUCCESS;
    };
    auto launch_all = [&]<bool STAGED, bool LDS>() -> int {
        for (int g0 = 0; g0 < G;) {
            const int take = std::min(gp, G - g0);
            int rc;
            if (take == 4) rc = launch.template operator()<STAGED, 4, LDS>(g0);
            else if (take == 3) rc = launch.template operator()<STAGED, 3, LDS>(g0);
            else if (take == 2) rc = launch.template operator()<STAGED, 2, LDS>(g0);
            else rc = launch.template operator()<STAGED, 1, LDS>(g0);
            if (rc) return rc;
            g0 += take;
        }
        return RT_SUCCESS;
    };
    if (n > 0) {
        int rc;
        if (staged) rc = a.use_lds ? launch_all.template operator()<true, true>() : launch_all.template operator()<true, false>();
        else rc = a.use_lds ? launch_all.template operator()<false, true>() : launch_all.template operator()<false, false>();
        if (rc) return rc;
        const int64_t nl = 2 * n * G;
        hipLaunchKernelGGL(rt::k_sweep_link, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, (const int32_t *)t->sw_src.p,
                           (const double *)t->sw_psi_out.p, t->sw_psi_in.p, 2 * n, G, n);
    }
    if (async_sweep) {
        RT_HIP(hipGetLastError());
        if (ms) *ms = 0.0;
        t->in_flight = true;  // (every accessor waits; a consumer with its own stream orders against rt_mesh_get_stream / rt_wait)
    } else {
        RT_HIP(hipEventRecord(t->ev[7], s));
        RT_HIP(wait_stream(s));
        RT_HIP(hipGetLastError());
        if (ms) { float f = 0; RT_HIP(hipEventElapsedTime(&f, t->ev[0], t->ev[7])); *ms = f; }
        t->in_flight = false;  // (the sweep waited for the stream)
    }
    t->sw_done = true;
    t->sw_last_input = staged ? 2 : 1; t->sw_last_gp = a.use_lds ? gp : 0; t->sw_last_passes = passes;
    return RT_SUCCESS;
}

int32_t rt_sweep_set_links(rt_tracks *t, const int64_t *next_fwd, const int64_t *next_bwd, const int8_t *dir_fwd,
                           const int8_t *dir_bwd, const int8_t *bc_fwd, const int8_t *bc_bwd) {
    try {
        return sweep_set_links_impl(t, next_fwd, next_bwd, dir_fwd, dir_bwd, bc_fwd, bc_bwd);
    } catch (const std::exception &e) {
        set_error("rt_sweep_set_links: %s", e.what());
        return RT_ERR_INVALID;
    }
}

int32_t rt_sweep(rt_tracks *t, int32_t n_groups, const double *sigma_t, const double *source, const double *track_weight,
                 const double *psi_in, int32_t input, double *ms) {
    try {
        return sweep_impl(t, n_groups, sigma_t, source, track_weight, psi_in, input, ms);
    } catch (const std::exception &e) {
        set_error("rt_sweep: %s", e.what());
        return RT_ERR_INVALID;
    }
}

int32_t rt_sweep_fetch(rt_tracks *t, double *phi, double *psi_out, double *psi_next) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->sw_done) { set_error("rt_sweep has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    const size_t npsi = (size_t)(2 * t->n * t->sw_groups), nphi = (size_t)t->mesh->n_cells * t->sw_groups;
    if (phi) RT_HIP(hipMemcpy(phi, t->sw_phi.p, nphi * sizeof(double), hipMemcpyDeviceToHost));
    if (psi_out && npsi) RT_HIP(hipMemcpy(psi_out, t->sw_psi_out.p, npsi * sizeof(double), hipMemcpyDeviceToHost));
    if (psi_next && npsi) RT_HIP(hipMemcpy(psi_next, t->sw_psi_in.p, npsi * sizeof(double), hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_sweep_info(rt_tracks *t, void **ptrs_dev, int32_t *info) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->sw_done) { set_error("rt_sweep has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    // (no wait here: addresses and counts only — under "async" the caller orders its reads against the mesh's stream or rt_wait)
    if (ptrs_dev) { ptrs_dev[0] = t->sw_phi.p; ptrs_dev[1] = t->sw_psi_out.p; ptrs_dev[2] = t->sw_psi_in.p; }
    if (info) { info[0] = t->sw_last_input; info[1] = t->sw_last_gp; info[2] = t->sw_last_passes; info[3] = t->sw_groups; }
    return RT_SUCCESS;
}

int32_t rt_sweep_xs_pointer(rt_tracks *t, void **xs_dev) {
    if (!t || !xs_dev) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->sw_has_xs) { set_error("rt_sweep has not been given cross sections yet"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    *xs_dev = t->sw_xs.p;
    return RT_SUCCESS;
}

int32_t rt_device_pointers(rt_tracks *t, void **p) {
    if (!t || !p) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    p[0] = t->offsets.p; p[1] = t->status.p; p[2] = t->spx.p; p[3] = t->spy.p; p[4] = t->sqx.p;
    p[5] = t->sqy.p; p[6] = t->sell.p; p[7] = t->element.p; p[8] = t->volumes.p;
    return RT_SUCCESS;
}

int32_t rt_mesh_info(rt_mesh *m, double *info, int32_t n_info, char *note, int32_t note_cap) {
    if (!m || (n_info > 0 && !info) || n_info < 0 || note_cap < 0) { set_error("rt_mesh_info: bad argument"); return RT_ERR_INVALID; }
    const double v[RT_MESH_INFO_COUNT] = {
        (double)(m->d.walk_ok ? 1 : 0), (double)m->n_records, (double)m->n_records_walk, m->eps_min, m->eps_max,
        m->d.d_vertex, m->d.l_min, (double)m->n_cells_fragile, (double)m->n_cells_wild, (double)m->n_edges_nonmanifold,
        (double)m->extras_max, m->prep_ms, m->kappa, (double)(m->walk_available ? 1 : 0),
        (double)(m->topo_available ? m->n_records_topo : 0), m->topo_tiny_max};
    for (int i = 0; i < n_info && i < RT_MESH_INFO_COUNT; ++i) info[i] = v[i];
    if (note && note_cap > 0) {
        strncpy(note, m->prep_note.c_str(), (size_t)note_cap - 1);
        note[note_cap - 1] = 0;
    }
    return RT_SUCCESS;
}

int32_t rt_last_stats(rt_tracks *t, int64_t *stats, int32_t n) {
    if (!t || !stats || n < 4) { set_error("rt_last_stats: bad argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    stats[0] = t->total;
    stats[1] = t->n_generic_records;
    stats[2] = t->chunks_needed_last;
    stats[3] = t->pool_chunks;
    if (n > 4) stats[4] = t->last_march_waves;
    if (n > 5) stats[5] = t->last_split;  // 0 whole tracks, 1 pieces, 2 hybrid (pieces for the longest waves only)
    if (n > 6) stats[6] = t->last_widek;
    if (n > 8) stats[8] = t->last_topo ? t->total - t->n_generic_records - t->n_exact_walk_records : 0;  // records made by cheap steps
    for (int b = 0; b < 9 && 9 + b < n; ++b) stats[9 + b] = t->refusals[b];
    if (n > 18) stats[18] = t->n_near_rtol;
    if (n > 19) stats[19] = t->n_restarts;
    if (n > 7) {  // device memory held by this handle: inputs, staging pools, tables, results
        auto b = [](const auto &d) { return (int64_t)(d.cap * sizeof(*d.p)); };
        stats[7] = b(t->in_arena) + b(t->cnt_slot) + b(t->off_slot) +
                   b(t->perm_whole) + b(t->counts) + b(t->status) + b(t->element) + b(t->offsets) + b(t->tile_sums) + b(t->ctl) + b(t->spx) +
                   b(t->spy) + b(t->sqx) + b(t->sqy) + b(t->sell) + b(t->volumes) + b(t->volumes_prev) + b(t->delta_s) + b(t->gpx) + b(t->gpy) +
                   b(t->gqx) + b(t->gqy) + b(t->gelement) + b(t->ctab) + b(t->cowner) + b(t->vorder) + b(t->vw_wave) + b(t->vw_k) +
                   b(t->w_base) + b(t->w_P) + b(t->s_el) + b(t->s_eq) + b(t->p_count) + b(t->p_flags) + b(t->p_valid) + b(t->p_rel) + b(t->s_px) +
                   b(t->s_py) + b(t->s_qx) + b(t->s_qy) + b(t->s_ell) + b(t->p_sum) + b(t->vacc) + b(t->fst_i) + b(t->fst_v) + b(t->tau) +
                   b(t->sigma_t) + b(t->sw_src) + b(t->sw_w) + b(t->sw_xs) + b(t->sw_psi_in) + b(t->sw_psi_out) + b(t->sw_phi) + b(t->sw_ell) +
                   b(t->sw_cell) + b(t->side_px) + b(t->side_py) + b(t->side_qx) + b(t->side_qy) + b(t->side_el) + b(t->marg);
    }
    return RT_SUCCESS;
}

int32_t rt_last_timing(rt_tracks *t, double *ms, int32_t n) {
    if (!t || !ms || n < 6) { set_error("bad argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    for (int i = 0; i < n && i < 8; ++i) ms[i] = t->ms[i];
    return RT_SUCCESS;
}

}  // extern "C"
