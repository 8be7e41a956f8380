// rt_internal.hpp — what the translation units of librt_segmentize.so share: the device-side argument structs, the two
// handles behind include/rt_segmentize.h and the host helpers that cross a TU boundary.  Not installed; not part of the C ABI.
//   rt_march.hip       k_march (+ k_seed, k_resolve) and their launchers
//   rt_records.hip     staging -> records: k_compact3, k_materialise, k_finish, the offsets scan, k_volumes, k_fill_tau
//   rt_materialise.hip k_materialise_lin: a two-phase call's records in output order
//   rt_sweep.hip       rt_sweep: k_sweep, k_sweep_link, the sweep's host code and entry points
//   rt_segmentize.hip  handles, rt_tracks_create, rt_segmentize (the call's host logic), fetches, statistics
#pragma once
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
extern thread_local std::string g_last_error;
void set_error(const char *fmt, ...);  // (defined in rt_segmentize.hip; shared with rt_host.cpp and rt_multi.hip)
}  // namespace rthost
using rthost::g_last_error;
using rthost::set_error;

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

// ------------------------------------------------------------------- device-side argument structs
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
constexpr int kTileAccStride = 32;  // DStage::tile_acc: one tile sum per 128-B line (experiment)
constexpr int kStaticRegions = 12;  // chunk indices with a reserved region (see DStage): 384 records per track

constexpr int32_t kWordExactTally = 1 << 30;  // staged word of a cheap record whose fill_volumes term k_materialise adds (see DStage)
struct DStage {
    RT_G double *qx, *qy;   // exit point of every record
    RT_G double *px, *py;   // entry point, only for records whose element is staged negative (see k_march)
    RT_G int32_t *element;
    RT_G int32_t *ctab;     // [n_waves][kMaxChunks] chunk ids
    RT_G int32_t *cowner;   // [pool_chunks] wave * kMaxChunks + j of the chunk's owner
    RT_G int32_t *cursor;   // [0] chunks handed out, [1] overflow flag
    int32_t pool_chunks;
    // Whole-track marches: chunk j of march wave w is chunk reg_base[j] + w for w < reg_cap[j] — reserved by the host from the
    // waves' estimated record counts (the waves are ordered longest first, so the waves that need a j-th chunk are a prefix), the
    // cursor starts behind the regions.  No atomic and — for k_materialise — no table lookup in front of a wave's first loads:
    // the chunk id follows from (w, j) and 2 x kStaticRegions kernel arguments.  Chunks beyond the estimate come from the
    // cursor as before; ctab / cowner are written for every chunk either way (k_compact3, rt_sweep read them).
    // k_march<TOPO> adds every wave's record count to the sum of its tile of kScanTile uids here (null: no): the offsets' scan is
    // then ONE kernel whose blocks each add up the tile sums in front of their own (k_scan_fused), not two
    RT_G int32_t *tile_acc;
    int32_t n_regions;      // regions in use (0: every chunk from the cursor)
    int32_t reg_cap[kStaticRegions], reg_base[kStaticRegions];
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
    // Records in COMPLETION order (round 6; option "record_order"): cq non-null — a march workgroup that has
    // ended takes the span of its tracks' records from the cursor in the control block (word kCtlCq), writes its tracks' offsets
    // (off_slot by march slot, tab_off by uid), releases and appends its index to this queue; the record kernel runs BESIDE the
    // march on a second stream and its workgroups take the units of the k-th march workgroup to finish.
    RT_G unsigned long long *cq;       // [8][cq_blocks], one queue per XCD: (epoch << 32) | march workgroup; entries of earlier calls carry older epochs
    RT_G int64_t *tab_off;             // [n] first record of every track, uid order
    unsigned long long *cq_started;    // pinned host memory, [8]: the grid's last eight workgroups store the epoch when they start
    uint32_t cq_epoch;
    int32_t cq_blocks;
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
constexpr int kCtlExactTally = 43;  // cheap records whose fill_volumes term k_materialise adds from the record's own length
constexpr int kCtlNearRtol = 42;  // tracks whose Σℓ check (src/track.jl:171) sits within summation-order noise of its threshold
constexpr int kCtlCq = 48;        // 48: records handed out to march workgroups (the completion-order cursor), 49: march workgroups that
                                  // have ended and queued themselves, 50: the record kernel beside the march gave up waiting (the attempt is
                                  // void), 51: units the record kernel has served
constexpr int kCtlCqXcd = 52;     // 52..59, one word per XCD as two int32: [0] march workgroups queued on the XCD, [1] units taken there
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

// ---- the march in three kernels ("lean" plan, round 6) -----------------------------------------
// The cheap loop of a lane needs a dozen values; the exact step (walk_step / generic: find_element + intersections) it almost
// never takes needs 200 registers.  In one kernel the two share a register budget (218 VGPRs, two waves per SIMD, scratch).  The
// lean plan runs them as three kernels that hand a lane's state through memory (src/track.jl:106-178: what the reference carries
// between iterations is xp, prev_element and i — here: the record the lane predicts, its last record's code, the signed distances
// of the entry edge's end points, the tally's positions and the counters):
//   k_first  = k_march<..., PHASE 1>: start band + every track's first record (generic step) + topo_enter; writes the state
//   k_cheap  (rt_cheap.hip): ONLY the cheap loop and its tally, four waves per SIMD; a lane whose cheap step refuses (or that has
//              no certified successor, or whose iteration bound reaches the cap) writes its state, queues its slot and LEAVES
//   k_serve  = k_march<..., PHASE 2>: a few persistent workgroups that claim queued slots and march those tracks to their end with
//              the full kernel (exact step, then cheap steps again) — behind k_cheap on its stream, or beside it on a second one.
// The staged words, the side list, counts / status / tile sums are what the one-kernel march leaves: scan, k_materialise_lin and
// k_finish do not change.  Results never depend on which kernel decided a record.
struct DLean {
    RT_G int32_t *pred, *last, *i, *it, *fl, *word, *prev_el, *wk_last;  // [n_slots] per march slot
    RT_G double *sp, *sn, *ttP, *ttN, *ttp, *dprev, *lqx, *lqy;           // [n_slots]
    RT_G int32_t *dump;       // [kChunkRows * 16 + 64] where the lanes of k_cheap that have left store (nobody reads it)
    RT_G int32_t *queue;      // [n_slots] march slots that need exact steps, in the order they were queued (-1: entry not written yet)
    RT_G int32_t *qctl;       // [0] queue tail, [1] head, [2] k_cheap workgroups that have ended, [3] k_serve gave up (timeout)
    int32_t n_cheap_wgs;      // workgroups of the k_cheap launch (k_serve ends when all of them have and the queue is empty)
    int32_t pad_;
};
// k_march's lane flags (see there)
constexpr uint32_t kFlCheap = 1, kFlUsed = 2, kFlMat = 4, kFlWait = 8, kFlDone = 16, kFlRestart = 32;
// state word `fl` of a slot: bits 0..5 the lane flags, and
constexpr int32_t kLnApos = 1 << 8;    // TopoState::apos
constexpr int32_t kLnFinal = 1 << 9;   // the lane has ended (its counts are written)
constexpr int32_t kLnExact = 1 << 10;  // queued: exact state (lqx, lqy, prev_el, wk_last) — no cheap step was possible behind the first record
constexpr int kLeanCtl = 44;           // control-block words 44..47 (as int32: 8 words): the queue's counters (DLean::qctl)

// Chunk j of march wave w for a kernel of the lean plan: the one the host reserved, the one ctab records (k_first cleared the wave's
// row to -1), or a fresh one from the pool's cursor — by compare-and-swap, because k_serve may run beside k_cheap and a lane it
// serves stages into its ORIGIN wave's columns.  -2: pool exhausted (the attempt is void, the host grows the pool and re-runs).
__device__ __forceinline__ int32_t lean_chunk(const RT_K DStage *sk, int64_t w, int j) {
    RT_G int32_t *e = sk->ctab + w * kMaxChunks + j;
    if (j < sk->n_regions && w < sk->reg_cap[j]) {
        const int32_t c = sk->reg_base[j] + (int32_t)w;
        *e = c;                                   // (rt_sweep reads every chunk of a wave through ctab; same value whoever writes)
        sk->cowner[c] = (int32_t)(w * kMaxChunks + j);
        return c;
    }
    int32_t c = __hip_atomic_load((int32_t *)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c != -1) return c;
    RT_G int32_t *cursor = sk->cursor;
    const int32_t nw = atomicAdd((int32_t *)&cursor[0], 1);
    if (nw >= sk->pool_chunks) { cursor[1] = 1; return -2; }
    const int32_t old = atomicCAS((int32_t *)e, -1, nw);
    if (old == -1) { sk->cowner[nw] = (int32_t)(w * kMaxChunks + j); return nw; }
    sk->cowner[nw] = -1;  // (lost the race: the chunk stays unused)
    return old;
}

// (sum_check_is_marginal and the Σℓ chain of k_materialise_lin: rt_device.hpp — host and device)

constexpr int kC3Pitch = kChunkRows + 4;  // doubles per track in a tile: slot 0 = carry, slots 1..32 = rows

// k_materialise's arguments (rt_records.hip)
struct DMat {
    const RT_G EdgeABC *etab;
    int32_t etab_bytes;           // 3 n_cells entries of 32 B (k_materialise_lin's buffer resource; < 2^32)
    const RT_G int32_t *corder;   // large batches: march waves in the order of their output addresses (as k_compact3)
    int64_t n_units;              // 4 per march wave
    double rtol;
    double coord_max;             // largest |coordinate| of the mesh's bounding box (the Σℓ chain's absolute band)
    int32_t tally;                // 1: Σℓ + status (the call's first pass over the codes); 0: records / rows only
    int32_t force_exact;          // tests: every track takes k_finish's left-to-right sum
    int32_t marg_cap;
    // QUEUE (records in completion order, beside the march): workgroup b takes unit b mod (4 q_waves) of the (b / (4 q_waves))-th
    // march workgroup in the completion queue (DStage::cq)
    int32_t q_waves;              // waves per march workgroup
    int32_t n_waves;              // march waves of the batch
    RT_G int32_t *marg;           // [0] count, [1 ...] march slots of the tracks k_finish has to sum exactly
    RT_G double *ell_rows;        // ROWS
    RT_G int32_t *cell_rows;
    RT_G double *vacc;            // fill_volumes' accumulator: the terms of the records the march flagged (kWordExactTally) are added here
    unsigned long long *ctl;      // the call's control block ([0] failed tracks, [1] first failing uid + 1)
    unsigned long long *dbg;      // development (-DRT_LIN_TIMING): cycle sums between the kernel's stamps
};

constexpr int kScanBlock = 256;
constexpr int kScanPer = 4;
constexpr int kScanTile = kScanBlock * kScanPer;
constexpr int kTauSegs = 2048;  // segments per workgroup

// k_sweep's arguments (rt_sweep.hip)
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
    int fetch_hugepages = 1;  // rt_fetch_* into the caller's arrays: ask for transparent huge pages on the destination (once per array)
    int sweep_rows = 1; // rt_sweep over the compact records: as (ℓ, cell) rows (the staging's, or made once from the records); 0: where they lie
    int sweep_ell = 1;  // rt_sweep over staged rows: keep ℓ of every row from the first pass for the later ones (0: every pass derives it)
    int sweep_debug = 0, compact_debug = 0;
    int mat_kernel = 0;      // records of a two-phase call: 0 k_materialise_lin (output order, 16-B stores), 1 k_materialise (chunk tiles; A/B)
    int march_waves = 0;     // 4 / 6: waves per workgroup of the fused march (0: automatic)
    int topo = 1;          // 1: cheap steps (k_march<..., TOPO>) for whole-track batches when the mesh allows it; 2: forced — also on
                           // meshes where fewer than 90 % of the walkable records carry a cheap certificate, and a wave that is
                           // refused often does not hand back to exact steps (tests and fuzzing: every cheap certificate is exercised)
    int async_calls = 0;   // 1: rt_segmentize returns once total, status summary and offsets' scan are known to the host; the
                           // compaction may still be running on the stream (every entry point that touches results waits)
    int record_order = 0;  // 0: the records in CSR order (uid order); 1: in completion order when the plan allows it and every march
                           // workgroup is resident at once (the record kernel then runs beside the march); 2: whenever the plan allows
    int lean = 0;          // the march of a two-phase call in three kernels (DLean): 0 no, 1 k_serve behind k_cheap, 2 beside it (second stream)
    int serve_blocks = 0;  // workgroups of k_serve (0: 64)
    int cheap_per_cu = 0;  // experiments: workgroups of k_cheap per CU (0: automatic)
    hipStream_t side_stream = nullptr;
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    int timing = 0;        // 1: record HIP events between the kernels of a call for rt_last_timing (≈4 µs of stream time each)
    bool topo_available = false;
    double topo_tiny_max = 0.0, topo_rmax = 0.0, topo_end_err = 0.0, tally_a = 0.0, tally_b = 0.0;
    bool fused_scan = true;  // two-phase calls: the march leaves the scan's tile sums, one scan kernel (0: the two-launch scan; A/B)
    int64_t test_reserved_pct = -1;  // tests only: the staging chunks reserved per wave as a percentage of the estimate (< 0: all of it)
    int64_t test_tally_tau = 0;  // tests, A/B: the relative error allowed to a cheap record's chord in fill_volumes, in 1e-12 (0: 8e-11; < 0: none — every cheap record tallied by k_materialise)
    int64_t n_records_topo = 0;
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
    DevView<double> As, Bs, Cs, Ls, Dxs, Dys;   // the track lines, lengths and directions (cos ϕ, sin ϕ) in march order (k_materialise)
    DevView<double> Pxs, Pys, Phis;             // start points and angles in march order (the whole-track march's first loads)
    DevView<int32_t> Azs;                       // ... and azimuthal indices
    DevView<int32_t> iperm;       // uid -> march slot
    DevBuf<int32_t> cnt_slot;     // record counts / CSR offsets in march-slot order (whole-track two-phase calls)
    DevBuf<int64_t> off_slot;
    DevBuf<double> w_slot;
    rt::DTracks d{};
    // results
    bool segmentized = false;
    int64_t total = 0;
    DevBuf<int32_t> counts, status, element;
    DevBuf<int64_t> offsets, tile_sums;
    DevBuf<int32_t> tile_acc;        // two halves of n_tiles record counts per tile, one per control block (see DStage)
    int64_t tile_acc_tiles = 0;
    bool tile_acc_clean[2] = {false, false};  // the half is zero (set by the scan that cleared it)
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
    int64_t pool_chunks = 0, chunks_needed_last = 0, total_last = 0;
    // split mode (pieces of tracks)
    int32_t n_vwaves = 0;
    int32_t reg_cap[rt::kStaticRegions] = {};  // reserved staging chunks per chunk index (DStage), from the waves' expected record counts
    DevBuf<int32_t> vorder, vw_wave, vw_k, w_base, w_P, s_el, s_eq, p_count, p_flags, p_valid, p_rel;
    DevBuf<double> s_px, s_py, s_qx, s_qy, s_ell, p_sum;
    double sum_ell = 0.0;
    int32_t azim_min = 1, azim_max = 0;  // range of azim_idx (checked against n_azim_2 by rt_segmentize)
    int64_t n_generic_records = 0;       // rt_last_stats
    bool force_unsplit = false;  // a track reached MAX_ITER segments in split mode: this track set marches whole from now on
    int32_t last_topo = 0;  // 1: the last call marched with cheap steps
    int32_t last_lean = 0;  // ... in three kernels (the option's value)
    int32_t last_record_kernel = 0;  // which kernel wrote the last call's records: 0 none yet, 1 k_compact3, 2 k_materialise, 3 k_materialise_lin, 4 k_materialise (rows only)
    bool lean_gave_up = false, lean_q_clean = false;
    int64_t n_lean_queued = 0;
    DevBuf<int32_t> lean_i;   // DLean: the lanes' state (8 int arrays), the queue, the dump row
    DevBuf<double> lean_d;    // ... 8 double arrays
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
        int march_waves = 4;  // waves per workgroup of the whole-track march
        double rtol = 0.0;
    } cplan;
    bool compacted = false;  // the six record arrays hold the last call's records in CSR order (uid order)
    // Records in COMPLETION order (option "record_order"): the six arrays hold the last call's records, every track's contiguous,
    // at tab_off[uid] (off_slot[march slot]) — the order in which the march's workgroups ended.  rt_device_table hands out the
    // table; every entry point that promises the CSR layout goes through ensure_compacted, which rewrites them once.
    bool completion_order = false;
    bool completion_gave_up = false;   // the record kernel beside the march once waited in vain: this handle stays with CSR order
    DevBuf<int64_t> tab_off;           // [n]
    DevBuf<unsigned long long> cq;     // [march workgroups] the completion queue (DStage::cq)
    unsigned long long *cq_started = nullptr;  // pinned, [8]
    unsigned long long cq_epoch_last = 0;
    int32_t last_attempts = 0;         // rt_last_stats: attempts of the last call
    int32_t last_completion = 0;       // rt_last_stats: the last call wrote its records beside the march
    bool in_flight = false;  // option "async": the last rt_segmentize returned while its compaction was still on the stream
    unsigned long long call_seq = 0;  // sequence number the scan writes behind its host copy of the control block
    // rt_sweep: the gather map of the cyclic linking, per-track weights, cross sections, boundary fluxes, tallies
    DevBuf<int32_t> sw_src;
    DevBuf<double> sw_w, sw_xs, sw_psi_in, sw_psi_out, sw_phi;
    DevBuf<double> sw_ell;      // ℓ of every staged row (slot-indexed like the staging pool), left by the first staged pass after a call
    bool sw_ell_valid = false;  // ... of the last rt_segmentize
    DevBuf<int32_t> sw_cell;    // codes: cell + 1 of every staged row, beside sw_ell (k_materialise<.., ROWS>)
    DevBuf<int32_t> sw_ctab, sw_plan;  // rows made from the COMPACT records (rt_sweep.hip ensure_rows_from_compact): their own chunk table
    bool sw_rowsc_valid = false;       // ... sw_ell / sw_cell hold those rows for the last rt_segmentize
    bool sw_links = false, sw_has_w = false, sw_has_xs = false, sw_done = false;
    int32_t sw_groups = 0, sw_last_input = 0, sw_last_gp = 0, sw_last_passes = 0, sw_last_rows = 0;
    int64_t refusals[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // cheap-step refusals of the last call by certificate term
    int64_t n_near_rtol = 0, n_restarts = 0, n_exact_tally = 0;
    int64_t n_failed = 0, first_failed_uid = 0;
    int32_t first_failed_status = 0;
};


// ------------------------------------------------------------------- host helpers that cross a TU boundary
namespace rtx {
// rt_segmentize.hip
hipError_t wait_stream(hipStream_t s);
hipError_t wait_seq(const unsigned long long *h_res, unsigned long long seq, hipStream_t s);
int finish_call(rt_tracks *t);  // every entry point that reads a call's results first waits for a call still on the stream
template <typename T>
int upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s) {
    RT_HIP(b.reserve(n > 0 ? n : 1));
    if (n) RT_HIP(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_SUCCESS;
}
// rt_records.hip
int reserve_records(rt_tracks *t, int64_t tot, rt::DOut &out);
int launch_materialise(rt_tracks *t, const rt::DOut &out, hipStream_t s, bool records, bool rows, bool tally, unsigned long long *d_ctl, bool queue = false);
bool lin_kernel_serves(const rt_tracks *t, const rt::DOut &out);  // k_materialise_lin can write this call's records
// rt_materialise.hip
void launch_materialise_lin(const rt::DTracks &d, int32_t *status, const rt::DStage &stg, const rt::DOut &out, const rt::DMat &a, hipStream_t s,
                            int n_cus, bool queue = false);
void launch_finish(rt_tracks *t, const rt::DOut &out, hipStream_t s, bool from_rows, bool scale_volumes, double n_azim_2,
                   unsigned long long *d_ctl, unsigned long long *h_res_dev, unsigned long long seq, bool completion_order = false);
void launch_compaction(rt_tracks *t, const rt::DOut &out, hipStream_t s);
int ensure_compacted(rt_tracks *t);
int ensure_rows(rt_tracks *t);
int ensure_rows_from_compact(rt_tracks *t);  // rt_sweep.hip
void launch_prologue(hipStream_t s, unsigned long long *ctl, double *volumes, int32_t n_cells, int32_t first_chunk, int32_t side_first);
// the exclusive scan of the counts (two kernels); see k_scan_tile_sums / k_scan_write for the optional pointers
void launch_scan_fused(hipStream_t s, rt_tracks *t, int64_t n_tiles, unsigned long long *d_ctl, const int32_t *tile_acc, int32_t *tile_acc_next,
                       unsigned long long *ctl_next, int32_t first_chunk_next, int32_t side_first_next, bool write_slots = true);
void launch_scan(hipStream_t s, rt_tracks *t, int64_t n_tiles, unsigned long long *d_ctl, unsigned long long *host_copy,
                 unsigned long long *ctl_next, int32_t first_chunk_next, int32_t side_first_next, unsigned long long seq,
                 double *scale_volumes, double n_azim_2, bool slot_order);
int launch_volumes_pass(hipStream_t s, rt_tracks *t, const int32_t *overflow, int64_t cap);  // fill_volumes over the compact records
void launch_scale_volumes(hipStream_t s, double *volumes, int32_t n_cells, double n_azim_2);
void launch_fill_tau(hipStream_t s, rt_tracks *t, int32_t n_groups);
void launch_slot_arrays(hipStream_t s, int64_t n, const rt::DTracks &d, double *As, double *Bs, double *Cs, double *Ls, double *Dx, double *Dy,
                        double *Pxs, double *Pys, double *Phis, int32_t *Azs, int32_t *iperm);
// rt_march.hip
int launch_march(int mode, int waves, bool split, bool widek, bool topo, unsigned blocks, size_t smem, hipStream_t s, const rt::DMesh &m,
                 const rt::DTracks &t, const rt::DParams &prm, int32_t *counts, int32_t *status, const int64_t *offsets, const rt::DOut &out,
                 const rt::DStage &stg, unsigned long long *fail_info, const rt::DSplit &sp, int phase = 0, const rt::DLean *lean = nullptr);
// rt_cheap.hip: the lean plan's cheap-only kernel (waves: 4, 8 or 16 per workgroup, by the size of the LDS copy of `volumes`)
int launch_cheap(int waves, unsigned blocks, size_t smem, hipStream_t s, const rt::DMesh &m, const rt::DTracks &t, const rt::DParams &prm,
                 int32_t *counts, int32_t *status, const rt::DOut &out, const rt::DStage &stg, unsigned long long *ctl, const rt::DLean &ln);
void launch_seed(bool widek, unsigned blocks, hipStream_t s, const rt::DMesh &m, const rt::DTracks &t, const rt::DParams &prm, const rt::DSplit &sp);
void launch_resolve(unsigned blocks, hipStream_t s, const rt::DTracks &t, const rt::DParams &prm, const rt::DSplit &sp, int32_t *counts,
                    int32_t *status, unsigned long long *fail_info);
}  // namespace rtx
