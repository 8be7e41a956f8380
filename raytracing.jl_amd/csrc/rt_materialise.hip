// rt_materialise.hip — codes -> records in OUTPUT ORDER (gfx950): the record-writing half of the two-phase march, round 5.
//
// k_march<..., TOPO> leaves one word per record (DStage); this kernel turns the words of a unit — the 16 tracks of a quarter of a
// march wave — into the reference's records:
//   q = intersection(track.ABC, general_form of the exit edge)   src/intersection.jl:127-138 (edge_exit_point; `etab` holds the
//       host's general forms, evaluated with the reference's operations — bit-identical),
//   p = the previous record's q (bit-identical to the reference's own intersection with the shared edge), or the side list's p,
//   ℓ = ‖p − q‖                                                  Segment ctor, src/segment.jl:31-33,
//   Σℓ per track and isapprox(track.ℓ, Σℓ; rtol)                 src/track.jl:171-175 (decided by margin, k_finish sums the rest).
// Same operands in the same operations as k_materialise (rt_records.hip), which it replaces for calls that write records only —
// that kernel's shape followed the staging chunks (wave k = chunk k of 32 rows, transposing LDS tiles, 8-B stores of 256-B runs):
//   * a unit of 5 or 6 chunks (54 % of C5's) kept one wave busy twice as long as the others;
//   * every store instruction moved 8 B per lane;
//   * the workgroup began with dependent trips to memory (uid -> length, counts -> words) and added up every ℓ per track.
// Here the unit's words are first transposed into an LDS array in OUTPUT ORDER ("linear slots": the tracks of the unit one
// after the other, as their records lie in the result arrays), and then the 256 threads walk that array: a lane owns the
// records of two consecutive slots (2m, 2m + 1) — an aligned pair in memory: the slots are shifted so that slot ≡ record index
// (mod 16) — gathers their two edges, computes q, takes p from the slot before (its own first record, or the neighbouring
// lane's second: one DPP shift), ℓ, and writes each f64 array with ONE 16-B store per lane (1 KB per wave-instruction, whole
// 128-B lines), the cell ids with one 8-B store.  Work is dealt by slots, not by chunks: whole iterations of 64 pairs, evenly.
// Tracks whose records are not adjacent in memory (a unit that straddles the packed partial wave of uids, sort modes 0 / 1,
// rounds of tracks longer than 256 records) form several RUN GROUPS, each padded to its own alignment; a pair never straddles
// two groups, pads read as "no record".
//
// gfx950 retires a wave's loads, stores and atomics through ONE in-order counter (a wait for a load is a wait for everything the
// wave issued before it), and the compiler's wait for a value must assume, behind a branch, the smaller count of the two paths.
// Hence, in the loop over a wave's pairs:
//   * every vector-memory instruction of the steady state is UNCONDITIONAL — record stores are buffer stores whose inactive lanes
//     hand in an offset beyond the array (the hardware drops them), gathers use clamped offsets, six dropped stores in front of
//     the loop make its entry look like its back edge: the wait for a gather then leaves exactly the younger stores in flight;
//   * the gathers of iteration i + 1 are issued before the stores of iteration i (two register sets, the loop unrolled by two);
//   * what is rare and conditional (half pairs at a run group's ends, the fill_volumes terms of marked records) collects in LDS
//     lists and is written out behind the loop.
// The unit's first trip to memory is ONE trip: counts, offsets, the table (lines, lengths, first records: three wave-loads in a
// (field, track) lane layout) and — for chunks the host reserved, whose ids follow from kernel arguments — the words themselves.
// Σℓ is not added up: a track's records lie head to tail (see s_qlast).
// Who issues first: the header (first trip, run groups, transposition) and the epilogue of a workgroup run with a raised issue
// priority (s_setprio), its store loop with the default one — the latency-bound parts do not queue for issue slots behind the other
// workgroups' FP64 loops.  The usual unit is one run group (a DPP shift and a ballot establish it: no scalar recurrence), each of
// the header's wave-loads is taken by another wave, and the transposition is one compare and two LDS writes per word.
//
// Measured and set aside (round 5, profiles/r05/exp_materialise_*.log): persistent workgroups that issue the next unit's header in
// the loop's tail (the in-order counter leaves at most two iterations of distance: the header still arrives late, and the extra
// registers cost a wave per SIMD); four compute waves fed by one or two LOADER waves through a double-buffered LDS image (the
// compute waves then never wait for a header, but two workgroups of six waves per CU leave the gathers' latency uncovered, and
// three do not fit the register file without spilling the loader's words).
#include "rt_internal.hpp"

namespace rt {

constexpr int kLinRows = 256;                      // rows of a track per round (a unit of longer tracks takes several rounds)
constexpr int kLinCap = 16 * kLinRows + 16 * 32;   // linear slots of a round: records + per run group < 16 pads in front, < 16 behind
static_assert(kLinRows % kChunkRows == 0 && kLinRows / kChunkRows == 8, "a wave takes two of a round's eight chunks");

// What the linear phase needs per track (LDS).  16-B aligned pieces: one ds_read_b128 each.
struct __attribute__((aligned(16))) LinTrack {
    double g0[4];          // A, B, C of the track's line; its length ℓ
    double g1[4];          // δs of its azimuthal angle; the first record's p (x, y) and q.x — the track's reserved side-list entry
    double g2[4];          // ... and q.y; the march's direction cos ϕ, sin ϕ (the signs of the Σℓ chain's corrections); unused
    int32_t goff;          // record index in the result arrays = goff + linear slot (this round; below 2^29, see the host's choice of kernel)
    int32_t last;          // linear slot of the track's last record, if it lies in this round (else -1): the end of the Σℓ chain
    double cx[2], cy[2];   // exit point of the last row of the previous round (rounds alternate)
    int32_t el0;           // cell + 1 of the first record
    int32_t lb;            // linear slot of the track's first row of this round
};
static_assert(sizeof(LinTrack) == 144, "LinTrack layout");

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v), o = __builtin_bit_cast(uint64_t, old);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)b, CTRL, ROWMASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(b >> 32), CTRL, ROWMASK, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// intersection(track.ABC, edge.ABC), src/intersection.jl:127-138, with ONE reciprocal for its two quotients.  The compiler's f64
// division is v_div_scale x 2, v_rcp, two Newton steps (four FMAs), q0 = n·r, e = fma(−d, q0, n), v_div_fmas (= fma(e, r, q0) when
// nothing was scaled), v_div_fixup (sign and special values): five of its eleven instructions depend on the denominator alone.
// With |d|, |n| in [2^-200, 2^200] v_div_scale scales nothing and v_div_fixup changes nothing: the sequence below is then the
// compiler's, operation for operation — the same bits.  A numerator that is exactly zero (an exit point ON x = 0 or y = 0) gives
// n·r, which carries the quotient's sign.  `ok` = false: the caller takes the two full divisions (wave-uniformly).
// Measured (round 6, profiles/r06/exp_shared_reciprocal.log, same box, records hash-identical): C3 0.1216 / 0.1197 / 0.1240 ms without,
// 0.1236 / 0.1197 / 0.1240 with; C5 1.237 / 1.218 against 1.221 / 1.233 — nothing: the guards that keep it bit-identical (zero
// numerators, exponent range, the wave-uniform fallback) cost what the two v_rcp_f64 and four v_div_scale_f64 save (44 -> 42
// instructions of the iteration's ≈450), and the kernel spills (20 B of scratch).  Off; -DRT_LIN_SHARED_RCP=1 builds it.
#ifndef RT_LIN_SHARED_RCP
#define RT_LIN_SHARED_RCP 0
#endif
__device__ __forceinline__ bool lin_range_ok(double v) {
    const uint32_t e = ((uint32_t)(__builtin_bit_cast(uint64_t, v) >> 52)) & 0x7ffu;
    return (e - (1023u - 200u)) <= 400u;
}
__device__ __forceinline__ bool edge_exit_point_shared(double tA, double tB, double tC, double eA, double eB, double eC, double &qx, double &qy) {
    const double a = tB * eA;
    const double b = eB * tA;
    const double det = a - b;
    const double nx = tC * eB - eC * tB, ny = tA * eC - eA * tC;
    double r = __builtin_amdgcn_rcp(det);
    r = __builtin_fma(r, __builtin_fma(-det, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-det, r, 1.0), r);
    const double x0 = nx * r, y0 = ny * r;
    const double x1 = __builtin_fma(__builtin_fma(-det, x0, nx), r, x0), y1 = __builtin_fma(__builtin_fma(-det, y0, ny), r, y0);
    qx = nx == 0.0 ? x0 : x1;
    qy = ny == 0.0 ? y0 : y1;
    return lin_range_ok(det) && (nx == 0.0 || lin_range_ok(nx)) && (ny == 0.0 || lin_range_ok(ny));
}

// A load of bytes the march wrote in THIS launch window (QUEUE: the march runs beside this kernel): agent scope — it bypasses the
// CU's L1, which nothing refreshes (the XCD's L2, where the march workgroup of this XCD left them, serves it).
template <bool QUEUE, typename T>
__device__ __forceinline__ T ld_handoff(const RT_G T *p) {
    if (QUEUE) return __hip_atomic_load((T *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

typedef double __attribute__((ext_vector_type(2))) lin_d2;
typedef int32_t __attribute__((ext_vector_type(2))) lin_i2;
typedef int32_t __attribute__((ext_vector_type(4))) lin_i4;
constexpr int kBufWord3 = 0x00020000;  // gfx950 buffer resource word 3: raw buffer, 32-bit data format
constexpr int kLinHalfCap = 16, kLinFlagCap = 64;
struct LinHalf { int64_t o; double px, py, qx, qy, l; int32_t cell, pad; };

constexpr int32_t kWordLast = 1 << 29;  // (side-list references are below it)
constexpr int32_t kWordCode = ~kWordExactTally;
// The record stores are NON-TEMPORAL (aux bit 1 = nt): whole 128-B lines leave the XCD's L2 without displacing the edges' general
// forms and the next units' words — same-box A/B: record kernel −12 % at C3, −7 % at C5, and the march of the NEXT call −3 % (it
// starts in a cache that is not full of dirty record lines).  (Round 3's compaction, whose 256-B runs ended in half-written lines,
// lost 30 % with them.)
#ifndef RT_LIN_STORE_AUX
#define RT_LIN_STORE_AUX 2
#endif
constexpr int kWaitVm0 = 0x0F70;  // s_waitcnt vmcnt(0) (gfx9 encoding: expcnt and lgkmcnt fields at their maxima)

#ifdef RT_LIN_TIMING
#define LIN_STAMP(k) do { if (lane == 0) stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LIN_STAMP(k) do { } while (0)
#endif

#ifndef RT_LIN_OCC
#define RT_LIN_OCC 4
#endif

// QUEUE (round 6, records in COMPLETION order): the kernel runs BESIDE the march, on a second stream, as PERSISTENT workgroups (four
// per CU).  A march workgroup that ends has taken its span of the result arrays from the cursor, written its tracks' offsets —
// off_slot is then not the CSR offset but the track's place in completion order; the 64 W tracks of a march workgroup (W waves) are
// one run of records, every unit the usual single run group, and everything below is unchanged — and queued itself ON ITS XCD
// (DStage::cq: one queue per XCD).  A workgroup here serves the queue of the XCD it runs on (HW_REG_XCC_ID): ticket t of the XCD's
// head = unit t mod (4 W) of the (t / (4 W))-th march workgroup that ended on this XCD.  Producer and consumer share that XCD's L2:
// the march workgroup only drains its stores (s_waitcnt) before it queues itself — no write-back of the L2, whose cost, paid by
// every ending workgroup, slowed the march's remaining chains by half (the first build of this round: one queue for the chip and
// an agent-scope release per march workgroup, march 146 -> 221 µs at C3) — and the consumer invalidates its CU's L1 (ONE agent-scope
// acquire by the polling lane, a wait, the workgroup's barrier; cdna guide, guideline 16) and reads with plain loads.  Nothing here
// assumes a placement: both sides read their XCD from the hardware.  A workgroup leaves when every march workgroup has ended and
// its XCD's queue holds no further unit; one that waits far beyond any march (~0.3 s), or sees another one's give-up flag, sets
// the flag in the control block and leaves: the attempt is void, the host marches again in CSR order.  The host also counts: a
// call whose units were not all served (no record workgroup on some XCD: never observed) writes its records again, in CSR order.
template <bool QUEUE>
__global__ __launch_bounds__(256, RT_LIN_OCC) void k_materialise_lin(DTracks t, int32_t *__restrict__ status, DStage stg, DOut out, DMat a) {
    __shared__ __attribute__((aligned(16))) int32_t s_meta[kLinCap];   // the round's words in output order (0: no record)
    __shared__ __attribute__((aligned(16))) uint8_t s_tmap[kLinCap];   // ... and which of the 16 tracks each belongs to
    __shared__ LinTrack s_trk[16];
    // Σℓ of a track (src/track.jl:171) without adding up its records: the records of a track lie head to tail on its line (p of a
    // record IS the q before it, bit for bit), so Σ‖p_i − q_i‖ = ‖p_first − q_last‖ up to the roundings of the n norms and of their
    // sum — n·2⁻⁵³·Σ, the margin the check already leaves to a sum in another order; only where a record keeps its own p (a
    // generic step's, behind tiny steps) the gap ‖p_i − q_(i−1)‖ is missing from the chain: those are added up here.  What the
    // margin cannot decide, k_finish sums left to right as before.
    __shared__ lin_d2 s_qlast[16];   // exit point of the track's last record
    __shared__ double s_gap[16];     // what the chain misses: Σ gaps in front of records that keep their own p (signed: an overlap counts
                                     // negative), minus twice the length of such a record if it walks backwards
    __shared__ LinHalf s_half[4][kLinHalfCap];   // per wave: half pairs for the epilogue
    __shared__ double s_fval[4][kLinFlagCap];    // per wave: fill_volumes terms of marked records (value, cell) for the epilogue
    __shared__ int32_t s_fcell[4][kLinFlagCap];
    __shared__ int32_t s_qblock, s_qr;
    if (!QUEUE && (stg.cursor[1] != 0 || stg.cursor[3] != 0)) return;  // pool / side list overflow: this attempt is void
    // A workgroup's header phase (first trip to memory, run groups, transposition, up to the barrier behind it) issues with priority
    // over the other workgroups' store loops: its few instructions no longer queue behind four waves of FP64 work per SIMD, its
    // loads leave sooner and the unit reaches its own loop sooner — record kernel −4 % at C3, −6.5 % at C4, −5 % at C5, same box
    // (profiles/r05/exp_issue_priority.log; priority kept until the loop or until behind the first gathers: 1-2 % less; the march
    // does not respond to priorities).  A/B: option "compact_debug" 8 switches it off.
    const bool hprio = !(out.dbg & 8);
    int xcc = 0;
    if (QUEUE) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc = (int)(x & 7u);
    }
    for (;;) {  // QUEUE: one pass per unit this workgroup takes from its XCD's queue; else a single pass
    int64_t q_unit = 0;
    if (QUEUE) {
        __syncthreads();  // (the previous unit's epilogue has read its tables; every wave has read s_qblock / s_qr)
        __builtin_amdgcn_s_setprio(0);
        if (threadIdx.x == 0) {
            // ---- the next unit of THIS XCD: ticket t = unit t mod (4 W) of the (t / (4 W))-th march workgroup that ended here
            const int per = 4 * a.q_waves;
            int32_t *ht = reinterpret_cast<int32_t *>(a.ctl + kCtlCqXcd + xcc);  // [0] tail (march workgroups queued), [1] head (units taken)
            const int32_t tk = atomicAdd(ht + 1, 1);
            const int32_t e = tk / per;
            const RT_G unsigned long long *entry = stg.cq + (int64_t)xcc * stg.cq_blocks + e;
            unsigned long long *gave_up = a.ctl + kCtlCq + 2;
            int32_t mb = -1;
            if (e < stg.cq_blocks) {
                for (unsigned spins = 0;; ++spins) {
                    const unsigned long long v = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(v >> 32) == stg.cq_epoch) { mb = (int32_t)(uint32_t)v; break; }
                    // every march workgroup has ended and fewer than e + 1 of them on this XCD: no such unit — the workgroup is done
                    // (the count of ended workgroups is added to behind the XCD's tail: once it is complete, the tail is final)
                    if (__hip_atomic_load(a.ctl + kCtlCq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)stg.cq_blocks) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (__hip_atomic_load(ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= e) break;
                    }
                    // an exit every workgroup reaches: the host launches this kernel when the march's last workgroups have started and
                    // every march workgroup that ends queues itself — this fires only if the march never does (its argument guard)
                    if ((spins & 63u) == 63u && __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > 300000u) { __hip_atomic_store(gave_up, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(32);
                }
            }
            // the march workgroup ran on THIS XCD: its stores are in this XCD's L2 (it drained them before it queued itself).  What
            // remains is this CU's L1 — which every load of handed-off bytes below bypasses (ld_handoff / non-temporal loads: an
            // acquire here, an invalidate of the L1 per unit, cost ≈7 µs of a unit's ≈14 at four workgroups per CU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (mb >= 0) atomicAdd(a.ctl + kCtlCq + 3, 1ull);  // (the host counts: every unit of every march workgroup has to be taken)
            // pool / side list overflow — flagged by a lane of the march workgroup itself before it queued itself (its chunk ids or
            // side-list references are then void).  ONE lane decides for the workgroup: the flags can change while this kernel runs
            // (another march workgroup overflowing), and 256 loads of them could disagree — threads of one workgroup on either side of
            // a `continue` meet different barriers.  (Round 6, fuzz seed 830802: a memory access fault from exactly that.)
            if (mb >= 0 && (__hip_atomic_load(&stg.cursor[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                            __hip_atomic_load(&stg.cursor[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) mb = -2;
            s_qblock = mb; s_qr = tk - e * per;
        }
        __syncthreads();
        const int32_t mb = s_qblock;
        if (mb == -1) return;
        if (mb < 0) continue;  // (a void unit: the attempt is void, the host re-runs it)
        const int r = s_qr;
        const int64_t wq = (int64_t)mb * a.q_waves + (r >> 2);
        if (wq >= a.n_waves) continue;  // (the batch's last march workgroup may hold fewer waves)
        q_unit = 4 * wq + (r & 3);
    }
    if (hprio) __builtin_amdgcn_s_setprio(3);
    const int kw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // (kw in a scalar register: uniform loops)
    const int tl = lane & 15, rr = lane >> 4;  // transposition: track tl, rows 4 i + rr of a chunk
#ifdef RT_LIN_TIMING
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LIN_STAMP(0);
#endif
    const int64_t unit = QUEUE ? q_unit : (int64_t)blockIdx.x;  // (a unit on the XCD whose march workgroup staged its words was tried: no difference — the words come from the
                                      //  memory-side cache either way, profiles/r05/exp_materialise_variants.log)
    if (unit >= a.n_units) { if (QUEUE) continue; return; }
    // the result arrays as buffer resources (raw, no stride, bounds = the arrays' capacity; the host takes this kernel only for
    // arrays below 4 GB): see the stores
    const int nb8 = (int)(uint32_t)((uint64_t)out.cap << 3), nb4 = (int)(uint32_t)((uint64_t)out.cap << 2);
    const __amdgpu_buffer_rsrc_t r_px = __builtin_amdgcn_make_buffer_rsrc((double *)out.px, 0, nb8, kBufWord3);
    const __amdgpu_buffer_rsrc_t r_py = __builtin_amdgcn_make_buffer_rsrc((double *)out.py, 0, nb8, kBufWord3);
    const __amdgpu_buffer_rsrc_t r_qx = __builtin_amdgcn_make_buffer_rsrc((double *)out.qx, 0, nb8, kBufWord3);
    const __amdgpu_buffer_rsrc_t r_qy = __builtin_amdgcn_make_buffer_rsrc((double *)out.qy, 0, nb8, kBufWord3);
    const __amdgpu_buffer_rsrc_t r_ell = __builtin_amdgcn_make_buffer_rsrc((double *)out.ell, 0, nb8, kBufWord3);
    const __amdgpu_buffer_rsrc_t r_el = __builtin_amdgcn_make_buffer_rsrc((int32_t *)out.element, 0, nb4, kBufWord3);
    const int32_t cap32 = (int32_t)out.cap;
    const bool nostore = (out.dbg & 1) != 0;
    // the exit edges' general forms: gathered through a buffer resource too (32-bit offsets: 3 n_cells < 2^27 entries of 32 B)
    const __amdgpu_buffer_rsrc_t r_etab = __builtin_amdgcn_make_buffer_rsrc((void *)a.etab, 0, a.etab_bytes, kBufWord3);
    const int32_t w = __builtin_amdgcn_readfirstlane((int32_t)((!QUEUE && a.corder) ? a.corder[unit >> 2] : (int32_t)(unit >> 2)));
    const int q = (int)(unit & 3);
    const int64_t slot0 = (int64_t)w * 64 + 16 * q;
    const int64_t slot = slot0 + tl;
    const bool have = slot < t.n;
    const int lane_q = 16 * q + tl;
    const RT_G int32_t *ctab = stg.ctab + (int64_t)w * kMaxChunks;
    // reserved chunks follow from (w, j) and kernel arguments (DStage): scalar loads.  A chunk from the cursor is looked up, and
    // waited for where it is looked up (a wait at the join would also wait for the other chunk's words)
    auto chunk_id = [&](const int j) -> int32_t {
        const int js = __builtin_amdgcn_readfirstlane(j);
        if (js < stg.n_regions && w < stg.reg_cap[js]) return stg.reg_base[js] + w;
        // (QUEUE: the table entry was written by the march beside this kernel — an agent-scope vector load, never the scalar path)
        return __builtin_amdgcn_readfirstlane(QUEUE ? __hip_atomic_load(&ctab[js], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ctab[js]);
    };
    // ---- the unit's first trip to memory, all of it at once: counts and offsets; the words of the wave's two chunks of the first
    // round, if the host reserved them (their ids follow from (w, j) and kernel arguments — whether or not the unit turns out to
    // need them); wave 0 the table: lane (field rr, track tl) reads field rr of (A, B, C, ℓ) and of (δs, first record's p.x, p.y, q.x)
    int32_t cnt = 0;
    int64_t off = 0;
    if (have) { cnt = ld_handoff<QUEUE>(&t.cnt_slot[slot]); off = ld_handoff<QUEUE>(&t.off_slot[slot]); }
    int32_t ve[2][8];
    bool spec[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int j = kw + 4 * c2;
        spec[c2] = j < stg.n_regions && w < stg.reg_cap[j];
        if (spec[c2]) {
            const int32_t c = stg.reg_base[j] + w;
#pragma unroll
            for (int i = 0; i < 8; ++i) ve[c2][i] = __builtin_nontemporal_load(&stg.element[stage_slot(c, 4 * i + rr, lane_q)]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) ve[c2][i] = 0;
        }
    }
    {
        // (one wave-load per wave: a wave that took all three waited for them — and, the counter being in order, for its words —
        //  before it could begin with the run groups, and the other three waited for it at the barrier)
        const int64_t sc = have ? slot : t.n - 1;
        const int64_t side_last = stg.side_cap > 0 ? stg.side_cap - 1 : 0;
        const int64_t ss = sc < side_last ? sc : side_last;
        if (kw == 0) {
            const RT_G double *pg0 = rr == 0 ? t.As : (rr == 1 ? t.Bs : (rr == 2 ? t.Cs : t.Ls));
            s_trk[tl].g0[rr] = pg0[sc];
        } else if (kw == 1) {
            const RT_G double *pg1 = rr == 0 ? (const RT_G double *)t.w_slot : (rr == 1 ? (const RT_G double *)stg.s_px : (rr == 2 ? (const RT_G double *)stg.s_py : (const RT_G double *)stg.s_qx));
            s_trk[tl].g1[rr] = ld_handoff<QUEUE>(&pg1[rr == 0 ? sc : ss]);
        } else if (kw == 2) {
            const RT_G double *pg2 = rr == 0 ? (const RT_G double *)stg.s_qy : (rr == 1 ? t.Dxs : t.Dys);
            s_trk[tl].g2[rr] = ld_handoff<QUEUE>(&pg2[rr == 0 ? ss : sc]);
        } else if (lane < 16) {
            s_trk[tl].el0 = ld_handoff<QUEUE>(&stg.s_el[ss]); s_gap[tl] = 0.0;
        }
    }
    int32_t gmax = cnt;
    for (int o = 8; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(gmax, o, 64);
        gmax = v > gmax ? v : gmax;
    }
    gmax = __builtin_amdgcn_readfirstlane(gmax);
    LIN_STAMP(1);
    const int nrounds = (gmax + kLinRows - 1) / kLinRows;
    const bool many_rounds = nrounds > 1;
    for (int s = 0; s < nrounds; ++s) {
        const int r0 = s * kLinRows;
        if (s > 0) {
            __syncthreads();  // the previous round's linear phase has read its slots
            if (hprio) __builtin_amdgcn_s_setprio(3);
        }
        // ---- the round's words: wave kw takes chunks 8 s + kw and 8 s + kw + 4 (lane = track tl, rows 4 i + rr)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int j = 8 * s + kw + 4 * c2;
            if (__builtin_expect((j << kChunkLog2) < gmax && !(s == 0 && spec[c2]), 0)) {  // a later round's chunk, or one from the pool's cursor: looked up, fetched now
                const int32_t c = chunk_id(j);
#pragma unroll
                for (int i = 0; i < 8; ++i) ve[c2][i] = __builtin_nontemporal_load(&stg.element[stage_slot(c, 4 * i + rr, lane_q)]);
            }
        }
        // ---- linear slots.  Track k's rows of this round follow track k - 1's when their records are adjacent in memory; else a
        // new run group starts, on the next multiple of 16 slots + (record index mod 16): slot ≡ record index (mod 16) everywhere.
        int32_t cr = cnt - r0;
        cr = cr < 0 ? 0 : (cr > kLinRows ? kLinRows : cr);
        const int64_t o = off + r0;
        int32_t end = 0, my_lb = 0, my_gap = 0;
        // The usual unit is ONE run group — sixteen tracks with records whose runs follow one another in memory: every track starts
        // where the one before it ends (one DPP shift), and its first slot is its distance from the first track's.  (The general
        // recurrence below is a chain of sixteen dependent scalar steps: 4,000 cycles of a workgroup's 30,000, in-kernel stamps.)
        const uint32_t o_lo = (uint32_t)(uint64_t)o, o_hi = (uint32_t)((uint64_t)o >> 32);
        const uint32_t po_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_lo, 0x111, 0xf, 0xf, false);  // row_shr:1 — track tl − 1's
        const uint32_t po_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_hi, 0x111, 0xf, 0xf, false);
        const int32_t pcr = __builtin_amdgcn_update_dpp(0, cr, 0x111, 0xf, 0xf, false);
        const int64_t po = (int64_t)(((uint64_t)po_hi << 32) | po_lo);
        const bool one_group = __ballot(cr > 0 && (tl == 0 || o == po + pcr)) == ~0ull;
        if (one_group) {
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)o_lo, 0);
            my_lb = (int32_t)(o0 & 15u) + (int32_t)(o_lo - o0);
            my_gap = tl == 0 ? 0 : my_lb;
            end = __builtin_amdgcn_readlane(my_lb + cr, 15);
        } else {
            int64_t next_o = -1;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int32_t crk = __builtin_amdgcn_readlane(cr, k);
                const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)o, k);
                const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)o >> 32), k);
                const int64_t ok = (int64_t)(((uint64_t)ohi << 32) | olo);
                const int32_t gapk = end;
                int32_t lbk = end;
                if (crk > 0) {
                    if (ok != next_o) lbk = ((end + 15) & ~15) + (int32_t)(olo & 15u);
                    end = lbk + crk;
                    next_o = ok + crk;
                }
                if (tl == k) { my_lb = lbk; my_gap = gapk; }
            }
        }
        const int Lp = (end + 1) & ~1;
        LIN_STAMP(2);
        if (kw == 3 && lane < 16) {
            const int lr = cnt - 1 - r0;
            s_trk[tl].goff = (int32_t)(o - my_lb);
            s_trk[tl].last = (lr >= 0 && lr < kLinRows) ? my_lb + lr : -1;
            s_trk[tl].lb = my_lb;
            for (int k = my_gap; k < my_lb; ++k) s_meta[k] = 0;  // pads in front of a run group
            if (tl == 0 && end < Lp) s_meta[end] = 0;           // ... and behind the last one (pairs)
        }
        // ---- transposition: the words to their slots — a lane's rows of a chunk lie four slots apart (immediate offsets), and
        //      the rows beyond its track's end are those from index nv on (round 5: 14 vector instructions per word -> 2; the
        //      track's last record is no longer marked in the word: its slot is in the table)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int j = 8 * s + kw + 4 * c2;
            const int rb = (j << kChunkLog2) + rr;  // the lane's first row of the chunk
            const int nv = (cnt - rb + 3) >> 2;     // its rows below the track's end (<= 0: none, >= 8: all)
            int32_t *pm = &s_meta[my_lb + (rb - r0)];
            uint8_t *pt = &s_tmap[my_lb + (rb - r0)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (__builtin_expect(i < nv, 1)) { pm[4 * i] = ve[c2][i]; pt[4 * i] = (uint8_t)tl; }
        }
        LIN_STAMP(3);
        __syncthreads();
        LIN_STAMP(4);
        if (hprio) __builtin_amdgcn_s_setprio(0);
        // ---- the linear phase: wave kw takes pairs [m0, m1) (a multiple of 8 pairs = whole cache lines of the f64 arrays)
        const int P = Lp >> 1;
        const int nit = (P + 63) >> 6;  // iterations of 64 pairs, dealt whole: wave kw takes [nit kw / 4, nit (kw + 1) / 4)
        const int m0 = ((nit * kw) >> 2) << 6;  // (rounded down: an iteration that does not divide goes to the higher waves — rounded up,
        const int m1e = ((nit * (kw + 1)) >> 2) << 6;  //  to the lower ones, C5 is 5 % slower and C3 the same: exp_materialise_header.log)
        const int m1 = P < m1e ? P : m1e;
        if (m0 >= m1) continue;
        struct Pre { int32_t w0, w1, t0, t1; double e0A, e0B, e0C, e1A, e1B, e1C; };
        auto prefetch = [&](const int m) -> Pre {
            Pre p;
            p.w0 = 0; p.w1 = 0; p.t0 = 0; p.t1 = 0;
            if (m < m1) {
                const lin_i2 ww = *(const lin_i2 *)&s_meta[2 * m];
                const uint32_t tt = *(const uint16_t *)&s_tmap[2 * m];
                p.w0 = ww.x; p.w1 = ww.y;
                p.t0 = p.w0 != 0 ? (int32_t)(tt & 15u) : 0;
                p.t1 = p.w1 != 0 ? (int32_t)((tt >> 8) & 15u) : 0;
            }
            const uint32_t o0 = (uint32_t)(p.w0 > 0 ? (p.w0 & kWordCode) - 1 : 0) << 5, o1 = (uint32_t)(p.w1 > 0 ? (p.w1 & kWordCode) - 1 : 0) << 5;
            const lin_d2 ab0 = __builtin_bit_cast(lin_d2, __builtin_amdgcn_raw_buffer_load_b128(r_etab, o0, 0, 0));
            const lin_d2 ab1 = __builtin_bit_cast(lin_d2, __builtin_amdgcn_raw_buffer_load_b128(r_etab, o1, 0, 0));
            p.e0A = ab0.x; p.e0B = ab0.y; p.e0C = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_etab, o0 + 16u, 0, 0));
            p.e1A = ab1.x; p.e1B = ab1.y; p.e1C = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_etab, o1 + 16u, 0, 0));
            return p;
        };
        // the exit point of the record in front of the wave's first pair (another wave's record: computed once more, by every lane alike)
        double carry_x = 0.0, carry_y = 0.0;
        int32_t wp = 0, tp = 0;
        if (m0 > 0) { wp = s_meta[2 * m0 - 1]; tp = wp != 0 ? (s_tmap[2 * m0 - 1] & 15) : 0; }
        const RT_G EdgeABC *ep = a.etab + (wp > 0 ? (wp & kWordCode) - 1 : 0);
        const double epA = ep->A, epB = ep->B, epC = ep->C;
        Pre cur = prefetch(m0 + lane);
        {
            edge_exit_point(s_trk[tp].g0[0], s_trk[tp].g0[1], s_trk[tp].g0[2], epA, epB, epC, carry_x, carry_y);
            if (__builtin_expect(wp < 0, 0)) {
                const int64_t idx = (-(int64_t)wp - 1) & (kWordLast - 1);
                if (idx == slot0 + tp) { carry_x = s_trk[tp].g1[3]; carry_y = s_trk[tp].g2[0]; }
                else { carry_x = ld_handoff<QUEUE>(&stg.s_qx[idx]); carry_y = ld_handoff<QUEUE>(&stg.s_qy[idx]); __builtin_amdgcn_s_waitcnt(kWaitVm0); }
            }
        }
        int n_half = 0, n_flag = 0;  // (wave-uniform: the lists' fill)
        auto flush_flags = [&]() {
            if (n_flag > 0) {
                for (int e = lane; e < n_flag; e += 64) unsafeAtomicAdd((double *)&a.vacc[s_fcell[kw][e]], s_fval[kw][e]);
                n_flag = 0;
            }
        };
#ifdef RT_LIN_LEAN
        auto process = [&](Pre &pre, const int mb) {
#else
        auto process = [&](const Pre &pre, const int mb) {
#endif
            const int m = mb + lane;
            const int32_t w0 = pre.w0, w1 = pre.w1;
            const int t0 = pre.t0, t1 = pre.t1;
            // the tracks' lines and offsets (mostly one track per wave: broadcast reads)
            const lin_d2 ab0 = *(const lin_d2 *)&s_trk[t0].g0[0], ab1 = *(const lin_d2 *)&s_trk[t1].g0[0];
            const double c0 = s_trk[t0].g0[2], c1 = s_trk[t1].g0[2];
            const lin_i2 gl0 = *(const lin_i2 *)&s_trk[t0].goff, gl1 = *(const lin_i2 *)&s_trk[t1].goff;  // (goff, last)
            const int32_t g0 = gl0.x, g1 = gl1.x;
            double q0x, q0y, q1x, q1y;
#if RT_LIN_SHARED_RCP
            {
                const bool ok0 = edge_exit_point_shared(ab0.x, ab0.y, c0, pre.e0A, pre.e0B, pre.e0C, q0x, q0y);
                const bool ok1 = edge_exit_point_shared(ab1.x, ab1.y, c1, pre.e1A, pre.e1B, pre.e1C, q1x, q1y);
                // (a word of 0 — no record — gathers entry 0 against whatever line: only lanes WITH a record count)
                if (__builtin_expect(__ballot((w0 > 0 && !ok0) || (w1 > 0 && !ok1)) != 0, 0)) {
                    edge_exit_point(ab0.x, ab0.y, c0, pre.e0A, pre.e0B, pre.e0C, q0x, q0y);
                    edge_exit_point(ab1.x, ab1.y, c1, pre.e1A, pre.e1B, pre.e1C, q1x, q1y);
                }
            }
#else
            edge_exit_point(ab0.x, ab0.y, c0, pre.e0A, pre.e0B, pre.e0C, q0x, q0y);  // src/intersection.jl:127-138
            edge_exit_point(ab1.x, ab1.y, c1, pre.e1A, pre.e1B, pre.e1C, q1x, q1y);
#endif
            int32_t cell0 = (int32_t)((uint32_t)(w0 > 0 ? (w0 & kWordCode) - 1 : 0) / 3u) + 1;
            int32_t cell1 = (int32_t)((uint32_t)(w1 > 0 ? (w1 & kWordCode) - 1 : 0) / 3u) + 1;
            // records that keep their own end points (the generic step's): every track's first one from the table, the others
            // (refusals' fall-backs) from the side list
            bool own0 = false, own1 = false, gap0 = false, gap1 = false;  // gap: a record with its own p that is not its track's first
            double o0x = 0.0, o0y = 0.0, o1x = 0.0, o1y = 0.0;
            if (w0 < 0 || w1 < 0) {
                bool slow = false;
                const int32_t i0 = (-w0 - 1) & (kWordLast - 1), i1 = (-w1 - 1) & (kWordLast - 1);  // side-list entries (if w < 0)
                if (w0 < 0) {
                    own0 = true;
                    if ((int64_t)i0 == slot0 + t0) { q0x = s_trk[t0].g1[3]; q0y = s_trk[t0].g2[0]; o0x = s_trk[t0].g1[1]; o0y = s_trk[t0].g1[2]; cell0 = s_trk[t0].el0; }
                    else slow = true;
                }
                if (w1 < 0) {
                    own1 = true;
                    if ((int64_t)i1 == slot0 + t1) { q1x = s_trk[t1].g1[3]; q1y = s_trk[t1].g2[0]; o1x = s_trk[t1].g1[1]; o1y = s_trk[t1].g1[2]; cell1 = s_trk[t1].el0; }
                    else slow = true;
                }
                if (__builtin_expect(slow, 0)) {
                    if (w0 < 0 && (int64_t)i0 != slot0 + t0) {
                        gap0 = true;
                        const int32_t idx = i0;
                        q0x = ld_handoff<QUEUE>(&stg.s_qx[idx]); q0y = ld_handoff<QUEUE>(&stg.s_qy[idx]); o0x = ld_handoff<QUEUE>(&stg.s_px[idx]);
                        o0y = ld_handoff<QUEUE>(&stg.s_py[idx]); cell0 = ld_handoff<QUEUE>(&stg.s_el[idx]);
                    }
                    if (w1 < 0 && (int64_t)i1 != slot0 + t1) {
                        gap1 = true;
                        const int32_t idx = i1;
                        q1x = ld_handoff<QUEUE>(&stg.s_qx[idx]); q1y = ld_handoff<QUEUE>(&stg.s_qy[idx]); o1x = ld_handoff<QUEUE>(&stg.s_px[idx]);
                        o1y = ld_handoff<QUEUE>(&stg.s_py[idx]); cell1 = ld_handoff<QUEUE>(&stg.s_el[idx]);
                    }
                    __builtin_amdgcn_s_waitcnt(kWaitVm0);  // (waited for here, not at the join with the hot path: that wait would cover the gathers in flight)
                }
            }
            if (__builtin_expect(many_rounds, 0)) {
                // the first row of a later round starts where the previous round's last row ended; this round's last rows are kept
                if (s > 0) {
                    if (w0 > 0 && 2 * m == s_trk[t0].lb) { own0 = true; o0x = s_trk[t0].cx[s & 1]; o0y = s_trk[t0].cy[s & 1]; }
                    if (w1 > 0 && 2 * m + 1 == s_trk[t1].lb) { own1 = true; o1x = s_trk[t1].cx[s & 1]; o1y = s_trk[t1].cy[s & 1]; }
                }
                if (w0 != 0 && 2 * m == s_trk[t0].lb + kLinRows - 1) { s_trk[t0].cx[(s + 1) & 1] = q0x; s_trk[t0].cy[(s + 1) & 1] = q0y; }
                if (w1 != 0 && 2 * m + 1 == s_trk[t1].lb + kLinRows - 1) { s_trk[t1].cx[(s + 1) & 1] = q1x; s_trk[t1].cy[(s + 1) & 1] = q1y; }
            }
            // p = the exit point of the slot before: the neighbouring lane's second record (lane 0: the carry)
            const double prev_x = dpp_f64<0x138, 0xf>(carry_x, q1x), prev_y = dpp_f64<0x138, 0xf>(carry_y, q1y);
            carry_x = readlane_f64(q1x, 63); carry_y = readlane_f64(q1y, 63);
            const double p0x = own0 ? o0x : prev_x, p0y = own0 ? o0y : prev_y;
            const double p1x = own1 ? o1x : q0x, p1y = own1 ? o1y : q0y;
            const double l0 = norm2(p0x - q0x, p0y - q0y);  // Segment ctor, src/segment.jl:31-33
            const double l1 = norm2(p1x - q1x, p1y - q1y);
            if (__builtin_expect(__ballot(gap0 || gap1) != 0, 0)) {  // (the gaps in the chain of Σℓ, see s_qlast)
                // (the record in front: the slot before — or, for the first row of a later round, the last row of the round before)
                double b0x = prev_x, b0y = prev_y, b1x = q0x, b1y = q0y;
                if (many_rounds && s > 0) {
                    if (gap0 && 2 * m == s_trk[t0].lb) { b0x = s_trk[t0].cx[s & 1]; b0y = s_trk[t0].cy[s & 1]; }
                    if (gap1 && 2 * m + 1 == s_trk[t1].lb) { b1x = s_trk[t1].cx[s & 1]; b1y = s_trk[t1].cy[s & 1]; }
                }
                // Signed along the MARCH (its direction cos ϕ, sin ϕ is in the table): a record that begins BEHIND the
                // exit point before it (cells that overlap within the locate's tolerance) adds its overlap to Σℓ instead of leaving a
                // gap; and a record whose own two points are in the wrong order — order_intersection_points compares x coordinates,
                // src/intersection.jl:151-159, which near ϕ = π/2 are equal to the last bit — walks BACKWARDS: the chain loses its length
                // twice.  (Records that take p from the record before are never reversed: the walk step's certificate 6.)
                if (a.tally && (gap0 || gap1)) {
                    const double d0x = s_trk[t0].g2[1], d0y = s_trk[t0].g2[2], d1x = s_trk[t1].g2[1], d1y = s_trk[t1].g2[2];
                    if (gap0) {
                        atomicAdd(&s_gap[t0], chain_gap_term(p0x, p0y, q0x, q0y, l0, b0x, b0y, d0x, d0y));
                        if (2 * m == gl0.y) { lin_d2 v; v.x = q0x; v.y = q0y; s_qlast[t0] = v; }  // ... and the chain's end, if it is the last record
                    }
                    if (gap1) {
                        atomicAdd(&s_gap[t1], chain_gap_term(p1x, p1y, q1x, q1y, l1, b1x, b1y, d1x, d1y));
                        if (2 * m + 1 == gl1.y) { lin_d2 v; v.x = q1x; v.y = q1y; s_qlast[t1] = v; }
                    }
                }
            }
            const int32_t oa = g0 + 2 * m, ob = g1 + 2 * m + 1;
            const bool v0 = w0 != 0 && oa < cap32 && !nostore, v1 = w1 != 0 && ob < cap32 && !nostore;
#ifdef RT_LIN_LEAN
            // (lean variant: ONE register set — the next pairs' words and gathers are fetched here, behind the last use of this
            //  iteration's and in front of its stores: fewer registers for more waves per SIMD, the gathers' latency is the other waves')
            pre = prefetch(mb + 64 + lane);
#endif
            {
                // Whole pairs: one 16-B store per f64 array, 8 B of cell ids.  BUFFER stores: a lane without a whole pair hands in an
                // offset beyond the array and the hardware drops its store — the six stores are unconditional instructions, so the
                // compiler's count of what is in flight behind the next gathers is exact (behind a branch it has to assume that no
                // store was issued, and the wait for the gathers then waits for the stores too).
                const bool full = v0 && v1;
                const uint32_t vo8 = full ? (uint32_t)oa << 3 : 0xffffffffu;
                const uint32_t vo4 = full ? (uint32_t)oa << 2 : 0xffffffffu;
                lin_d2 v;
                v.x = p0x; v.y = p1x; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lin_i4, v), r_px, vo8, 0, RT_LIN_STORE_AUX);
                v.x = p0y; v.y = p1y; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lin_i4, v), r_py, vo8, 0, RT_LIN_STORE_AUX);
                v.x = q0x; v.y = q1x; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lin_i4, v), r_qx, vo8, 0, RT_LIN_STORE_AUX);
                v.x = q0y; v.y = q1y; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lin_i4, v), r_qy, vo8, 0, RT_LIN_STORE_AUX);
                v.x = l0; v.y = l1; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lin_i4, v), r_ell, vo8, 0, RT_LIN_STORE_AUX);
                lin_i2 c;
                c.x = cell0; c.y = cell1; __builtin_amdgcn_raw_buffer_store_b64(c, r_el, vo4, 0, RT_LIN_STORE_AUX);
            }
            // Half pairs — the first or last record of a run group whose neighbour in memory is not this unit's — wait in LDS for the
            // wave's epilogue (at most two per run group; a list that is full writes directly)
            const bool h0 = v0 && !v1, h1 = v1 && !v0;
            const unsigned long long hm = __ballot(h0 || h1);
            if (__builtin_expect(hm != 0, 0)) {
                if (h0 || h1) {
                    const int pos = n_half + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                    const int64_t oh = h0 ? oa : ob;
                    const double hpx = h0 ? p0x : p1x, hpy = h0 ? p0y : p1y, hqx = h0 ? q0x : q1x, hqy = h0 ? q0y : q1y, hl = h0 ? l0 : l1;
                    const int32_t hc = h0 ? cell0 : cell1;
                    if (pos < kLinHalfCap) {
                        LinHalf &e = s_half[kw][pos];
                        e.o = oh; e.px = hpx; e.py = hpy; e.qx = hqx; e.qy = hqy; e.l = hl; e.cell = hc;
                    } else {
                        out.px[oh] = hpx; out.py[oh] = hpy; out.qx[oh] = hqx; out.qy[oh] = hqy; out.ell[oh] = hl; out.element[oh] = hc;
                    }
                }
                n_half += __popcll(hm);
            }
            if (a.tally) {
                // Σℓ: the chain's end (see s_qlast)
                const bool z0 = w0 > 0 && 2 * m == gl0.y, z1 = w1 > 0 && 2 * m + 1 == gl1.y;
                if (__ballot(z0 || z1)) {
                    lin_d2 v;
                    if (z0) { v.x = q0x; v.y = q0y; s_qlast[t0] = v; }
                    if (z1) { v.x = q1x; v.y = q1y; s_qlast[t1] = v; }
                }
                // fill_volumes (src/trackgenerator.jl:382) for the records the march left out (δs[azim]·ℓ with the record's own
                // length): collected in LDS, added by the wave's epilogue — atomics inside this loop would be conditional
                // vector-memory instructions that most iterations execute (3.7 % of C3's records are marked)
                const bool f0 = w0 > 0 && (w0 & kWordExactTally) != 0, f1 = w1 > 0 && (w1 & kWordExactTally) != 0;
                const unsigned long long fm0 = __ballot(f0), fm1 = __ballot(f1);
                if (fm0 | fm1) {
                    const int n0 = __popcll(fm0), n1 = __popcll(fm1);
                    if (n_flag + n0 + n1 > kLinFlagCap) { flush_flags(); }
                    if (__builtin_expect(n0 + n1 > kLinFlagCap, 0)) {  // (a mesh far from the origin: every record is marked)
                        if (f0) unsafeAtomicAdd((double *)&a.vacc[cell0 - 1], s_trk[t0].g1[0] * l0);
                        if (f1) unsafeAtomicAdd((double *)&a.vacc[cell1 - 1], s_trk[t1].g1[0] * l1);
                    } else {
                        if (f0) {
                            const int pos = n_flag + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm0, 0u));
                            s_fval[kw][pos] = s_trk[t0].g1[0] * l0; s_fcell[kw][pos] = cell0 - 1;
                        }
                        if (f1) {
                            const int pos = n_flag + n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm1, 0u));
                            s_fval[kw][pos] = s_trk[t1].g1[0] * l1; s_fcell[kw][pos] = cell1 - 1;
                        }
                        n_flag += n0 + n1;
                    }
                }
            }
        };
        // two iterations per trip, two sets of registers: the gathers of the next 64 pairs are issued before this iteration's
        // stores, and nothing copies a value that is still in flight.  The prefetch is unconditional (beyond the wave's range it
        // gathers entry 0): behind a branch the compiler's wait for the CURRENT gathers would cover the next ones too
        {
            // (six stores that the hardware drops, so that what is in flight behind the first gathers looks at the loop's entry as it
            //  does on its back edge — gathers, six stores, gathers: the compiler takes the smaller count of the two paths)
            const lin_i4 z4 = {0, 0, 0, 0};
            const lin_i2 z2 = {0, 0};
            __builtin_amdgcn_raw_buffer_store_b128(z4, r_px, 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z4, r_py, 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z4, r_qx, 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z4, r_qy, 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z4, r_ell, 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(z2, r_el, 0xffffffffu, 0, 0);
        }
        LIN_STAMP(5);
#ifdef RT_LIN_LEAN
        for (int mb = m0; mb < m1; mb += 64) process(cur, mb);
#else
        for (int mb = m0; mb < m1; mb += 128) {
            const Pre nxt = prefetch(mb + 64 + lane);
            process(cur, mb);
            if (mb + 64 >= m1) break;
            cur = prefetch(mb + 128 + lane);
            process(nxt, mb + 64);
        }
#endif
        LIN_STAMP(6);
        // (the epilogue — half pairs, flagged terms, the barrier and the Σℓ check of the unit's tracks — with priority again: a
        //  workgroup that has stored its records should leave its slots, not wait for issue cycles; C5 −3.4 %, C3 −0.5 %, same box)
        if (hprio) __builtin_amdgcn_s_setprio(3);
        // ---- the wave's epilogue: half pairs, marked records' fill_volumes terms
        if (__builtin_expect(n_half != 0, 0)) {
            const int ne = n_half < kLinHalfCap ? n_half : kLinHalfCap;
            if (lane < ne) {
                const LinHalf &e = s_half[kw][lane];
                const int64_t oh = e.o;
                out.px[oh] = e.px; out.py[oh] = e.py; out.qx[oh] = e.qx; out.qy[oh] = e.qy; out.ell[oh] = e.l; out.element[oh] = e.cell;
            }
        }
        flush_flags();
    }
#ifdef RT_LIN_TIMING
    if (lane == 0 && a.dbg) {
        LIN_STAMP(7);
        if (kw < 2) {  // (a slot per workgroup and wave 0 / 1: no atomics — 260 k waves on one line would be the measurement)
            unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 2 + kw) * 8;
            for (int k = 0; k < 8; ++k) d[k] = stamp[k];
        }
    }
#endif
    if (a.tally) {
        __syncthreads();
        if (threadIdx.x < 16 && have) {
            // Σℓ = first record + chain from its q to the last record's q − gaps (a track of one record: the first record alone)
            const double fx = s_trk[tl].g1[1], fy = s_trk[tl].g1[2], gx = s_trk[tl].g1[3], gy = s_trk[tl].g2[0];
            const lin_d2 ql = s_qlast[tl];
            // (the chain as a PROJECTION on the march direction: a last record that walks backwards may end behind the first one's q)
            const double S = chain_sum(fx, fy, gx, gy, ql.x, ql.y, s_trk[tl].g2[1], s_trk[tl].g2[2], s_gap[tl], cnt);
            const double L = s_trk[tl].g0[3];
            // (rt_device.hpp, chain_status: inside the band of what a sum in another order — or the chain's own error — could decide
            //  differently, k_finish sums left to right)
            const int cs = chain_status(L, S, a.rtol, cnt, a.coord_max);
            if (a.force_exact || cs == 2) {
                const int32_t e = atomicAdd((int32_t *)&a.marg[0], 1);
                if (e < a.marg_cap) a.marg[1 + e] = (int32_t)slot;  // (marg_cap = every march slot: cannot overflow)
            } else if (cs == 1) {  // src/track.jl:171-175
                const int32_t u = t.perm[slot];
                if (ld_handoff<QUEUE>((const RT_G int32_t *)&status[u]) == RT_TRACK_OK) {
                    status[u] = RT_TRACK_LENGTH_MISMATCH;
                    atomicAdd(&a.ctl[0], 1ull);
                    atomicMin(&a.ctl[1], (unsigned long long)(u + 1));
                }
            }
        }
    }
    if (!QUEUE) break;
    }  // for (;;): the units of a QUEUE workgroup
}

}  // namespace rt

namespace rtx {

// The launch of k_materialise_lin for the plan of the last two-phase call (records only; rows for rt_sweep: k_materialise).
void launch_materialise_lin(const rt::DTracks &d, int32_t *status, const rt::DStage &stg, const rt::DOut &out, const rt::DMat &a_in, hipStream_t s,
                            int n_cus, bool queue) {
    rt::DMat a = a_in;
    const unsigned blocks = (unsigned)a.n_units;
#ifdef RT_LIN_TIMING
    // development: the kernel's stamps per workgroup (waves 0 and 1), averaged per launch
    static unsigned long long *dbg = nullptr;
    static size_t dbg_cap = 0;
    const size_t need = (size_t)blocks * 16;
    if (dbg_cap < need) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void **)&dbg, need * sizeof(unsigned long long)); dbg_cap = need; }
    (void)hipMemsetAsync(dbg, 0, need * sizeof(unsigned long long), s);
    a.dbg = dbg;
#endif
    if (queue) {
        // persistent workgroups: as many as fit on the chip at once (four per CU), never more than there are units
        const unsigned qblocks = (unsigned)std::min<int64_t>((int64_t)std::max(1, n_cus) * RT_LIN_OCC, a.n_units);
        hipLaunchKernelGGL(rt::k_materialise_lin<true>, dim3(qblocks), dim3(256), 0, s, d, status, stg, out, a);
    } else {
        hipLaunchKernelGGL(rt::k_materialise_lin<false>, dim3(blocks), dim3(256), 0, s, d, status, stg, out, a);
    }
#ifdef RT_LIN_TIMING
    static int calls = 0;
    if (++calls % 16 == 0) {
        std::vector<unsigned long long> h(need);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), dbg, need * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int g = 0; g < 2; ++g) {
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            size_t n = 0;
            for (size_t b = 0; b < blocks; ++b) {
                const unsigned long long *st = &h[(b * 2 + g) * 8];
                if (!st[0] || !st[7]) continue;
                ++n;
                tmin = std::min(tmin, st[0]); tmax = std::max(tmax, st[7]);
                for (int k = 1; k < 8; ++k) sum[k] += st[k] > st[k - 1] && st[k - 1] ? (double)(st[k] - st[k - 1]) : 0.0;
                sum[0] += (double)(st[7] - st[0]);
            }
            const double nn = n ? (double)n : 1.0;
            fprintf(stderr, "[lin timing wave %d] n=%zu ticks: cnt %.0f | words+bases %.0f | scatter %.0f | barrier %.0f | pred+issue %.0f | loop %.0f | epilogue %.0f | total %.0f | kernel span %.0f\n",
                    g, n, sum[1] / nn, sum[2] / nn, sum[3] / nn, sum[4] / nn, sum[5] / nn, sum[6] / nn, sum[7] / nn, sum[0] / nn, (double)(tmax - tmin));
        }
    }
#endif
}

}  // namespace rtx
