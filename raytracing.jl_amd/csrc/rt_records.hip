// rt_records.hip — from the march's staging to the results (gfx950): the exclusive scan of the counts (CSR offsets),
// k_compact3 (exact-step staging rows -> records), k_materialise + k_finish (the two-phase march's words -> records, Σℓ,
// status), fill_volumes as its own pass, Segment.τ.  Launched through the helpers at the end of this file (rt_internal.hpp).
#include "rt_internal.hpp"

namespace rt {

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
#ifndef RT_MAT_OCC
#define RT_MAT_OCC 4  // waves per SIMD the kernel is compiled for (131 VGPRs unforced: six spilled values, four workgroups per CU; measured -1 … -2 % per step against 3)
#endif
// LDS: one tile per wave, 16 tracks x (1 + 32) slots of 16 B.  A slot first holds a row's exit point (x, y) — slot 0 of a
// track: the row before the chunk, i.e. the first row's entry point —, written in the load mapping and read back in the store
// mapping (one ds_read_b128 / ds_write_b128 per point: the first version kept x and y in two tiles of doubles, twice the LDS
// instructions, and was LDS-issue-bound beside its arithmetic), then the row's (ℓ, cell).  Pitch 33 slots: the 16 lanes of a
// row of the load mapping fall on different banks.
constexpr int kMatPitch = kChunkRows + 1;
typedef double __attribute__((ext_vector_type(2))) rt_d2;
template <bool RECORDS, bool ROWS>
__global__ __launch_bounds__(256, RT_MAT_OCC) void k_materialise(DTracks t, const int32_t *__restrict__ counts, int32_t *__restrict__ status,
                                                                const int64_t *__restrict__ offsets, DStage stg, DOut out, DMat a) {
    static_assert(kChunkRows == 32, "k_materialise moves 32-row chunks");
    __shared__ rt_d2 tiles[4][16 * kMatPitch];
    __shared__ double s_sum[16];   // Σℓ of the unit's tracks
    __shared__ rt_d2 s_track[16];  // per track: CSR offset and record count (as bit patterns)
    __shared__ double s_w[16];     // per track: δs of its azimuthal angle (fill_volumes terms of marked records)
    if (stg.cursor[1] != 0 || stg.cursor[3] != 0) return;  // pool / side list overflow: this attempt is void
    const int kw = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = lane & 15, rr = lane >> 4;     // load mapping: track tl, rows rr, rr + 4, ...
    const int rowL = lane & 31, sub = lane >> 5;  // store mapping: row rowL of tracks sub, sub + 2, ...
    typedef __attribute__((address_space(3))) volatile rt_d2 lds_d2;
    lds_d2 *T = (lds_d2 *)tiles[kw];
    lds_d2 *strk = (lds_d2 *)s_track;
    const int64_t unit = blockIdx.x;
    const int64_t w = a.corder ? a.corder[unit >> 2] : (unit >> 2);
    const int q = (int)(unit & 3);
    const int64_t slot = w * 64 + 16 * q + tl;
    // (everything a unit needs first is read in march-slot order, side by side: counts, offsets, lines, the wave's first chunk id)
    int32_t cnt = 0, u = 0;
    int64_t off = 0;
    double tA = 0.0, tB = 0.0, tC = 0.0;
    const bool have = slot < t.n;
    const RT_G int32_t *ctab = stg.ctab + w * kMaxChunks;
    // chunk j of this march wave: reserved chunks follow from (w, j) and kernel arguments — the wave's words can be fetched with its
    // very first loads; only chunks beyond the host's estimate are looked up (a load the words would have to wait for)
    auto chunk_id = [&](const int j) -> int32_t {
        const int js = __builtin_amdgcn_readfirstlane(j);
        if (js < stg.n_regions && w < stg.reg_cap[js]) return stg.reg_base[js] + (int32_t)w;
        return ctab[js];
    };
    const int32_t c_first = chunk_id(kw);
    if (have) {
        u = t.perm[slot];
        cnt = t.cnt_slot[slot];
        off = t.off_slot[slot];
        tA = t.As[slot]; tB = t.Bs[slot]; tC = t.Cs[slot];
    }
    if (threadIdx.x < 16) {
        s_w[tl] = (have && a.tally) ? t.w_slot[slot] : 0.0;
        rt_d2 v;
        v.x = __builtin_bit_cast(double, off); v.y = __builtin_bit_cast(double, (int64_t)cnt);
        strk[tl] = v;
        s_sum[tl] = 0.0;
    }
    int32_t gmax = cnt;
    for (int o = 8; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(gmax, o, 64);
        gmax = v > gmax ? v : gmax;
    }
    gmax = __shfl(gmax, 0, 64);
    __syncthreads();
    const int lane_q = 16 * q + tl;
    const int tb = tl * kMatPitch;
    double acc = 0.0;  // Σℓ of this lane's rows of its load-mapping track
    for (int j = kw; (j << kChunkLog2) < gmax; j += 4) {
        const int r0 = j << kChunkLog2;
        const int32_t c = j == kw ? c_first : chunk_id(j);
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
            if (lane < 16 && cnt > r0) hw = stg.element[stage_slot(chunk_id(j - 1), kChunkRows - 1, lane_q)];
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
                rt_d2 qv;
                double qx, qy;
                edge_exit_point(tA, tB, tC, eA[i2], eB[i2], eC[i2], qx, qy);  // src/intersection.jl:127-138
                if (__builtin_expect(ve[i] < 0, 0)) {  // a record of the generic step: its own q (and cell)
                    const int32_t idx = -ve[i] - 1;
                    qx = stg.s_qx[idx]; qy = stg.s_qy[idx];
                    if (i == 0 && rr == 0) ve[i] = 3 * (stg.s_el[idx] - 1) + 1;  // (its p sits in slot 0: from here on an ordinary word)
                    else slow = true;
                }
                qv.x = qx; qv.y = qy;
                T[tb + 1 + 4 * i + rr] = qv;
            }
        }
        if (lane < 16) { rt_d2 hv; hv.x = hx; hv.y = hy; T[tb] = hv; }
        const bool any_slow = __ballot(slow) != 0;
        __builtin_amdgcn_wave_barrier();
        // ---- ℓ = ‖p − q‖ (Segment ctor, src/segment.jl:31-33) in the load mapping; p, q to the output in the store mapping
        double dl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = 4 * i + rr;
            rt_d2 pv = T[tb + rl];
            const rt_d2 qv = T[tb + rl + 1];
            if (__builtin_expect(any_slow, 0))
                if (ve[i] < 0) { pv.x = stg.s_px[-ve[i] - 1]; pv.y = stg.s_py[-ve[i] - 1]; }
            dl[i] = norm2(pv.x - qv.x, pv.y - qv.y);
            acc += ve[i] != 0 ? dl[i] : 0.0;
        }
        if (__builtin_expect(any_flagged, 0)) {
            // fill_volumes (src/trackgenerator.jl:382) for the records the march left out: δs[azim]·ℓ with the record's own length
            const double wt = s_w[tl];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((fmask >> i) & 1) unsafeAtomicAdd((double *)&a.vacc[(int32_t)((uint32_t)(ve[i] - 1) / 3u)], wt * dl[i]);
        }
        if (RECORDS) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int tt = 2 * g + sub;
                const rt_d2 trk = strk[tt];
                const int row = r0 + rowL;
                const int64_t o = __builtin_bit_cast(int64_t, (double)trk.x) + row;
                const rt_d2 pv = T[tt * kMatPitch + rowL], qv = T[tt * kMatPitch + rowL + 1];
                // plain stores: the partial lines at the ends of a run wait in L2 for the sibling wave's half
                if (row < (int32_t)__builtin_bit_cast(int64_t, (double)trk.y) && o < out.cap && !((out.dbg & 1) && pv.x != -1.25)) {
                    out.px[o] = pv.x; out.py[o] = pv.y; out.qx[o] = qv.x; out.qy[o] = qv.y;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lengths and cells through the tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int32_t wd = ve[i];
            if (__builtin_expect(any_slow, 0))
                if (wd < 0) wd = 3 * (stg.s_el[-wd - 1] - 1) + 1;
            const int32_t cell = (int32_t)((uint32_t)(wd > 0 ? wd - 1 : 0) / 3u) + 1;
            rt_d2 lv;
            lv.x = dl[i]; lv.y = __builtin_bit_cast(double, (int64_t)cell);
            T[tb + 1 + 4 * i + rr] = lv;
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
                const rt_d2 trk = strk[tt];
                const int row = r0 + rowL;
                const int64_t o = __builtin_bit_cast(int64_t, (double)trk.x) + row;
                const rt_d2 lv = T[tt * kMatPitch + 1 + rowL];
                if (row < (int32_t)__builtin_bit_cast(int64_t, (double)trk.y) && o < out.cap && !((out.dbg & 1) && lv.x != -1.25)) {
                    out.ell[o] = lv.x; out.element[o] = (int32_t)__builtin_bit_cast(int64_t, (double)lv.y);
                }
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
        __builtin_amdgcn_wave_barrier();  // the tile is rewritten if this wave has a further chunk
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

// After k_materialise, ONE workgroup: volumes ./= n_azim_2; the tracks whose Σℓ check a sum in another order cannot decide are
// summed left to right — from the records, or from the ℓ rows when the call wrote no records — and checked as the reference
// does (src/track.jl:171-175), the statistic of rt_last_stats (tracks within 64 summation-order bands of the threshold) is
// counted; then the control block is copied to the host and the call's sequence number written behind it (what a two-phase
// call's host waits for).  (A single workgroup: no ticket between blocks — the list is empty in practice and the volumes are a
// few thousand values.)
constexpr int kFinishThreads = 1024;
__global__ __launch_bounds__(kFinishThreads) void k_finish(DTracks t, const int32_t *__restrict__ counts, int32_t *__restrict__ status,
                                                           const int64_t *__restrict__ offsets, const double *__restrict__ ell, int64_t cap,
                                                           DStage stg, const double *__restrict__ ell_rows, double rtol, int32_t *__restrict__ marg,
                                                           double *__restrict__ volumes, double *__restrict__ vacc, int32_t n_cells, double n_azim_2,
                                                           unsigned long long *__restrict__ ctl, unsigned long long *__restrict__ host_copy,
                                                           unsigned long long seq, const int64_t *__restrict__ off_by_slot) {
    // (off_by_slot: the records lie in completion order — a track's first record is off_slot[march slot], not the CSR offset)
    const bool void_attempt = stg.cursor[1] != 0 || stg.cursor[3] != 0;
    // volumes ./= n_azim_2 (src/trackgenerator.jl:386): the march accumulated into `vacc` (k_materialise added the terms of the
    // records the march left to it), which is read, scaled into `volumes` and left ZERO for the next call's march
    // (eight independent loads per thread and round: `vacc` was written by atomics from every XCD, each load is a trip to memory)
    if (volumes)
        for (int c0 = threadIdx.x; c0 < n_cells; c0 += 8 * kFinishThreads) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k * kFinishThreads;
                v[k] = c < n_cells ? vacc[c] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k * kFinishThreads;
                if (c < n_cells) { volumes[c] = v[k] / n_azim_2; vacc[c] = 0.0; }
            }
        }
    if (!void_attempt) {
        const int32_t nm = marg[0];
        for (int32_t e = threadIdx.x; e < nm; e += kFinishThreads) {
            const int32_t slot = marg[1 + e];
            if (slot < 0) continue;  // done by an earlier pass
            const int32_t u = t.perm[slot];
            const int32_t cnt = counts[u];
            const int64_t off = off_by_slot ? off_by_slot[slot] : offsets[u];
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
    __syncthreads();  // (the counters above are device-scope atomics; what this kernel wrote to device memory is flushed at its end)
    if (threadIdx.x >= 64) return;  // (one wave copies: a system-scope release per wave is a cache write-back per wave)
    if (threadIdx.x == 0 && __hip_atomic_load(&ctl[kCtlDeferred], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) marg[0] = 0;  // the list is consumed
    if (host_copy) {
        static_assert(kCtlWords <= 64, "one wave copies the control block");
        host_copy[threadIdx.x] = __hip_atomic_load(&ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the lanes' stores and lane 0's release store are one wave's instructions, in order; the release waits for them)
        if (threadIdx.x == 0) __hip_atomic_store(&host_copy[kCtlWords], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- exclusive scan of per-track counts (int32) into CSR offsets (int64) ----------------

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

// The offsets' scan of a two-phase call in ONE launch: k_march<TOPO> has already added every wave's record count to the sum of
// its tile of kScanTile uids (DStage::tile_acc), so a block needs no other block's result — it adds up the tile sums in front of
// its own tile (n_tiles values: 128 at the headline configuration, 1,019 for a BWR assembly), scans its own 1,024 counts and writes
// its offsets (uid order, and — through iperm — march-slot order for k_materialise).  The last tile's block writes the total and
// resets the OTHER control block; every block clears its entry of the OTHER tile-sum buffer: calls alternate between two, so that
// a call in the steady state starts with its march.  (A look-back scan over published tile sums was measured too: its chain of
// waits made it slower than the two launches it replaced, profiles/r04/exp_by_side_state_and_single_scan.log.)
__global__ __launch_bounds__(kScanBlock) void k_scan_fused(const int32_t *__restrict__ counts, int64_t n, const int32_t *__restrict__ tile_acc,
                                                           int32_t *__restrict__ tile_acc_next, int64_t n_tiles, int64_t *__restrict__ total,
                                                           int64_t *__restrict__ offsets, const int32_t *__restrict__ iperm,
                                                           int64_t *__restrict__ off_slot, unsigned long long *__restrict__ ctl,
                                                           unsigned long long *__restrict__ ctl_next, int32_t first_chunk_next,
                                                           int32_t side_first_next) {
    __shared__ int64_t wsum[kScanBlock / 64];
    __shared__ int64_t bsum[kScanBlock / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    // the tiles in front of this one
    int64_t pre = 0;
    for (int64_t i = threadIdx.x; i < b; i += kScanBlock) pre += tile_acc[i * kTileAccStride];
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if (lane == 0) bsum[wv] = pre;
    const int64_t i0 = (b * kScanBlock + threadIdx.x) * kScanPer;
    int64_t c[kScanPer];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        c[j] = (i0 + j < n) ? counts[i0 + j] : 0;
        s += c[j];
    }
    int64_t incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int64_t wave_off = 0, tile = 0, base = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) {
        if (w < wv) wave_off += wsum[w];
        tile += wsum[w];
        base += bsum[w];
    }
    int64_t run = base + wave_off + incl - s;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        if (i0 + j < n) {
            offsets[i0 + j] = run;
            if (iperm) off_slot[iperm[i0 + j]] = run;
        }
        run += c[j];
    }
    if (threadIdx.x < 3 && tile_acc_next) tile_acc_next[b * kTileAccStride + threadIdx.x] = 0;
    if (b != n_tiles - 1) return;
    if (threadIdx.x == 0) { offsets[n] = base + tile; *total = base + tile; }
    {   // the waves' statistics (k_march left them on their tiles' lines): records by the generic step, cheap records tallied from
        // their lengths -> the control block's words, where the host reads them
        unsigned long long ng = 0, ne = 0;
        for (int64_t i = threadIdx.x; i < n_tiles; i += kScanBlock) { ng += (unsigned)tile_acc[i * kTileAccStride + 1]; ne += (unsigned)tile_acc[i * kTileAccStride + 2]; }
        for (int off = 32; off > 0; off >>= 1) { ng += __shfl_xor(ng, off, 64); ne += __shfl_xor(ne, off, 64); }
        if (lane == 0 && ng) atomicAdd(&ctl[15], ng);
        if (lane == 0 && ne) atomicAdd(&ctl[kCtlExactTally], ne);
    }
    if (ctl_next && threadIdx.x < kCtlWords) ctl_next[threadIdx.x] = ctl_reset_word(threadIdx.x, first_chunk_next, side_first_next);
}

// Records in completion order -> CSR order on demand (ensure_compacted): the CSR offsets in march-slot order, for the record
// kernel's second run.
__global__ __launch_bounds__(256) void k_offsets_to_slots(const int64_t *__restrict__ offsets, const int32_t *__restrict__ iperm, int64_t n,
                                                          int64_t *__restrict__ off_slot) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) off_slot[iperm[u]] = offsets[u];
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


__global__ void k_scale_volumes(double *__restrict__ vol, int32_t n_cells, double n_azim_2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) vol[i] = vol[i] / n_azim_2;  // volumes ./= n_azim_2, src/trackgenerator.jl:386
}


// rt_tracks_create: what the kernels read per march slot — the lines' coefficients, lengths, directions, start points, angles and
// azimuthal indices in march-slot order, and the inverse of the march order — from the uploaded arrays (84 B per track that need
// not cross PCIe).
__global__ void k_slot_arrays(int64_t n, DTracks t, double *__restrict__ As, double *__restrict__ Bs, double *__restrict__ Cs, double *__restrict__ Ls,
                              double *__restrict__ Dx, double *__restrict__ Dy, double *__restrict__ Pxs, double *__restrict__ Pys,
                              double *__restrict__ Phis, int32_t *__restrict__ Azs, int32_t *__restrict__ iperm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t u = t.perm[i];  // slot i holds track perm[i]
    As[i] = t.A[u]; Bs[i] = t.B[u]; Cs[i] = t.C[u]; Ls[i] = t.ell[u]; Dx[i] = t.cs[u]; Dy[i] = t.sn[u];
    Pxs[i] = t.px[u]; Pys[i] = t.py[u]; Phis[i] = t.phi[u]; Azs[i] = t.azim[u];
    iperm[u] = (int32_t)i;
}

}  // namespace rt

// ------------------------------------------------------------------- launchers -------------
namespace rtx {

void launch_slot_arrays(hipStream_t s, int64_t n, const rt::DTracks &d, double *As, double *Bs, double *Cs, double *Ls, double *Dx, double *Dy,
                        double *Pxs, double *Pys, double *Phis, int32_t *Azs, int32_t *iperm) {
    if (n > 0)
        hipLaunchKernelGGL(rt::k_slot_arrays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d, As, Bs, Cs, Ls, Dx, Dy, Pxs, Pys, Phis, Azs, iperm);
}

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

// (k_materialise_lin addresses the result arrays through 32-bit buffer offsets: arrays below 4 GB, i.e. 2^29 records)
bool lin_kernel_serves(const rt_tracks *t, const rt::DOut &out) {
    return t->mesh->mat_kernel != 1 && out.cap < ((int64_t)1 << 29) - 64 && t->cplan.stg.side_cap > 0;
}

// Codes -> records and / or (ℓ, cell) rows (k_materialise) for the plan of the last two-phase call.  tally: the call's first
// pass over the codes — Σℓ and status (k_finish completes them).
int launch_materialise(rt_tracks *t, const rt::DOut &out, hipStream_t s, bool records, bool rows, bool tally, unsigned long long *d_ctl, bool queue) {
    using rt::as_global;
    rt_mesh *m = t->mesh;
    const rt_tracks::CompactPlan &c = t->cplan;
    if (t->n <= 0 || c.n_whole_waves <= 0) return RT_SUCCESS;
    rt::DMat a{};
    a.q_waves = c.march_waves; a.n_waves = (int32_t)std::min<int64_t>(c.n_whole_waves, 0x7fffffff);
    a.etab = m->d.etab; a.corder = as_global(c.corder);
    a.etab_bytes = (int32_t)(uint32_t)std::min<uint64_t>((uint64_t)3 * (uint64_t)m->n_cells * sizeof(rt::EdgeABC), 0xffffffffull);
    a.n_units = 4 * c.n_whole_waves; a.rtol = c.rtol; a.tally = tally ? 1 : 0;
    a.coord_max = std::max(std::max(fabs(m->d.bx0), fabs(m->d.bx1)), std::max(fabs(m->d.by0), fabs(m->d.by1)));
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
    if (records && !rows && lin_kernel_serves(t, out)) {
        launch_materialise_lin(c.d_whole, t->status.p, c.stg, out, a, s, m->n_cus, queue);
        if (tally) t->last_record_kernel = 3;
        return RT_SUCCESS;
    }
    if (queue) { set_error("records in completion order need k_materialise_lin"); return RT_ERR_INVALID; }
    if (tally) t->last_record_kernel = rows && !records ? 4 : 2;
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
                   unsigned long long *d_ctl, unsigned long long *h_res_dev, unsigned long long seq, bool completion_order) {
    const rt_tracks::CompactPlan &c = t->cplan;
    hipLaunchKernelGGL(rt::k_finish, dim3(1), dim3(rt::kFinishThreads), 0, s, c.d_whole, (const int32_t *)t->counts.p, t->status.p,
                       (const int64_t *)t->offsets.p, from_rows ? (const double *)nullptr : (const double *)t->sell.p, out.cap, c.stg,
                       from_rows ? (const double *)t->sw_ell.p : (const double *)nullptr, c.rtol, t->marg.p,
                       scale_volumes ? t->volumes.p : (double *)nullptr, t->vacc.p, t->mesh->n_cells, n_azim_2, d_ctl, h_res_dev, seq,
                       completion_order ? (const int64_t *)t->off_slot.p : (const int64_t *)nullptr);
}

// Staged rows -> compact CSR records for the plan of the last single-pass call: k_compact3 over (q, ±cell) rows, or — codes —
// k_materialise without its tallies.
void launch_compaction(rt_tracks *t, const rt::DOut &out, hipStream_t s) {
    const rt_tracks::CompactPlan &c = t->cplan;
    if (c.codes) { (void)launch_materialise(t, out, s, true, false, false, nullptr); return; }
    t->last_record_kernel = 1;
    if (t->n > 0 && !c.split_all && c.n_whole_waves > 0)
        hipLaunchKernelGGL(rt::k_compact3<false>, dim3(4u * (unsigned)c.n_whole_waves), dim3(256), 0, s, c.d_whole,
                           (const int32_t *)t->counts.p, (const int64_t *)t->offsets.p, c.stg, out, c.sp, c.corder);
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
    if (t->completion_order && t->n > 0) {
        // The last call left its records in completion order (the per-track table: rt_device_table); whoever asks for the CSR
        // layout gets it here, once: the CSR offsets in slot order, and the record kernel again over the staged words.
        hipLaunchKernelGGL(rt::k_offsets_to_slots, dim3((unsigned)((t->n + 255) / 256)), dim3(256), 0, t->mesh->stream, (const int64_t *)t->offsets.p,
                           (const int32_t *)t->iperm.p, t->n, t->off_slot.p);
        t->completion_order = false;
    }
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
    t->sw_rowsc_valid = false;  // (the same buffers)
    return RT_SUCCESS;
}


void launch_prologue(hipStream_t s, unsigned long long *ctl, double *volumes, int32_t n_cells, int32_t first_chunk, int32_t side_first) {
    hipLaunchKernelGGL(rt::k_prologue, dim3((unsigned)((std::max(n_cells, rt::kCtlWords) + 255) / 256)), dim3(256), 0, s, ctl, volumes, n_cells,
                       first_chunk, side_first);
}

void launch_scan_fused(hipStream_t s, rt_tracks *t, int64_t n_tiles, unsigned long long *d_ctl, const int32_t *tile_acc, int32_t *tile_acc_next,
                       unsigned long long *ctl_next, int32_t first_chunk_next, int32_t side_first_next, bool write_slots) {
    hipLaunchKernelGGL(rt::k_scan_fused, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, (const int32_t *)t->counts.p, t->n, tile_acc,
                       tile_acc_next, n_tiles, reinterpret_cast<int64_t *>(d_ctl + 16), t->offsets.p,
                       write_slots ? (const int32_t *)t->iperm.p : (const int32_t *)nullptr, t->off_slot.p,
                       d_ctl, ctl_next, first_chunk_next, side_first_next);
}

void launch_scan(hipStream_t s, rt_tracks *t, int64_t n_tiles, unsigned long long *d_ctl, unsigned long long *host_copy,
                 unsigned long long *ctl_next, int32_t first_chunk_next, int32_t side_first_next, unsigned long long seq,
                 double *scale_volumes, double n_azim_2, bool slot_order) {
    int64_t *const d_total = reinterpret_cast<int64_t *>(d_ctl + 16);
    hipLaunchKernelGGL(rt::k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, t->n, t->tile_sums.p, n_tiles,
                       d_total, reinterpret_cast<unsigned int *>(d_ctl + 20), (const unsigned long long *)d_ctl, host_copy, ctl_next,
                       first_chunk_next, side_first_next, seq);
    hipLaunchKernelGGL(rt::k_scan_write, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, t->n, t->tile_sums.p, d_total,
                       t->offsets.p, scale_volumes, t->mesh->n_cells, n_azim_2, t->vacc.p,
                       slot_order ? (const int32_t *)t->iperm.p : (const int32_t *)nullptr, t->off_slot.p);
}

int launch_volumes_pass(hipStream_t s, rt_tracks *t, const int32_t *overflow, int64_t cap) {
    rt_mesh *m = t->mesh;
    const int64_t n = t->n;
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
    hipLaunchKernelGGL(rt::k_volumes, dim3((unsigned)nb), dim3(1024), shmem, s, (const int64_t *)t->offsets.p, n, (const int32_t *)t->azim.p,
                       (const double *)t->delta_s.p, (const int32_t *)t->element.p, (const double *)t->sell.p, t->volumes.p, m->n_cells, tpb,
                       use_lds, overflow, cap);
    return RT_SUCCESS;
}

void launch_scale_volumes(hipStream_t s, double *volumes, int32_t n_cells, double n_azim_2) {
    hipLaunchKernelGGL(rt::k_scale_volumes, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, s, volumes, n_cells, n_azim_2);
}

void launch_fill_tau(hipStream_t s, rt_tracks *t, int32_t n_groups) {
    const unsigned blocks = (unsigned)((t->total + rt::kTauSegs - 1) / rt::kTauSegs);
    const uint32_t inv = n_groups == 1 ? 0u : (uint32_t)(0x100000000ull / (uint64_t)n_groups) + 1u;  // ≥ 2^32 / G; 0 = one group
    hipLaunchKernelGGL(rt::k_fill_tau, dim3(blocks), dim3(256), 0, s, (const double *)t->sell.p, (const int32_t *)t->element.p,
                       (const double *)t->sigma_t.p, t->total, n_groups, inv, t->tau.p);
}

}  // namespace rtx
