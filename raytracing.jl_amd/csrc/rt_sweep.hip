// rt_sweep.hip — rt_sweep: the transport sweep over the cyclic tracks, kernels, host code and entry points.
#include "rt_internal.hpp"

namespace rt {

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
            double wd[GP], tau[GP];
            bool thin = true;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                tau[g] = st[g] * ell;
                thin = thin && tau[g] < kThinTau;
            }
            // −expm1(−τ) to within an ulp (rt_device.hpp): where every lane's segment is optically thin in every group of the pass —
            // a wave-uniform branch — by the series alone (10 instructions per group instead of 24)
            // (The choice is per WAVE-row: a segment takes the series when the other 63 lanes' segments are thin too, else the general
            //  form — the two agree to 2 ulp, so ψ_out is NOT bitwise invariant across march orders, sort modes or shardings of the
            //  same problem; the tests compare at 1e-12.  "sweep_debug" 4 = the general form everywhere: the reproducible mode.)
            if (__ballot(!thin) == 0 && !(a.debug & 4)) {
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    const double d = (psi[g] - qs[g]) * one_minus_exp_neg_thin(tau[g], poly);
                    psi[g] = psi[g] - d;
                    wd[g] = w * d;
                }
            } else {
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    const double d = (psi[g] - qs[g]) * one_minus_exp_neg(tau[g], poly);
                    psi[g] = psi[g] - d;
                    wd[g] = w * d;
                }
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
// ---- (ℓ, cell) rows from the COMPACT records (round 6) --------------------------------------------------------------
// The sweep's fast input is the coalesced one: rows of a march wave, lane = track (k_sweep<true, ..., ELLROWS>).  A handle whose
// call left no whole-track staging (tracks marched in pieces: C1, C2), or a caller that names the compact CSR records — the
// reference's own layout, `for segment in track.segments: ℓ, element` (README.md:127-135) — used to sweep those records where they
// lie: one lane walks one track's run, a wave-load touches 64 lines (0.60 ms and 1.39 GB per sweep at C3 / 7 groups against 0.25 ms
// over rows).  Now the first sweep after a segmentation transposes the compact (ℓ, cell) into rows ONCE — a chunk table of its own:
// wave w gets ⌈max count / 32⌉ chunks in a row, `k_rows_plan` — and every sweep reads the rows.
__global__ __launch_bounds__(256) void k_rows_count(const int32_t *__restrict__ counts, const int32_t *__restrict__ perm, int64_t n, int32_t n_waves,
                                                    int32_t *__restrict__ nch) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const int64_t slot = w * 64 + (threadIdx.x & 63);
    int32_t mc = slot < n ? counts[perm[slot]] : 0;
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(mc, o, 64);
        mc = v > mc ? v : mc;
    }
    if ((threadIdx.x & 63) == 0) nch[w] = (mc + kChunkRows - 1) >> kChunkLog2;
}
// one workgroup: exclusive scan of the waves' chunk counts -> first[w]; total[0] = chunks in all
__global__ __launch_bounds__(1024) void k_rows_plan(const int32_t *__restrict__ nch, int32_t n_waves, int32_t *__restrict__ first, int32_t *__restrict__ total) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int32_t base = 0; base < n_waves; base += 1024) {
        const int32_t i = base + (int32_t)threadIdx.x;
        const int32_t v = i < n_waves ? nch[i] : 0;
        int32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int32_t woff = 0;
        for (int k = 0; k < wv; ++k) woff += wsum[k];
        if (i < n_waves) first[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}
// one four-wave workgroup per march wave: lane = track, wave k of the workgroup takes the rows r ≡ k (mod 4)
__global__ __launch_bounds__(256) void k_rows_fill(const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets, const int32_t *__restrict__ perm,
                                                   int64_t n, const double *__restrict__ ell, const int32_t *__restrict__ element,
                                                   const int32_t *__restrict__ first, const int32_t *__restrict__ nch, int32_t *__restrict__ ctab,
                                                   double *__restrict__ ell_rows, int32_t *__restrict__ cell_rows) {
    const int64_t w = blockIdx.x;
    const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6;
    const int64_t slot = w * 64 + lane;
    const bool have = slot < n;
    const int32_t u = have ? perm[slot] : 0;
    const int32_t cnt = have ? counts[u] : 0;
    const int64_t off = have ? offsets[u] : 0;
    const int32_t c0 = first[w], nc = nch[w];
    for (int j = threadIdx.x; j < nc; j += 256) ctab[w * kMaxChunks + j] = c0 + j;
    const int maxr = nc << kChunkLog2;
    for (int r = kw; r < maxr; r += 4) {
        if (r < cnt) {
            const int64_t sl = stage_slot(c0 + (r >> kChunkLog2), r & (kChunkRows - 1), lane);
            ell_rows[sl] = ell[off + r];
            cell_rows[sl] = element[off + r];
        }
    }
}

}  // namespace rt

namespace rtx {
// The rows of the last segmentation from its compact records (see k_rows_fill): built once, `sw_rowsc_valid`.
int ensure_rows_from_compact(rt_tracks *t) {
    if (t->sw_rowsc_valid) return RT_SUCCESS;
    if (int rc = ensure_compacted(t)) return rc;
    hipStream_t s = t->mesh->stream;
    const int64_t n = t->n;
    const int32_t n_waves = (int32_t)((n + 63) / 64);
    if (n_waves == 0) { t->sw_rowsc_valid = true; return RT_SUCCESS; }
    RT_HIP(t->sw_plan.reserve(2 * (size_t)n_waves + 8));
    RT_HIP(t->sw_ctab.reserve((size_t)n_waves * rt::kMaxChunks));
    int32_t *nch = t->sw_plan.p, *first = nch + n_waves, *total = first + n_waves;
    hipLaunchKernelGGL(rt::k_rows_count, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, s, (const int32_t *)t->counts.p, (const int32_t *)t->perm.p, n, n_waves, nch);
    hipLaunchKernelGGL(rt::k_rows_plan, dim3(1), dim3(1024), 0, s, (const int32_t *)nch, n_waves, first, total);
    int32_t h_total = 0;
    RT_HIP(hipMemcpyAsync(&h_total, total, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    const size_t slots = (size_t)std::max(1, h_total) * rt::kChunkRows * 64;
    RT_HIP(t->sw_ell.reserve(slots)); RT_HIP(t->sw_cell.reserve(slots));
    t->sw_ell_valid = false;  // (the buffers now hold rows in THIS chunk table's layout, not the staging pool's)
    hipLaunchKernelGGL(rt::k_rows_fill, dim3((unsigned)n_waves), dim3(256), 0, s, (const int32_t *)t->counts.p, (const int64_t *)t->offsets.p,
                       (const int32_t *)t->perm.p, n, (const double *)t->sell.p, (const int32_t *)t->element.p, (const int32_t *)first, (const int32_t *)nch,
                       t->sw_ctab.p, t->sw_ell.p, t->sw_cell.p);
    RT_HIP(hipGetLastError());
    t->sw_rowsc_valid = true;
    return RT_SUCCESS;
}
}  // namespace rtx



using namespace rtx;

extern "C" {

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
    bool staged = input == 2 || (input == 0 && staged_ok);
    // The compact records asked for (or all there is: tracks marched in pieces): swept as ROWS all the same (option "sweep_rows",
    // on by default) — the staging's, while the handle still has them; else rows made once from the compact records
    // (ensure_rows_from_compact).  "sweep_rows" 0: the records where they lie (k_sweep<false, ...>; A/B and tests).
    bool rows_compact = false;
    if (!staged && m->sweep_rows) {
        if (staged_ok && m->sweep_rows != 2) staged = true;  // ("sweep_rows" 2, tests / A/B: always from the compact records)
        else rows_compact = true;
    }
    if (!staged)
        if (int rc = ensure_compacted(t)) return rc;
    if (rows_compact)
        if (int rc = ensure_rows_from_compact(t)) return rc;
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
    if (rows_compact) {
        a.stg = rt::DStage{};
        a.stg.ctab = as_global(t->sw_ctab.p); a.stg.element = as_global(t->sw_cell.p);
        ell_rows = true;
    } else if (staged && t->cplan.codes) {
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
        if (STAGED && ell_rows && (t->sw_ell_valid || rows_compact)) {
            if constexpr (STAGED) {
                if (smem > 48 * 1024)
                    RT_HIP(hipFuncSetAttribute((const void *)rt::k_sweep<true, GP, LDS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                hipLaunchKernelGGL((rt::k_sweep<true, GP, LDS, true>), dim3(blocks), dim3(64 * W), smem, s, a);
            }
        } else {
            if (smem > 48 * 1024)
                RT_HIP(hipFuncSetAttribute((const void *)rt::k_sweep<STAGED, GP, LDS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((rt::k_sweep<STAGED, GP, LDS, false>), dim3(blocks), dim3(64 * W), smem, s, a);
            if (STAGED && ell_rows) { t->sw_ell_valid = true; t->sw_rowsc_valid = false; }  // (the forward waves of this pass have written every row's ℓ)
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
        if (staged || rows_compact) rc = a.use_lds ? launch_all.template operator()<true, true>() : launch_all.template operator()<true, false>();
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
    // (what the caller's records were: 1 the compact CSR records — named, or all there is —, 2 the staging; and how they were read)
    t->sw_last_input = (input == 1 || rows_compact || !staged) ? 1 : 2;
    t->sw_last_rows = rows_compact ? 2 : (staged && ell_rows ? 1 : 0); t->sw_last_gp = a.use_lds ? gp : 0; t->sw_last_passes = passes;
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

int32_t rt_sweep_rows_kind(rt_tracks *t) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->sw_done) { set_error("rt_sweep has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    return t->sw_last_rows;
}

int32_t rt_sweep_xs_pointer(rt_tracks *t, void **xs_dev) {
    if (!t || !xs_dev) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->sw_has_xs) { set_error("rt_sweep has not been given cross sections yet"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    *xs_dev = t->sw_xs.p;
    return RT_SUCCESS;
}

}  // extern "C"
