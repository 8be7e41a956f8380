// host_march.hip — TEST INFRASTRUCTURE.  Runs the device march's per-lane logic on the host.
//
// The geometry of the march (csrc/rt_device.hpp: walk_step and its certificates, find_element,
// intersections) and the mesh preprocessing (csrc/rt_mesh_prep.hpp) are plain FP64 arithmetic that
// compiles for the host as well.  This file drives them with the control flow of one lane of k_march
// (csrc/rt_segmentize.hip; reference: _segmentize_track!, src/track.jl:106-178), so that the walk
// step's certificates can be fuzzed against the CPU checker on thousands of meshes without a GPU:
// walk on == walk off == checker, bit for bit (tests/test_walk_certificates_cpu.py, tools/fuzz_cpu.py).
// It is not part of the product: the library never marches on the host, nothing in raytracing.jl_amd/
// builds or loads this file, and the GPU parity tests go through the C ABI on the real kernels.
//
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (tests/hostmarch.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../raytracing.jl_amd/csrc/rt_device.hpp"
#include "../raytracing.jl_amd/csrc/rt_mesh_prep.hpp"

namespace {

struct Rec { double px, py, qx, qy, ell; int32_t element; int32_t own; };  // own: a generic step's record (it keeps its own p)

struct Result {
    std::vector<int64_t> offsets;
    std::vector<int32_t> status;
    std::vector<Rec> recs;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
Result g_res;

// Why did walk_step refuse?  Re-evaluates its sub-certificates (development statistics only).
// reasons: 0 record off / extras, 1 vertex clearance, 2 entry edge not crossed, 3 isolation margin (c1/c2), 4 entry side (c0),
// 5 short chord, 6 order, 7 tie / other
int refusal_reason(const rt::DMesh &m, const rt::Walk &w, const rt::NextRec &nr, int kk, double phi, double tA, double tB, double tC,
                   double xpx, double xpy, double ppx, double ppy) {
    using namespace rt;
    if (rec_extras(nr.hdr) > kk) return 0;
    const bool same = rec_same(nr.hdr);
    const double x0 = same ? w.ax : w.bx, y0 = same ? w.ay : w.by, x1 = same ? w.bx : w.ax, y1 = same ? w.by : w.ay;
    const double x2 = nr.x2, y2 = nr.y2;
    const double s0 = tA * x0 + tB * y0 + tC, s1 = tA * x1 + tB * y1 + tC, s2 = tA * x2 + tB * y2 + tC;
    if (!(fabs(s0) >= m.d_vertex && fabs(s1) >= m.d_vertex && fabs(s2) >= m.d_vertex)) return 1;
    if ((s0 > 0) == (s1 > 0)) return 2;
    const bool exit1 = (s1 > 0) != (s2 > 0);
    const double area2 = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    const double sg = area2 > 0 ? 1.0 : -1.0, aa = fabs(area2);
    const double c0 = sg * ((x1 - x0) * (xpy - y0) - (y1 - y0) * (xpx - x0));
    const double c1 = sg * ((x2 - x1) * (xpy - y1) - (y2 - y1) * (xpx - x1));
    const double c2 = sg * ((x0 - x2) * (xpy - y2) - (y0 - y2) * (xpx - x2));
    const double eps_aa = rec_eps(nr.hdr) * aa;
    if (!(c1 >= eps_aa && c2 >= eps_aa)) return 3;
    if (!(c0 >= -0.25 * kRtolDefault * aa)) return 4;
    const double eA = exit1 ? nr.e1A : nr.e2A, eB = exit1 ? nr.e1B : nr.e2B, eC = exit1 ? nr.e1C : nr.e2C;
    const double det = tB * eA - eB * tA;
    const double qx = (tC * eB - eC * tB) / det, qy = (tA * eC - eA * tC) / det;
    if (!(norm2(ppx - qx, ppy - qy) >= m.l_min)) return 5;
    if (!(phi < kHalfPi ? ppx < qx : ppx > qx)) return 6;
    return 7;
}
int64_t g_reasons[8];

// counters: [0] walk emits, [1] walk skips (incl. creep passes), [2] generic emits, [3] generic iterations in total,
// [4] generic iterations taken although a prediction existed (a certificate refused)
// `tt0.on`: cheap steps (topo_step) wherever their certificates hold, the record's arithmetic as the march evaluates it
// (edge_exit_point with the cell's edge table, p = previous q, ℓ = ‖p − q‖); cnt[5] cheap emits,
// cnt[6] cheap refusals (mid-track), cnt[7] restarts with exact steps only (iteration bound reached the cap).
void march_track(const rt::DMesh &m, const rt::DGeo &g, double px0, double py0, double phi, double cs, double sn, double tA,
                 double tB, double tC, double track_ell, double tiny, int k, double rtol, int64_t iter_cap,
                 std::vector<Rec> &out, int32_t &status, int64_t *cnt, rt::TopoTrack tt0 = rt::TopoTrack{}) {
    using namespace rt;
    const size_t out_base = out.size();
    TopoTrack tt = tt0;
    TopoState ts{-1, -1, 0.0, 0.0, false};
    bool cheap = false, used_cheap = false;
restart:
    const double sx = tiny * cs, sy = tiny * sn;  // advance_step, src/point.jl:43
    double xpx = px0 + sx, xpy = py0 + sy;        // src/track.jl:114
    int i = 0;
    int64_t it = 0;
    int32_t prev_element = -1;
    int st = 0;
    double sum_ell = 0.0;
    Walk wk;
    wk.T = -1; wk.pred = -1;
    wk.ax = wk.ay = wk.bx = wk.by = wk.cx = wk.cy = 0.0; wk.dT = 1.0;
    const int kk = k > 2 ? (k < kExtrasNever - 1 ? k : kExtrasNever - 1) : 2;
    double lqx = 0.0, lqy = 0.0;
    NextRec nr;
    while (st == 0 && inboundary(m, xpx, xpy, tiny)) {  // start band
        if (++it > iter_cap) { st = 4; break; }
        xpx = xpx + sx; xpy = xpy + sy;
    }
    while (st == 0 && i < kMaxIter) {
        if (tt.on && used_cheap && it > iter_cap) {
            // the iteration count is only an upper bound after cheap steps: march this track again, exactly
            out.resize(out_base);
            tt.on = false; cheap = false; used_cheap = false;
            ++cnt[7];
            goto restart;
        }
        double px, py, qx, qy, ell;
        int32_t element = -1;
        if (cheap) {
            const RT_G TopoRec *R = m.trec + ts.pred;
            int32_t code, kub;
            const int32_t last_before = ts.last;
            const int r = topo_step(tt, ts, R->hdr, R->x2, R->y2, R->c01, R->c23, kk, tA, tB, tC, code, kub);
            if (r != kTopoFull) {
                const RT_G EdgeABC *e = m.etab + code;
                edge_exit_point(tA, tB, tC, e->A, e->B, e->C, qx, qy);
                px = lqx; py = lqy;
                ell = norm2(px - qx, py - qy);
                out.push_back({px, py, qx, qy, ell, code / 3 + 1, 0});
                sum_ell += ell;
                lqx = qx; lqy = qy;
                ++i; it += kub; ++cnt[5];
                used_cheap = true;
                if (r == kTopoEnd) break;
                if (ts.pred >= 0) continue;
            } else {
                ++cnt[6];
                (void)last_before;
            }
            // exact steps from the last emitted record
            cheap = false;
            double mqx, mqy;
            topo_materialize(m, g, ts.last, tA, tB, tC, wk, mqx, mqy);
            lqx = mqx; lqy = mqy;
            xpx = lqx + sx; xpy = lqy + sy;
            prev_element = wk.T;
            if (!(i < kMaxIter)) break;
        }
        if (++it > iter_cap) { if (tt.on && used_cheap) continue; st = 4; break; }
        if (inboundary(m, xpx, xpy, tiny)) {
            if (i == 0) { xpx = xpx + sx; xpy = xpy + sy; continue; }
            break;
        }
        load_next(m, wk.pred, nr);
        int res = walk_step(m, wk, nr, kk, phi, tA, tB, tC, xpx, xpy, lqx, lqy, qx, qy, ell);
        if (res == kWalkSkip) {
            ++cnt[1];
            xpx = xpx + sx; xpy = xpy + sy;
            while (it < iter_cap && !inboundary(m, xpx, xpy, tiny) && walk_still_skip(m, wk, nr, xpx, xpy)) {
                ++it; ++cnt[1];
                xpx = xpx + sx; xpy = xpy + sy;
            }
            continue;
        }
        px = lqx; py = lqy; element = wk.T;
        if (res == kWalkGeneric) {
            ++cnt[3];
            if (m.walk_ok && wk.pred >= 0) {
                ++cnt[4];
                __atomic_fetch_add(&g_reasons[refusal_reason(m, wk, nr, kk, phi, tA, tB, tC, xpx, xpy, lqx, lqy)], 1, __ATOMIC_RELAXED);
            }
            Tri tri;
            element = k > kMaxK ? find_element<true>(g, xpx, xpy, k, tri) : find_element<false>(g, xpx, xpy, k, tri);
            if (element < 0) { st = 1; break; }
            if (element == prev_element) { xpx = xpx + sx; xpy = xpy + sy; continue; }
            int eq;
            if (!intersections(tri, phi, tA, tB, tC, px, py, qx, qy, eq)) { st = 3; break; }
            if (isapprox_v2(px, py, qx, qy)) { xpx = xpx + sx; xpy = xpy + sy; continue; }
            ell = norm2(px - qx, py - qy);
            if (m.walk_ok && eq >= 0) walk_enter(m, tri, wk, element, eq);
            else { wk.T = element; wk.pred = -1; }
            ++cnt[2];
        } else {
            ++cnt[0];
        }
        out.push_back({px, py, qx, qy, ell, element + 1, res == kWalkGeneric ? 1 : 0});
        sum_ell += ell;
        lqx = qx; lqy = qy;
        xpx = qx + sx; xpy = qy + sy;
        prev_element = element;
        ++i;
        if (tt.on && wk.T == element) cheap = topo_enter(m, tt, wk, tA, tB, tC, ts);
    }
    if (st == 0 && !isapprox_s(track_ell, sum_ell, rtol)) st = 2;
    status = st;
}

}  // namespace

extern "C" {

// Preprocess the mesh (rtprep::prepare) and march every track on the host.  Ids as at the C ABI: cell_nodes and
// node_cells_data 1-based, node_cells_ptrs 0-based.  walk = 0: literal step only.  Returns the number of records
// (held until the next call; hostmarch_fetch copies them) or -1.
int64_t hostmarch_run(const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                      const int32_t *ncp, const int32_t *ncd1, const double *bb, int64_t n_tracks, const double *px,
                      const double *py, const double *phi, const double *cs, const double *sn, const double *A,
                      const double *B, const double *C, const double *ell, double tiny, int32_t k, double rtol,
                      int64_t iter_cap, int32_t walk, int32_t n_threads, double *info /* [8] or NULL */) {
    std::vector<int32_t> cn(3 * (size_t)n_cells), ncd((size_t)ncp[n_nodes] > 0 ? ncp[n_nodes] : 1);
    for (size_t i = 0; i < cn.size(); ++i) cn[i] = cell_nodes[i] - 1;
    for (int32_t i = 0; i < ncp[n_nodes]; ++i) ncd[i] = ncd1[i] - 1;
    rtprep::Prep P = rtprep::prepare(x, y, n_nodes, cn.data(), n_cells, bb);
    rt::DGeo g{};
    g.x = rt::as_global(x); g.y = rt::as_global(y); g.cn = rt::as_global((const int32_t *)cn.data());
    g.ncp = rt::as_global(ncp); g.ncd = rt::as_global((const int32_t *)ncd.data());
    g.gstart = rt::as_global((const int32_t *)P.gstart.data()); g.gnode = rt::as_global((const int32_t *)P.gnode.data());
    g.c3start = rt::as_global((const int32_t *)P.c3start.data()); g.c3node = rt::as_global((const int32_t *)P.c3node.data());
    g.c3x = rt::as_global((const double *)P.c3x.data()); g.c3y = rt::as_global((const double *)P.c3y.data());
    std::vector<rt::FanEntry> fan((size_t)std::max(ncp[n_nodes], 1));
    for (int32_t i = 0; i < ncp[n_nodes]; ++i) {
        const int32_t c = ncd[i];
        rt::FanEntry &e = fan[i];
        e.x1 = x[cn[3 * c]]; e.y1 = y[cn[3 * c]]; e.x2 = x[cn[3 * c + 1]]; e.y2 = y[cn[3 * c + 1]]; e.x3 = x[cn[3 * c + 2]]; e.y3 = y[cn[3 * c + 2]];
        e.cell = c;
        for (int q = 0; q < 3; ++q) e.adj[q] = P.adjr[(size_t)3 * c + q];
    }
    g.fan = rt::as_global((const rt::FanEntry *)fan.data());
    g.gx0 = bb[0]; g.gy0 = bb[1]; g.gh = P.gh; g.ginv = P.ginv; g.gnx = P.gnx; g.gny = P.gny; g.n_nodes = n_nodes;
    rt::DMesh m{};
    m.wrec = rt::as_global(reinterpret_cast<const rt::WalkRec *>(P.wrec.data()));
    m.adjr = rt::as_global((const int32_t *)P.adjr.data());
    m.d_vertex = P.d_vertex; m.l_min = P.l_min; m.walk_ok = (P.walk_ok && walk) ? 1 : 0; m.n_cells = n_cells;
    m.bx0 = bb[0]; m.by0 = bb[1]; m.bx1 = bb[2]; m.by1 = bb[3];
    m.geo = nullptr;
    m.trec = rt::as_global(reinterpret_cast<const rt::TopoRec *>(P.trec.data()));
    m.etab = rt::as_global(reinterpret_cast<const rt::EdgeABC *>(P.etab.data()));
    const bool topo = walk == 2 && P.walk_ok && P.topo_ok;
    if (info) {
        info[0] = P.walk_ok ? 1 : 0; info[1] = (double)P.n_records; info[2] = (double)P.n_records_walk; info[3] = P.eps_min;
        info[4] = P.eps_max; info[5] = P.d_vertex; info[6] = (double)P.n_cells_fragile; info[7] = (double)P.n_cells_wild;
    }
    const int nt = n_threads > 0 ? n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::vector<Rec>> part(nt);
    std::vector<int64_t> counts(n_tracks, 0);
    g_res = Result();
    g_res.status.assign(n_tracks, 0);
    std::vector<std::vector<int64_t>> cnt(nt, std::vector<int64_t>(8, 0));
    // contiguous blocks of tracks per thread, so that the concatenation is in uid order
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            const int64_t u0 = n_tracks * t / nt, u1 = n_tracks * (t + 1) / nt;
            for (int64_t u = u0; u < u1; ++u) {
                const size_t before = part[t].size();
                int32_t st = 0;
                march_track(m, g, px[u], py[u], phi[u], cs[u], sn[u], A[u], B[u], C[u], ell[u], tiny, k, rtol, iter_cap,
                            part[t], st, cnt[t].data(),
                            rt::topo_track(topo, P.d_vertex, P.topo_tiny_max, P.topo_rmax, P.topo_end_err, tiny, cs[u], sn[u]));
                counts[u] = (int64_t)(part[t].size() - before);
                g_res.status[u] = st;
            }
        });
    for (auto &t : th) t.join();
    g_res.offsets.assign(n_tracks + 1, 0);
    for (int64_t u = 0; u < n_tracks; ++u) g_res.offsets[u + 1] = g_res.offsets[u] + counts[u];
    for (int t = 0; t < nt; ++t) {
        g_res.recs.insert(g_res.recs.end(), part[t].begin(), part[t].end());
        for (int j = 0; j < 8; ++j) g_res.stats[j] += cnt[t][j];
    }
    return (int64_t)g_res.recs.size();
}

void hostmarch_reasons(int64_t *out, int32_t reset) {
    for (int i = 0; i < 8; ++i) { out[i] = g_reasons[i]; if (reset) g_reasons[i] = 0; }
}

void hostmarch_fetch(int64_t *offsets, int32_t *status, double *px, double *py, double *qx, double *qy, double *ell,
                     int32_t *element, int64_t *stats) {
    memcpy(offsets, g_res.offsets.data(), sizeof(int64_t) * g_res.offsets.size());
    memcpy(status, g_res.status.data(), sizeof(int32_t) * g_res.status.size());
    for (size_t i = 0; i < g_res.recs.size(); ++i) {
        const Rec &r = g_res.recs[i];
        px[i] = r.px; py[i] = r.py; qx[i] = r.qx; qy[i] = r.qy; ell[i] = r.ell; element[i] = r.element;
    }
    memcpy(stats, g_res.stats, sizeof(g_res.stats));
}

// The Σℓ check of k_materialise_lin on the records of the last hostmarch_run: per track, the chain (rt_device.hpp chain_gap_term /
// chain_sum / chain_status — the very functions the kernel calls) over its records in march order, the gaps added left to right.
// status_chain[u]: 0 OK, 1 LENGTH_MISMATCH, 2 inside the band (k_finish sums left to right); S_chain[u]: the chain's Σℓ;
// S_exact[u]: the left-to-right sum of the records' lengths (what src/track.jl:171 compares, and what k_finish forms).
void hostmarch_chain(int64_t n_tracks, const double *cs, const double *sn, const double *ell, double rtol, double coord_max,
                     int32_t *status_chain, double *S_chain, double *S_exact) {
    for (int64_t u = 0; u < n_tracks; ++u) {
        const int64_t o = g_res.offsets[u], cnt = g_res.offsets[u + 1] - o;
        double gap = 0.0, sum = 0.0;
        for (int64_t i = 0; i < cnt; ++i) {
            const Rec &r = g_res.recs[(size_t)(o + i)];
            sum += r.ell;
            if (i > 0 && r.own) {
                const Rec &b = g_res.recs[(size_t)(o + i - 1)];
                gap += rt::chain_gap_term(r.px, r.py, r.qx, r.qy, r.ell, b.qx, b.qy, cs[u], sn[u]);
            }
        }
        double S = 0.0;
        if (cnt > 0) {
            const Rec &f = g_res.recs[(size_t)o], &l = g_res.recs[(size_t)(o + cnt - 1)];
            S = rt::chain_sum(f.px, f.py, f.qx, f.qy, l.qx, l.qy, cs[u], sn[u], gap, (int)cnt);
        }
        S_chain[u] = S; S_exact[u] = sum;
        status_chain[u] = rt::chain_status(ell[u], S, rtol, (int)cnt, coord_max);
    }
}

// Host-only view of the preprocessing (no march): the per-record certificate fields, for tests.
// extras[3*n_cells], epscode[3*n_cells] (-1 where the walk step is off), cls[n_cells].
int32_t hostmarch_prep(const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                       const double *bb, int32_t *extras, int32_t *epscode, int32_t *cls, double *info /* [8] */,
                       char *note, int32_t note_cap) {
    std::vector<int32_t> cn(3 * (size_t)n_cells);
    for (size_t i = 0; i < cn.size(); ++i) cn[i] = cell_nodes[i] - 1;
    rtprep::Prep P = rtprep::prepare(x, y, n_nodes, cn.data(), n_cells, bb);
    for (size_t r = 0; r < P.wrec.size(); ++r) {
        const uint64_t h = P.wrec[r].hdr;
        const int ex = (int)(h >> 54) & 15, code = (int)(h >> 58) & 31;
        extras[r] = ex;
        epscode[r] = ex >= rtprep::kExtrasNever ? -1 : code;
    }
    for (int32_t c = 0; c < n_cells; ++c) cls[c] = P.rec[c].cls;
    info[0] = P.walk_ok ? 1 : 0; info[1] = (double)P.n_records; info[2] = (double)P.n_records_walk; info[3] = P.eps_min;
    info[4] = P.eps_max; info[5] = P.d_vertex; info[6] = (double)P.n_cells_fragile; info[7] = (double)P.n_cells_wild;
    if (note && note_cap > 0) { strncpy(note, P.note.c_str(), (size_t)note_cap - 1); note[note_cap - 1] = 0; }
    return 0;
}

// rt_sweep's attenuation factor 1 - exp(-tau) (rt_device.hpp, one_minus_exp_neg), for its accuracy test.
void hostmarch_one_minus_exp_neg(const double *tau, int64_t n, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = rt::one_minus_exp_neg(tau[i]);
}
// ... and its form for optically thin segments (tau < rt::kThinTau)
void hostmarch_one_minus_exp_neg_thin(const double *tau, int64_t n, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = rt::one_minus_exp_neg_thin(tau[i]);
}

// bf16_up / bf16_value of the preprocessing (the cheap step's four per-record constants are stored as bfloat16 rounded UP).
void hostmarch_bf16(const double *v, int64_t n, uint16_t *pattern, double *value) {
    for (int64_t i = 0; i < n; ++i) { pattern[i] = rtprep::bf16_up(v[i]); value[i] = rtprep::bf16_value(pattern[i]); }
}

// The cheap-step records of a mesh as the device decodes them (rt_device.hpp: rec_extras, rec_eps, bf16_lo / bf16_hi):
// per record extras (15: no cheap step), E = 2^(code - 20), g1, k2, dtf, lc; scalars[6] = tiny_max, rmax, end_err, l_min,
// d_vertex, walk_ok && topo_ok.
int32_t hostmarch_topo(const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                       const double *bb, int32_t *extras, double *E, double *g1, double *k2, double *dtf, double *lc,
                       double *scalars) {
    std::vector<int32_t> cn(3 * (size_t)n_cells);
    for (size_t i = 0; i < cn.size(); ++i) cn[i] = cell_nodes[i] - 1;
    rtprep::Prep P = rtprep::prepare(x, y, n_nodes, cn.data(), n_cells, bb);
    for (size_t r = 0; r < (size_t)3 * n_cells; ++r) {
        const rt::TopoRec &T = reinterpret_cast<const rt::TopoRec *>(P.trec.data())[r];
        extras[r] = rt::rec_extras(T.hdr);
        E[r] = rt::rec_eps(T.hdr);
        g1[r] = rt::bf16_lo(T.c01); k2[r] = rt::bf16_hi(T.c01); dtf[r] = rt::bf16_lo(T.c23); lc[r] = rt::bf16_hi(T.c23);
    }
    scalars[0] = P.topo_tiny_max; scalars[1] = P.topo_rmax; scalars[2] = P.topo_end_err; scalars[3] = P.l_min;
    scalars[4] = P.d_vertex; scalars[5] = (P.walk_ok && P.topo_ok) ? 1.0 : 0.0;
    return 0;
}

}  // extern "C"
