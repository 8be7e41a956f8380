// rt_mesh_prep.hpp — one-time host preprocessing of the flattened mesh (C++17, no HIP).
//
// Builds what the device march needs beyond the reference's own tables:
//  * the uniform node grid used by the exact (k-)nearest-node search (replaces the kd-tree,
//    src/mesh.jl:38-42);
//  * rotated walk records per (cell, entry edge) for the walk step of the march (csrc/rt_device.hpp,
//    `WalkRec`): successor records across the two possible exit edges, the vertex opposite the
//    entry edge, and the exit edges' normalised general forms computed exactly as `general_form`
//    does (src/intersection.jl:11-18) so that the device reproduces the reference's intersection
//    points bit for bit;
//  * the walk step's certificates, PER RECORD (a sliver somewhere in the mesh only costs the walk
//    step in its own neighbourhood):
//      - `extras`: how many non-vertex nodes can be nearer to a point of the cell than the cell's
//        nearest vertex (bounds the rank at which `find_element`'s node scan, src/mesh.jl:107-132,
//        reaches the cell);
//      - `eps`: the barycentric isolation margin — with xp at least eps (barycentric) inside the
//        record's cell T' from its two exit edges (and not more than tol/4 outside its entry edge),
//        no cell other than T' and the predecessor T can pass the reference's √eps barycentric test
//        (src/mesh.jl:166-174) at xp.  Computed by clipping every nearby cell's acceptance region
//        (the cell scaled by 1 + 3·tol' about its centroid, tol' = √eps + the rounding noise of the
//        reference's own evaluation in that cell) against the record's region;
//      - cells whose own barycentric test is too noisy at the √eps level (tiny area far from the
//        origin) or degenerate are never walked into; degenerate cells also switch off the records
//        around them.
// This TU is compiled with -ffp-contract=off, like everything else in the library.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <exception>
#include <thread>
#include <unordered_map>
#include <vector>

namespace rtprep {

struct CellRecHost {
    int32_t adj[3];   // neighbour cell across edge k = (v_k, v_{k+1 mod 3}); -1 on the boundary / non-manifold edge
    int32_t extras;   // extras bound, capped at kExtrasNever
    double vx[3], vy[3];
    double eA[3], eB[3], eC[3];
    int32_t cls;      // 0 normal, 1 fragile (not walked into; acceptance region inflated), 2 wild (degenerate)
    double fp_err;    // bound on |λ computed by the reference - λ exact| for points near the cell
    double lmax, area2;
};

// Rotated walk record for (cell, entry edge e): vertices rotated cyclically so that rotated
// edge 0 = (v0, v1) is the entry edge, in the cell's own edge orientation.  Must match
// rt::WalkRec (rt_device.hpp): 80 bytes = five 16-B loads per lane (the fetch costs the march
// ~50-250 cycles per load instruction, tools/micro/bench_gather.hip).  v0 and v1 are not stored:
// they are the endpoints of the predecessor's exit edge, which the walk state already holds
// (same nodes, hence the same bits), in the same or the opposite order (`same` flag).
constexpr int kWalkIdBits = 27;      // record ids + 1 must fit: 3 * n_cells + 1 < 2^27
constexpr int kExtrasNever = 15;     // extras field value that no `k` satisfies: walk step off for the record
constexpr int kEpsCodeMin = 0;       // eps = 2^(code - 20): 2^-20 ≈ 9.5e-7 ... 2^-2
constexpr int kEpsCodeMax = 18;      // larger margins: record disabled (μ0, μ1 ≥ 1/2 cannot both hold inside)
struct WalkRecHost {
    uint64_t hdr;   // bits 0..26 next1 + 1, 27..53 next2 + 1 (record 3*cell' + entry' across rotated edge 1 / 2,
                    // 0 on the boundary), 54..57 extras bound (15: no walk), 58..62 eps code, 63: v0 is the `a` of the
                    // predecessor's exit edge (a, b) — else v0 = b
    double dT;      // det of the barycentric system in the ORIGINAL node order (reference operation order)
    double x2, y2;  // the vertex opposite the entry edge
    double e1A, e1B, e1C, e2A, e2B, e2C;  // general_form of rotated edges 1 = (v1,v2) and 2 = (v2,v0)
};
static_assert(sizeof(WalkRecHost) == 80, "WalkRec layout");

// Topological walk record for (cell T', entry edge e) — the "cheap step" of the march (rt_device.hpp,
// `topo_step`): what a lane needs to decide, from the signed distances of the cell's vertices to the track line alone,
// that the reference emits its next segment in T' with entry on the shared edge and exit on one other edge — without
// computing any point.  32 B = two 16-B loads.  Must match rt::TopoRec.
//   hdr: as WalkRecHost::hdr, except that a successor field may hold kTopoEndV / kTopoEndH (the exit edge lies on a
//        vertical / horizontal border of the bounding box) and that extras / eps code are the cheap step's own;
//   x2, y2: the vertex opposite the entry edge;
//   g1, k2, dtf, lc: certificate constants, bfloat16 bit patterns rounded UP (see `prepare`).
constexpr uint32_t kTopoEndV = (1u << kWalkIdBits) - 1;
constexpr uint32_t kTopoEndH = (1u << kWalkIdBits) - 2;
constexpr int kTopoKcap = 4096;  // tiny steps the reference may take between two emitted segments under a cheap step
struct TopoRecHost {
    uint64_t hdr;
    double x2, y2;
    uint16_t g1, k2, dtf, lc;
};
static_assert(sizeof(TopoRecHost) == 32, "TopoRec layout");
struct EdgeABCHost { double A, B, C, pad; };  // general_form of edge k of cell c at [3*c + k], reference operations
static_assert(sizeof(EdgeABCHost) == 32, "EdgeABC layout");

// smallest bfloat16 (as its 16-bit pattern) that is >= v, for v > 0; 0x7f80 (inf) when out of range
inline uint16_t bf16_up(double v) {
    if (!(v > 0)) return 0;
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    uint32_t b;
    std::memcpy(&b, &f, 4);
    uint32_t hi = b >> 16;
    if ((b & 0xffffu) != 0) ++hi;
    if (hi >= 0x7f80u) return 0x7f80u;
    return (uint16_t)hi;
}
inline double bf16_value(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &b, 4);
    return (double)f;
}

struct Prep {
    std::vector<WalkRecHost> wrec;  // [3*n_cells]
    std::vector<int32_t> adjr;      // [3*n_cells] record index reached across edge k of cell c, -1 on the boundary
    std::vector<TopoRecHost> trec;  // [3*n_cells] cheap-step records
    std::vector<EdgeABCHost> etab;  // [3*n_cells] general forms of the cells' edges
    bool topo_ok = false;           // some record can take the cheap step
    double topo_tiny_max = 0.0;     // the cheap step's certificates hold for tiny_step <= this
    double topo_rmax = 0.0;         // order guard: a track needs |s_v| >= lc * max(1, topo_rmax / |cos ϕ|)
    double topo_end_err = 0.0;      // |computed exit coordinate - border| bound on border edges
    double tally_a = INFINITY, tally_b = INFINITY;  // fill_volumes of cheap records from the vertices' distances (k_march): the chord is
                                    // within a + b / (end points' distance across the track line) of the record's length (see prepare)
    int64_t n_records_topo = 0;
    // node grid
    int gnx = 1, gny = 1;
    double gh = 1.0, ginv = 1.0;
    std::vector<int32_t> gstart, gnode;
    // per bucket, the nodes of its 3x3 block of buckets with their coordinates, contiguous: the first two rings of the
    // nearest-node search as ONE range whose ids and coordinates load in parallel (the ring search walks ~13 ranges with
    // three dependent loads each: most of the ≈10 µs a literal step costs)
    std::vector<int32_t> c3start, c3node;
    std::vector<double> c3x, c3y;
    // records
    std::vector<CellRecHost> rec;
    // certificate margins that stay global
    double d_vertex = 1e-7;  // absolute clearance of the track line from a cell's vertices
    double l_min = 1e-6;     // minimum chord length handled by the walk step
    bool walk_ok = true;     // false: no record can be walked (see note)
    double kappa = 0.0;      // expected segments per unit track length: Σ cell perimeters / (π · area) (Cauchy–Crofton)
    std::string note;
    // diagnostics (rt_mesh_info)
    int64_t n_records = 0, n_records_walk = 0;
    int32_t n_cells_fragile = 0, n_cells_wild = 0, n_edges_nonmanifold = 0, extras_max = 0;
    double eps_min = 0.0, eps_max = 0.0;  // over the records the walk step can use
};

struct P2 { double x, y; };

// The two expensive loops of `prepare` (polygon clipping per cell) are independent per cell: contiguous cell ranges on a few
// host threads.  Results do not depend on the number of threads.  RT_PREP_THREADS overrides (1: serial).
template <typename F>
inline void parallel_cells(int32_t n_cells, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *env = std::getenv("RT_PREP_THREADS")) nt = (unsigned)std::max(1, atoi(env));
    nt = std::max(1u, std::min({nt, 16u, (unsigned)(n_cells / 256 + 1)}));
    if (nt == 1) { f(0, n_cells); return; }
    std::vector<std::thread> th;
    th.reserve(nt);
    // what a worker (or the caller's own share) throws — bad_alloc from its scratch vectors — is carried to the caller's
    // thread and rethrown after every thread has been joined: the C ABI's guards turn it into rt_last_error
    std::vector<std::exception_ptr> err(nt);
    int32_t done = 0;
    try {
        for (unsigned t = 0; t + 1 < nt; ++t) {
            const int32_t c0 = (int32_t)((int64_t)n_cells * t / nt), c1 = (int32_t)((int64_t)n_cells * (t + 1) / nt);
            std::exception_ptr *slot = &err[t];
            th.emplace_back([=, &f]() {
                try { f(c0, c1); } catch (...) { *slot = std::current_exception(); }
            });
            done = c1;
        }
    } catch (...) {  // no thread to be had: the caller's thread does the rest
    }
    try { f(done, n_cells); } catch (...) { err[nt - 1] = std::current_exception(); }
    for (auto &t : th) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

// A convex polygon of a few vertices (a triangle clipped by at most three half-planes), without heap traffic: the
// preprocessing clips ~10^5 – 10^6 of them.
struct Poly {
    P2 v[12];
    int n = 0;
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    void push_back(const P2 &p) { if (n < 12) v[n++] = p; }
    void assign(const P2 *a, const P2 *b) { n = 0; for (; a != b; ++a) push_back(*a); }
    const P2 &operator[](size_t i) const { return v[i]; }
    const P2 *begin() const { return v; }
    const P2 *end() const { return v + n; }
    void swap(Poly &o) { Poly t = o; o = *this; *this = t; }
};

// Clip polygon by half-plane  n·p <= c  (Sutherland–Hodgman).
inline void clip(Poly &poly, double nx, double ny, double c) {
    Poly out;
    const size_t n = poly.size();
    for (size_t i = 0; i < n; ++i) {
        const P2 a = poly[i], b = poly[(i + 1) % n];
        const double da = nx * a.x + ny * a.y - c, db = nx * b.x + ny * b.y - c;
        if (da <= 0) out.push_back(a);
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            const double t = da / (da - db);
            out.push_back({a.x + t * (b.x - a.x), a.y + t * (b.y - a.y)});
        }
    }
    poly.swap(out);
}

constexpr double kTol = 1.4901161193847656e-8;  // sqrt(eps(Float64)): the reference's barycentric tolerance
constexpr double kUlp = 1.1102230246251565e-16;

inline Prep prepare(const double *x, const double *y, int32_t n_nodes, const int32_t *cn /*0-based*/,
                    int32_t n_cells, const double *bb) {
    Prep P;
    const double W = bb[2] - bb[0], H = bb[3] - bb[1];
    // ---- uniform node grid, about one node per bucket
    double gh = std::sqrt(W * H / std::max(1, n_nodes));
    int gnx = std::min(2048, std::max(1, (int)std::ceil(W / gh)));
    int gny = std::min(2048, std::max(1, (int)std::ceil(H / gh)));
    gh = std::max(W / gnx, H / gny);
    P.gnx = gnx; P.gny = gny; P.gh = gh; P.ginv = 1.0 / gh;
    P.gstart.assign((size_t)gnx * gny + 1, 0);
    P.gnode.assign(std::max(1, n_nodes), 0);
    std::vector<int32_t> bucket(n_nodes);
    auto bucket_of = [&](double px, double py, int &ix, int &iy) {
        double fx = std::floor((px - bb[0]) * P.ginv), fy = std::floor((py - bb[1]) * P.ginv);
        ix = !(fx > 0) ? 0 : (fx > gnx - 1 ? gnx - 1 : (int)fx);
        iy = !(fy > 0) ? 0 : (fy > gny - 1 ? gny - 1 : (int)fy);
    };
    for (int32_t i = 0; i < n_nodes; ++i) {
        int ix, iy;
        bucket_of(x[i], y[i], ix, iy);
        bucket[i] = iy * gnx + ix;
        P.gstart[bucket[i] + 1]++;
    }
    for (size_t b = 0; b < (size_t)gnx * gny; ++b) P.gstart[b + 1] += P.gstart[b];
    {
        std::vector<int32_t> cur(P.gstart.begin(), P.gstart.end() - 1);
        for (int32_t i = 0; i < n_nodes; ++i) P.gnode[cur[bucket[i]]++] = i;
    }

    {
        P.c3start.assign((size_t)gnx * gny + 1, 0);
        for (int by = 0; by < gny; ++by)
            for (int bx = 0; bx < gnx; ++bx) {
                int32_t cnt = 0;
                for (int yy = std::max(0, by - 1); yy <= std::min(gny - 1, by + 1); ++yy)
                    cnt += P.gstart[(size_t)yy * gnx + std::min(gnx - 1, bx + 1) + 1] - P.gstart[(size_t)yy * gnx + std::max(0, bx - 1)];
                P.c3start[(size_t)by * gnx + bx + 1] = cnt;
            }
        for (size_t b = 0; b < (size_t)gnx * gny; ++b) P.c3start[b + 1] += P.c3start[b];
        const size_t tot = (size_t)P.c3start[(size_t)gnx * gny];
        P.c3node.resize(std::max<size_t>(1, tot)); P.c3x.resize(std::max<size_t>(1, tot)); P.c3y.resize(std::max<size_t>(1, tot));
        for (int by = 0; by < gny; ++by)
            for (int bx = 0; bx < gnx; ++bx) {
                size_t o = (size_t)P.c3start[(size_t)by * gnx + bx];
                for (int yy = std::max(0, by - 1); yy <= std::min(gny - 1, by + 1); ++yy)
                    for (int32_t q = P.gstart[(size_t)yy * gnx + std::max(0, bx - 1)]; q < P.gstart[(size_t)yy * gnx + std::min(gnx - 1, bx + 1) + 1]; ++q) {
                        P.c3node[o] = P.gnode[q]; P.c3x[o] = x[P.gnode[q]]; P.c3y[o] = y[P.gnode[q]]; ++o;
                    }
            }
    }

    // ---- adjacency through an edge map; an edge shared by more than two cells has no neighbour on any side
    //      (the walk step never crosses it)
    P.rec.assign(n_cells, CellRecHost{});
    std::unordered_map<uint64_t, int64_t> edge_owner;  // key -> cell*3 + k of the first owner; -1: paired; -2: non-manifold
    edge_owner.reserve((size_t)n_cells * 2);
    auto key = [](int32_t a, int32_t b) { return ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b); };
    for (int32_t c = 0; c < n_cells; ++c)
        for (int k = 0; k < 3; ++k) P.rec[c].adj[k] = -1;
    std::vector<uint64_t> bad_edges;
    for (int32_t c = 0; c < n_cells; ++c) {
        for (int k = 0; k < 3; ++k) {
            const int32_t a = cn[3 * c + k], b = cn[3 * c + (k + 1) % 3];
            const uint64_t kk = key(a, b);
            auto it = edge_owner.find(kk);
            if (it == edge_owner.end()) edge_owner.emplace(kk, (int64_t)c * 3 + k);
            else if (it->second == -1) { it->second = -2; bad_edges.push_back(kk); }
            else if (it->second >= 0) {
                const int32_t c2 = (int32_t)(it->second / 3), k2 = (int32_t)(it->second % 3);
                P.rec[c].adj[k] = c2;
                P.rec[c2].adj[k2] = c;
                it->second = -1;
            }
        }
    }
    if (!bad_edges.empty()) {
        P.n_edges_nonmanifold = (int32_t)bad_edges.size();
        for (int32_t c = 0; c < n_cells; ++c)
            for (int k = 0; k < 3; ++k) {
                auto it = edge_owner.find(key(cn[3 * c + k], cn[3 * c + (k + 1) % 3]));
                if (it != edge_owner.end() && it->second == -2) {
                    const int32_t nb = P.rec[c].adj[k];
                    if (nb >= 0)
                        for (int q = 0; q < 3; ++q)
                            if (P.rec[nb].adj[q] == c) P.rec[nb].adj[q] = -1;
                    P.rec[c].adj[k] = -1;
                }
            }
        P.note = "edge shared by more than two cells (not crossed by the walk step)";
    }

    // ---- per-cell geometry, shape statistics, noise class
    const double cmax_x = std::max(std::fabs(bb[0]), std::fabs(bb[2])), cmax_y = std::max(std::fabs(bb[1]), std::fabs(bb[3]));
    double l_max = 0, perim = 0, area = 0;
    std::vector<int32_t> wild;
    for (int32_t c = 0; c < n_cells; ++c) {
        CellRecHost &R = P.rec[c];
        for (int k = 0; k < 3; ++k) { R.vx[k] = x[cn[3 * c + k]]; R.vy[k] = y[cn[3 * c + k]]; }
        double lmax_c = 0;
        for (int k = 0; k < 3; ++k) {
            const int j = (k + 1) % 3;
            // general_form(p1, p2), src/intersection.jl:11-18 — same operations, same order
            const double A = R.vy[k] - R.vy[j];
            const double B = R.vx[j] - R.vx[k];
            const double C = R.vx[k] * R.vy[j] - R.vx[j] * R.vy[k];
            const double nrm = std::sqrt(A * A + B * B + C * C);
            R.eA[k] = A / nrm; R.eB[k] = B / nrm; R.eC[k] = C / nrm;
            const double len = std::hypot(R.vx[k] - R.vx[j], R.vy[k] - R.vy[j]);
            lmax_c = std::max(lmax_c, len);
            perim += len;
        }
        l_max = std::max(l_max, lmax_c);
        const double area2 = std::fabs((R.vx[1] - R.vx[0]) * (R.vy[2] - R.vy[0]) - (R.vx[2] - R.vx[0]) * (R.vy[1] - R.vy[0]));
        area += 0.5 * area2;
        R.lmax = lmax_c; R.area2 = area2;
        // Rounding noise of the reference's λ = ((y2-y3)·x + (x3-x2)·y + (x2·y3 - x3·y2)) / d near this cell: the
        // products x_i·y_j carry an absolute error of u·|x||y| each — the formula is not translation invariant — and
        // d carries the same; |λ| ≤ 2 in the region of interest.
        const double ax = std::max({std::fabs(R.vx[0]), std::fabs(R.vx[1]), std::fabs(R.vx[2])}) + lmax_c;
        const double ay = std::max({std::fabs(R.vy[0]), std::fabs(R.vy[1]), std::fabs(R.vy[2])}) + lmax_c;
        const double Lx = std::max({R.vx[0], R.vx[1], R.vx[2]}) - std::min({R.vx[0], R.vx[1], R.vx[2]});
        const double Ly = std::max({R.vy[0], R.vy[1], R.vy[2]}) - std::min({R.vy[0], R.vy[1], R.vy[2]});
        const double err_abs = kUlp * (3.0 * ax * ay + 6.0 * (Ly * ax + Lx * ay));
        R.fp_err = area2 > 0 ? 3.0 * err_abs / area2 : INFINITY;
        R.cls = 0;
        if (!(R.fp_err <= 0.5 * kTol) || !(area2 >= 1e-5 * lmax_c * lmax_c)) R.cls = 1;
        if (!(R.fp_err <= 0.05) || !std::isfinite(area2) || !(area2 > 0)) R.cls = 2;
        if (R.cls == 1) ++P.n_cells_fragile;
        if (R.cls == 2) { ++P.n_cells_wild; wild.push_back(c); }
    }
    P.kappa = area > 0 ? perim / (3.141592653589793 * area) : 0.0;
    {
        // fill_volumes (src/trackgenerator.jl:376-386) sums δs·ℓ per cell and is compared at 1e-10, not bit for bit.  For a cheap
        // record the march adds the chord between the two edge crossings, each interpolated from the signed distances s and the
        // positions t along the line of the edge's end points: t = (s_p·t_q − s_q·t_p)/(s_p − s_q).  With R the largest
        // coordinate norm and D_x = |s_p − s_q| (the end points' distance across the line):
        //  * the inputs carry |δs| <= 3uR, |δt| <= 2uR, the expression 3uR, and ∂t/∂s <= l_max/D_x: 2uR·(7 + 6·l_max/D_x) for the
        //    chord's two ends;
        //  * the REFERENCE's own point is the track line intersected with the edge's general_form (src/intersection.jl:11-18), whose
        //    C = x_i·y_o − x_o·y_i carries u·R² of rounding against ‖(A, B)‖ = |edge|: the line it represents lies up to
        //    uR²/|edge| + uR beside the edge through the vertices, i.e. (uR² + uR·l_max)/D_x along the track (1/sin of the crossing
        //    angle = |edge|/D_x), and Cramer's rule adds ≈2uR·|edge|/D_x — two ends: 2uR·(R + 3·l_max)/D_x.  (Without this term the
        //    first version's bound was exceeded far from the origin: 5.1e-11 on a lattice 40 units out, profiles/r04/.)
        // A chord is therefore within a + b/D_x of the record's ‖p − q‖, a = 16uR, b = 2uR·(10·l_max + 1.25·R), D_x the smaller of
        // its two crossings'.  The march uses the chord only where that is a fraction ε of THE CHORD ITSELF (rt_segmentize: ε =
        // 8e-11 of north_star's 1e-10, an eighth for a, the rest for b); every other cheap record is marked and k_materialise adds
        // δs·ℓ from the record's own length.  A bound per term is a bound for every cell's sum, however few or short the chords a
        // cell happens to get.  Far from the origin everything is tallied from the lengths — slower, never wrong.
        const double Rfar = std::hypot(cmax_x, cmax_y);
        P.tally_a = 16.0 * kUlp * Rfar;
        P.tally_b = 2.0 * kUlp * Rfar * (10.0 * l_max + 1.25 * Rfar);
    }
    if (wild.size() > 4096) { P.walk_ok = false; P.note = "too many degenerate cells for the walk certificates"; }
    {
        // The track line must clear every vertex of the cell by d_vertex.  What depends on it:
        //  * the edge the line misses is rejected by point_in_segment (src/segment.jl:39-44), which accepts hits up to
        //    0.75e-8·|edge| beyond an endpoint — the miss is at least d_vertex beyond it, whatever the angle;
        //  * the two crossed edges are not "parallel" to the track (isapprox(a, b), src/intersection.jl:128-131, fires
        //    below sin ≈ 1.5e-8): endpoints on opposite sides at ≥ d_vertex give sin ≥ 2·d_vertex/|edge|;
        //  * rounding of the computed intersection points (≈1.5e-15·|far corner| / sin) is 1e7 times below the distance of
        //    a crossing from the edge's endpoints (≥ d_vertex / sin) — added here in absolute terms as well.
        // 2e-7·l_max keeps a factor ≥ 13 over both tolerances (the walk step's refusals cost a full literal step each:
        // 1e-6·l_max refused 5x as often and cost the march 4 % at the headline configuration).
        const double corner = std::hypot(cmax_x, cmax_y);
        P.d_vertex = std::max(2e-7 * l_max, 1e-9 * corner);
        // chords shorter than this go to the generic step (the reference's isapprox(p, q) skip,
        // src/track.jl:156, triggers below ~1.5e-8 * |p|)
        P.l_min = std::max(1e-6 * l_max, 8.0 * kTol * corner);
    }

    // ---- extras bound: non-vertex nodes m that beat all three vertices somewhere in the
    //      (slightly inflated) cell:  |p-m|^2 < |p-v_i|^2  <=>  2 p·(v_i - m) < |v_i|^2 - |m|^2
    if (P.walk_ok) parallel_cells(n_cells, [&](int32_t c_begin, int32_t c_end) {
    Poly poly;
    for (int32_t c = c_begin; c < c_end; ++c) {
        CellRecHost &R = P.rec[c];
        if (R.cls != 0) { R.extras = kExtrasNever; continue; }
        const double cx = (R.vx[0] + R.vx[1] + R.vx[2]) / 3, cy = (R.vy[0] + R.vy[1] + R.vy[2]) / 3;
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        P2 tri[3];
        for (int k = 0; k < 3; ++k) {
            tri[k] = {cx + (R.vx[k] - cx) * 1.001, cy + (R.vy[k] - cy) * 1.001};
            xmin = std::min(xmin, tri[k].x); xmax = std::max(xmax, tri[k].x);
            ymin = std::min(ymin, tri[k].y); ymax = std::max(ymax, tri[k].y);
        }
        int ix0, iy0, ix1, iy1;
        bucket_of(xmin - R.lmax, ymin - R.lmax, ix0, iy0);
        bucket_of(xmax + R.lmax, ymax + R.lmax, ix1, iy1);
        int extras = 0;
        for (int by = iy0; by <= iy1 && extras < kExtrasNever; ++by)
            for (int32_t s = P.gstart[by * gnx + ix0]; s < P.gstart[by * gnx + ix1 + 1] && extras < kExtrasNever; ++s) {
                const int32_t m = P.gnode[s];
                if (m == cn[3 * c] || m == cn[3 * c + 1] || m == cn[3 * c + 2]) continue;
                poly.assign(tri, tri + 3);
                const double mm = x[m] * x[m] + y[m] * y[m];
                for (int k = 0; k < 3 && !poly.empty(); ++k) {
                    const double nx = 2 * (R.vx[k] - x[m]), ny = 2 * (R.vy[k] - y[m]);
                    const double cc = R.vx[k] * R.vx[k] + R.vy[k] * R.vy[k] - mm;
                    clip(poly, nx, ny, cc + 1e-12 * (std::fabs(cc) + 1.0));  // slack: count borderline nodes
                }
                if (poly.size() >= 1) ++extras;
            }
        R.extras = std::min(extras, kExtrasNever);
    }
    });

    // ---- isolation margin per record.  Cell buckets: every cell is listed in the buckets its acceptance region's
    //      bounding box touches, so a query with T''s bounding box finds every cell whose region can reach T'.
    std::vector<int8_t> epscode((size_t)3 * n_cells, (int8_t)(kEpsCodeMax + 1));
    if (P.walk_ok) {
        auto accept_scale = [&](const CellRecHost &U) { return 1.0 + 3.0 * 1.01 * (kTol + 2.0 * U.fp_err) + 1e-13; };
        std::vector<int32_t> cstart((size_t)gnx * gny + 1, 0), clist;
        auto cell_box = [&](int32_t c, int &ix0, int &iy0, int &ix1, int &iy1) {
            const CellRecHost &U = P.rec[c];
            const double ux = (U.vx[0] + U.vx[1] + U.vx[2]) / 3, uy = (U.vy[0] + U.vy[1] + U.vy[2]) / 3;
            const double s = U.cls == 2 ? 1.0 : accept_scale(U);
            double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
            for (int k = 0; k < 3; ++k) {
                const double px = ux + (U.vx[k] - ux) * s, py = uy + (U.vy[k] - uy) * s;
                xmin = std::min(xmin, px); xmax = std::max(xmax, px); ymin = std::min(ymin, py); ymax = std::max(ymax, py);
            }
            const double pad = 1e-12 * (std::fabs(xmin) + std::fabs(xmax) + std::fabs(ymin) + std::fabs(ymax) + 1.0);
            bucket_of(xmin - pad, ymin - pad, ix0, iy0);
            bucket_of(xmax + pad, ymax + pad, ix1, iy1);
        };
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<int32_t> cur;
            if (pass == 1) {
                for (size_t b = 0; b < (size_t)gnx * gny; ++b) cstart[b + 1] += cstart[b];
                clist.assign((size_t)cstart[(size_t)gnx * gny], 0);
                cur.assign(cstart.begin(), cstart.end() - 1);
            }
            for (int32_t c = 0; c < n_cells; ++c) {
                if (P.rec[c].cls == 2) continue;  // degenerate cells are handled through `wild`
                int ix0, iy0, ix1, iy1;
                cell_box(c, ix0, iy0, ix1, iy1);
                for (int by = iy0; by <= iy1; ++by)
                    for (int bx = ix0; bx <= ix1; ++bx) {
                        if (pass == 0) cstart[(size_t)by * gnx + bx + 1]++;
                        else clist[cur[(size_t)by * gnx + bx]++] = c;
                    }
            }
        }
        const double theta = 0.25 * kTol * 1.5;  // the record's region reaches this far (barycentric) beyond its entry edge
        parallel_cells(n_cells, [&](int32_t c_begin, int32_t c_end) {
        std::vector<int32_t> stamp(n_cells, -1), cand;
        Poly acc, pa, pb;
        for (int32_t c = c_begin; c < c_end; ++c) {
            const CellRecHost &R = P.rec[c];
            if (R.cls != 0 || R.extras >= kExtrasNever) continue;
            // a degenerate cell within the scan reach of this cell (its nodes can precede the cell's own in the node
            // scan of find_element) makes the reference's result unpredictable here
            bool near_wild = false;
            const double bx0 = std::min({R.vx[0], R.vx[1], R.vx[2]}), bx1 = std::max({R.vx[0], R.vx[1], R.vx[2]});
            const double by0 = std::min({R.vy[0], R.vy[1], R.vy[2]}), by1 = std::max({R.vy[0], R.vy[1], R.vy[2]});
            for (int32_t wc : wild) {
                const CellRecHost &U = P.rec[wc];
                const double reach = 2.0 * R.lmax + 1e-9 * (std::fabs(bx0) + std::fabs(bx1) + std::fabs(by0) + std::fabs(by1));
                const double ux0 = std::min({U.vx[0], U.vx[1], U.vx[2]}), ux1 = std::max({U.vx[0], U.vx[1], U.vx[2]});
                const double uy0 = std::min({U.vy[0], U.vy[1], U.vy[2]}), uy1 = std::max({U.vy[0], U.vy[1], U.vy[2]});
                if (!(ux0 > bx1 + reach || ux1 < bx0 - reach || uy0 > by1 + reach || uy1 < by0 - reach)) { near_wild = true; break; }
            }
            if (near_wild) continue;
            // candidates: cells whose acceptance region's bounding box touches this cell's (slightly padded) box
            cand.clear();
            {
                const double pad = 1e-7 * R.lmax + 1e-12 * (std::fabs(bx0) + std::fabs(bx1) + std::fabs(by0) + std::fabs(by1) + 1.0);
                int ix0, iy0, ix1, iy1;
                bucket_of(bx0 - pad, by0 - pad, ix0, iy0);
                bucket_of(bx1 + pad, by1 + pad, ix1, iy1);
                for (int by = iy0; by <= iy1; ++by)
                    for (int bx = ix0; bx <= ix1; ++bx)
                        for (int32_t s = cstart[(size_t)by * gnx + bx]; s < cstart[(size_t)by * gnx + bx + 1]; ++s) {
                            const int32_t u = clist[s];
                            if (u != c && stamp[u] != c) { stamp[u] = c; cand.push_back(u); }
                        }
            }
            // barycentric coordinates of this cell: mu_k(p) = (nk·p - ck), zero on the edge opposite vertex k
            const double a2s = (R.vx[1] - R.vx[0]) * (R.vy[2] - R.vy[0]) - (R.vx[2] - R.vx[0]) * (R.vy[1] - R.vy[0]);
            double mnx[3], mny[3], mc[3];
            for (int k = 0; k < 3; ++k) {
                const int i1 = (k + 1) % 3, i2 = (k + 2) % 3;  // the edge opposite vertex k runs i1 -> i2
                mnx[k] = -(R.vy[i2] - R.vy[i1]) / a2s;
                mny[k] = (R.vx[i2] - R.vx[i1]) / a2s;
                mc[k] = mnx[k] * R.vx[i1] + mny[k] * R.vy[i1];
            }
            double need[3] = {0, 0, 0};
            // only the part of a neighbour's acceptance region inside this cell (reaching theta beyond an entry edge) can raise
            // a margin: a candidate whose scaled bounding box misses the cell's padded box is skipped before any clipping
            const double rpad = 1e-6 * R.lmax + 1e-12 * (std::fabs(bx0) + std::fabs(bx1) + std::fabs(by0) + std::fabs(by1) + 1.0);
            for (int32_t u : cand) {
                const CellRecHost &U = P.rec[u];
                const double ux = (U.vx[0] + U.vx[1] + U.vx[2]) / 3, uy = (U.vy[0] + U.vy[1] + U.vy[2]) / 3;
                const double s = accept_scale(U);
                {
                    double sx0 = INFINITY, sx1 = -INFINITY, sy0 = INFINITY, sy1 = -INFINITY;
                    for (int k = 0; k < 3; ++k) {
                        const double qx = ux + (U.vx[k] - ux) * s, qy = uy + (U.vy[k] - uy) * s;
                        sx0 = std::min(sx0, qx); sx1 = std::max(sx1, qx); sy0 = std::min(sy0, qy); sy1 = std::max(sy1, qy);
                    }
                    if (sx0 > bx1 + rpad || sx1 < bx0 - rpad || sy0 > by1 + rpad || sy1 < by0 - rpad) continue;
                }
                for (int e = 0; e < 3; ++e) {
                    if (u == R.adj[e]) continue;  // the predecessor T of this record is evaluated exactly on the device
                    // rotated roles: entry edge (v_e, v_e+1) is opposite vertex e+2; exit edges opposite vertices e and e+1
                    const int k0 = e, k1 = (e + 1) % 3, k2 = (e + 2) % 3;
                    acc.clear();
                    for (int k = 0; k < 3; ++k) acc.push_back({ux + (U.vx[k] - ux) * s, uy + (U.vy[k] - uy) * s});
                    clip(acc, -mnx[k2], -mny[k2], -(mc[k2] - theta));  // mu_k2 >= -theta
                    if (acc.empty()) continue;
                    pa = acc;
                    clip(pa, mnx[k0] - mnx[k1], mny[k0] - mny[k1], mc[k0] - mc[k1]);  // mu_k0 <= mu_k1: min is mu_k0
                    for (const P2 &p : pa) need[e] = std::max(need[e], mnx[k0] * p.x + mny[k0] * p.y - mc[k0]);
                    pb = acc;
                    clip(pb, mnx[k1] - mnx[k0], mny[k1] - mny[k0], mc[k1] - mc[k0]);  // mu_k1 <= mu_k0: min is mu_k1
                    for (const P2 &p : pb) need[e] = std::max(need[e], mnx[k1] * p.x + mny[k1] * p.y - mc[k1]);
                }
            }
            for (int e = 0; e < 3; ++e) {
                const double eps = 2.0 * need[e] + 1e-9;
                int code = kEpsCodeMin;
                while (code <= kEpsCodeMax && std::ldexp(1.0, code - 20) < eps) ++code;
                epscode[(size_t)3 * c + e] = (int8_t)code;
            }
        }
        });
    }

    // ---- rotated walk records
    {
        uint64_t limit = 1ull << kWalkIdBits;
        if (const char *env = std::getenv("RT_TEST_WALK_RECORD_LIMIT")) limit = std::strtoull(env, nullptr, 10);  // tests only
        if ((uint64_t)3 * (uint64_t)n_cells + 1 >= limit) { P.walk_ok = false; P.note = "too many cells for the packed walk records"; }
    }
    P.wrec.assign((size_t)3 * n_cells, WalkRecHost{});
    P.adjr.assign((size_t)3 * n_cells, -1);
    for (int32_t c = 0; c < n_cells; ++c)
        for (int k = 0; k < 3; ++k) {
            const int32_t nb = P.rec[c].adj[k];
            if (nb < 0) continue;
            int ki = -1;
            for (int q = 0; q < 3; ++q)
                if (P.rec[nb].adj[q] == c) {
                    // the shared edge: same node pair
                    const int32_t a = cn[3 * c + k], b = cn[3 * c + (k + 1) % 3];
                    const int32_t a2 = cn[3 * nb + q], b2 = cn[3 * nb + (q + 1) % 3];
                    if ((a == a2 && b == b2) || (a == b2 && b == a2)) ki = q;
                }
            P.adjr[3 * c + k] = ki >= 0 ? 3 * nb + ki : -1;
        }
    P.n_records = (int64_t)3 * n_cells;
    P.eps_min = INFINITY; P.eps_max = 0.0;
    for (int32_t c = 0; c < n_cells; ++c) {
        const CellRecHost &R = P.rec[c];
        const double x1 = R.vx[0], y1 = R.vy[0], x2 = R.vx[1], y2 = R.vy[1], x3 = R.vx[2], y3 = R.vy[2];
        // det of [x1 x2 x3; y1 y2 y3; 1 1 1] as StaticArrays evaluates it (src/mesh.jl:166-168)
        const double dT = x1 * (y2 - y3) + y1 * (x3 - x2) + (x2 * y3 - y2 * x3);
        for (int e = 0; e < 3; ++e) {
            WalkRecHost &Wr = P.wrec[3 * c + e];
            const int i1 = (e + 1) % 3, i2 = (e + 2) % 3;
            // orientation of the entry edge relative to the neighbour's exit edge (a, b) = (its v_q, v_q+1)
            bool same = false;
            const int32_t back = P.adjr[3 * c + e];  // record (neighbour, its edge q) across the entry edge
            if (back >= 0) same = cn[3 * c + e] == cn[3 * (back / 3) + back % 3];
            int code = epscode[(size_t)3 * c + e];
            int extras = R.extras;
            if (!P.walk_ok || code > kEpsCodeMax || back < 0) { extras = kExtrasNever; code = kEpsCodeMax; }
            if (extras < kExtrasNever) {
                ++P.n_records_walk;
                const double eps = std::ldexp(1.0, code - 20);
                P.eps_min = std::min(P.eps_min, eps); P.eps_max = std::max(P.eps_max, eps);
                P.extras_max = std::max(P.extras_max, extras);
            }
            Wr.hdr = (uint64_t)(P.adjr[3 * c + i1] + 1) | ((uint64_t)(P.adjr[3 * c + i2] + 1) << kWalkIdBits) |
                     ((uint64_t)(extras & 15) << (2 * kWalkIdBits)) | ((uint64_t)(code & 31) << (2 * kWalkIdBits + 4)) |
                     ((uint64_t)(same ? 1 : 0) << 63);
            Wr.dT = dT;
            Wr.x2 = R.vx[i2]; Wr.y2 = R.vy[i2];
            Wr.e1A = R.eA[i1]; Wr.e1B = R.eB[i1]; Wr.e1C = R.eC[i1];
            Wr.e2A = R.eA[i2]; Wr.e2B = R.eB[i2]; Wr.e2C = R.eC[i2];
        }
    }
    if (P.n_records_walk == 0) { P.eps_min = 0.0; if (P.walk_ok) { P.walk_ok = false; if (P.note.empty()) P.note = "no cell passes the walk certificates"; } }

    // ---- cheap-step ("topological") records.  rt_device.hpp `topo_step` states what the
    //      lane checks; here are the constants, each a sufficient bound with room to spare:
    //  With s_i the signed (scaled, |s| <= distance) distances of v0, v1, v2 from the track line, D = |s0| + |s1|,
    //  m = min(|s0|, |s1|): the entry point has barycentric coordinates (1-u, u, 0), u = |s0| / D, in T'.
    //   * δ, the error of the computed entry point (= the exit point computed for T) plus the rounding of up to
    //     kTopoKcap additions of the step vector: |δ| <= κ0·lmax/D + δ_add; k2 = κ0·lmax·g / (0.3·tol) makes
    //     g·κ0·lmax/D <= 0.3·tol (g = 1 / smallest altitude of T'), and records with g·δ_add > 0.075·tol are left out:
    //     xp is never more than 0.375·tol (barycentric) outside the entry edge — the depth the isolation regions
    //     above were computed for, and T' passes its own test there.
    //   * the reference keeps stepping while it locates T again, which ends for sure once xp is 1.5·tol·h_T (+|δ|)
    //     beyond the shared edge, after a path of at most tol·dtf/D + tiny with dtf = 1.5·|dT| + 0.375·|dT'|;
    //     D·(kTopoKcap - 2)·tiny/tol >= dtf bounds the number of those steps by kTopoKcap.
    //   * along that path the two exit-side coordinates stay >= E (isolation margin of the walk records, and the
    //     distance > tiny from every border that `inboundary` needs) if m >= (E + 0.375·tol + g·tiny_max)·D + g·tol·dtf:
    //     the record's eps code holds the first factor, g1 = g·tol·dtf.
    //   * lc: the chord in T' is at least 2·tan(γ/2)·|s_v| (v: the vertex shared by entry and exit edge, γ its angle);
    //     |s_v| >= lc keeps it above l_min and (with the per-track factor topo_rmax/|cos ϕ|) above twice the rounding of
    //     both end points divided by |cos ϕ| — the order of the pair cannot flip (src/intersection.jl:151-159).
    P.trec.assign((size_t)3 * std::max(n_cells, 1), TopoRecHost{});
    P.etab.assign((size_t)3 * std::max(n_cells, 1), EdgeABCHost{});
    for (int32_t c = 0; c < n_cells; ++c)
        for (int k = 0; k < 3; ++k) P.etab[(size_t)3 * c + k] = {P.rec[c].eA[k], P.rec[c].eB[k], P.rec[c].eC[k], 0.0};
    {
        const double corner = std::hypot(cmax_x, cmax_y);
        const double kappa0 = 20.0 * kUlp * corner;
        const double dadd = 3.0 * kTopoKcap * kUlp * corner;
        // the certificates hold for tiny_step up to this bound (it enters the margins below): relative to the mesh, but
        // never below 4e-8 — the reference's default tiny_step is 1e-8, and a fine mesh must not lose the cheap step to it
        // (cells too small for the resulting margins simply get no cheap record)
        P.topo_tiny_max = std::max(1e-6 * l_max, 4e-8);
        const bool ids_fit = (uint64_t)3 * (uint64_t)n_cells + 3 < (1ull << kWalkIdBits);
        double rmax = 0.0, end_err = 0.0;
        for (int32_t c = 0; c < n_cells; ++c) {
            const CellRecHost &R = P.rec[c];
            for (int e = 0; e < 3; ++e) {
                const WalkRecHost &Wr = P.wrec[(size_t)3 * c + e];
                TopoRecHost &Tr = P.trec[(size_t)3 * c + e];
                const int i0 = e, i1 = (e + 1) % 3, i2 = (e + 2) % 3;
                int extras = (int)((Wr.hdr >> (2 * kWalkIdBits)) & 15);
                int code = (int)((Wr.hdr >> (2 * kWalkIdBits + 4)) & 31);
                uint64_t n1 = Wr.hdr & ((1ull << kWalkIdBits) - 1), n2 = (Wr.hdr >> kWalkIdBits) & ((1ull << kWalkIdBits) - 1);
                // border exits: the edge lies exactly on a side of the bounding box
                auto end_code = [&](int ia, int ib) -> uint64_t {
                    const double xa = R.vx[ia], ya = R.vy[ia], xb = R.vx[ib], yb = R.vy[ib];
                    if (xa == xb && (xa == bb[0] || xa == bb[2]) && ya != yb) {
                        end_err = std::max(end_err, std::fabs(xa) * (2 * kUlp * (std::fabs(ya) + std::fabs(yb)) / std::fabs(ya - yb) + 12 * kUlp));
                        return kTopoEndV;
                    }
                    if (ya == yb && (ya == bb[1] || ya == bb[3]) && xa != xb) {
                        end_err = std::max(end_err, std::fabs(ya) * (2 * kUlp * (std::fabs(xa) + std::fabs(xb)) / std::fabs(xa - xb) + 12 * kUlp));
                        return kTopoEndH;
                    }
                    return 0;
                };
                if (n1 == 0 && R.adj[i1] < 0) n1 = end_code(i1, i2);
                if (n2 == 0 && R.adj[i2] < 0) n2 = end_code(i2, i0);
                double g1 = 0, k2 = 0, dtf = 0, lc = 0;
                bool ok = ids_fit && P.walk_ok && extras < kExtrasNever && R.cls == 0 && R.adj[e] >= 0 && P.rec[R.adj[e]].cls == 0;
                if (ok) {
                    const CellRecHost &Tp = P.rec[R.adj[e]];
                    const double hmin = R.area2 / R.lmax, g = 1.0 / hmin;
                    ok = g * dadd <= 0.075 * kTol;
                    // isolation margin, border clearance (inboundary(xp, tiny) must stay false along the path)
                    double E = std::ldexp(1.0, code - 20);
                    const double b0[4] = {R.vx[i0] - bb[0], bb[2] - R.vx[i0], R.vy[i0] - bb[1], bb[3] - R.vy[i0]};
                    const double b1[4] = {R.vx[i1] - bb[0], bb[2] - R.vx[i1], R.vy[i1] - bb[1], bb[3] - R.vy[i1]};
                    const double b2[4] = {R.vx[i2] - bb[0], bb[2] - R.vx[i2], R.vy[i2] - bb[1], bb[3] - R.vy[i2]};
                    for (int q = 0; q < 4 && ok; ++q) {
                        const double den = b0[q] + b1[q];
                        if (!(den > 0)) { ok = false; break; }
                        E = std::max(E, (1.001 * P.topo_tiny_max + 16 * kUlp * corner + 0.375 * kTol * std::fabs(b2[q])) / den);
                    }
                    const double E1 = 1.01 * (E + 0.375 * kTol + g * P.topo_tiny_max);
                    int tcode = kEpsCodeMin;
                    while (tcode <= kEpsCodeMax && std::ldexp(1.0, tcode - 20) < E1) ++tcode;
                    ok = ok && tcode <= kEpsCodeMax;
                    code = std::min(tcode, kEpsCodeMax);
                    dtf = 1.01 * (1.5 * Tp.area2 + 0.375 * R.area2);
                    g1 = 1.01 * g * kTol * dtf;
                    k2 = 1.01 * kappa0 * R.lmax * g / (0.3 * kTol);
                    // angles at v0 and v1 (the ends of the entry edge)
                    auto half_tan = [&](int iv, int ia, int ib) {
                        const double ux = R.vx[ia] - R.vx[iv], uy = R.vy[ia] - R.vy[iv], wx = R.vx[ib] - R.vx[iv], wy = R.vy[ib] - R.vy[iv];
                        const double cr = std::fabs(ux * wy - uy * wx), dt = ux * wx + uy * wy;
                        return std::tan(0.5 * std::atan2(cr, dt));
                    };
                    const double cv = 2.0 * std::min(half_tan(i0, i1, i2), half_tan(i1, i2, i0)) * (1.0 - 1e-9);
                    ok = ok && cv > 1e-3;
                    if (ok) {
                        lc = 1.05 * (P.l_min + 1.2 * kTol * hmin) / cv;
                        rmax = std::max(rmax, 1.2 * kTol * hmin / (P.l_min + 1.2 * kTol * hmin));
                    }
                    ok = ok && bf16_up(g1) < 0x7f80u && bf16_up(k2) < 0x7f80u && bf16_up(dtf) < 0x7f80u && bf16_up(lc) < 0x7f80u;
                }
                if (!ok) { extras = kExtrasNever; code = kEpsCodeMax; }
                else ++P.n_records_topo;
                Tr.hdr = n1 | (n2 << kWalkIdBits) | ((uint64_t)(extras & 15) << (2 * kWalkIdBits)) |
                         ((uint64_t)(code & 31) << (2 * kWalkIdBits + 4)) | (Wr.hdr & (1ull << 63));
                Tr.x2 = R.vx[i2]; Tr.y2 = R.vy[i2];
                Tr.g1 = bf16_up(g1); Tr.k2 = bf16_up(k2); Tr.dtf = bf16_up(dtf); Tr.lc = bf16_up(lc);
            }
        }
        P.topo_rmax = 1.01 * rmax;
        P.topo_end_err = end_err;
        P.topo_ok = P.n_records_topo > 0;
    }
    return P;
}

}  // namespace rtprep
