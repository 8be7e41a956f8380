// rt_device.hpp — device-side geometry of the segmentize! march (gfx950, FP64, no MFMA).
//
// Exact-decision emulation of the reference's per-track march: every predicate below is
// evaluated with the reference's own operation order in IEEE double (the TU is compiled
// with -ffp-contract=off; f64 div/sqrt are correctly rounded), so element ids, segment
// counts and skip/emit decisions follow the reference's tolerance- and order-dependent
// rules, not "pure geometry".  Citations are into /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Device pointers carry the global address space explicitly: pointers that reach the march
// through by-reference structs would otherwise be loaded with flat_* instructions (which wait
// on both vmcnt and lgkmcnt) instead of global_load_*.
#define RT_G __attribute__((address_space(1)))
// The geometry below is plain arithmetic: it also compiles for the host, where tests/host_march.hip runs it
// against the CPU checker without a GPU (test infrastructure; the library itself never marches on the host).
#define RT_HD __host__ __device__

namespace rt {

template <typename T>
__host__ __device__ inline RT_G T *as_global(T *p) { return (RT_G T *)p; }

constexpr double kRtolDefault = 1.4901161193847656e-8;  // sqrt(eps(Float64)) = Base.rtoldefault
constexpr double kHalfPi = 1.5707963267948966;          // Float64(pi)/2, src/intersection.jl:153
constexpr int kMaxIter = 10000;                         // const MAX_ITER, src/track.jl:104
constexpr int kMaxK = 8;                                // width of the in-register k-best list; larger `k` stream on

// Walk record for (cell, entry edge): the cell's vertices rotated cyclically so that rotated
// edge 0 = (v0, v1) is the entry edge, in the cell's own edge orientation (built on the host,
// rt_mesh_prep.hpp).  80 B: one lane fetches its next cell with five independent 16-B loads — every load
// instruction of such a per-lane gather costs the wave ~50 cycles when the lanes of a quad read the same
// line and up to ~250 when they all differ (tools/micro/bench_gather.hip), so the record holds only what
// cannot be had otherwise: v0 and v1 are the endpoints (a, b) of the predecessor's exit edge, already in the
// walk state, and the cell id is record index / 3.
constexpr int kWalkIdBits = 27;
constexpr int kExtrasNever = 15;  // extras field of a record the walk step must not use (rt_mesh_prep.hpp)
struct __attribute__((aligned(16))) WalkRec {
    uint64_t hdr;          // bits 0..26 next1 + 1, 27..53 next2 + 1 (0: boundary), 54..57 extras bound (15: never),
                           // 58..62 isolation margin code (eps = 2^(code - 20)), 63: v0 == a
    double dT;             // det of the barycentric system in the ORIGINAL node order, reference operation order
    double x2, y2;         // the vertex opposite the entry edge
    double e1A, e1B, e1C, e2A, e2B, e2C;   // general_form (src/intersection.jl:11-18) of rotated edges 1, 2
};
static_assert(sizeof(WalkRec) == 80, "WalkRec layout");

// Flattened mesh in HBM (SoA; all ids 0-based on the device, converted at upload).
// What only the generic step reads (locate + intersections): kept behind a pointer in constant address space,
// so that the march loop, which runs the walk step >99.9 % of the time, does not hold ~30 SGPRs of pointers
// and grid parameters it never uses there (the kernel was spilling SGPRs to VGPR lanes in its hot path).
// One incident cell of a node, as find_element's scan needs it: the cell id and its three vertices in the cell's own node
// order (the values of x[], y[] behind cn[]) — 64 B, four independent 16-B loads instead of the chain ncd -> cn -> x, y.
struct __attribute__((aligned(16))) FanEntry {
    double x1, y1, x2, y2, x3, y3;
    int32_t cell, adj[3];  // adj[k]: the walk record reached across edge k of the cell (adjr[3*cell + k]), -1 on the boundary
};
static_assert(sizeof(FanEntry) == 64, "FanEntry layout");
struct Tri { double x1, y1, x2, y2, x3, y3; int32_t adj[3]; };  // a cell's vertices in its node order + its three successors

// Cheap-step record for (cell, entry edge): 32 B, two 16-B loads (rt_mesh_prep.hpp, TopoRecHost).
constexpr uint32_t kTopoEndV = (1u << kWalkIdBits) - 1;  // successor field: the exit edge lies on a vertical border
constexpr uint32_t kTopoEndH = (1u << kWalkIdBits) - 2;  // ... on a horizontal border
constexpr int kTopoKcap = 4096;
struct __attribute__((aligned(16))) TopoRec {
    uint64_t hdr;       // as WalkRec::hdr; successor fields may hold kTopoEndV / kTopoEndH; own extras / eps code
    double x2, y2;      // the vertex opposite the entry edge
    uint32_t c01, c23;  // bfloat16 patterns: g1 | k2 << 16, dtf | lc << 16
};
static_assert(sizeof(TopoRec) == 32, "TopoRec layout");
struct __attribute__((aligned(16))) EdgeABC { double A, B, C, pad; };
static_assert(sizeof(EdgeABC) == 32, "EdgeABC layout");

struct DGeo {
    const RT_G double *x;        // [n_nodes]
    const RT_G double *y;        // [n_nodes]
    const RT_G int32_t *cn;      // [3*n_cells] cell -> nodes, reference order
    const RT_G int32_t *ncp;     // [n_nodes+1] node -> cells CSR offsets
    const RT_G int32_t *ncd;     // node -> cells, ascending cell id per node
    const RT_G int32_t *gstart;  // [gnx*gny+1] uniform node grid CSR (row-major, y-major rows)
    const RT_G int32_t *gnode;   // node ids grouped by bucket
    const RT_G int32_t *c3start; // [gnx*gny+1] per bucket: the nodes of its 3x3 block of buckets, contiguous ...
    const RT_G int32_t *c3node;  // ... their ids ...
    const RT_G double *c3x, *c3y;  // ... and coordinates (same values as x[], y[]: the distances come out bit-identical)
    const RT_G struct FanEntry *fan;  // [ncp[n_nodes]] node -> incident cells WITH their vertex coordinates, in ncd's order
    double gx0, gy0, gh, ginv;   // grid origin, bucket size and its inverse
    int32_t gnx, gny;
    int32_t n_nodes, pad_;
};
#define RT_K __attribute__((address_space(4)))

struct DMesh {
    const RT_G WalkRec *wrec;    // [3*n_cells] rotated walk records
    const RT_G int32_t *adjr;    // [3*n_cells] record index reached across edge k of cell c; -1 on the boundary
    double d_vertex, l_min;      // mesh-wide certificate margins of the walk step (the isolation margin is per record)
    int32_t walk_ok;
    int32_t n_cells;
    double bx0, by0, bx1, by1;   // bounding box (bb_min, bb_max)
    const RT_K DGeo *geo;        // device copy of the generic step's data
    const RT_G struct TopoRec *trec;  // [3*n_cells] cheap-step records (topo_step)
    const RT_G struct EdgeABC *etab;  // [3*n_cells] general_form of edge k of cell c at 3*c + k
};

// The generic step's data, fetched with scalar loads where it is needed.
RT_HD __forceinline__ DGeo load_geo(const RT_K DGeo *p) {
    DGeo g;
    g.x = p->x; g.y = p->y; g.cn = p->cn; g.ncp = p->ncp; g.ncd = p->ncd; g.gstart = p->gstart; g.gnode = p->gnode;
    g.c3start = p->c3start; g.c3node = p->c3node; g.c3x = p->c3x; g.c3y = p->c3y; g.fan = p->fan;
    g.gx0 = p->gx0; g.gy0 = p->gy0; g.gh = p->gh; g.ginv = p->ginv;
    g.gnx = p->gnx; g.gny = p->gny; g.n_nodes = p->n_nodes; g.pad_ = 0;
    return g;
}

// Per-track inputs in HBM (SoA, uid order) + the march order.
struct DTracks {
    const RT_G double *px, *py, *phi, *cs, *sn;
    const RT_G double *A, *B, *C, *ell;
    const RT_G int32_t *azim;    // 1-based azimuthal index
    const RT_G int32_t *perm;    // march slot -> track (longest tracks first)
    int64_t n;
    // in MARCH-SLOT order, for k_materialise (one level of dependent loads less than through perm): the track lines, the record
    // counts (written by the whole-track march beside counts[uid]) and the CSR offsets (k_scan_write, through iperm: uid -> slot)
    const RT_G double *As, *Bs, *Cs, *Ls, *Dxs, *Dys;  // (Ls: the tracks' lengths ℓ; Dxs, Dys: cos ϕ, sin ϕ)
    const RT_G double *Pxs, *Pys, *Phis;               // start points and angles in march-slot order: the whole-track march reads
    const RT_G int32_t *Azs;                           // everything a lane starts from by slot — one trip, none behind perm[slot]
    const RT_G int32_t *iperm;
    RT_G int32_t *cnt_slot;
    RT_G int64_t *off_slot;
    RT_G double *w_slot;    // δs of the track's azimuthal angle, march-slot order (k_march leaves it for k_materialise's fill_volumes terms)
};

struct DParams {
    double tiny_step;
    double rtol;
    int32_t k;
    int32_t n_azim_2;
    int64_t iter_cap;
    double topo_tiny_max, topo_rmax, topo_end_err;  // cheap steps (topo_track); unused elsewhere
    int32_t topo_force;  // 1: option "topo" = 2 — a wave that is refused often does NOT hand back to exact steps
    int32_t pad_;
    double tally_c1, tally_c2;  // fill_volumes of a cheap record from the vertices' distances (k_march) only if chord >= c1 and
                                // chord · (smaller |s_p − s_q| of its two crossings) >= c2; else k_materialise adds the term
                                // from the record's own length (rt_mesh_prep.hpp); c1 = ∞: all of them
};

// ---------------------------------------------------------------- Base.isapprox ----------
RT_HD __forceinline__ bool isfin(double v) { return fabs(v) <= 1.7976931348623157e308; }

// isapprox(x, y; rtol) scalar form with atol = 0
RT_HD __forceinline__ bool isapprox_s(double x, double y, double rtol) {
    if (x == y) return true;
    if (!(isfin(x) && isfin(y))) return false;
    const double ax = fabs(x), ay = fabs(y);
    return fabs(x - y) <= rtol * (ax > ay ? ax : ay);
}
RT_HD __forceinline__ double norm2(double a, double b) { return sqrt(a * a + b * b); }

// isapprox(p, q) for Point2D with default tolerances (array form: 2-norms)
RT_HD __forceinline__ bool isapprox_v2(double px, double py, double qx, double qy) {
    const double d = norm2(px - qx, py - qy);
    if (isfin(d)) {
        const double np = norm2(px, py), nq = norm2(qx, qy);
        return d <= kRtolDefault * (np > nq ? np : nq);
    }
    return isapprox_s(px, qx, kRtolDefault) && isapprox_s(py, qy, kRtolDefault);
}

// inboundary(mesh, x, atol) — src/mesh.jl:91-95: four scalar isapprox(x[i], bb[i]; atol), i.e.
// x == b || (isfinite(x) && isfinite(b) && |x - b| <= max(atol, rtol·max(|x|,|b|))) with rtol = 0 when
// atol > 0 and √eps otherwise.  Evaluated without branches (it runs once per march iteration and a
// short-circuit form costs ~25 divergent branches there); the bounding box is finite.
RT_HD __forceinline__ bool near_bb(double x, double b, double atol) {
    const double ax = fabs(x), ab = fabs(b);
    const double tol = atol > 0.0 ? atol : kRtolDefault * (ax > ab ? ax : ab);
    return ((int)(x == b) | ((int)isfin(x) & (int)(fabs(x - b) <= tol))) != 0;
}
// atol == 0 (never in practice): out of line, so that its constants do not occupy scalar registers in the march
inline RT_HD __noinline__ bool inboundary_general(double bx0, double by0, double bx1, double by1, double x, double y, double atol) {
    return ((int)near_bb(x, bx1, atol) | (int)near_bb(x, bx0, atol) | (int)near_bb(y, by1, atol) | (int)near_bb(y, by0, atol)) != 0;
}
RT_HD __forceinline__ bool inboundary(const DMesh &m, double x, double y, double atol) {
    // atol > 0 (every real call: atol = tiny_step; wave-uniform): rtol = 0 and the box is finite (rt_mesh_create
    // checks), so x == b || (isfinite(x) && |x - b| <= atol) is |x - b| <= atol — 8 instead of ~60 instructions
    // per march iteration
    if (atol > 0.0)
        return ((int)(fabs(x - m.bx1) <= atol) | (int)(fabs(x - m.bx0) <= atol) | (int)(fabs(y - m.by1) <= atol) |
                (int)(fabs(y - m.by0) <= atol)) != 0;
    return inboundary_general(m.bx0, m.by0, m.bx1, m.by1, x, y, atol);
}

// ------------------------------------------------------- point_in_triangle ---------------
// src/mesh.jl:158-176: λ = [x1 x2 x3; y1 y2 y3; 1 1 1] \ [x, y, 1] by the closed form
// StaticArrays uses for 3x3 (cofactors / det, det = col1 · (col2 × col3)); inside iff every
// λ ∈ [0 - tol, 1 + tol], tol = sqrt(eps).
RT_HD __forceinline__ bool point_in_triangle(const Tri &t, double x, double y) {
    const double x1 = t.x1, y1 = t.y1, x2 = t.x2, y2 = t.y2, x3 = t.x3, y3 = t.y3;
    const double d = x1 * (y2 - y3) + y1 * (x3 - x2) + (x2 * y3 - y2 * x3);
    const double l1 = ((y2 - y3) * x + (x3 - x2) * y + (x2 * y3 - x3 * y2)) / d;
    const double l2 = ((y3 - y1) * x + (x1 - x3) * y + (x3 * y1 - x1 * y3)) / d;
    const double l3 = ((y1 - y2) * x + (x2 - x1) * y + (x1 * y2 - x2 * y1)) / d;
    const double lo = 0.0 - kRtolDefault, hi = 1.0 + kRtolDefault;
    return (lo <= l1 && l1 <= hi) && (lo <= l2 && l2 <= hi) && (lo <= l3 && l3 <= hi);
}

// ------------------------------------------------------- exact (k-)nearest nodes ---------
// Replaces NearestNeighbors' kd-tree (src/mesh.jl:38-42,107,123) by an exact ring search on
// a uniform bucket grid: rings of buckets around the query are visited until the k-th best
// distance is provably smaller than the distance to anything unvisited.
struct KBest {
    double d2[kMaxK];
    int32_t id[kMaxK];
    int32_t n, k;
};
// Nodes are ranked by (squared distance, id): a total order, so that the result does not depend on the order in
// which a search structure happens to visit exactly equidistant nodes.
RT_HD __forceinline__ bool node_before(double d2a, int32_t ida, double d2b, int32_t idb) {
    return d2a < d2b || (d2a == d2b && ida < idb);
}
RT_HD __forceinline__ void kbest_push(KBest &b, double d2, int32_t id) {
    if (b.n == b.k && !node_before(d2, id, b.d2[b.n - 1], b.id[b.n - 1])) return;
    int i = (b.n < b.k) ? b.n++ : b.n - 1;
    while (i > 0 && node_before(d2, id, b.d2[i - 1], b.id[i - 1])) {
        b.d2[i] = b.d2[i - 1];
        b.id[i] = b.id[i - 1];
        --i;
    }
    b.d2[i] = d2;
    b.id[i] = id;
}

// Lower bound on the distance from (qx,qy) to any node outside the visited block of buckets
// [ix-r, ix+r] x [iy-r, iy+r]; +inf once the block covers the whole grid.
RT_HD __forceinline__ double ring_bound(const DGeo &m, double qx, double qy, int ix, int iy, int r) {
    const double inf = __builtin_huge_val();
    double lb = inf;
    if (ix - r > 0) lb = fmin(lb, qx - (m.gx0 + (double)(ix - r) * m.gh));
    if (ix + r < m.gnx - 1) lb = fmin(lb, (m.gx0 + (double)(ix + r + 1) * m.gh) - qx);
    if (iy - r > 0) lb = fmin(lb, qy - (m.gy0 + (double)(iy - r) * m.gh));
    if (iy + r < m.gny - 1) lb = fmin(lb, (m.gy0 + (double)(iy + r + 1) * m.gh) - qy);
    return lb;
}

RT_HD __forceinline__ void bucket_of(const DGeo &m, double qx, double qy, int &ix, int &iy) {
    double fx = floor((qx - m.gx0) * m.ginv), fy = floor((qy - m.gy0) * m.ginv);
    fx = fx < 0.0 ? 0.0 : fx;
    fy = fy < 0.0 ? 0.0 : fy;
    ix = fx > (double)(m.gnx - 1) ? m.gnx - 1 : (int)fx;
    iy = fy > (double)(m.gny - 1) ? m.gny - 1 : (int)fy;
}

// nn(kdtree, x): the nearest node (0-based id).
RT_HD __forceinline__ int32_t nearest_node(const DGeo &m, double qx, double qy) {
    int ix, iy;
    bucket_of(m, qx, qy, ix, iy);
    double best = __builtin_huge_val();
    int32_t best_id = 0x7fffffff;
    const int rmax = m.gnx > m.gny ? m.gnx : m.gny;
    {   // rings 0 and 1 in one go: the bucket's 3x3 block as one contiguous range of (id, x, y)
        const int b = iy * m.gnx + ix;
        // Three candidates per round trip: a lane of the march walks this range alone, and with one node per iteration every
        // iteration is a dependent memory round trip (≈12 of them, the longest stretch of a track's first step; four per
        // trip would take the march without cheap steps from 168 to 169 VGPRs).  The minimum under the (distance, id) order
        // does not depend on the visiting order; a clamped repeat of the last node changes nothing.
        const int32_t s1 = m.c3start[b + 1];
        for (int32_t s = m.c3start[b]; s < s1; s += 3) {
            int32_t idb[3];
            double xb[3], yb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int32_t q = s + j < s1 ? s + j : s1 - 1;
                idb[j] = m.c3node[q]; xb[j] = m.c3x[q]; yb[j] = m.c3y[q];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dx = qx - xb[j], dy = qy - yb[j];
                const double d2 = dx * dx + dy * dy;
                if (node_before(d2, idb[j], best, best_id)) { best = d2; best_id = idb[j]; }
            }
        }
        const double lb = ring_bound(m, qx, qy, ix, iy, 1) - 1e-9 * m.gh;
        if (lb == __builtin_huge_val() || (best_id != 0x7fffffff && lb > 0.0 && best < lb * lb)) return best_id == 0x7fffffff ? -1 : best_id;
    }
    for (int r = 2; r <= rmax; ++r) {
        const int y0 = iy - r, y1 = iy + r;
        for (int by = (y0 < 0 ? 0 : y0); by <= (y1 >= m.gny ? m.gny - 1 : y1); ++by) {
            const int xl = ix - r < 0 ? 0 : ix - r, xr = ix + r >= m.gnx ? m.gnx - 1 : ix + r;
            const bool full = (by == y0) || (by == y1);
            // full rows: one contiguous CSR range; inner rows: only the two end buckets
            for (int part = 0; part < (full ? 1 : 2); ++part) {
                int b0, b1;
                if (full) { b0 = by * m.gnx + xl; b1 = by * m.gnx + xr + 1; }
                else {
                    const int bx = part == 0 ? ix - r : ix + r;
                    if (bx < 0 || bx >= m.gnx) continue;
                    b0 = by * m.gnx + bx; b1 = b0 + 1;
                }
                for (int32_t s = m.gstart[b0]; s < m.gstart[b1]; ++s) {
                    const int32_t id = m.gnode[s];
                    const double dx = qx - m.x[id], dy = qy - m.y[id];
                    const double d2 = dx * dx + dy * dy;
                    if (node_before(d2, id, best, best_id)) { best = d2; best_id = id; }
                }
            }
        }
        const double lb = ring_bound(m, qx, qy, ix, iy, r) - 1e-9 * m.gh;
        if (lb == __builtin_huge_val()) break;
        if (best_id != 0x7fffffff && lb > 0.0 && best < lb * lb) break;
    }
    return best_id == 0x7fffffff ? -1 : best_id;
}

// knn(kdtree, x, k, true, i -> i == skip): the k nearest nodes other than `skip`, ascending.  AFTER: only nodes that
// follow (d2_after, id_after) in the (squared distance, id) order — widths beyond the in-register list are served
// in batches, the next kMaxK nodes after the last one of the previous batch.
template <bool AFTER>
RT_HD __noinline__ void knearest_nodes(const DGeo &m, double qx, double qy, int k, int32_t skip, double d2_after,
                                       int32_t id_after, KBest &kb) {
    kb.n = 0;
    kb.k = k > kMaxK ? kMaxK : k;  // (callers pass k <= kMaxK; wider searches stream, see find_element_fallback)
    if (kb.k <= 0) return;
    int ix, iy;
    bucket_of(m, qx, qy, ix, iy);
    const int rmax = m.gnx > m.gny ? m.gnx : m.gny;
    for (int r = 0; r <= rmax; ++r) {
        const int y0 = iy - r, y1 = iy + r;
        for (int by = (y0 < 0 ? 0 : y0); by <= (y1 >= m.gny ? m.gny - 1 : y1); ++by) {
            const int xl = ix - r < 0 ? 0 : ix - r, xr = ix + r >= m.gnx ? m.gnx - 1 : ix + r;
            const bool full = (by == y0) || (by == y1);
            for (int part = 0; part < (full ? 1 : 2); ++part) {
                int b0, b1;
                if (full) { b0 = by * m.gnx + xl; b1 = by * m.gnx + xr + 1; }
                else {
                    const int bx = part == 0 ? ix - r : ix + r;
                    if (bx < 0 || bx >= m.gnx) continue;
                    b0 = by * m.gnx + bx; b1 = b0 + 1;
                }
                for (int32_t s = m.gstart[b0]; s < m.gstart[b1]; ++s) {
                    const int32_t id = m.gnode[s];
                    if (id == skip) continue;
                    const double dx = qx - m.x[id], dy = qy - m.y[id];
                    const double d2 = dx * dx + dy * dy;
                    if (!AFTER || node_before(d2_after, id_after, d2, id)) kbest_push(kb, d2, id);
                }
            }
        }
        const double lb = ring_bound(m, qx, qy, ix, iy, r) - 1e-9 * m.gh;
        if (lb == __builtin_huge_val()) break;
        if (kb.n == kb.k && lb > 0.0 && kb.d2[kb.n - 1] < lb * lb) break;
    }
}

RT_HD __forceinline__ Tri load_tri(const DGeo &m, int32_t cell) {
    const int32_t n1 = m.cn[3 * cell], n2 = m.cn[3 * cell + 1], n3 = m.cn[3 * cell + 2];
    Tri t;
    t.x1 = m.x[n1]; t.y1 = m.y[n1]; t.x2 = m.x[n2]; t.y2 = m.y[n2]; t.x3 = m.x[n3]; t.y3 = m.y[n3];
    t.adj[0] = t.adj[1] = t.adj[2] = -2;  // not loaded: walk_enter reads adjr
    return t;
}
// The cells of node_cells[node] in stored order, first hit wins (src/mesh.jl:110-118); `tri` receives the hit's vertices.
RT_HD __forceinline__ int32_t first_cell_containing(const DGeo &m, int32_t node, double x, double y, Tri &tri) {
    for (int32_t s = m.ncp[node]; s < m.ncp[node + 1]; ++s) {
        const RT_G FanEntry *e = m.fan + s;
        Tri t;
        t.x1 = e->x1; t.y1 = e->y1; t.x2 = e->x2; t.y2 = e->y2; t.x3 = e->x3; t.y3 = e->y3;
        if (point_in_triangle(t, x, y)) {
            t.adj[0] = e->adj[0]; t.adj[1] = e->adj[1]; t.adj[2] = e->adj[2];
            tri = t;
            return e->cell;
        }
    }
    return -1;
}

// The fallback half of find_element (src/mesh.jl:123-132) for find_element(mesh, xp) [k=2]
// followed, on failure, by find_element(mesh, xp, k) (src/track.jl:122,139).  The second
// call repeats the first one's tests and then looks at nodes 3..k of the same sorted list,
// so one sorted list of max(2,k) nodes serves both.
// WIDEK (k > kMaxK; src/mesh.jl:123 takes any k): the same sorted node list, kMaxK nodes at a time.  A separate
// instantiation, so that the march of the usual k keeps its register budget.
template <bool WIDEK>
RT_HD __noinline__ int32_t find_element_fallback(const DGeo &m, double x, double y, int k, int32_t nn_id, Tri &tri) {
    KBest kb;
    const int kk = k > 2 ? k : 2;
    if (WIDEK) {
        double d2p = -1.0;
        int32_t idp = -1;
        for (int done = 0; done < kk; done += kMaxK) {
            knearest_nodes<true>(m, x, y, kk - done < kMaxK ? kk - done : kMaxK, nn_id, d2p, idp, kb);
            for (int j = 0; j < kb.n; ++j) {
                const int32_t c = first_cell_containing(m, kb.id[j], x, y, tri);
                if (c >= 0) return c;
            }
            if (kb.n < kMaxK) break;  // no more nodes
            d2p = kb.d2[kb.n - 1]; idp = kb.id[kb.n - 1];
        }
        return -1;
    }
    knearest_nodes<false>(m, x, y, kk, nn_id, -1.0, -1, kb);
    const int first = kb.n < 2 ? kb.n : 2;
    for (int j = 0; j < first; ++j) {
        const int32_t c = first_cell_containing(m, kb.id[j], x, y, tri);
        if (c >= 0) return c;
    }
    const int lim = kb.n < k ? kb.n : k;  // second call: knn(k) — only new nodes can succeed
    for (int j = first; j < lim; ++j) {
        const int32_t c = first_cell_containing(m, kb.id[j], x, y, tri);
        if (c >= 0) return c;
    }
    return -1;
}

// find_element (src/mesh.jl:103-146), 0-based cell id or -1.
template <bool WIDEK = false>
RT_HD __forceinline__ int32_t find_element(const DGeo &m, double x, double y, int k, Tri &tri) {
    const int32_t nn_id = nearest_node(m, x, y);
    if (nn_id < 0) return -1;
    const int32_t c = first_cell_containing(m, nn_id, x, y, tri);
    if (c >= 0) return c;
    return find_element_fallback<WIDEK>(m, x, y, k, nn_id, tri);
}

// ------------------------------------------------------- intersections -------------------
// general_form (src/intersection.jl:11-18) of edge p1->p2, intersection with the track line
// (src/intersection.jl:127-138) and point_in_segment (src/segment.jl:39-44).
// Returns 0 = no valid intersection, 1 = valid (x,y), 2 = parallel.
RT_HD __forceinline__ int edge_hit(double tA, double tB, double tC, double p1x, double p1y, double p2x,
                                        double p2y, double &x, double &y) {
    double eA = p1y - p2y;
    double eB = p2x - p1x;
    double eC = p1x * p2y - p2x * p1y;
    const double nrm = sqrt(eA * eA + eB * eB + eC * eC);
    eA = eA / nrm;
    eB = eB / nrm;
    eC = eC / nrm;
    const double a = tB * eA;
    const double b = eB * tA;
    if (isapprox_s(a, b, kRtolDefault)) return 2;
    const double det = a - b;
    x = (tC * eB - eC * tB) / det;
    y = (tA * eC - eA * tC) / det;
    const double lpx = norm2(p1x - x, p1y - y);
    const double lqx = norm2(p2x - x, p2y - y);
    const double lpq = norm2(p1x - p2x, p1y - p2y);
    return isapprox_s(lpx + lqx, lpq, kRtolDefault) ? 1 : 0;
}

// order_intersection_points (src/intersection.jl:151-159)
RT_HD __forceinline__ bool order_points(double phi, double x1, double y1, double x2, double y2, double &px,
                                             double &py, double &qx, double &qy) {
    const bool first = (phi < kHalfPi) ? (x1 < x2) : (x1 > x2);
    px = first ? x1 : x2;
    py = first ? y1 : y2;
    qx = first ? x2 : x1;
    qy = first ? y2 : y1;
    return first;  // true: (x1,y1) is the entry point
}

// intersections(mesh, cell_id, track) (src/intersection.jl:34-119).  Returns false for the
// branch in which the reference reads an unassigned variable (n_int == 3, all coincident).
// `eq` receives the index (0..2) of the cell edge the exit point q lies on (-1 if q was not
// produced): the walk step uses it to predict the next cell through the adjacency table.
// The three edges' results (edge_hit of edges (n1,n2), (n2,n3), (n3,n1), src/intersection.jl:48-54) combined into the pair
// (p, q): src/intersection.jl:56-119.  Separate from the edge tests so that k_first can evaluate the three edges on three lanes.
RT_HD __forceinline__ bool intersections_combine(int h0, double ex0, double ey0, int h1, double ex1, double ey1, int h2, double ex2,
                                                 double ey2, double phi, double &px, double &py, double &qx, double &qy, int &eq);

RT_HD __forceinline__ bool intersections(const Tri &t, double phi, double tA, double tB,
                                              double tC, double &px, double &py, double &qx, double &qy, int &eq) {
    const double x1 = t.x1, y1 = t.y1, x2 = t.x2, y2 = t.y2, x3 = t.x3, y3 = t.y3;
    double ex0 = 0, ey0 = 0, ex1 = 0, ey1 = 0, ex2 = 0, ey2 = 0;
    const int h0 = edge_hit(tA, tB, tC, x1, y1, x2, y2, ex0, ey0);
    const int h1 = edge_hit(tA, tB, tC, x2, y2, x3, y3, ex1, ey1);
    const int h2 = edge_hit(tA, tB, tC, x3, y3, x1, y1, ex2, ey2);
    return intersections_combine(h0, ex0, ey0, h1, ex1, ey1, h2, ex2, ey2, phi, px, py, qx, qy, eq);
}

RT_HD __forceinline__ bool intersections_combine(int h0, double ex0, double ey0, int h1, double ex1, double ey1, int h2, double ex2,
                                                 double ey2, double phi, double &px, double &py, double &qx, double &qy, int &eq) {
    eq = -1;
    const bool v0 = h0 == 1, v1 = h1 == 1, v2 = h2 == 1;
    const bool parallel_found = (h0 == 2) || (h1 == 2) || (h2 == 2);
    const int n_int = (int)v0 + (int)v1 + (int)v2;
    if (n_int == 3) {
        // farthest pair over (1,2), (1,3), (2,3) with a strict `>` (src/intersection.jl:81-94)
        double l = 0.0, a1x = 0, a1y = 0, a2x = 0, a2y = 0;
        bool have = false;
        int e1 = -1, e2 = -1;
        double li = norm2(ex0 - ex1, ey0 - ey1);
        if (li > l) { a1x = ex0; a1y = ey0; a2x = ex1; a2y = ey1; l = li; have = true; e1 = 0; e2 = 1; }
        li = norm2(ex0 - ex2, ey0 - ey2);
        if (li > l) { a1x = ex0; a1y = ey0; a2x = ex2; a2y = ey2; l = li; have = true; e1 = 0; e2 = 2; }
        li = norm2(ex1 - ex2, ey1 - ey2);
        if (li > l) { a1x = ex1; a1y = ey1; a2x = ex2; a2y = ey2; l = li; have = true; e1 = 1; e2 = 2; }
        if (!have) return false;
        eq = order_points(phi, a1x, a1y, a2x, a2y, px, py, qx, qy) ? e2 : e1;
        return true;
    }
    if (n_int == 2) {
        const double f_x = v0 ? ex0 : ex1, f_y = v0 ? ey0 : ey1;  // first valid edge in edge order
        const double s_x = v2 ? ex2 : ex1, s_y = v2 ? ey2 : ey1;  // second valid edge
        const int fe = v0 ? 0 : 1, se = v2 ? 2 : 1;
        if (!parallel_found && isapprox_v2(f_x, f_y, s_x, s_y)) {
            px = f_x; py = f_y; qx = s_x; qy = s_y;  // unordered; the caller skips it (src/intersection.jl:107-110)
        } else {
            eq = order_points(phi, f_x, f_y, s_x, s_y, px, py, qx, qy) ? se : fe;
        }
        return true;
    }
    px = py = qx = qy = 0.0;  // n_int in {0,1}: the caller moves a tiny step (src/intersection.jl:114-118)
    return true;
}

// The generic step as ONE out-of-line function: its register demand (ring search, k-best list,
// three edge intersections) then lives in the callee's frame instead of inflating the march
// loop, which runs the walk step >98 % of the time.  Returns 0: segment (px..ell, element, eq)
// produced; 1: the reference takes a `continue` branch (advance by tiny_step); 2: locate failed
// (src/track.jl:140-143); 3: undefined intersection (src/intersection.jl:82-94).
struct GenericOut {
    double px, py, qx, qy, ell;
    int32_t element, eq;
};
template <bool WIDEK>
RT_HD __noinline__ int generic_step(const DGeo &m, double xpx, double xpy, int k, int32_t prev_element, double phi,
                                         double tA, double tB, double tC, GenericOut &o) {
    Tri tri;
    const int32_t element = find_element<WIDEK>(m, xpx, xpy, k, tri);  // src/track.jl:122 and :138-139
    o.element = element;
    if (element < 0) return 2;
    if (element == prev_element) return 1;  // :147-150
    int eq;
    if (!intersections(tri, phi, tA, tB, tC, o.px, o.py, o.qx, o.qy, eq)) return 3;  // :153
    o.eq = eq;
    if (isapprox_v2(o.px, o.py, o.qx, o.qy)) return 1;  // :156-159
    o.ell = norm2(o.px - o.qx, o.py - o.qy);            // Segment ctor, src/segment.jl:31-33
    return 0;
}

// Does the reference's iteration at xp end in a tiny step (`continue` at src/track.jl:147-150 or :156-159)?
// False when it emits a segment or fails to locate / intersect.  Used by the cooperative creep of k_march.
template <bool WIDEK>
RT_HD __noinline__ bool generic_tiny_step(const DGeo &m, double xpx, double xpy, int k, int32_t prev_element,
                                               double phi, double tA, double tB, double tC) {
    Tri tri;
    const int32_t element = find_element<WIDEK>(m, xpx, xpy, k, tri);
    if (element < 0) return false;
    if (element == prev_element) return true;
    double px, py, qx, qy;
    int eq;
    if (!intersections(tri, phi, tA, tB, tC, px, py, qx, qy, eq)) return false;
    return isapprox_v2(px, py, qx, qy);
}

// ======================================================================= walk step ========
// After a segment has been emitted in cell T with its exit point q on edge `ko`, the reference
// re-seeds at xp = q + tiny_step·(cos ϕ, sin ϕ) and locates from scratch (src/track.jl:165,
// 122).  The walk step predicts the located cell as T' = adj[T][ko] and PROVES, with
// certificates whose margins are orders of magnitude above the reference's tolerances and
// above FP64 rounding, that the reference's own procedure returns T' (or returns T again and
// takes its `prev_element == element` branch) and that `intersections(T')` yields exactly
// (entry on the shared edge, exit on one other edge).  Everything that reaches the output or
// the next state (λ of T at xp, q', ℓ', xp) is computed with the reference's formulas and
// operation order; the certificates themselves may use any arithmetic.  When a certificate
// fails the lane runs the generic step (find_element + intersections above) for that
// iteration, so results never depend on which path was taken.
//
// Why only T and T' can contain xp (up to the √eps barycentric tolerance): xp lies within
// tiny_step of the interior of the shared edge and, by the isolation certificate, at least
// eps (barycentric) away from the other two edges of T'; eps is computed per record on the host
// (rt_mesh_prep.hpp) by clipping the acceptance region of every nearby cell — the cell scaled by
// 1 + 3·(√eps + rounding noise) about its centroid — against the record's region, so that no cell
// other than T and T' can pass the reference's test at xp.  find_element scans the cell
// lists of nodes in order of distance and returns the first cell that passes; T' always
// passes (its own λ are ≥ -√eps/4 by certificate 2, and cells whose λ are too noisy for that
// statement are never walked into); T passes iff its three exact λ below are within
// [-√eps, 1 + √eps] ("shallow crossing").  T is scanned before T' iff the nearest of
// {a, b, c, c'} is c, or it is a or b and T < T' (lists are ascending in cell id).  `extras`
// bounds how many other nodes can precede, so the scan stays inside the window
// find_element(xp) / find_element(xp, k) covers.
struct Walk {
    int32_t T;       // cell of the last emitted segment (prev_element), -1 at the start
    int32_t pred;    // walk record index of the predicted next cell (3*cell' + entry edge), -1: none
    double ax, ay, bx, by;  // endpoints of T's exit edge, in T's edge orientation
    double cx, cy;          // T's vertex opposite the exit edge
    double dT;              // det of T's barycentric system in T's node order (exact reference ops)
    int32_t last;           // 3·T + exit edge (T's own edge numbering) of the last emitted segment: its staging code (k_march<TOPO>)
};

// State for the next walk step after a segment was emitted by the generic step in `cell` with
// its exit point on edge `ko` (0..2).
RT_HD __forceinline__ void walk_enter(const DMesh &m, const Tri &t, Walk &w, int32_t cell, int ko) {
    const double x1 = t.x1, y1 = t.y1, x2 = t.x2, y2 = t.y2, x3 = t.x3, y3 = t.y3;
    w.T = cell;
    // det of the barycentric system in the cell's node order, with the reference's operations (src/mesh.jl:166-168): the
    // same expression the host evaluated for the walk records, hence the same bits — no load
    w.dT = x1 * (y2 - y3) + y1 * (x3 - x2) + (x2 * y3 - y2 * x3);
    w.ax = ko == 0 ? x1 : (ko == 1 ? x2 : x3);
    w.ay = ko == 0 ? y1 : (ko == 1 ? y2 : y3);
    w.bx = ko == 0 ? x2 : (ko == 1 ? x3 : x1);
    w.by = ko == 0 ? y2 : (ko == 1 ? y3 : y1);
    w.cx = ko == 0 ? x3 : (ko == 1 ? x1 : x2);
    w.cy = ko == 0 ? y3 : (ko == 1 ? y1 : y2);
    const int32_t a = ko == 0 ? t.adj[0] : (ko == 1 ? t.adj[1] : t.adj[2]);
    w.pred = a != -2 ? a : m.adjr[3 * cell + ko];
    w.last = 3 * cell + ko;
}

enum WalkResult { kWalkGeneric = 0, kWalkSkip = 1, kWalkEmit = 2 };

// Register copy of the walk record the lane needs next.
struct NextRec {
    uint64_t hdr;
    double dT, x2, y2, e1A, e1B, e1C, e2A, e2B, e2C;
};
RT_HD __forceinline__ void load_next(const DMesh &m, int32_t pred, NextRec &r) {
    const RT_G WalkRec *R = m.wrec + (pred >= 0 ? pred : 0);
    r.hdr = R->hdr; r.dT = R->dT;
    r.x2 = R->x2; r.y2 = R->y2;
    r.e1A = R->e1A; r.e1B = R->e1B; r.e1C = R->e1C; r.e2A = R->e2A; r.e2B = R->e2B; r.e2C = R->e2C;
}
RT_HD __forceinline__ int32_t rec_next1(uint64_t hdr) { return (int32_t)(hdr & ((1u << kWalkIdBits) - 1)) - 1; }
RT_HD __forceinline__ int32_t rec_next2(uint64_t hdr) { return (int32_t)((hdr >> kWalkIdBits) & ((1u << kWalkIdBits) - 1)) - 1; }
RT_HD __forceinline__ int32_t rec_extras(uint64_t hdr) { return (int32_t)(hdr >> (2 * kWalkIdBits)) & 15; }
// the record's isolation margin 2^(code - 20), assembled as a double: exponent field 1023 - 20 + code
RT_HD __forceinline__ double rec_eps(uint64_t hdr) {
    const uint32_t hi = (((uint32_t)(hdr >> 32) >> 26) & 31u) + 1003u;
    return __builtin_bit_cast(double, (uint64_t)hi << 52);
}
RT_HD __forceinline__ bool rec_same(uint64_t hdr) { return (hdr >> 63) != 0; }

// One walk step at xp for the lane's predicted record.  On kWalkEmit: (qx,qy) is the exit point,
// `ell` the segment length (entry point = previous exit point, bit-identical by symmetry of the
// edge's general form), and `w` is advanced to the new cell.  On kWalkSkip the reference takes
// its `prev_element == element` branch (src/track.jl:147-150).  kWalkGeneric: no decision.
// `kk` = min(max(k, 2), 14): the node window of find_element(xp) followed by find_element(xp, k);
// `phi` = ϕ, for order_intersection_points (src/intersection.jl:151-159).
// Written straight-line (all certificates are folded into one predicate; a lane without a
// prediction reads record 0 and is masked out) except for the rare exact shallow-crossing test:
// on a 64-wide wave, selects are cheaper than divergent early exits.
// Is fl(num / d) inside [0 - √eps, 1 + √eps] (src/mesh.jl:171-174)?  The quotient is within one rounding of num/d, so
// away from the two thresholds the answer follows from products and compares; next to a threshold (a band of relative
// width 1e-12 below -√eps; anything above 1) the division itself decides — practically never, in a cold branch.
// r = num·sign(d), ad = |d|: num/d = r/ad bit for bit.
constexpr double kLamLoOut = -kRtolDefault * (1.0 + 1e-12);  // r < kLamLoOut·ad  =>  fl(r/ad) < -√eps
constexpr double kLamLoIn = -kRtolDefault * (1.0 - 1e-12);   // kLamLoIn·ad <= r <= ad  =>  inside
RT_HD __forceinline__ bool lambda_exact(double num, double d) {
    const double l = num / d;
    return (0.0 - kRtolDefault) <= l && l <= (1.0 + kRtolDefault);
}
// λ expected well inside (0, 1): inside for sure when 0 <= r <= ad
RT_HD __forceinline__ bool lambda_inner_in_range(double num, double d, double ad) {
    const double r = d > 0 ? num : -num;
    bool in = r >= 0.0 && r <= ad && ad > 0.0;
    if (__builtin_expect(!in, 0)) {
        asm volatile("" ::: "memory");  // a real, cold branch
        in = lambda_exact(num, d);
    }
    return in;
}

// The shallow crossing: T still passes the reference's barycentric test at xp iff all three λ, exactly as
// point_in_triangle evaluates them (src/mesh.jl:166-174; a, b, c are a cyclic rotation of the cell's nodes, which maps
// the three closed forms onto each other), lie in [0 - √eps, 1 + √eps]; rT / adT: the first one's numerator and
// denominator, known not to be surely below -√eps.  Then the scan order of find_element: nearest of {a, b, c, c'}; T is
// met before T' iff that is c, or it is a or b and T < T' (node -> cells lists ascend in cell id).
RT_HD __forceinline__ bool shallow_T_first(const Walk &w, double numT, double rT, double adT, double x2, double y2, int32_t Tn,
                                           double xpx, double xpy, bool &tie) {
    bool pass = rT >= kLamLoIn * adT && rT <= adT && adT > 0.0;
    if (__builtin_expect(!pass, 0)) {
        asm volatile("" ::: "memory");  // a real, cold branch
        pass = lambda_exact(numT, w.dT);
    }
    const double numA = (w.by - w.cy) * xpx + (w.cx - w.bx) * xpy + (w.bx * w.cy - w.cx * w.by);
    const double numB = (w.cy - w.ay) * xpx + (w.ax - w.cx) * xpy + (w.cx * w.ay - w.ax * w.cy);
    pass = pass && lambda_inner_in_range(numA, w.dT, adT) && lambda_inner_in_range(numB, w.dT, adT);
    const double da = (xpx - w.ax) * (xpx - w.ax) + (xpy - w.ay) * (xpy - w.ay);
    const double db = (xpx - w.bx) * (xpx - w.bx) + (xpy - w.by) * (xpy - w.by);
    const double dc = (xpx - w.cx) * (xpx - w.cx) + (xpy - w.cy) * (xpy - w.cy);
    const double dcp = (xpx - x2) * (xpx - x2) + (xpy - y2) * (xpy - y2);
    const double dab = da < db ? da : db;
    tie = pass && (dc == dab || dcp == dab || dc == dcp);  // exactly equidistant nodes: generic step
    return pass && ((dc < dab && dc < dcp) || (dab < dc && dab < dcp && w.T < Tn));
}

RT_HD __forceinline__ int walk_step(const DMesh &m, Walk &w, const NextRec &nr, int kk, double phi, double tA, double tB,
                                         double tC, double xpx, double xpy, double ppx, double ppy, double &qx,
                                         double &qy, double &ell) {
    const bool has = m.walk_ok && w.pred >= 0;
    const int32_t n1 = rec_next1(nr.hdr), n2 = rec_next2(nr.hdr), Tn = (int32_t)((uint32_t)(w.pred >= 0 ? w.pred : 0) / 3u);
    const double dTn = nr.dT;
    // the entry edge (v0, v1) of T' is the exit edge (a, b) of T, in T''s orientation
    const bool same = rec_same(nr.hdr);
    const double x0 = same ? w.ax : w.bx, y0 = same ? w.ay : w.by, x1 = same ? w.bx : w.ax, y1 = same ? w.by : w.ay;
    const double x2 = nr.x2, y2 = nr.y2;
    const double e1A = nr.e1A, e1B = nr.e1B, e1C = nr.e1C, e2A = nr.e2A, e2B = nr.e2B, e2C = nr.e2C;
    bool ok = has && rec_extras(nr.hdr) <= kk;
    // --- certificate 1: the track line clears every vertex of T' by d_vertex and crosses the entry edge
    //     (certificates may use any arithmetic — only what reaches an output follows the reference's operation order —
    //      so these use fused multiply-adds: half the instructions)
    const double s0 = __builtin_fma(tA, x0, __builtin_fma(tB, y0, tC)), s1 = __builtin_fma(tA, x1, __builtin_fma(tB, y1, tC)),
                 s2 = __builtin_fma(tA, x2, __builtin_fma(tB, y2, tC));
    ok = ok && fabs(s0) >= m.d_vertex && fabs(s1) >= m.d_vertex && fabs(s2) >= m.d_vertex;
    const bool p0 = s0 > 0, p1 = s1 > 0, p2 = s2 > 0;
    ok = ok && (p0 != p1);
    const bool exit1 = p1 != p2;  // the line leaves through rotated edge 1 = (v1,v2), else edge 2 = (v2,v0)
    // --- certificate 2: xp is inside T', at least the record's eps (barycentric) from edges 1 and 2
    const double ex = x1 - x0, ey = y1 - y0, fx = x2 - x0, fy = y2 - y0, gx = xpx - x0, gy = xpy - y0;
    const double area2 = __builtin_fma(ex, fy, -(fx * ey));
    const double aa = fabs(area2);
    // signed areas of (entry edge, xp), (edge 1, xp), (edge 2, xp), in units of the cell's signed double area
    const double d0 = __builtin_fma(ex, gy, -(ey * gx));          // ~ distance from the entry edge
    const double d2 = __builtin_fma(gx, fy, -(gy * fx));          // from edge 2 = (v2, v0)
    const double d1 = area2 - d0 - d2;                            // from edge 1 = (v1, v2): the three sum to area2
    const bool pos = area2 > 0;
    const double c0 = pos ? d0 : -d0, c1 = pos ? d1 : -d1, c2 = pos ? d2 : -d2;
    const double eps_aa = rec_eps(nr.hdr) * aa;
    ok = ok && c0 >= -0.25 * kRtolDefault * aa && c1 >= eps_aa && c2 >= eps_aa;
    // --- exit point: intersection(track.ABC, ABC) — src/intersection.jl:127-138
    const double eA = exit1 ? e1A : e2A, eB = exit1 ? e1B : e2B, eC = exit1 ? e1C : e2C;
    const double a = tB * eA;
    const double b = eB * tA;
    const double det = a - b;
    qx = (tC * eB - eC * tB) / det;
    qy = (tA * eC - eA * tC) / det;
    ell = norm2(ppx - qx, ppy - qy);  // Segment ctor, src/segment.jl:31-33
    ok = ok && ell >= m.l_min;
    // order_intersection_points (src/intersection.jl:151-159) compares the two x coordinates: the entry point stays
    // the entry point only if it is strictly on the expected side (near ϕ = π/2 rounding may decide otherwise)
    ok = ok && (phi < kHalfPi ? ppx < qx : ppx > qx);
    // --- shallow crossing: does T still pass the reference's barycentric test at xp?  λ of T for the
    //     vertex opposite its exit edge (src/mesh.jl:166-168) is below -√eps for all but the flattest ≈3 % of
    //     crossings; for those the other two λ and the scan order of find_element decide
    const double numT = (w.ay - w.by) * xpx + (w.bx - w.ax) * xpy + (w.ax * w.by - w.bx * w.ay);
    const double adT = fabs(w.dT), rT = w.dT > 0 ? numT : -numT;
    int res = ok ? kWalkEmit : kWalkGeneric;
    if (ok && !(rT < kLamLoOut * adT)) {
        asm volatile("" ::: "memory");  // a real branch
        bool tie;
        const bool t_first = shallow_T_first(w, numT, rT, adT, x2, y2, Tn, xpx, xpy, tie);
        res = tie ? kWalkGeneric : (t_first ? kWalkSkip : kWalkEmit);
    }
    // --- advance the state to T' (only when emitting)
    if (res == kWalkEmit) {
        asm volatile("" ::: "memory");  // keep this a branch: as selects the update is twice the instructions
        {   // the exit edge in T''s own numbering: the record is (T', entry edge e), rotated edge 1 / 2 = edge e + 1 / e + 2
            const int32_t e = (w.pred >= 0 ? w.pred : 0) - 3 * Tn;
            int32_t ko = e + (exit1 ? 1 : 2);
            ko = ko >= 3 ? ko - 3 : ko;
            w.last = 3 * Tn + ko;
        }
        w.T = Tn;
        w.dT = dTn;
        w.ax = exit1 ? x1 : x2; w.ay = exit1 ? y1 : y2;
        w.bx = exit1 ? x2 : x0; w.by = exit1 ? y2 : y0;
        w.cx = exit1 ? x0 : x1; w.cy = exit1 ? y0 : y1;
        w.pred = exit1 ? n1 : n2;
    }
    return res;
}


// After walk_step returned kWalkSkip at the previous xp: would it return kWalkSkip again at this xp
// (same T, same predicted T')?  Only what depends on xp is re-evaluated — certificate 2, the exact
// shallow-crossing λ and the scan-order rule, with the expressions of walk_step; the track line,
// the exit point and ℓ do not change while the lane creeps by tiny_step.  A track that crosses an
// edge at a very small angle takes hundreds of such steps (src/track.jl:147-150); this keeps each
// of them to a few dozen instructions instead of a full march iteration.
RT_HD __forceinline__ bool walk_still_skip(const DMesh &m, const Walk &w, const NextRec &nr, double xpx, double xpy) {
    const bool same = rec_same(nr.hdr);
    const double x0 = same ? w.ax : w.bx, y0 = same ? w.ay : w.by, x1 = same ? w.bx : w.ax, y1 = same ? w.by : w.ay;
    const double x2 = nr.x2, y2 = nr.y2;
    const double area2 = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    const double sg = area2 > 0 ? 1.0 : -1.0;
    const double aa = fabs(area2);
    const double c0 = sg * ((x1 - x0) * (xpy - y0) - (y1 - y0) * (xpx - x0));
    const double c1 = sg * ((x2 - x1) * (xpy - y1) - (y2 - y1) * (xpx - x1));
    const double c2 = sg * ((x0 - x2) * (xpy - y2) - (y0 - y2) * (xpx - x2));
    const double eps_aa = rec_eps(nr.hdr) * aa;
    if (!(c0 >= -0.25 * kRtolDefault * aa && c1 >= eps_aa && c2 >= eps_aa)) return false;
    const double numT = (w.ay - w.by) * xpx + (w.bx - w.ax) * xpy + (w.ax * w.by - w.bx * w.ay);
    const double adT = fabs(w.dT), rT = w.dT > 0 ? numT : -numT;
    if (rT < kLamLoOut * adT) return false;
    bool tie;
    const bool t_first = shallow_T_first(w, numT, rT, adT, x2, y2, (int32_t)((uint32_t)(w.pred >= 0 ? w.pred : 0) / 3u), xpx, xpy, tie);
    return !tie && t_first;
}

// ------------------------------------------------------------------------- cheap step -
// The cheap step (k_march<..., TOPO>) splits a walk step into the DECISION — which cell the
// reference emits next, entered and left through which edges — and the ARITHMETIC of the record (exit point,
// length), which the march evaluates one record behind, off the dependent chain, with the reference's formulas.  The decision
// needs no point at all: with s_i the signed distances of the predicted cell's vertices from the track line,
//   * the line leaves T' through the edge whose end points lie on opposite sides (|s_i| >= d_vertex: certificate 1
//     of walk_step, unchanged);
//   * every position xp the reference visits before it emits in T' — q + tiny·d and further tiny steps while it
//     still locates T (src/track.jl:147-150) — stays where only T and T' can pass the reference's test, T' does
//     pass, and `inboundary` is false.  Sufficient, in terms of m = min(|s0|, |s1|) and D = |s0| + |s1| (the entry
//     point has barycentric coordinates (|s1|, |s0|, 0) / D in T'):  m >= E·D + g1,  D >= k2,  D·c1 >= dtf
//     (rt_mesh_prep.hpp derives the per-record constants E, g1, k2, dtf, lc);
//   * the record is (previous exit point, exit point on the predicted edge): the chord is >= lc-guarded above l_min
//     and above the rounding that could flip order_intersection_points (src/intersection.jl:151-159).
// Whether the reference took 0 or 300 tiny steps before emitting does not reach the output, so unlike walk_step
// the cheap step does not decide skip/emit per position; it bounds the number of those steps (kTopoKcap) and adds
// the bound to the iteration counter — a track whose bound reaches the iteration cap is marched again with exact
// steps only.  A lane whose cheap step refuses takes the exact step (walk_step / generic) for that record:
// results never depend on which path was taken.
struct TopoTrack {   // per-track constants
    double dv;       // vertex clearance
    double c1;       // (kTopoKcap - 2) · tiny / √eps
    float c2;        // √eps / tiny · 1.01: tiny steps per unit of dtf / D
    double lcf;      // max(1, topo_rmax / |cos ϕ|)
    bool end_v, end_h;  // an exit through an edge on a vertical / horizontal border ends the track for sure
    bool on;         // the cheap step may be used for this track
};
struct TopoState {
    int32_t pred;    // record of the predicted cell (3·T' + entry edge), -1: none
    int32_t last;    // 3·T + exit edge of the last emitted record
    // signed distances from the track line of the end points of T's exit edge (= T''s entry edge), kept BY SIDE: sp of the end point
    // on the positive side, sn of the other one (the line crosses the edge: sp > 0 >= sn).  The predicted cell's third vertex
    // replaces the end point on its own side — the line leaves through the edge of the two that are then on opposite sides — so a
    // step moves one value instead of choosing four (round 4: the selects by edge orientation were a fifth of the loop's vector
    // instructions); which of the two is the edge's first end point in T's orientation is one bit.
    double sp, sn;
    bool apos;       // the edge's first end point (a) is the one on the positive side
};
enum TopoResult { kTopoFull = 0, kTopoEmit = 1, kTopoEnd = 2 };

RT_HD __forceinline__ double bf16_lo(uint32_t w) { return (double)__builtin_bit_cast(float, w << 16); }
RT_HD __forceinline__ double bf16_hi(uint32_t w) { return (double)__builtin_bit_cast(float, w & 0xffff0000u); }

// intersection(track.ABC, edge.ABC) — src/intersection.jl:127-138 (the expression walk_step uses for its exit point)
RT_HD __forceinline__ void edge_exit_point(double tA, double tB, double tC, double eA, double eB, double eC, double &qx, double &qy) {
    const double a = tB * eA;
    const double b = eB * tA;
    const double det = a - b;
    qx = (tC * eB - eC * tB) / det;
    qy = (tA * eC - eA * tC) / det;
}

// The cheap step in three pieces (k_march issues the loads of the NEXT record between the first and the second):
// topo_geo — which edge the line leaves T' through and what comes behind it; topo_certified — the certificates;
// topo_commit — advance the state.  topo_step is the three in a row (tests/host_march.hip).
struct TopoGeo {
    double s2;          // signed distance of v2 (the vertex opposite the entry edge) from the track line
    bool p2;            // v2 lies on the positive side
    bool pv1;           // v1 (the entry edge's second vertex in T''s orientation) lies on the positive side
    bool exit1;         // the line leaves through rotated edge 1 = (v1, v2), else edge 2 = (v2, v0)
    uint32_t nx;        // successor field behind the exit edge: record + 1, 0, kTopoEndV / kTopoEndH
    int32_t code;       // 3·T' + exit edge (in T''s own edge numbering)
    int32_t cell;       // T'
};
RT_HD __forceinline__ TopoGeo topo_geo(const TopoState &ts, uint64_t hdr, double x2, double y2, double tA, double tB, double tC) {
    TopoGeo g;
    g.s2 = __builtin_fma(tA, x2, __builtin_fma(tB, y2, tC));
    g.p2 = g.s2 > 0;
    g.pv1 = rec_same(hdr) != ts.apos;  // (v0, v1) = (a, b) if the record's entry edge runs as T's exit edge does, else (b, a)
    g.exit1 = g.pv1 != g.p2;           // v1 and v2 on opposite sides
    g.nx = g.exit1 ? (uint32_t)(hdr & ((1u << kWalkIdBits) - 1)) : (uint32_t)((hdr >> kWalkIdBits) & ((1u << kWalkIdBits) - 1));
    const int32_t pr = ts.pred >= 0 ? ts.pred : 0;
    const int32_t Tn = (int32_t)((uint32_t)pr / 3u), e = pr - 3 * Tn;
    int ko = e + (g.exit1 ? 1 : 2);
    ko = ko >= 3 ? ko - 3 : ko;
    g.code = 3 * Tn + ko;
    g.cell = Tn;
    return g;
}
// successor record behind the exit edge, -1: none (boundary, or no certified record)
RT_HD __forceinline__ int32_t topo_next(const TopoGeo &g) { return (g.nx - 1u) < (kTopoEndH - 1u) ? (int32_t)g.nx - 1 : -1; }
// `kub`: upper bound of the reference's iterations for this record — 1 + at most √eps·dtf / (tiny·D) + 2 tiny steps (<= kTopoKcap)
RT_HD __forceinline__ bool topo_certified(const TopoTrack &tt, const TopoState &ts, const TopoGeo &g, uint64_t hdr, uint32_t c01,
                                          uint32_t c23, int kk, int32_t &kub) {
    const double ap = fabs(ts.sp), an = fabs(ts.sn), a2 = fabs(g.s2);
    const double D = ap + an, m = ap < an ? ap : an;
    const double sv = g.p2 ? an : ap, Dx = a2 + sv;  // (the end point that stays: the one across the line from v2)
    const double g1 = bf16_lo(c01), k2 = bf16_hi(c01), dtf = bf16_lo(c23), lc = bf16_hi(c23);
    // (evaluated without short-circuits: a branch here would also pull the record's second load behind the first compare)
    const int ok = (int)(ts.pred >= 0) & (int)(rec_extras(hdr) <= kk) & (int)(a2 >= tt.dv) & (int)((ts.sp > 0) != (ts.sn > 0)) &
                   (int)(m >= __builtin_fma(rec_eps(hdr), D, g1)) & (int)(D >= k2) & (int)(Dx >= k2) & (int)(D * tt.c1 >= dtf) &
                   (int)(sv >= lc * tt.lcf);
#if defined(__HIP_DEVICE_COMPILE__)
    float kf = (float)dtf * tt.c2 * __builtin_amdgcn_rcpf((float)D) * 1.001f;
#else
    float kf = (float)dtf * tt.c2 / (float)D * 1.001f;
#endif
    kf = kf < (float)kTopoKcap ? kf : (float)kTopoKcap;  // (D·c1 >= dtf: never more than kTopoKcap; also absorbs a NaN)
    kub = (int32_t)kf + 4;
    return ok != 0;
}
// The nine terms of topo_certified as a bit mask of the ones that FAIL (bit order = order in the expression above):
// 0 no predicted record, 1 scan window (extras > k), 2 |s2| < d_vertex, 3 entry edge not crossed, 4 m < E·D + g1,
// 5 D < k2, 6 Dx < k2, 7 D·c1 < dtf, 8 |s_v| < lc·lcf.  Statistics only (cold path of k_march, tests).
RT_HD __forceinline__ uint32_t topo_refusal_terms(const TopoTrack &tt, const TopoState &ts, const TopoGeo &g, uint64_t hdr, uint32_t c01,
                                                  uint32_t c23, int kk) {
    const double ap = fabs(ts.sp), an = fabs(ts.sn), a2 = fabs(g.s2);
    const double D = ap + an, m = ap < an ? ap : an;
    const double sv = g.p2 ? an : ap, Dx = a2 + sv;
    const double g1 = bf16_lo(c01), k2 = bf16_hi(c01), dtf = bf16_lo(c23), lc = bf16_hi(c23);
    return (uint32_t)!(ts.pred >= 0) | (uint32_t)!(rec_extras(hdr) <= kk) << 1 | (uint32_t)!(a2 >= tt.dv) << 2 |
           (uint32_t)!((ts.sp > 0) != (ts.sn > 0)) << 3 | (uint32_t)!(m >= __builtin_fma(rec_eps(hdr), D, g1)) << 4 |
           (uint32_t)!(D >= k2) << 5 | (uint32_t)!(Dx >= k2) << 6 | (uint32_t)!(D * tt.c1 >= dtf) << 7 |
           (uint32_t)!(sv >= lc * tt.lcf) << 8;
}
// kTopoEmit: on to the successor (pred = -1 when there is none with a certificate: exact steps from `last`);
// kTopoEnd: the exit edge lies on the border and the track ends for sure after this record
// The exit edge's end points by side: v2 replaces the end point on its own side.  (The edge's first end point in T''s orientation is
// v1 if the line leaves through (v1, v2) and v2 otherwise — v2 then lies where v1 does — so the orientation bit is `pv1` either
// way.)  k_march calls this for every lane, committed or not: a lane that does not commit leaves the cheap loop and comes back
// through topo_enter.
RT_HD __forceinline__ void topo_advance(TopoState &ts, const TopoGeo &g) {
    ts.sp = g.p2 ? g.s2 : ts.sp;
    ts.sn = g.p2 ? ts.sn : g.s2;
    ts.apos = g.pv1;
}
RT_HD __forceinline__ int topo_commit(const TopoTrack &tt, TopoState &ts, const TopoGeo &g) {
    ts.last = g.code;
    ts.pred = topo_next(g);
    return ((g.nx == kTopoEndV && tt.end_v) || (g.nx == kTopoEndH && tt.end_h)) ? kTopoEnd : kTopoEmit;
}
// One cheap step.  kTopoEmit / kTopoEnd: `code` = 3·T' + exit edge of the emitted record, `kub` bounds the reference's
// iterations for it, the state is advanced.  kTopoFull: nothing changed.
RT_HD __forceinline__ int topo_step(const TopoTrack &tt, TopoState &ts, uint64_t hdr, double x2, double y2, uint32_t c01,
                                    uint32_t c23, int kk, double tA, double tB, double tC, int32_t &code, int32_t &kub) {
    const TopoGeo g = topo_geo(ts, hdr, x2, y2, tA, tB, tC);
    if (!topo_certified(tt, ts, g, hdr, c01, c23, kk, kub)) return kTopoFull;
    code = g.code;
    topo_advance(ts, g);
    return topo_commit(tt, ts, g);
}

// Can the lane take cheap steps after an exact step left it in cell wk.T with the prediction wk.pred?
RT_HD __forceinline__ bool topo_enter(const DMesh &m, const TopoTrack &tt, const Walk &wk, double tA, double tB, double tC, TopoState &ts) {
    if (!(tt.on && wk.pred >= 0)) return false;
    const double sa = __builtin_fma(tA, wk.ax, __builtin_fma(tB, wk.ay, tC)), sb = __builtin_fma(tA, wk.bx, __builtin_fma(tB, wk.by, tC));
    if (!(fabs(sa) >= tt.dv && fabs(sb) >= tt.dv && ((sa > 0) != (sb > 0)))) return false;
    ts.pred = wk.pred; ts.apos = sa > 0; ts.sp = ts.apos ? sa : sb; ts.sn = ts.apos ? sb : sa;
    ts.last = m.adjr[wk.pred];  // the record reached back across T''s entry edge: 3·T + exit edge
    return true;
}

// The exact step's state after cheap steps: the last record was emitted in cell last / 3 with exit edge last % 3.
RT_HD __forceinline__ void topo_materialize(const DMesh &m, const DGeo &g, int32_t last, double tA, double tB, double tC, Walk &wk,
                                            double &lqx, double &lqy) {
    const int32_t cell = (int32_t)((uint32_t)last / 3u), ko = last - 3 * cell;
    walk_enter(m, load_tri(g, cell), wk, cell, ko);
    const RT_G EdgeABC *e = m.etab + last;
    edge_exit_point(tA, tB, tC, e->A, e->B, e->C, lqx, lqy);
}

RT_HD __forceinline__ TopoTrack topo_track(bool mesh_on, double d_vertex, double tiny_max, double rmax, double end_err, double tiny,
                                           double cs, double sn) {
    TopoTrack tt;
    tt.dv = d_vertex;
    tt.c1 = (double)(kTopoKcap - 2) * tiny / kRtolDefault;
    tt.c2 = (float)(1.01 * kRtolDefault / tiny);
    const double ac = fabs(cs), as = fabs(sn);
    tt.lcf = ac * 1.0 >= rmax ? 1.0 : rmax / ac;
    tt.end_v = tiny * (1.0 - ac) >= 2.0 * end_err + 1e-300;
    tt.end_h = tiny * (1.0 - as) >= 2.0 * end_err + 1e-300;
    tt.on = mesh_on && tiny > 0.0 && tiny <= tiny_max && ac > 0.0 && tt.lcf < 1e6;
    return tt;
}

// ------------------------------------------------------------------ the Σℓ check -
// The Σℓ check `isapprox(track.ℓ, sum(ℓ.(segments)); rtol)` (src/track.jl:171) is decided here (and in the CPU checker) with a
// left-to-right sum; Julia's `sum` reassociates (pairwise blocks, @simd lanes), so its Σℓ can differ by a few ulp·n.  A
// track whose |ℓ − Σℓ| lies within 64·ulp·n·max(ℓ, Σℓ) of the threshold rtol·max(ℓ, Σℓ) could get the other status there:
// such tracks are counted (rt_last_stats) so that a caller knows when this cannot be pinned.
// `abs_band`: an absolute widening of the band (the Σℓ chain below: n · a few ulp of the largest coordinate).
RT_HD __forceinline__ bool sum_check_is_marginal(double ell, double sum, double rtol, int n, double band = 64.0, double abs_band = 0.0) {
    const double big = fabs(ell) > fabs(sum) ? fabs(ell) : fabs(sum);
    return fabs(fabs(ell - sum) - rtol * big) <= band * 1.1102230246251565e-16 * (double)(n > 1 ? n : 1) * big + abs_band;
}

// Σℓ of a track WITHOUT adding up its records (k_materialise_lin, round 5): the records of a track lie head to tail on its line —
// p of a record IS the q before it, bit for bit — so Σ‖p_i − q_i‖ = ‖p_0 − q_0‖ + (q_last − q_0)·d, d = (cos ϕ, sin ϕ) the march's
// direction, up to the roundings of n norms and their sum.  Only a record that keeps its OWN p (a generic step's, behind tiny steps)
// breaks the chain; `chain_gap_term` is what such a record i contributes to the correction `gap` that `chain_sum` subtracts:
//   * the gap in front of it, (p_i − b_i)·d (b_i = the exit point of the record before; a record that begins BEHIND that point —
//     cells that overlap within the locate's tolerance — counts negative: its overlap is IN Σℓ),
//   * and minus twice its own length if its two points are in the wrong order along d (order_intersection_points compares x
//     coordinates, src/intersection.jl:151-159, which near ϕ = π/2 are equal to the last bit): such a record walks backwards, the
//     projection subtracts its length where Σℓ adds it.
// `chain_status` decides like the kernel: 2 = inside the band in which a sum in another order (or this chain's own error: the
// exit points lie off the line by a few ulp of the coordinates, which a near-zero-length record turns into a first-order term —
// `coord_max`) could decide the other way: k_finish sums those left to right; 1 = LENGTH_MISMATCH for sure; 0 = OK for sure.
// Host twin: tests/host_march.hip drives these on the checker's records (tests/test_sum_chain_cpu.py, tools/fuzz_cpu.py).
RT_HD __forceinline__ double chain_gap_term(double px, double py, double qx, double qy, double ell, double bx, double by, double dx, double dy) {
    // The gap's PROJECTION on d, not its length: the span (q_last − q_first)·d that chain_sum starts from contains exactly this
    // projection.  (Round 5 took ±‖p − b‖.  The two differ when the gap is not along the line: the exit points of a SHALLOW crossing
    // — an edge within 1e-7 … 1e-3 rad of the track — lie off the line by u·R / sin, 1e-10 on a mesh 13 units from the origin, and a gap
    // of 4.5e-10 between two such points was 1.5e-11 longer than its projection: outside the band, found by the host twin of this
    // function at a tolerance tuned to the track, tools/fuzz_cpu.py seed 710227.)
    double acc = (px - bx) * dx + (py - by) * dy;
    if ((qx - px) * dx + (qy - py) * dy < 0.0) acc -= 2.0 * ell;
    return acc;
}
RT_HD __forceinline__ double chain_sum(double fpx, double fpy, double fqx, double fqy, double lqx, double lqy, double dx, double dy, double gap, int cnt) {
    double S = cnt > 0 ? norm2(fpx - fqx, fpy - fqy) : 0.0;
    if (cnt > 1) S += ((lqx - fqx) * dx + (lqy - fqy) * dy) - gap;
    return S;
}
constexpr double kChainBands = 96.0;      // any-order sum against the left-to-right one: n·2^-53·Σ each; 96 hold the statistic's 64
constexpr double kChainCoordUlps = 64.0;  // ... + n · this many ulp of the largest coordinate
RT_HD __forceinline__ int chain_status(double L, double S, double rtol, int cnt, double coord_max) {
    const double abs_band = kChainCoordUlps * 2.220446049250313e-16 * coord_max * (double)(cnt > 1 ? cnt : 1);
    if (sum_check_is_marginal(L, S, rtol, cnt, kChainBands, abs_band)) return 2;
    return isapprox_s(L, S, rtol) ? 0 : 1;  // src/track.jl:171-175
}

// --------------------------------------------------------------------- transport sweep -
// 1 − e^{−τ} for τ >= 0 (rt_sweep: the attenuation factor of a segment), accurate to a few ulp over the whole range —
// also where e^{−τ} ≈ 1 and 1 − e^{−τ} would cancel.  The device library's expm1 serves every argument (overflow,
// NaN, huge negatives) with quarter-rate instructions (v_rndne_f64, v_cvt_i32_f64, two v_ldexp_f64): ≈150 issue cycles per
// wave where this takes ≈95, and the sweep is bound by instruction issue.  x = −τ = n·ln2 + r, |r| <= ln2/2:
// n by the 1.5·2^52 trick (two additions), r with ln2 split in two, expm1(r) by its Taylor polynomial to r^13 (the next term is
// below 4e-18·|r|), 2^n assembled from its exponent bits, and e^x − 1 = 2^n·expm1(r) + (2^n − 1) in one fma (2^n − 1 is exact).
// Beyond τ = 41.5, e^{−τ} < 2^−59 and the result is 1.
// The six highest coefficients of the polynomial below (1/13! ... 1/8!) can be handed in: k_sweep keeps them in vector
// registers, where its scalar file is short (13 coefficients = 26 scalar registers; with them 23 scalar values of its loop were
// spilled to vector lanes, −2 % sweep time without).
struct ExpPoly { double c[6]; };
RT_HD __forceinline__ ExpPoly exp_poly() {
    return ExpPoly{{1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06,
                    2.48015873015873e-05}};
}
RT_HD __forceinline__ double one_minus_exp_neg(double tau, const ExpPoly &hi = exp_poly()) {
    const double x = -__builtin_fmin(tau, 41.5);  // (one v_min_f64; τ is never NaN)
    const double kMagic = 6755399441055744.0;  // 1.5 · 2^52: adding it rounds to an integer, whose low word is that integer
    const double t = __builtin_fma(x, 1.4426950408889634074, kMagic);
    const double n = t - kMagic;
    const int32_t ni = (int32_t)(uint32_t)__builtin_bit_cast(uint64_t, t);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);  // ln2 = hi + lo, hi with 21 trailing zero bits: n·hi is exact
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double q = hi.c[0];                               // 1/13!
    q = __builtin_fma(q, r, hi.c[1]);                 // 1/12!
    q = __builtin_fma(q, r, hi.c[2]);                 // 1/11!
    q = __builtin_fma(q, r, hi.c[3]);                 // 1/10!
    q = __builtin_fma(q, r, hi.c[4]);                 // 1/9!
    q = __builtin_fma(q, r, hi.c[5]);                 // 1/8!
    q = __builtin_fma(q, r, 1.984126984126984e-04);   // 1/7!
    q = __builtin_fma(q, r, 1.3888888888888889e-03);  // 1/6!
    q = __builtin_fma(q, r, 8.333333333333333e-03);   // 1/5!
    q = __builtin_fma(q, r, 4.1666666666666664e-02);  // 1/4!
    q = __builtin_fma(q, r, 1.6666666666666666e-01);  // 1/3!
    q = __builtin_fma(q, r, 0.5);                     // 1/2!
    const double p = __builtin_fma(r * r, q, r);      // expm1(r)
    const double s2 = __builtin_bit_cast(double, (uint64_t)(uint32_t)(ni + 1023) << 52);  // 2^n, n in [-60, 0]
    return -__builtin_fma(s2, p, s2 - 1.0);
}
// The same for an optically THIN segment, 0 <= τ < 1/8 (kThinTau) — the usual case on a mesh fine enough for flat sources: no
// range reduction, −expm1(−τ) = τ·(1 + x/2! + … + x^9/10!) with x = −τ; the first term left out, x^10/11!, is below 2.4e-17 of the sum.
// 10 instructions where the general form takes 24 (the sweep is bound by instruction issue).  Same coefficients as above.
constexpr double kThinTau = 0.125;
RT_HD __forceinline__ double one_minus_exp_neg_thin(double tau, const ExpPoly &hi = exp_poly()) {
    const double x = -tau;
    double q = hi.c[3];                               // 1/10!
    q = __builtin_fma(q, x, hi.c[4]);                 // 1/9!
    q = __builtin_fma(q, x, hi.c[5]);                 // 1/8!
    q = __builtin_fma(q, x, 1.984126984126984e-04);   // 1/7!
    q = __builtin_fma(q, x, 1.3888888888888889e-03);  // 1/6!
    q = __builtin_fma(q, x, 8.333333333333333e-03);   // 1/5!
    q = __builtin_fma(q, x, 4.1666666666666664e-02);  // 1/4!
    q = __builtin_fma(q, x, 1.6666666666666666e-01);  // 1/3!
    q = __builtin_fma(q, x, 0.5);                     // 1/2!
    q = __builtin_fma(q, x, 1.0);
    return tau * q;
}

}  // namespace rt
