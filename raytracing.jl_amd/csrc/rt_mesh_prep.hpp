// rt_mesh_prep.hpp — one-time host preprocessing of the flattened mesh (C++17, no HIP).
//
// Builds what the device march needs beyond the reference's own tables:
//  * the uniform node grid used by the exact (k-)nearest-node search (replaces the kd-tree,
//    src/mesh.jl:38-42);
//  * per-cell records for the walk step of the march (csrc/rt_device.hpp, `CellRec`):
//    neighbour across each edge, vertex coordinates, and each edge's normalised general
//    form computed exactly as `general_form` does (src/intersection.jl:11-18) so that the
//    device reproduces the reference's intersection points bit for bit;
//  * per cell, an upper bound on how many non-vertex nodes can be nearer to a point of the
//    cell than the cell's nearest vertex (bounds the rank at which `find_element`'s node scan,
//    src/mesh.jl:107-132, reaches the cell);
//  * global certificate margins derived from the mesh's shape statistics.
// This TU is compiled with -ffp-contract=off, like everything else in the library.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace rtprep {

struct CellRecHost {  // must match rt::CellRec (rt_device.hpp): 144 bytes
    int32_t adj[3];   // neighbour cell across edge k = (v_k, v_{k+1 mod 3}); -1 on the boundary
    int32_t meta;     // bits 0..7: extras bound (255 = walk step disabled for this cell)
    double vx[3], vy[3];
    double eA[3], eB[3], eC[3];
    double pad;
};
static_assert(sizeof(CellRecHost) == 144, "CellRec layout");

// Rotated walk record for (cell, entry edge e): vertices rotated cyclically so that rotated
// edge 0 = (v0, v1) is the entry edge, in the cell's own edge orientation.  Must match
// rt::WalkRec (rt_device.hpp): 80 bytes = five 16-B loads per lane (the fetch costs the march
// ~50-250 cycles per load instruction, tools/micro/bench_gather.hip).  v0 and v1 are not stored:
// they are the endpoints of the predecessor's exit edge, which the walk state already holds
// (same nodes, hence the same bits), in the same or the opposite order (`same` flag).
constexpr int kWalkIdBits = 27;  // record ids + 1 must fit: 3 * n_cells + 1 < 2^27
struct WalkRecHost {
    uint64_t hdr;   // bits 0..26 next1 + 1, 27..53 next2 + 1 (record 3*cell' + entry' across rotated edge 1 / 2,
                    // 0 on the boundary), 54..61 extras bound (255: no walk), 62: v0 is the `a` of the
                    // predecessor's exit edge (a, b) — else v0 = b
    double dT;      // det of the barycentric system in the ORIGINAL node order (reference operation order)
    double x2, y2;  // the vertex opposite the entry edge
    double e1A, e1B, e1C, e2A, e2B, e2C;  // general_form of rotated edges 1 = (v1,v2) and 2 = (v2,v0)
};
static_assert(sizeof(WalkRecHost) == 80, "WalkRec layout");

struct Prep {
    std::vector<WalkRecHost> wrec;  // [3*n_cells]
    std::vector<int32_t> adjr;      // [3*n_cells] record index reached across edge k of cell c, -1 on the boundary
    // node grid
    int gnx = 1, gny = 1;
    double gh = 1.0, ginv = 1.0;
    std::vector<int32_t> gstart, gnode;
    // records
    std::vector<CellRecHost> rec;
    // certificate margins
    double eps_iso = 1e-6;   // barycentric isolation margin
    double d_vertex = 1e-7;  // absolute clearance of the track line from a cell's vertices
    double l_min = 1e-6;     // minimum chord length handled by the walk step
    bool walk_ok = true;     // false: mesh is not an edge-manifold triangulation -> generic path only
    double kappa = 0.0;      // expected segments per unit track length: Σ cell perimeters / (π · area) (Cauchy–Crofton)
    std::string note;
};

struct P2 { double x, y; };

// Clip polygon by half-plane  n·p <= c  (Sutherland–Hodgman).
inline void clip(std::vector<P2> &poly, double nx, double ny, double c) {
    std::vector<P2> out;
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
        ix = fx < 0 ? 0 : (fx > gnx - 1 ? gnx - 1 : (int)fx);
        iy = fy < 0 ? 0 : (fy > gny - 1 ? gny - 1 : (int)fy);
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

    // ---- adjacency through an edge map
    P.rec.assign(n_cells, CellRecHost{});
    std::unordered_map<uint64_t, int64_t> edge_owner;  // key -> cell*3 + k of the first owner
    edge_owner.reserve((size_t)n_cells * 2);
    auto key = [](int32_t a, int32_t b) { return ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b); };
    for (int32_t c = 0; c < n_cells; ++c)
        for (int k = 0; k < 3; ++k) P.rec[c].adj[k] = -1;
    for (int32_t c = 0; c < n_cells && P.walk_ok; ++c) {
        for (int k = 0; k < 3; ++k) {
            const int32_t a = cn[3 * c + k], b = cn[3 * c + (k + 1) % 3];
            const uint64_t kk = key(a, b);
            auto it = edge_owner.find(kk);
            if (it == edge_owner.end()) edge_owner.emplace(kk, (int64_t)c * 3 + k);
            else if (it->second < 0) { P.walk_ok = false; P.note = "edge shared by more than two cells"; break; }
            else {
                const int32_t c2 = (int32_t)(it->second / 3), k2 = (int32_t)(it->second % 3);
                P.rec[c].adj[k] = c2;
                P.rec[c2].adj[k2] = c;
                it->second = -1;
            }
        }
    }

    // ---- per-cell geometry + shape statistics
    double alt_min = INFINITY, alt_max = 0, sin_min = 1.0, l_max = 0, perim = 0, area = 0;
    for (int32_t c = 0; c < n_cells; ++c) {
        CellRecHost &R = P.rec[c];
        for (int k = 0; k < 3; ++k) { R.vx[k] = x[cn[3 * c + k]]; R.vy[k] = y[cn[3 * c + k]]; }
        double len[3];
        for (int k = 0; k < 3; ++k) {
            const int j = (k + 1) % 3;
            // general_form(p1, p2), src/intersection.jl:11-18 — same operations, same order
            const double A = R.vy[k] - R.vy[j];
            const double B = R.vx[j] - R.vx[k];
            const double C = R.vx[k] * R.vy[j] - R.vx[j] * R.vy[k];
            const double nrm = std::sqrt(A * A + B * B + C * C);
            R.eA[k] = A / nrm; R.eB[k] = B / nrm; R.eC[k] = C / nrm;
            len[k] = std::hypot(R.vx[k] - R.vx[j], R.vy[k] - R.vy[j]);
            l_max = std::max(l_max, len[k]);
            perim += len[k];
        }
        const double area2 = std::fabs((R.vx[1] - R.vx[0]) * (R.vy[2] - R.vy[0]) - (R.vx[2] - R.vx[0]) * (R.vy[1] - R.vy[0]));
        area += 0.5 * area2;
        if (!(area2 > 0)) { R.meta = 255; continue; }
        for (int k = 0; k < 3; ++k) {
            const double alt = area2 / len[k];
            alt_min = std::min(alt_min, alt); alt_max = std::max(alt_max, alt);
            const double s = area2 / (len[k] * len[(k + 2) % 3]);  // sin of the angle at vertex k
            sin_min = std::min(sin_min, s);
        }
    }
    P.kappa = area > 0 ? perim / (3.141592653589793 * area) : 0.0;
    if (!(alt_min > 0) || !std::isfinite(alt_min)) { P.walk_ok = false; P.note = "degenerate cells"; }
    const double tol = 1.4901161193847656e-8;
    if (P.walk_ok) {
        P.eps_iso = std::max(1e-6, 8.0 * tol * (alt_max / alt_min) / std::max(sin_min, 1e-3));
        P.d_vertex = 1e-6 * l_max;
        // chords shorter than this go to the generic step (the reference's isapprox(p, q) skip,
        // src/track.jl:156, triggers below ~1.5e-8 * |p|)
        P.l_min = std::max(1e-6 * l_max, 8.0 * tol * std::hypot(std::max(std::fabs(bb[0]), std::fabs(bb[2])), std::max(std::fabs(bb[1]), std::fabs(bb[3]))));
        if (P.eps_iso > 1e-3) { P.walk_ok = false; P.note = "mesh too distorted for the walk certificates"; }
    }

    // ---- extras bound: non-vertex nodes m that beat all three vertices somewhere in the
    //      (slightly inflated) cell:  |p-m|^2 < |p-v_i|^2  <=>  2 p·(v_i - m) < |v_i|^2 - |m|^2
    for (int32_t c = 0; c < n_cells && P.walk_ok; ++c) {
        CellRecHost &R = P.rec[c];
        if (R.meta == 255) continue;
        const double cx = (R.vx[0] + R.vx[1] + R.vx[2]) / 3, cy = (R.vy[0] + R.vy[1] + R.vy[2]) / 3;
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, lmax = 0;
        P2 tri[3];
        for (int k = 0; k < 3; ++k) {
            tri[k] = {cx + (R.vx[k] - cx) * 1.001, cy + (R.vy[k] - cy) * 1.001};
            xmin = std::min(xmin, tri[k].x); xmax = std::max(xmax, tri[k].x);
            ymin = std::min(ymin, tri[k].y); ymax = std::max(ymax, tri[k].y);
            lmax = std::max(lmax, std::hypot(R.vx[k] - R.vx[(k + 1) % 3], R.vy[k] - R.vy[(k + 1) % 3]));
        }
        int ix0, iy0, ix1, iy1;
        bucket_of(xmin - lmax, ymin - lmax, ix0, iy0);
        bucket_of(xmax + lmax, ymax + lmax, ix1, iy1);
        int extras = 0;
        for (int by = iy0; by <= iy1; ++by)
            for (int32_t s = P.gstart[by * gnx + ix0]; s < P.gstart[by * gnx + ix1 + 1]; ++s) {
                const int32_t m = P.gnode[s];
                if (m == cn[3 * c] || m == cn[3 * c + 1] || m == cn[3 * c + 2]) continue;
                std::vector<P2> poly(tri, tri + 3);
                const double mm = x[m] * x[m] + y[m] * y[m];
                for (int k = 0; k < 3 && !poly.empty(); ++k) {
                    const double nx = 2 * (R.vx[k] - x[m]), ny = 2 * (R.vy[k] - y[m]);
                    const double cc = R.vx[k] * R.vx[k] + R.vy[k] * R.vy[k] - mm;
                    clip(poly, nx, ny, cc + 1e-12 * (std::fabs(cc) + 1.0));  // slack: count borderline nodes
                }
                if (poly.size() >= 1) ++extras;
            }
        R.meta = extras > 254 ? 254 : extras;
    }
    // ---- rotated walk records
    if ((uint64_t)3 * (uint64_t)n_cells + 1 >= (1ull << kWalkIdBits)) { P.walk_ok = false; P.note = "too many cells for the packed walk records"; }
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
            Wr.hdr = (uint64_t)(P.adjr[3 * c + i1] + 1) | ((uint64_t)(P.adjr[3 * c + i2] + 1) << kWalkIdBits) |
                     ((uint64_t)(R.meta & 255) << (2 * kWalkIdBits)) | ((uint64_t)(same ? 1 : 0) << (2 * kWalkIdBits + 8));
            Wr.dT = dT;
            Wr.x2 = R.vx[i2]; Wr.y2 = R.vy[i2];
            Wr.e1A = R.eA[i1]; Wr.e1B = R.eB[i1]; Wr.e1C = R.eC[i1];
            Wr.e2A = R.eA[i2]; Wr.e2B = R.eB[i2]; Wr.e2C = R.eC[i2];
        }
    }
    return P;
}

}  // namespace rtprep
