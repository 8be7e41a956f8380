// rt_host.cpp — native host-side pieces around the device path (C++17, no HIP):
//   * TrackGenerator's per-angle track counts          src/trackgenerator.jl:96-110
//   * trace! and next_tracks                            src/trackgenerator.jl:134-348
//     (with AzimuthalQuadrature / init_weights!         src/azimuthal_quad.jl:21-63,
//      boundary_condition                               src/boundary.jl:48-63)
//   * mesh ingest: gmsh 4.1 ASCII -> the flat arrays rt_mesh_create takes, i.e. what
//     Mesh(model) extracts from Gridap (src/mesh.jl:24-69)
// These are the rows SURVEY.md §8f ranks next after the hot path; they stay on the host as in
// the reference, but no longer need Julia (or Python) to produce the device path's inputs.
// Compiled with -ffp-contract=off like the rest of the library: the per-track inputs must be
// the values the reference's own arithmetic produces.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_segmentize.h"

namespace rthost {
void set_error(const char *fmt, ...);  // defined in rt_segmentize.hip (thread-local last error)

constexpr double kPi = 3.141592653589793;  // Float64(π)
constexpr double kRtol = 1.4901161193847656e-8;

inline double norm2(double a, double b) { return std::sqrt(a * a + b * b); }
inline bool isapprox(double x, double y) {
    if (x == y) return true;
    if (!(std::isfinite(x) && std::isfinite(y))) return false;
    return std::fabs(x - y) <= kRtol * std::max(std::fabs(x), std::fabs(y));
}
// point_in_segment, src/segment.jl:39-44
inline bool point_in_segment(double px, double py, double qx, double qy, double x, double y) {
    return isapprox(norm2(px - x, py - y) + norm2(qx - x, qy - y), norm2(px - qx, py - qy));
}
}  // namespace rthost

using namespace rthost;

extern "C" {

int64_t rt_trace_counts(double width, double height, int32_t n_azim, double delta, int64_t *n_tracks_x,
                        int64_t *n_tracks_y) {
    // argument validation of AzimuthalQuadrature, src/azimuthal_quad.jl:21-25
    if (!(n_azim > 0)) { set_error("DomainError: number of azimuthal angles must be positive."); return RT_ERR_INVALID; }
    if (n_azim % 4 != 0) { set_error("DomainError: number of azimuthal angles must be a multiple of 4."); return RT_ERR_INVALID; }
    if (!(delta > 0)) { set_error("DomainError: azimuthal spacing must be positive."); return RT_ERR_INVALID; }
    if (!n_tracks_x || !n_tracks_y) { set_error("null output"); return RT_ERR_INVALID; }
    const int n2 = n_azim / 2, n4 = n_azim / 4;
    for (int i = 1; i <= n4; ++i) {
        const double phi = kPi / n2 * (i - 1.0 / 2);
        const int64_t nx = (int64_t)std::floor(width / delta * std::fabs(std::sin(phi))) + 1;
        const int64_t ny = (int64_t)std::floor(height / delta * std::fabs(std::cos(phi))) + 1;
        const int j = n2 - i + 1;  // suplementary_idx
        n_tracks_x[i - 1] = n_tracks_x[j - 1] = nx;
        n_tracks_y[i - 1] = n_tracks_y[j - 1] = ny;
    }
    int64_t total = 0;
    for (int i = 0; i < n2; ++i) total += n_tracks_x[i] + n_tracks_y[i];
    return total;
}

int32_t rt_trace(const double *bb, int32_t n_azim, const int64_t *ntx, const int64_t *nty, const int32_t *bcs,
                 double *phis, double *delta_s, double *omega, int32_t *azim_idx, int32_t *track_idx, double *px,
                 double *py, double *qx, double *qy, double *phi, double *cos_phi, double *sin_phi, double *ell,
                 double *A, double *B, double *C, int8_t *bc_fwd, int8_t *bc_bwd, int8_t *dir_fwd, int8_t *dir_bwd,
                 int64_t *next_fwd, int64_t *next_bwd) {
    if (!bb || !ntx || !nty || !bcs || n_azim <= 0 || n_azim % 4) { set_error("rt_trace: bad arguments"); return RT_ERR_INVALID; }
    if (!phis || !delta_s || !omega || !azim_idx || !track_idx || !px || !py || !qx || !qy || !phi || !cos_phi || !sin_phi || !ell ||
        !A || !B || !C || !bc_fwd || !bc_bwd || !dir_fwd || !dir_bwd || !next_fwd || !next_bwd) {
        set_error("rt_trace: null output array");
        return RT_ERR_INVALID;
    }
    for (int i = 0; i < n_azim / 2; ++i)
        if (ntx[i] <= 0 || nty[i] <= 0) { set_error("rt_trace: track counts must be positive (angle %d)", i + 1); return RT_ERR_INVALID; }
    try {
    const int n2 = n_azim / 2, n4 = n_azim / 4;
    const double Dx = bb[2] - bb[0], Dy = bb[3] - bb[1];
    std::vector<double> dxs(n2), dys(n2), tans(n2), coss(n2), sins(n2);
    std::vector<int64_t> off(n2 + 1, 0);
    // effective angles and spacings, src/trackgenerator.jl:150-166
    for (int i = 1; i <= n4; ++i) {
        const double ph = std::atan((Dy * (double)ntx[i - 1]) / (Dx * (double)nty[i - 1]));
        const int j = n2 - i + 1;
        phis[i - 1] = ph;
        phis[j - 1] = kPi - ph;
        dxs[i - 1] = dxs[j - 1] = Dx / (double)ntx[i - 1];
        dys[i - 1] = dys[j - 1] = Dy / (double)nty[i - 1];
        delta_s[i - 1] = delta_s[j - 1] = dxs[i - 1] * std::sin(ph);
    }
    // init_weights!, src/azimuthal_quad.jl:35-53
    for (int i = 1; i <= n4; ++i) {
        double w;
        if (i == 1) w = phis[i] - phis[i - 1];
        else if (i == n4) w = kPi - phis[i - 1] - phis[i - 2];
        else w = phis[i] - phis[i - 2];
        w /= 4 * kPi;
        omega[i - 1] = omega[n2 - i] = w;
    }
    for (int i = 0; i < n2; ++i) {
        off[i + 1] = off[i] + ntx[i] + nty[i];
        tans[i] = std::tan(phis[i]); coss[i] = std::cos(phis[i]); sins[i] = std::sin(phis[i]);
    }
    const int TOP = bcs[0], BOTTOM = bcs[1], RIGHT = bcs[2], LEFT = bcs[3];
    const double x0 = bb[0], y0 = bb[1], x1 = bb[2], y1 = bb[3];
    auto side_bc = [&](double x, double y, int &bc) -> bool {  // boundary_condition, src/boundary.jl:48-63
        if (point_in_segment(x0, y1, x1, y1, x, y)) bc = TOP;          // top    = (p2, p3)
        else if (point_in_segment(x1, y0, x0, y0, x, y)) bc = BOTTOM;  // bottom = (p4, p1)
        else if (point_in_segment(x1, y1, x1, y0, x, y)) bc = RIGHT;   // right  = (p3, p4)
        else if (point_in_segment(x0, y0, x0, y1, x, y)) bc = LEFT;    // left   = (p1, p2)
        else return false;
        return true;
    };
    int rc = RT_SUCCESS;
    for (int i = 1; i <= n2 && rc == RT_SUCCESS; ++i) {
        const bool right = i <= n4;  // points_right
        const int64_t nx = ntx[i - 1], ny = nty[i - 1], n = nx + ny;
        const int k = n2 - i + 1;
        const double ph = phis[i - 1], m = tans[i - 1];
        for (int64_t j = 1; j <= n; ++j) {
            const int64_t u = off[i - 1] + j - 1;
            // origin, src/trackgenerator.jl:188-200
            double ox, oy;
            if (j <= nx) { ox = right ? dxs[i - 1] * ((double)(nx - j) + 1.0 / 2) : dxs[i - 1] * ((double)j - 1.0 / 2); oy = 0; }
            else { ox = right ? 0.0 : Dx; oy = dys[i - 1] * ((double)(j - nx) - 1.0 / 2); }
            // exit, src/trackgenerator.jl:203-221
            double ex = ox - (oy - Dy) / m, ey = Dy;
            if (!(0 <= ex && ex <= Dx)) {
                if (right) { ex = Dx; ey = oy + m * (Dx - ox); }
                else { ex = 0; ey = oy - m * ox; }
                if (!(0 <= ey && ey <= Dy)) { set_error("DomainError: could not found track exit point."); rc = RT_ERR_INVALID; break; }
            }
            ox += x0; oy += y0; ex += x0; ey += y0;  // :224-225
            // general_form(p, q), src/intersection.jl:11-18
            const double gA = oy - ey, gB = ex - ox, gC = ox * ey - ex * oy;
            const double gn = std::sqrt(gA * gA + gB * gB + gC * gC);
            int bf, bw;
            if (!side_bc(ex, ey, bf) || !side_bc(ox, oy, bw)) { set_error("Point do not lie in the boundary."); rc = RT_ERR_INVALID; break; }
            const int bf1 = right ? (j <= ny ? RIGHT : TOP) : (j <= ny ? LEFT : TOP);       // :235-241
            const int bw1 = right ? (j <= nx ? BOTTOM : LEFT) : (j <= nx ? BOTTOM : RIGHT);
            if (bf != bf1 || bw != bw1) { set_error("Boundaries do not match!"); rc = RT_ERR_INVALID; break; }
            const bool per_f = bf == 2, per_b = bw == 2;
            azim_idx[u] = i; track_idx[u] = (int32_t)j;
            px[u] = ox; py[u] = oy; qx[u] = ex; qy[u] = ey;
            phi[u] = ph; cos_phi[u] = coss[i - 1]; sin_phi[u] = sins[i - 1];
            ell[u] = norm2(ox - ex, oy - ey);
            A[u] = gA / gn; B[u] = gB / gn; C[u] = gC / gn;
            bc_fwd[u] = (int8_t)bf; bc_bwd[u] = (int8_t)bw;
            dir_fwd[u] = (int8_t)(j <= ny ? 0 : (per_f ? 0 : 1));  // :247-255  (0 = Forward, 1 = Backward)
            dir_bwd[u] = (int8_t)(j <= nx ? (per_b ? 1 : 0) : 1);  // :257-265
            // next_track_fwd / next_track_bwd, src/trackgenerator.jl:294-348 (1-based uids)
            next_fwd[u] = j <= ny ? (per_f ? off[i - 1] + j + nx : off[k - 1] + j + nx)
                                  : (per_f ? off[i - 1] + j - ny : off[k - 1] + n + ny - j + 1);
            next_bwd[u] = j <= nx ? (per_b ? off[i - 1] + j + ny : off[k - 1] + nx - j + 1)
                                  : (per_b ? off[i - 1] + j - nx : off[k - 1] + j - nx);
        }
    }
    return rc;
    } catch (const std::exception &e) {  // no C++ exception may cross the C ABI (a Julia ccall / ctypes caller would abort)
        set_error("rt_trace: %s", e.what());
        return RT_ERR_INVALID;
    }
}

// ------------------------------------------------------------------ mesh ingest -----------
}  // extern "C" (the helpers below are C++)

struct rt_msh {
    std::vector<double> x, y;
    std::vector<int32_t> cells;              // 3 per cell, 1-based, ascending per cell
    std::vector<int32_t> nc_ptrs, nc_data;   // node -> cells CSR (0-based ptrs, 1-based ascending cell ids)
    double bb[4] = {0, 0, 0, 0};
};

// Loads the 2-D triangles of a gmsh 4.1 ASCII file with Gridap's numbering for such a file
// (GmshDiscreteModel(msh; renumber=true) + oriented grid): node tags are the ids, triangles keep
// file order, each cell's node ids are sorted ascending (verified against demo/pincell.json).
// ---- Gridap JSON (DiscreteModelFromFile, v0.15 dict layout): "grid": {"node_coordinates": [x1, y1, x2, ...],
//      "cell_node_ids": {"ptrs": [...], "data": [...]}} — a targeted scan, not a general JSON parser
static const char *json_find(const std::string &s, size_t from, size_t to, const char *key) {
    const std::string k = std::string("\"") + key + "\"";
    const size_t p = s.find(k, from);
    return (p == std::string::npos || p >= to) ? nullptr : s.c_str() + p + k.size();
}
static size_t json_match(const std::string &s, size_t open) {  // index of the bracket closing s[open]
    const char o = s[open], c = o == '{' ? '}' : ']';
    int depth = 0;
    bool in_str = false;
    for (size_t i = open; i < s.size(); ++i) {
        const char ch = s[i];
        if (in_str) { if (ch == '\\') ++i; else if (ch == '"') in_str = false; continue; }
        if (ch == '"') in_str = true;
        else if (ch == o) ++depth;
        else if (ch == c && --depth == 0) return i;
    }
    return std::string::npos;
}
template <typename T, typename Conv>
static bool json_numbers(const std::string &, const char *after_key, std::vector<T> &out, Conv conv) {
    const char *p = after_key;
    while (*p && *p != '[') { if (*p != ':' && !isspace((unsigned char)*p)) return false; ++p; }
    if (*p != '[') return false;
    ++p;
    for (;;) {
        while (*p && (isspace((unsigned char)*p) || *p == ',')) ++p;
        if (*p == ']') return true;
        char *end = nullptr;
        out.push_back(conv(p, &end));
        if (end == p) return false;
        p = end;
    }
}
static bool load_gridap_json(const std::string &s, rt_msh *M, std::string &why) {
    const char *g = json_find(s, 0, s.size(), "grid");
    if (!g) { why = "no \"grid\" object"; return false; }
    size_t gb = s.find('{', g - s.c_str());
    const size_t ge = gb == std::string::npos ? gb : json_match(s, gb);
    if (ge == std::string::npos) { why = "unbalanced \"grid\" object"; return false; }
    const char *nc = json_find(s, gb, ge, "node_coordinates");
    std::vector<double> xy;
    if (!nc || !json_numbers(s, nc, xy, [](const char *p, char **e) { return strtod(p, e); }) || xy.size() % 2) {
        why = "bad grid.node_coordinates"; return false;
    }
    const char *ci = json_find(s, gb, ge, "cell_node_ids");
    if (!ci) { why = "no grid.cell_node_ids"; return false; }
    const size_t cb = s.find('{', ci - s.c_str());
    const size_t ce = cb == std::string::npos ? cb : json_match(s, cb);
    if (ce == std::string::npos || ce > ge) { why = "unbalanced grid.cell_node_ids"; return false; }
    std::vector<long> ptrs, data;
    const char *pp = json_find(s, cb, ce, "ptrs"), *pd = json_find(s, cb, ce, "data");
    auto tol = [](const char *p, char **e) { return strtol(p, e, 10); };
    if (!pp || !pd || !json_numbers(s, pp, ptrs, tol) || !json_numbers(s, pd, data, tol) || ptrs.size() < 2) {
        why = "bad grid.cell_node_ids table"; return false;
    }
    for (size_t c = 0; c + 1 < ptrs.size(); ++c)
        if (ptrs[c + 1] - ptrs[c] != 3) { why = "only triangular cells are supported"; return false; }
    if ((size_t)(ptrs.back() - ptrs.front()) != data.size()) { why = "cell_node_ids ptrs / data mismatch"; return false; }
    const size_t nn = xy.size() / 2;
    M->x.resize(nn); M->y.resize(nn);
    for (size_t i = 0; i < nn; ++i) { M->x[i] = xy[2 * i]; M->y[i] = xy[2 * i + 1]; }
    M->cells.assign(data.begin(), data.end());  // 1-based, Gridap's order per cell (no renumbering)
    return true;
}

// node -> cells table and bounding box of a filled rt_msh (shared by both file formats)
static bool finish_msh(rt_msh *M, std::string &why) {
    if (M->x.empty() || M->cells.empty()) { why = "no 2-D triangles found"; return false; }
    const int32_t nn = (int32_t)M->x.size(), nc = (int32_t)(M->cells.size() / 3);
    for (int32_t v : M->cells) if (v < 1 || v > nn) { why = "cell refers to an unknown node"; return false; }
    // node -> cells (get_faces(topology, 0, 2), src/mesh.jl:27): filled in cell order => ascending
    M->nc_ptrs.assign(nn + 1, 0);
    for (int32_t v : M->cells) M->nc_ptrs[v]++;
    for (int32_t i = 0; i < nn; ++i) M->nc_ptrs[i + 1] += M->nc_ptrs[i];
    M->nc_data.assign(M->cells.size(), 0);
    std::vector<int32_t> cur(M->nc_ptrs.begin(), M->nc_ptrs.end() - 1);
    for (int32_t c = 0; c < nc; ++c)
        for (int k = 0; k < 3; ++k) M->nc_data[cur[M->cells[3 * c + k] - 1]++] = c + 1;
    // bounding_box, src/mesh.jl:53-69: plain min / max of the node coordinates
    M->bb[0] = *std::min_element(M->x.begin(), M->x.end()); M->bb[2] = *std::max_element(M->x.begin(), M->x.end());
    M->bb[1] = *std::min_element(M->y.begin(), M->y.end()); M->bb[3] = *std::max_element(M->y.begin(), M->y.end());
    return true;
}

extern "C" {

static rt_msh *msh_load_impl(const char *path);

rt_msh *rt_msh_load(const char *path) {
    if (!path) { set_error("rt_msh_load: null path"); return nullptr; }
    try {
        return msh_load_impl(path);
    } catch (const std::exception &e) {  // bad_alloc / length_error from a hostile header must not cross the C ABI
        set_error("rt_msh_load(%s): %s", path, e.what());
        return nullptr;
    }
}

static rt_msh *msh_load_impl(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error("rt_msh_load: cannot open %s", path); return nullptr; }
    f.seekg(0, std::ios::end);
    const long file_size = (long)f.tellg();  // every node / element takes at least two bytes of the file: bounds the counts
    f.seekg(0, std::ios::beg);
    struct Guard { rt_msh *p; ~Guard() { delete p; } } guard{new rt_msh()};
    rt_msh *M = guard.p;
    {   // a file that starts with '{' is a Gridap JSON model
        int ch;
        while ((ch = f.peek()) != EOF && isspace(ch)) f.get();
        if (ch == '{') {
            std::stringstream buf;
            buf << f.rdbuf();
            std::string why;
            if (!load_gridap_json(buf.str(), M, why) || !finish_msh(M, why)) {
                set_error("rt_msh_load(%s): %s", path, why.c_str());
                return nullptr;
            }
            guard.p = nullptr;
            return M;
        }
    }
    std::string line;
    bool fmt_ok = false;
    auto fail = [&](const char *why) -> rt_msh * { set_error("rt_msh_load(%s): %s", path, why); return nullptr; };
    while (std::getline(f, line)) {
        if (line.rfind("$MeshFormat", 0) == 0) {
            std::getline(f, line);
            fmt_ok = line.rfind("4.1", 0) == 0;
            if (!fmt_ok) return fail("expected gmsh format 4.1 ASCII");
        } else if (line.rfind("$Nodes", 0) == 0) {
            long nb, nn, mn, mx;
            f >> nb >> nn >> mn >> mx;
            // (gmsh writes a block for every entity, also those without nodes — e.g. "1 1 0 0" for a curve with no interior node — so
            //  a coarse mesh has more blocks than nodes: the block count is bounded by the file size, a header line being >= 8 bytes)
            if (!f || nb < 0 || nn <= 0 || nn > file_size / 2 || nb > file_size / 8) return fail("bad $Nodes header");
            M->x.assign(nn, 0.0); M->y.assign(nn, 0.0);
            std::vector<char> seen(nn, 0);
            for (long b = 0; b < nb; ++b) {
                long dim, tag, par, cnt;
                f >> dim >> tag >> par >> cnt;
                if (!f || cnt < 0 || cnt > nn) return fail("bad $Nodes block header");
                std::vector<long> tags(cnt);
                for (long k = 0; k < cnt; ++k) f >> tags[k];
                for (long k = 0; k < cnt; ++k) {
                    double px, py, pz;
                    f >> px >> py >> pz;
                    if (tags[k] < 1 || tags[k] > nn) return fail("node tags must be 1..n_nodes");
                    M->x[tags[k] - 1] = px; M->y[tags[k] - 1] = py; seen[tags[k] - 1] = 1;
                }
            }
            if (!f) return fail("truncated $Nodes");
            for (char c : seen) if (!c) return fail("missing node tags");
        } else if (line.rfind("$Elements", 0) == 0) {
            long nb, ne, mn, mx;
            f >> nb >> ne >> mn >> mx;
            if (!f || nb < 0 || ne < 0 || ne > file_size / 2 || nb > file_size / 8) return fail("bad $Elements header");
            for (long b = 0; b < nb; ++b) {
                long dim, tag, type, cnt;
                f >> dim >> tag >> type >> cnt;
                if (!f || cnt < 0 || cnt > ne) return fail("bad $Elements block header");
                std::getline(f, line);
                for (long k = 0; k < cnt; ++k) {
                    std::getline(f, line);
                    if (type != 2) continue;
                    std::istringstream is(line);
                    long et, a, b2, c;
                    is >> et >> a >> b2 >> c;
                    int32_t t3[3] = {(int32_t)a, (int32_t)b2, (int32_t)c};
                    std::sort(t3, t3 + 3);
                    M->cells.insert(M->cells.end(), t3, t3 + 3);
                }
            }
            if (!f) return fail("truncated $Elements");
        }
    }
    if (!fmt_ok) return fail("no 2-D triangles found");
    {
        std::string why;
        if (!finish_msh(M, why)) return fail(why.c_str());
    }
    guard.p = nullptr;
    return M;
}

int32_t rt_msh_sizes(rt_msh *msh, int32_t *n_nodes, int32_t *n_cells, int32_t *nnz) {
    if (!msh) { set_error("null handle"); return RT_ERR_INVALID; }
    if (n_nodes) *n_nodes = (int32_t)msh->x.size();
    if (n_cells) *n_cells = (int32_t)(msh->cells.size() / 3);
    if (nnz) *nnz = (int32_t)msh->nc_data.size();
    return RT_SUCCESS;
}

int32_t rt_msh_fetch(rt_msh *msh, double *x, double *y, int32_t *cell_nodes, int32_t *node_cells_ptrs,
                     int32_t *node_cells_data, double *bb) {
    if (!msh) { set_error("null handle"); return RT_ERR_INVALID; }
    if (x) memcpy(x, msh->x.data(), sizeof(double) * msh->x.size());
    if (y) memcpy(y, msh->y.data(), sizeof(double) * msh->y.size());
    if (cell_nodes) memcpy(cell_nodes, msh->cells.data(), sizeof(int32_t) * msh->cells.size());
    if (node_cells_ptrs) memcpy(node_cells_ptrs, msh->nc_ptrs.data(), sizeof(int32_t) * msh->nc_ptrs.size());
    if (node_cells_data) memcpy(node_cells_data, msh->nc_data.data(), sizeof(int32_t) * msh->nc_data.size());
    if (bb) memcpy(bb, msh->bb, sizeof(double) * 4);
    return RT_SUCCESS;
}

void rt_msh_free(rt_msh *msh) { delete msh; }

}  // extern "C"
