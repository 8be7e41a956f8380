/*
 * rt_oracle.c — CPU restatement of RayTracing.jl's segmentize! path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle and the "port" CPU baseline.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
 * load or call it.  The shipped path (raytracing.jl_amd/csrc) never links or calls it.
 *
 * It follows the reference branch for branch (all citations are into /root/reference):
 *   _segmentize_track!        src/track.jl:106-178
 *   advance_step              src/point.jl:43
 *   inboundary                src/mesh.jl:91-95
 *   find_element              src/mesh.jl:103-146
 *   point_in_triangle         src/mesh.jl:158-176   (StaticArrays 3x3 `\`, closed form)
 *   general_form              src/intersection.jl:11-18
 *   intersections             src/intersection.jl:34-119
 *   intersection              src/intersection.jl:127-138
 *   order_intersection_points src/intersection.jl:151-159
 *   Segment ctor, point_in_segment   src/segment.jl:31-44
 *   segmentize!, fill_volumes src/trackgenerator.jl:357-386
 *   TrackGenerator ctor, trace!, next_tracks   src/trackgenerator.jl:80-125,134-348
 *   AzimuthalQuadrature, init_weights!         src/azimuthal_quad.jl:21-63
 *
 * Third-party arithmetic restated from its published algorithm (packages are not vendored
 * in the reference; compat pins from Project.toml): NearestNeighbors 0.4 `nn`/`knn` = exact
 * (k-)nearest neighbours by Euclidean distance, knn with sortres=true ascending and a skip
 * predicate; StaticArrays 1.9 `\` for 3x3 = adjugate / determinant closed form, `norm` =
 * sqrt(sum abs2), left-to-right; Base.isapprox scalar/array forms with rtol=sqrt(eps) iff
 * atol==0; Gridap 0.19 vertex->cells table ascending in cell id (built by the caller).
 *
 * PARITY PIN STATUS: the reference is Julia and cannot be executed in the build container
 * or on the GPU box, and its test-suite holds no golden segment lists.  This oracle is
 * pinned against every known answer the reference's tests do hold for this path
 * (test/runtests.jl:14-28 track counts / δs / ϕs, :30-43 entry/exit and Σℓ properties,
 * :46-334 linking known answers) — see tests/test_oracle_*.py.  Per-segment element ids,
 * endpoints and volumes are "parity unpinned" against the real reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp] -shared -fPIC (oracle/Makefile).
 * All arithmetic is IEEE double without contraction, as Julia evaluates it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_LOCATE_FAILED 1   /* src/track.jl:141 */
#define ORC_LENGTH_MISMATCH 2 /* src/track.jl:172 */
#define ORC_UNDEF_INTERSECTION 3 /* src/intersection.jl:82-94: x_int1 never assigned -> UndefVarError */
#define ORC_ITER_CAP 4        /* oracle's own guard on the unbounded `continue` paths */

#define ORC_MAX_ITER 10000    /* const MAX_ITER, src/track.jl:104 */

static const double RTOL_DEFAULT = 1.4901161193847656e-8; /* sqrt(eps(Float64)) */

/* ------------------------------------------------------------------ Base.isapprox ---- */
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* isapprox(x, y) scalar, atol = 0, rtol given */
static inline int isapprox_s(double x, double y, double rtol) {
    if (x == y) return 1;
    if (!(isfinite(x) && isfinite(y))) return 0;
    return fabs(x - y) <= dmax(0.0, rtol * dmax(fabs(x), fabs(y)));
}
/* isapprox(x, y; atol) scalar with atol > 0  =>  rtol = 0 */
static inline int isapprox_atol(double x, double y, double atol) {
    if (x == y) return 1;
    if (!(isfinite(x) && isfinite(y))) return 0;
    return fabs(x - y) <= dmax(atol, 0.0);
}
static inline double norm2(double a, double b) { return sqrt(a * a + b * b); }
/* isapprox(p, q) for 2-vectors, default tolerances */
static inline int isapprox_v2(double px, double py, double qx, double qy) {
    double d = norm2(px - qx, py - qy);
    if (isfinite(d)) return d <= dmax(0.0, RTOL_DEFAULT * dmax(norm2(px, py), norm2(qx, qy)));
    return isapprox_s(px, qx, RTOL_DEFAULT) && isapprox_s(py, qy, RTOL_DEFAULT);
}

/* ------------------------------------------------------------------------ mesh ------- */
typedef struct {
    int32_t n_nodes, n_cells;
    double *x, *y;           /* node coordinates */
    int32_t *cell_nodes;     /* 3 per cell, 1-based */
    int32_t *nc_ptrs;        /* n_nodes+1, 0-based offsets */
    int32_t *nc_data;        /* 1-based cell ids */
    double bb[4];            /* xmin ymin xmax ymax */
    /* kd-tree over the nodes (exact nearest-neighbour queries) */
    int32_t *kd_idx;         /* permutation of node ids (0-based) */
    int32_t *kd_split_dim;   /* per tree node: -1 = leaf */
    double *kd_split_val;
    int32_t *kd_lo, *kd_hi;  /* range in kd_idx */
    int32_t *kd_left, *kd_right;
    int32_t kd_n;
    int use_bruteforce;
    /* results of the last orc_segmentize */
    int64_t n_seg;
    double *spx, *spy, *sqx, *sqy, *sell;
    int32_t *selem;
} orc_mesh;

#define KD_LEAF 10 /* NearestNeighbors default leafsize */

static int kd_build(orc_mesh *m, int32_t lo, int32_t hi) {
    int32_t id = m->kd_n++;
    m->kd_lo[id] = lo; m->kd_hi[id] = hi;
    m->kd_left[id] = m->kd_right[id] = -1;
    if (hi - lo <= KD_LEAF) { m->kd_split_dim[id] = -1; return id; }
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int32_t i = lo; i < hi; i++) {
        double x = m->x[m->kd_idx[i]], y = m->y[m->kd_idx[i]];
        if (x < xmin) xmin = x; if (x > xmax) xmax = x;
        if (y < ymin) ymin = y; if (y > ymax) ymax = y;
    }
    int dim = (xmax - xmin >= ymax - ymin) ? 0 : 1;
    const double *c = dim == 0 ? m->x : m->y;
    /* median split by simple selection (insertion sort on the range: build is one-time) */
    int32_t mid = (lo + hi) / 2;
    /* nth_element via repeated partition */
    int32_t l = lo, r = hi - 1;
    while (l < r) {
        double pv = c[m->kd_idx[(l + r) / 2]];
        int32_t i = l, j = r;
        while (i <= j) {
            while (c[m->kd_idx[i]] < pv) i++;
            while (c[m->kd_idx[j]] > pv) j--;
            if (i <= j) { int32_t t = m->kd_idx[i]; m->kd_idx[i] = m->kd_idx[j]; m->kd_idx[j] = t; i++; j--; }
        }
        if (j < mid) l = i;
        if (mid < i) r = j;
    }
    m->kd_split_dim[id] = dim;
    m->kd_split_val[id] = c[m->kd_idx[mid]];
    int32_t L = kd_build(m, lo, mid);
    int32_t R = kd_build(m, mid, hi);
    m->kd_left[id] = L; m->kd_right[id] = R;
    return id;
}

/* best-k list, ascending by (squared distance, node id): a total order, so that exactly equidistant nodes rank
 * the same way whatever order a search structure visits them in (NearestNeighbors' own tie order is not
 * specified; the device ranks the same way, csrc/rt_device.hpp node_before).  `k` is any width. */
typedef struct { int k, n; double *d2; int32_t *id; } knn_heap;
static inline int node_before(double d2a, int32_t ida, double d2b, int32_t idb) {
    return d2a < d2b || (d2a == d2b && ida < idb);
}
static inline void knn_push(knn_heap *h, double d2, int32_t id) {
    if (h->k <= 0) return;
    if (h->n == h->k && !node_before(d2, id, h->d2[h->n - 1], h->id[h->n - 1])) return;
    int i = h->n < h->k ? h->n++ : h->n - 1;
    while (i > 0 && node_before(d2, id, h->d2[i - 1], h->id[i - 1])) { h->d2[i] = h->d2[i - 1]; h->id[i] = h->id[i - 1]; i--; }
    h->d2[i] = d2; h->id[i] = id;
}
static void kd_query(const orc_mesh *m, int32_t node, double qx, double qy, knn_heap *h, int32_t skip) {
    int dim = m->kd_split_dim[node];
    if (dim < 0) {
        for (int32_t i = m->kd_lo[node]; i < m->kd_hi[node]; i++) {
            int32_t id = m->kd_idx[i];
            if (id == skip) continue;
            double dx = qx - m->x[id], dy = qy - m->y[id];
            knn_push(h, dx * dx + dy * dy, id);
        }
        return;
    }
    double diff = (dim == 0 ? qx : qy) - m->kd_split_val[node];
    int32_t near = diff < 0 ? m->kd_left[node] : m->kd_right[node];
    int32_t far = diff < 0 ? m->kd_right[node] : m->kd_left[node];
    kd_query(m, near, qx, qy, h, skip);
    if (h->n < h->k || diff * diff <= h->d2[h->n - 1]) kd_query(m, far, qx, qy, h, skip);
}
/* k nearest nodes (0-based ids) other than `skip`, ascending by distance; `out` holds k entries */
static int knn_nodes(const orc_mesh *m, double qx, double qy, int k, int32_t skip, int32_t *out) {
    if (k <= 0) return 0;
    double d2_small[16]; int32_t id_small[16];
    knn_heap h; h.k = k; h.n = 0;
    h.d2 = k <= 16 ? d2_small : (double *)malloc(sizeof(double) * (size_t)k);
    h.id = k <= 16 ? id_small : (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    if (m->use_bruteforce) {
        for (int32_t id = 0; id < m->n_nodes; id++) {
            if (id == skip) continue;
            double dx = qx - m->x[id], dy = qy - m->y[id];
            knn_push(&h, dx * dx + dy * dy, id);
        }
    } else {
        kd_query(m, 0, qx, qy, &h, skip);
    }
    for (int i = 0; i < h.n; i++) out[i] = h.id[i];
    if (k > 16) { free(h.d2); free(h.id); }
    return h.n;
}

void *orc_mesh_create(const double *x, const double *y, int32_t n_nodes,
                      const int32_t *cell_nodes, int32_t n_cells,
                      const int32_t *nc_ptrs, const int32_t *nc_data, const double *bb) {
    orc_mesh *m = (orc_mesh *)calloc(1, sizeof(orc_mesh));
    m->n_nodes = n_nodes; m->n_cells = n_cells;
    m->x = (double *)malloc(sizeof(double) * n_nodes);
    m->y = (double *)malloc(sizeof(double) * n_nodes);
    memcpy(m->x, x, sizeof(double) * n_nodes);
    memcpy(m->y, y, sizeof(double) * n_nodes);
    m->cell_nodes = (int32_t *)malloc(sizeof(int32_t) * 3 * n_cells);
    memcpy(m->cell_nodes, cell_nodes, sizeof(int32_t) * 3 * n_cells);
    m->nc_ptrs = (int32_t *)malloc(sizeof(int32_t) * (n_nodes + 1));
    memcpy(m->nc_ptrs, nc_ptrs, sizeof(int32_t) * (n_nodes + 1));
    int32_t nnz = nc_ptrs[n_nodes];
    m->nc_data = (int32_t *)malloc(sizeof(int32_t) * (nnz > 0 ? nnz : 1));
    memcpy(m->nc_data, nc_data, sizeof(int32_t) * nnz);
    memcpy(m->bb, bb, sizeof(double) * 4);
    int32_t cap = 2 * n_nodes + 8;
    m->kd_idx = (int32_t *)malloc(sizeof(int32_t) * n_nodes);
    for (int32_t i = 0; i < n_nodes; i++) m->kd_idx[i] = i;
    m->kd_split_dim = (int32_t *)malloc(sizeof(int32_t) * cap);
    m->kd_split_val = (double *)malloc(sizeof(double) * cap);
    m->kd_lo = (int32_t *)malloc(sizeof(int32_t) * cap);
    m->kd_hi = (int32_t *)malloc(sizeof(int32_t) * cap);
    m->kd_left = (int32_t *)malloc(sizeof(int32_t) * cap);
    m->kd_right = (int32_t *)malloc(sizeof(int32_t) * cap);
    m->kd_n = 0;
    kd_build(m, 0, n_nodes);
    return m;
}
void orc_mesh_set_bruteforce(void *mv, int on) { ((orc_mesh *)mv)->use_bruteforce = on; }
static void free_segments(orc_mesh *m) {
    free(m->spx); free(m->spy); free(m->sqx); free(m->sqy); free(m->sell); free(m->selem);
    m->spx = m->spy = m->sqx = m->sqy = m->sell = NULL; m->selem = NULL; m->n_seg = 0;
}
void orc_mesh_destroy(void *mv) {
    orc_mesh *m = (orc_mesh *)mv;
    if (!m) return;
    free_segments(m);
    free(m->x); free(m->y); free(m->cell_nodes); free(m->nc_ptrs); free(m->nc_data);
    free(m->kd_idx); free(m->kd_split_dim); free(m->kd_split_val);
    free(m->kd_lo); free(m->kd_hi); free(m->kd_left); free(m->kd_right);
    free(m);
}

/* ------------------------------------------------- point location: src/mesh.jl ------- */
/* point_in_triangle, src/mesh.jl:158-176.  λ = R \ r with R = [x1 x2 x3; y1 y2 y3; 1 1 1],
 * r = [x, y, 1]; StaticArrays closed form (det = x0·(x1 × x2) over the columns, solution by
 * cofactors / det; products with the literal 1 entries are exact and folded away). */
static int point_in_triangle(const orc_mesh *m, int32_t cell /*1-based*/, double x, double y) {
    const int32_t *nd = m->cell_nodes + 3 * (cell - 1);
    double x1 = m->x[nd[0] - 1], y1 = m->y[nd[0] - 1];
    double x2 = m->x[nd[1] - 1], y2 = m->y[nd[1] - 1];
    double x3 = m->x[nd[2] - 1], y3 = m->y[nd[2] - 1];
    double d = x1 * (y2 - y3) + y1 * (x3 - x2) + (x2 * y3 - y2 * x3);
    double l1 = ((y2 - y3) * x + (x3 - x2) * y + (x2 * y3 - x3 * y2)) / d;
    double l2 = ((y3 - y1) * x + (x1 - x3) * y + (x3 * y1 - x1 * y3)) / d;
    double l3 = ((y1 - y2) * x + (x2 - x1) * y + (x1 * y2 - x2 * y1)) / d;
    const double tol = RTOL_DEFAULT;          /* sqrt(eps(T)), src/mesh.jl:172 */
    const double lo = 0.0 - tol, hi = 1.0 + tol;
    return (lo <= l1 && l1 <= hi) && (lo <= l2 && l2 <= hi) && (lo <= l3 && l3 <= hi);
}

/* find_element, src/mesh.jl:103-146.  Returns 1-based cell id or -1. */
static int32_t find_element(const orc_mesh *m, double x, double y, int k) {
    int32_t nn_id;
    if (knn_nodes(m, x, y, 1, -1, &nn_id) < 1) return -1;
    for (int32_t i = m->nc_ptrs[nn_id]; i < m->nc_ptrs[nn_id + 1]; i++) {
        int32_t cell = m->nc_data[i];
        if (point_in_triangle(m, cell, x, y)) return cell;
    }
    int32_t ids_small[16];
    int32_t *ids = k <= 16 ? ids_small : (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    int n = knn_nodes(m, x, y, k, nn_id, ids); /* knn(kdtree, x, k, true, i -> i == nn_id): any k */
    int32_t found = -1;
    for (int j = 0; j < n && found < 0; j++) {
        int32_t node = ids[j];
        for (int32_t i = m->nc_ptrs[node]; i < m->nc_ptrs[node + 1]; i++) {
            int32_t cell = m->nc_data[i];
            if (point_in_triangle(m, cell, x, y)) { found = cell; break; }
        }
    }
    if (k > 16) free(ids);
    return found;
}

/* inboundary, src/mesh.jl:91-95 (atol > 0 => rtol = 0) */
static int inboundary(const orc_mesh *m, double x, double y, double atol) {
    if (atol > 0.0)
        return isapprox_atol(x, m->bb[2], atol) || isapprox_atol(x, m->bb[0], atol) ||
               isapprox_atol(y, m->bb[3], atol) || isapprox_atol(y, m->bb[1], atol);
    return isapprox_s(x, m->bb[2], RTOL_DEFAULT) || isapprox_s(x, m->bb[0], RTOL_DEFAULT) ||
           isapprox_s(y, m->bb[3], RTOL_DEFAULT) || isapprox_s(y, m->bb[1], RTOL_DEFAULT);
}

/* ------------------------------------------- intersections: src/intersection.jl ------ */
static inline void general_form(double xi, double yi, double xo, double yo, double *ABC) {
    double A = yi - yo;
    double B = xo - xi;
    double C = xi * yo - xo * yi;
    double n = sqrt(A * A + B * B + C * C);
    ABC[0] = A / n; ABC[1] = B / n; ABC[2] = C / n;
}
/* intersection(ABC1, ABC2), src/intersection.jl:127-138 */
static inline int intersection(const double *t, const double *e, double *x, double *y) {
    double a = t[1] * e[0];
    double b = e[1] * t[0];
    *x = 0.0; *y = 0.0;
    int par = isapprox_s(a, b, RTOL_DEFAULT);
    if (!par) {
        double det = a - b;
        *x = (t[2] * e[1] - e[2] * t[1]) / det;
        *y = (t[0] * e[2] - e[0] * t[2]) / det;
    }
    return par;
}
/* point_in_segment, src/segment.jl:39-44 */
static inline int point_in_segment(double px, double py, double qx, double qy, double x, double y) {
    double lpx = norm2(px - x, py - y);
    double lqx = norm2(qx - x, qy - y);
    double lpq = norm2(px - qx, py - qy);
    return isapprox_s(lpx + lqx, lpq, RTOL_DEFAULT);
}
static const double HALF_PI = 1.5707963267948966; /* Float64(π)/2 */
/* order_intersection_points, src/intersection.jl:151-159 */
static inline void order_points(double phi, double x1, double y1, double x2, double y2, double *pq) {
    int first;
    if (phi < HALF_PI) first = x1 < x2; else first = x1 > x2;
    if (first) { pq[0] = x1; pq[1] = y1; pq[2] = x2; pq[3] = y2; }
    else       { pq[0] = x2; pq[1] = y2; pq[2] = x1; pq[3] = y1; }
}
/* intersections(mesh, cell_id, track), src/intersection.jl:34-119.  Returns 0, or
 * ORC_UNDEF_INTERSECTION for the branch where the reference would hit an undefined variable. */
static int intersections(const orc_mesh *m, int32_t cell, double phi, const double *tABC, double *pq) {
    const int32_t *nd = m->cell_nodes + 3 * (cell - 1);
    double ix[4] = {0, 0, 0, 0}, iy[4] = {0, 0, 0, 0};
    int n_int = 0, parallel_found = 0;
    for (int i = 0; i < 3; i++) {
        int j = (i == 2) ? 0 : i + 1;
        double p1x = m->x[nd[i] - 1], p1y = m->y[nd[i] - 1];
        double p2x = m->x[nd[j] - 1], p2y = m->y[nd[j] - 1];
        double e[3];
        general_form(p1x, p1y, p2x, p2y, e);
        double x, y;
        int par = intersection(tABC, e, &x, &y);
        if (par) { parallel_found = 1; continue; }
        else if (!point_in_segment(p1x, p1y, p2x, p2y, x, y)) continue;
        else { ix[n_int] = x; iy[n_int] = y; n_int++; }
    }
    if (n_int == 3 || n_int == 4) {
        double l = 0.0; int have = 0;
        double a1x = 0, a1y = 0, a2x = 0, a2y = 0;
        for (int i = 2; i <= n_int; i++)
            for (int j = i; j <= n_int; j++) {
                double x1 = ix[i - 2], y1 = iy[i - 2], x2 = ix[j - 1], y2 = iy[j - 1];
                double li = norm2(x1 - x2, y1 - y2);
                if (li > l) { a1x = x1; a1y = y1; a2x = x2; a2y = y2; l = li; have = 1; }
            }
        if (!have) return ORC_UNDEF_INTERSECTION;
        order_points(phi, a1x, a1y, a2x, a2y, pq);
        return 0;
    } else if (n_int == 2 && parallel_found) {
        order_points(phi, ix[0], iy[0], ix[1], iy[1], pq);
        return 0;
    } else if (n_int == 2 && !parallel_found) {
        if (isapprox_v2(ix[0], iy[0], ix[1], iy[1])) {
            pq[0] = ix[0]; pq[1] = iy[0]; pq[2] = ix[1]; pq[3] = iy[1];
        } else {
            order_points(phi, ix[0], iy[0], ix[1], iy[1], pq);
        }
        return 0;
    }
    /* n_int in {0, 1}: the parent moves a tiny step further */
    pq[0] = pq[1] = pq[2] = pq[3] = 0.0;
    return 0;
}

/* ---------------------------------------------------- the march: src/track.jl -------- */
typedef struct { double *px, *py, *qx, *qy, *ell; int32_t *el; int64_t n, cap; } segbuf;
static void segbuf_push(segbuf *b, double px, double py, double qx, double qy, double ell, int32_t el) {
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 256;
        b->px = (double *)realloc(b->px, sizeof(double) * b->cap);
        b->py = (double *)realloc(b->py, sizeof(double) * b->cap);
        b->qx = (double *)realloc(b->qx, sizeof(double) * b->cap);
        b->qy = (double *)realloc(b->qy, sizeof(double) * b->cap);
        b->ell = (double *)realloc(b->ell, sizeof(double) * b->cap);
        b->el = (int32_t *)realloc(b->el, sizeof(int32_t) * b->cap);
    }
    b->px[b->n] = px; b->py[b->n] = py; b->qx[b->n] = qx; b->qy[b->n] = qy;
    b->ell[b->n] = ell; b->el[b->n] = el; b->n++;
}

/* _segmentize_track!, src/track.jl:106-178.  cs/sn: cos ϕ / sin ϕ if the caller supplies
 * them (the product takes them from the host), else NULL => libm per advance_step call,
 * as the reference's advance_step does. */
static int segmentize_track(const orc_mesh *m, double tpx, double tpy, double phi, const double *cs,
                            const double *sn, const double *tABC, double tell, double tiny_step, int k,
                            double rtol, int64_t iter_cap, segbuf *out, int64_t *n_iter) {
#define ADVANCE(X, Y) do { double c_ = cs ? *cs : cos(phi), s_ = sn ? *sn : sin(phi); \
        (X) = (X) + tiny_step * c_; (Y) = (Y) + tiny_step * s_; } while (0)
    int64_t first = out->n;
    double xpx = tpx, xpy = tpy;
    ADVANCE(xpx, xpy);                                   /* :114 */
    int i = 0;
    int64_t it = 0;
    int32_t element = -1, prev_element = -1;
    int status = ORC_OK;
    while (i < ORC_MAX_ITER) {                           /* :119 */
        if (++it > iter_cap) { status = ORC_ITER_CAP; break; }
        element = find_element(m, xpx, xpy, 2);          /* :122 */
        if (inboundary(m, xpx, xpy, tiny_step)) {        /* :125 */
            if (out->n == first) { ADVANCE(xpx, xpy); continue; }   /* :126-129 */
            else break;                                  /* :130-132 */
        }
        if (element == -1) {                             /* :138-144 */
            element = find_element(m, xpx, xpy, k);
            if (element == -1) { status = ORC_LOCATE_FAILED; break; }
        }
        if (prev_element == element) { ADVANCE(xpx, xpy); continue; }   /* :147-150 */
        double pq[4];
        int rc = intersections(m, element, phi, tABC, pq);               /* :153 */
        if (rc) { status = rc; break; }
        if (isapprox_v2(pq[0], pq[1], pq[2], pq[3])) { ADVANCE(xpx, xpy); continue; }  /* :156-159 */
        segbuf_push(out, pq[0], pq[1], pq[2], pq[3], norm2(pq[0] - pq[2], pq[1] - pq[3]), element); /* :161-162 */
        xpx = pq[2]; xpy = pq[3];
        ADVANCE(xpx, xpy);                               /* :165 */
        prev_element = element;                          /* :166 */
        i += 1;                                          /* :168 */
    }
#undef ADVANCE
    if (n_iter) *n_iter = it;
    if (status != ORC_OK) return status;
    double s = 0.0;                                      /* :171 sum(ℓ.(segments)) */
    for (int64_t j = first; j < out->n; j++) s += out->ell[j];
    if (!isapprox_s(tell, s, rtol)) return ORC_LENGTH_MISMATCH;
    return ORC_OK;
}

/* segmentize! over all tracks, src/trackgenerator.jl:357-369 (fill_volumes separately).
 * Outputs: seg_offsets[n_tracks+1], status[n_tracks], n_iters[n_tracks] (may be NULL).
 * Segment records are kept behind the mesh handle; read them with orc_fetch.  A failed
 * track keeps the segments it produced before failing (the reference would have thrown).
 * Returns the total number of segments. */
int64_t orc_segmentize(void *mv, int64_t n_tracks, const double *px, const double *py, const double *phi,
                       const double *cosphi, const double *sinphi, const double *A, const double *B,
                       const double *C, const double *ell, double tiny_step, int32_t k, double rtol,
                       int64_t iter_cap, int32_t n_threads, int64_t *seg_offsets, int32_t *status,
                       int64_t *n_iters) {
    orc_mesh *m = (orc_mesh *)mv;
    free_segments(m);
    if (iter_cap <= 0) iter_cap = (int64_t)1 << 40;
    int nthr = 1;
#ifdef _OPENMP
    nthr = n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    (void)n_threads;
    (void)nthr;
#endif
    /* contiguous uid chunks (one buffer each) keep per-track order trivially; chunks are
     * handed out dynamically so that long and short angles balance across threads */
    const int64_t CH = 256;
    int nt = (int)((n_tracks + CH - 1) / CH);
    if (nt < 1) nt = 1;
    segbuf *bufs = (segbuf *)calloc(nt, sizeof(segbuf));
    int64_t *counts = (int64_t *)calloc(n_tracks + 1, sizeof(int64_t));
    int64_t *lo = (int64_t *)malloc(sizeof(int64_t) * (nt + 1));
    for (int t = 0; t <= nt; t++) { lo[t] = (int64_t)t * CH; if (lo[t] > n_tracks) lo[t] = n_tracks; }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthr) schedule(dynamic, 1)
#endif
    for (int t = 0; t < nt; t++) {
        segbuf *b = &bufs[t];
        for (int64_t u = lo[t]; u < lo[t + 1]; u++) {
            double tABC[3] = {A[u], B[u], C[u]};
            int64_t before = b->n, it = 0;
            status[u] = segmentize_track(m, px[u], py[u], phi[u], cosphi ? &cosphi[u] : NULL,
                                         sinphi ? &sinphi[u] : NULL, tABC, ell[u], tiny_step, k, rtol,
                                         iter_cap, b, &it);
            counts[u] = b->n - before;
            if (n_iters) n_iters[u] = it;
        }
    }
    int64_t total = 0;
    for (int64_t u = 0; u < n_tracks; u++) { seg_offsets[u] = total; total += counts[u]; }
    seg_offsets[n_tracks] = total;
    m->n_seg = total;
    size_t nn = total > 0 ? (size_t)total : 1;
    m->spx = (double *)malloc(sizeof(double) * nn); m->spy = (double *)malloc(sizeof(double) * nn);
    m->sqx = (double *)malloc(sizeof(double) * nn); m->sqy = (double *)malloc(sizeof(double) * nn);
    m->sell = (double *)malloc(sizeof(double) * nn); m->selem = (int32_t *)malloc(sizeof(int32_t) * nn);
    for (int t = 0; t < nt; t++) {
        int64_t off = seg_offsets[lo[t]];
        segbuf *b = &bufs[t];
        memcpy(m->spx + off, b->px, sizeof(double) * b->n); memcpy(m->spy + off, b->py, sizeof(double) * b->n);
        memcpy(m->sqx + off, b->qx, sizeof(double) * b->n); memcpy(m->sqy + off, b->qy, sizeof(double) * b->n);
        memcpy(m->sell + off, b->ell, sizeof(double) * b->n); memcpy(m->selem + off, b->el, sizeof(int32_t) * b->n);
        free(b->px); free(b->py); free(b->qx); free(b->qy); free(b->ell); free(b->el);
    }
    free(bufs); free(counts); free(lo);
    return total;
}
int64_t orc_fetch(void *mv, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element) {
    orc_mesh *m = (orc_mesh *)mv;
    size_t n = (size_t)m->n_seg;
    if (px) memcpy(px, m->spx, sizeof(double) * n);
    if (py) memcpy(py, m->spy, sizeof(double) * n);
    if (qx) memcpy(qx, m->sqx, sizeof(double) * n);
    if (qy) memcpy(qy, m->sqy, sizeof(double) * n);
    if (ell) memcpy(ell, m->sell, sizeof(double) * n);
    if (element) memcpy(element, m->selem, sizeof(int32_t) * n);
    return m->n_seg;
}

/* fill_volumes, src/trackgenerator.jl:371-386 (the tail :388-397 is dead code). */
void orc_fill_volumes(void *mv, int64_t n_tracks, const int64_t *seg_offsets, const int32_t *azim_idx,
                      const double *delta_s, int32_t n_azim_2, double *volumes) {
    orc_mesh *m = (orc_mesh *)mv;
    for (int32_t c = 0; c < m->n_cells; c++) volumes[c] = 0.0;
    for (int64_t u = 0; u < n_tracks; u++) {
        int32_t a = azim_idx[u];
        for (int64_t s = seg_offsets[u]; s < seg_offsets[u + 1]; s++) {
            int32_t e = m->selem[s];
            volumes[e - 1] += delta_s[a - 1] * m->sell[s];
        }
    }
    for (int32_t c = 0; c < m->n_cells; c++) volumes[c] /= (double)n_azim_2;
}

/* single-point probes used by the tests */
int32_t orc_find_element(void *mv, double x, double y, int32_t k) { return find_element((orc_mesh *)mv, x, y, k); }
int32_t orc_nn(void *mv, double x, double y) { int32_t id = -1; knn_nodes((orc_mesh *)mv, x, y, 1, -1, &id); return id + 1; }
int32_t orc_knn(void *mv, double x, double y, int32_t k, int32_t skip1, int32_t *out) {
    /* `out` holds k entries */
    int n = knn_nodes((orc_mesh *)mv, x, y, k, skip1 - 1, out);
    for (int i = 0; i < n; i++) out[i] = out[i] + 1;
    return n;
}
int32_t orc_point_in_triangle(void *mv, int32_t cell, double x, double y) { return point_in_triangle((orc_mesh *)mv, cell, x, y); }
int32_t orc_intersections(void *mv, int32_t cell, double phi, const double *tABC, double *pq) {
    return intersections((orc_mesh *)mv, cell, phi, tABC, pq);
}

/* ------------------------------------ track generation: src/trackgenerator.jl -------- */
static const double PI = 3.141592653589793; /* Float64(π) */

/* TrackGenerator ctor, src/trackgenerator.jl:96-110.  ntx/nty: n_azim/2 entries. Returns
 * n_total_tracks or -1 on invalid arguments (src/azimuthal_quad.jl:21-25). */
int64_t orc_track_counts(double Dx, double Dy, int32_t n_azim, double delta, int64_t *ntx, int64_t *nty) {
    if (!(n_azim > 0) || n_azim % 4 != 0 || !(delta > 0)) return -1;
    int n2 = n_azim / 2, n4 = n_azim / 4;
    int64_t total = 0;
    for (int i = 1; i <= n4; i++) {
        double phi = PI / n2 * (i - 1.0 / 2);
        ntx[i - 1] = (int64_t)floor(Dx / delta * fabs(sin(phi))) + 1;
        nty[i - 1] = (int64_t)floor(Dy / delta * fabs(cos(phi))) + 1;
        int j = n2 - i + 1;
        ntx[j - 1] = ntx[i - 1]; nty[j - 1] = nty[i - 1];
    }
    for (int i = 0; i < n2; i++) total += ntx[i] + nty[i];
    return total;
}

/* trace!, src/trackgenerator.jl:134-280 + next_tracks :282-348.  bcs = {top,bottom,right,left}
 * with 0=Vacuum 1=Reflective 2=Periodic.  Per-angle outputs (n_azim/2): phis, delta_s,
 * omega.  Per-track outputs in uid order.  Returns 0, -2 DomainError (exit point), -3
 * "Boundaries do not match!" / point not on the boundary. */
int32_t orc_trace(const double *bb, int32_t n_azim, const int64_t *ntx, const int64_t *nty, const int32_t *bcs,
                  double *phis, double *delta_s, double *omega, int32_t *azim_idx, int32_t *track_idx,
                  double *px, double *py, double *qx, double *qy, double *phi_t, double *ell, double *A,
                  double *B, double *C, int8_t *bc_fwd, int8_t *bc_bwd, int8_t *dir_fwd, int8_t *dir_bwd,
                  int64_t *next_fwd, int64_t *next_bwd) {
    int n2 = n_azim / 2, n4 = n_azim / 4;
    double Dx = bb[2] - bb[0], Dy = bb[3] - bb[1];
    double *dxs = (double *)malloc(sizeof(double) * n2), *dys = (double *)malloc(sizeof(double) * n2);
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (n2 + 1));
    for (int i = 1; i <= n4; i++) {
        double ph = atan((Dy * (double)ntx[i - 1]) / (Dx * (double)nty[i - 1]));
        phis[i - 1] = ph;
        dxs[i - 1] = Dx / (double)ntx[i - 1];
        dys[i - 1] = Dy / (double)nty[i - 1];
        delta_s[i - 1] = dxs[i - 1] * sin(ph);
        int j = n2 - i + 1;
        phis[j - 1] = PI - ph; dxs[j - 1] = dxs[i - 1]; dys[j - 1] = dys[i - 1]; delta_s[j - 1] = delta_s[i - 1];
    }
    for (int i = 1; i <= n4; i++) { /* init_weights!, src/azimuthal_quad.jl:35-53 */
        double w;
        if (i == 1) w = phis[i] - phis[i - 1];
        else if (i == n4) w = PI - phis[i - 1] - phis[i - 2];
        else w = phis[i] - phis[i - 2];
        w /= 4 * PI;
        omega[i - 1] = w; omega[n2 - i] = w;
    }
    off[0] = 0;
    for (int i = 0; i < n2; i++) off[i + 1] = off[i] + ntx[i] + nty[i];
    const int32_t TOP = bcs[0], BOTTOM = bcs[1], RIGHT = bcs[2], LEFT = bcs[3];
    /* sides: top=(p2,p3) bottom=(p4,p1) right=(p3,p4) left=(p1,p2) */
    double p1x = bb[0], p1y = bb[1], p2x = bb[0], p2y = bb[3], p3x = bb[2], p3y = bb[3], p4x = bb[2], p4y = bb[1];
    int rc = 0;
    int64_t u = 0;
    for (int i = 1; i <= n2 && rc == 0; i++) {
        double ph = phis[i - 1];
        int right = i <= n4;
        int64_t nx = ntx[i - 1], ny = nty[i - 1], n = nx + ny;
        int k = n2 - i + 1;
        for (int64_t j = 1; j <= n; j++, u++) {
            double ox, oy;
            if (j <= nx) {
                if (right) { ox = dxs[i - 1] * ((double)(nx - j) + 1.0 / 2); oy = 0; }
                else { ox = dxs[i - 1] * ((double)j - 1.0 / 2); oy = 0; }
            } else {
                if (right) { ox = 0; oy = dys[i - 1] * ((double)(j - nx) - 1.0 / 2); }
                else { ox = Dx; oy = dys[i - 1] * ((double)(j - nx) - 1.0 / 2); }
            }
            double mm = tan(ph);
            double ex = ox - (oy - Dy) / mm, ey = Dy;
            if (!(0 <= ex && ex <= Dx)) {
                if (right) { ex = Dx; ey = oy + mm * (Dx - ox); }
                else { ex = 0; ey = oy - mm * ox; }
                if (!(0 <= ey && ey <= Dy)) { rc = -2; break; }
            }
            ox += bb[0]; oy += bb[1]; ex += bb[0]; ey += bb[1];
            double abc[3];
            general_form(ox, oy, ex, ey, abc);
            int bf, bbw;
            /* boundary_condition, src/boundary.jl:48-63 */
            if (point_in_segment(p2x, p2y, p3x, p3y, ex, ey)) bf = TOP;
            else if (point_in_segment(p4x, p4y, p1x, p1y, ex, ey)) bf = BOTTOM;
            else if (point_in_segment(p3x, p3y, p4x, p4y, ex, ey)) bf = RIGHT;
            else if (point_in_segment(p1x, p1y, p2x, p2y, ex, ey)) bf = LEFT;
            else { rc = -3; break; }
            if (point_in_segment(p2x, p2y, p3x, p3y, ox, oy)) bbw = TOP;
            else if (point_in_segment(p4x, p4y, p1x, p1y, ox, oy)) bbw = BOTTOM;
            else if (point_in_segment(p3x, p3y, p4x, p4y, ox, oy)) bbw = RIGHT;
            else if (point_in_segment(p1x, p1y, p2x, p2y, ox, oy)) bbw = LEFT;
            else { rc = -3; break; }
            int bf1, bb1;
            if (right) { bf1 = j <= ny ? RIGHT : TOP; bb1 = j <= nx ? BOTTOM : LEFT; }
            else { bf1 = j <= ny ? LEFT : TOP; bb1 = j <= nx ? BOTTOM : RIGHT; }
            if (bf != bf1 || bbw != bb1) { rc = -3; break; }
            int df, db;
            if (j <= ny) df = 0; else df = (bf == 2) ? 0 : 1;
            if (j <= nx) db = (bbw == 2) ? 1 : 0; else db = 1;
            int64_t nf, nb;
            if (j <= ny) nf = (bf == 2) ? off[i - 1] + j + nx : off[k - 1] + j + nx;
            else nf = (bf == 2) ? off[i - 1] + j - ny : off[k - 1] + n + ny - j + 1;
            if (j <= nx) nb = (bbw == 2) ? off[i - 1] + j + ny : off[k - 1] + nx - j + 1;
            else nb = (bbw == 2) ? off[i - 1] + j - nx : off[k - 1] + j - nx;
            azim_idx[u] = i; track_idx[u] = (int32_t)j;
            px[u] = ox; py[u] = oy; qx[u] = ex; qy[u] = ey; phi_t[u] = ph;
            ell[u] = norm2(ox - ex, oy - ey);
            A[u] = abc[0]; B[u] = abc[1]; C[u] = abc[2];
            bc_fwd[u] = (int8_t)bf; bc_bwd[u] = (int8_t)bbw; dir_fwd[u] = (int8_t)df; dir_bwd[u] = (int8_t)db;
            next_fwd[u] = nf; next_bwd[u] = nb;
        }
    }
    free(dxs); free(dys); free(off);
    return rc;
}

int32_t orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
