// rt_segmentize.hip — HIP kernels + C ABI of the MI355X-native segmentize! path (gfx950).
//
// Replaces, behind include/rt_segmentize.h, the reference's
//   segmentize!            src/trackgenerator.jl:357-369
//   _segmentize_track!     src/track.jl:106-178
//   find_element & co.     src/mesh.jl:91-176
//   intersections & co.    src/intersection.jl:11-159, src/segment.jl:31-44
//   fill_volumes           src/trackgenerator.jl:371-386
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (csrc/Makefile).
// There is no CPU fallback in this library: without a GPU every compute entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/rt_segmentize.h"
#include "rt_device.hpp"

namespace {

thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

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

}  // namespace

// ------------------------------------------------------------------- kernels -------------
namespace rt {

struct DOut {
    double *__restrict__ px, *__restrict__ py, *__restrict__ qx, *__restrict__ qy, *__restrict__ ell;
    int32_t *__restrict__ element;
    double *__restrict__ volumes;  // accumulated δs·ℓ per cell (un-normalised)
    const double *__restrict__ delta_s;
};

// One lane marches one track (_segmentize_track!, src/track.jl:106-178).  FILL=false counts
// segments and sets the track status; FILL=true re-runs the identical march and writes the
// records at the track's CSR offset (+ fused fill_volumes accumulation).
template <bool FILL>
__global__ __launch_bounds__(64) void k_march(DMesh m, DTracks t, DParams prm, int32_t *__restrict__ counts,
                                              int32_t *__restrict__ status,
                                              const int64_t *__restrict__ offsets, DOut out,
                                              unsigned long long *__restrict__ fail_info) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= t.n) return;
    const int32_t u = t.perm[slot];
    const double tA = t.A[u], tB = t.B[u], tC = t.C[u];
    const double phi = t.phi[u];
    // advance_step (src/point.jl:43): x + step * Point2D(cos ϕ, sin ϕ)
    const double sx = prm.tiny_step * t.cs[u];
    const double sy = prm.tiny_step * t.sn[u];
    double xpx = t.px[u] + sx, xpy = t.py[u] + sy;  // src/track.jl:114
    int64_t base = 0;
    double w = 0.0;
    if (FILL) {
        base = offsets[u];
        w = out.delta_s[t.azim[u] - 1];
    }
    int i = 0;
    int64_t it = 0;
    int32_t prev_element = -1;
    int st = RT_TRACK_OK;
    double sum_ell = 0.0;
    while (i < kMaxIter) {  // :119
        if (++it > prm.iter_cap) { st = RT_TRACK_ITER_CAP; break; }
        // The reference locates first and tests the boundary second (:122-125); the locate
        // result is unused on both boundary branches, so the order is swapped here.
        if (inboundary(m, xpx, xpy, prm.tiny_step)) {  // :125
            if (i == 0) { xpx = xpx + sx; xpy = xpy + sy; continue; }  // :126-129
            break;                                                      // :130-132
        }
        const int32_t element = find_element(m, xpx, xpy, prm.k);  // :122 and :138-139
        if (element < 0) { st = RT_TRACK_LOCATE_FAILED; break; }   // :140-143
        if (element == prev_element) { xpx = xpx + sx; xpy = xpy + sy; continue; }  // :147-150
        double px, py, qx, qy;
        if (!intersections(m, element, phi, tA, tB, tC, px, py, qx, qy)) {  // :153
            st = RT_TRACK_UNDEF_INTERSECTION;
            break;
        }
        if (isapprox_v2(px, py, qx, qy)) { xpx = xpx + sx; xpy = xpy + sy; continue; }  // :156-159
        const double ell = norm2(px - qx, py - qy);  // Segment ctor, src/segment.jl:31-33
        if (FILL) {
            const int64_t o = base + i;
            out.px[o] = px; out.py[o] = py; out.qx[o] = qx; out.qy[o] = qy;
            out.ell[o] = ell;
            out.element[o] = element + 1;
            unsafeAtomicAdd(&out.volumes[element], w * ell);  // fill_volumes, src/trackgenerator.jl:382
        } else {
            sum_ell += ell;
        }
        xpx = qx + sx; xpy = qy + sy;  // :165
        prev_element = element;        // :166
        ++i;                           // :168
    }
    if (!FILL) {
        // :171 isapprox(track.ℓ, sum(ℓ.(segments)); rtol)
        if (st == RT_TRACK_OK && !isapprox_s(t.ell[u], sum_ell, prm.rtol)) st = RT_TRACK_LENGTH_MISMATCH;
        counts[u] = i;
        status[u] = st;
        if (st != RT_TRACK_OK) {
            atomicAdd(&fail_info[0], 1ull);
            atomicMin(&fail_info[1], (unsigned long long)(u + 1));
        }
    }
}

// ---- exclusive scan of per-track counts (int32) into CSR offsets (int64) ----------------
constexpr int kScanBlock = 256;
constexpr int kScanPer = 4;
constexpr int kScanTile = kScanBlock * kScanPer;

__global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(const int32_t *__restrict__ counts, int64_t n,
                                                               int64_t *__restrict__ tile_sums) {
    __shared__ int64_t red[kScanBlock / 64];
    const int64_t i0 = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j)
        if (i0 + j < n) s += counts[i0 + j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t tot = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) tot += red[w];
        tile_sums[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(1024) void k_scan_tiles(int64_t *__restrict__ tile_sums, int64_t n_tiles,
                                                     int64_t *__restrict__ total) {
    __shared__ int64_t buf[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n_tiles ? tile_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            int64_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + buf[threadIdx.x] - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_write(const int32_t *__restrict__ counts, int64_t n,
                                                           const int64_t *__restrict__ tile_offsets,
                                                           const int64_t *__restrict__ total,
                                                           int64_t *__restrict__ offsets) {
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
        if (i0 + j < n) offsets[i0 + j] = run;
        run += c[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = *total;
}

__global__ void k_scale_volumes(double *__restrict__ vol, int32_t n_cells, double n_azim_2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) vol[i] = vol[i] / n_azim_2;  // volumes ./= n_azim_2, src/trackgenerator.jl:386
}

}  // namespace rt

// ------------------------------------------------------------------- handles -------------
struct rt_mesh {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int32_t n_nodes = 0, n_cells = 0;
    DevBuf<double> x, y;
    DevBuf<int32_t> cn, ncp, ncd, gstart, gnode;
    rt::DMesh d{};
    int64_t iter_cap = 4000000;
};

struct rt_tracks {
    rt_mesh *mesh = nullptr;
    int64_t n = 0;
    DevBuf<double> px, py, phi, cs, sn, A, B, C, ell;
    DevBuf<int32_t> azim, perm;
    rt::DTracks d{};
    // results
    bool segmentized = false;
    int64_t total = 0;
    DevBuf<int32_t> counts, status, element;
    DevBuf<int64_t> offsets, tile_sums, scalars;  // scalars[0] = total
    DevBuf<unsigned long long> fail_info;         // [0] n_failed, [1] first failing uid
    DevBuf<double> spx, spy, sqx, sqy, sell, volumes, delta_s;
    hipEvent_t ev[8] = {};
    double ms[8] = {};
    int64_t n_failed = 0, first_failed_uid = 0;
    int32_t first_failed_status = 0;
};

namespace {

template <typename T>
int upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s) {
    RT_HIP(b.reserve(n > 0 ? n : 1));
    if (n) RT_HIP(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_SUCCESS;
}

int build_mesh(rt_mesh *m, const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes,
               int32_t n_cells, const int32_t *ncp_in, const int32_t *ncd_in, const double *bb) {
    // --- ids to 0-based
    std::vector<int32_t> cn(3 * (size_t)n_cells);
    for (size_t i = 0; i < cn.size(); ++i) {
        const int32_t v = cell_nodes[i] - 1;
        if (v < 0 || v >= n_nodes) { set_error("cell_nodes[%zu] = %d out of range", i, cell_nodes[i]); return RT_ERR_INVALID; }
        cn[i] = v;
    }
    const int32_t p0 = ncp_in[0];  // 0- or 1-based CSR offsets
    if (p0 != 0 && p0 != 1) { set_error("node_cells_ptrs must start at 0 or 1"); return RT_ERR_INVALID; }
    std::vector<int32_t> ncp(n_nodes + 1);
    for (int32_t i = 0; i <= n_nodes; ++i) {
        ncp[i] = ncp_in[i] - p0;
        if (ncp[i] < 0 || (i > 0 && ncp[i] < ncp[i - 1])) { set_error("node_cells_ptrs not monotone"); return RT_ERR_INVALID; }
    }
    const int32_t nnz = ncp[n_nodes];
    std::vector<int32_t> ncd(nnz > 0 ? nnz : 1);
    for (int32_t i = 0; i < nnz; ++i) {
        const int32_t v = ncd_in[i] - 1;
        if (v < 0 || v >= n_cells) { set_error("node_cells_data[%d] = %d out of range", i, ncd_in[i]); return RT_ERR_INVALID; }
        ncd[i] = v;
    }
    // --- uniform node grid for exact nearest-node queries (replaces the kd-tree)
    const double W = bb[2] - bb[0], H = bb[3] - bb[1];
    if (!(W > 0) || !(H > 0)) { set_error("empty bounding box"); return RT_ERR_INVALID; }
    double gh = std::sqrt(W * H / std::max(1, n_nodes));
    int gnx = std::min(2048, std::max(1, (int)std::ceil(W / gh)));
    int gny = std::min(2048, std::max(1, (int)std::ceil(H / gh)));
    gh = std::max(W / gnx, H / gny);
    const double ginv = 1.0 / gh;
    std::vector<int32_t> gstart((size_t)gnx * gny + 1, 0), gnode(std::max(1, n_nodes)), bucket(n_nodes);
    for (int32_t i = 0; i < n_nodes; ++i) {
        double fx = std::floor((x[i] - bb[0]) * ginv), fy = std::floor((y[i] - bb[1]) * ginv);
        int ix = fx < 0 ? 0 : (fx > gnx - 1 ? gnx - 1 : (int)fx);
        int iy = fy < 0 ? 0 : (fy > gny - 1 ? gny - 1 : (int)fy);
        bucket[i] = iy * gnx + ix;
        gstart[bucket[i] + 1]++;
    }
    for (size_t b = 0; b < (size_t)gnx * gny; ++b) gstart[b + 1] += gstart[b];
    {
        std::vector<int32_t> cur(gstart.begin(), gstart.end() - 1);
        for (int32_t i = 0; i < n_nodes; ++i) gnode[cur[bucket[i]]++] = i;
    }
    hipStream_t s = m->stream;
    int rc;
    if ((rc = upload(m->x, x, n_nodes, s))) return rc;
    if ((rc = upload(m->y, y, n_nodes, s))) return rc;
    if ((rc = upload(m->cn, cn.data(), cn.size(), s))) return rc;
    if ((rc = upload(m->ncp, ncp.data(), ncp.size(), s))) return rc;
    if ((rc = upload(m->ncd, ncd.data(), (size_t)nnz, s))) return rc;
    if ((rc = upload(m->gstart, gstart.data(), gstart.size(), s))) return rc;
    if ((rc = upload(m->gnode, gnode.data(), (size_t)n_nodes, s))) return rc;
    RT_HIP(hipStreamSynchronize(s));  // host vectors die at return
    m->n_nodes = n_nodes;
    m->n_cells = n_cells;
    rt::DMesh &d = m->d;
    d.x = m->x.p; d.y = m->y.p; d.cn = m->cn.p; d.ncp = m->ncp.p; d.ncd = m->ncd.p;
    d.gstart = m->gstart.p; d.gnode = m->gnode.p;
    d.gx0 = bb[0]; d.gy0 = bb[1]; d.gh = gh; d.ginv = ginv; d.gnx = gnx; d.gny = gny;
    d.bx0 = bb[0]; d.by0 = bb[1]; d.bx1 = bb[2]; d.by1 = bb[3];
    d.n_nodes = n_nodes; d.n_cells = n_cells;
    return RT_SUCCESS;
}

void free_mesh(rt_mesh *m) {
    m->x.release(); m->y.release(); m->cn.release(); m->ncp.release(); m->ncd.release();
    m->gstart.release(); m->gnode.release();
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

void free_tracks(rt_tracks *t) {
    t->px.release(); t->py.release(); t->phi.release(); t->cs.release(); t->sn.release();
    t->A.release(); t->B.release(); t->C.release(); t->ell.release(); t->azim.release(); t->perm.release();
    t->counts.release(); t->status.release(); t->element.release(); t->offsets.release();
    t->tile_sums.release(); t->scalars.release(); t->fail_info.release();
    t->spx.release(); t->spy.release(); t->sqx.release(); t->sqy.release(); t->sell.release();
    t->volumes.release(); t->delta_s.release();
    for (auto &e : t->ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

}  // namespace

// ------------------------------------------------------------------- C ABI ---------------
extern "C" {

int32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char *rt_last_error(void) { return g_last_error.c_str(); }

const char *rt_status_message(int32_t status) {
    switch (status) {
        case RT_TRACK_OK: return "";
        case RT_TRACK_LOCATE_FAILED:
            return "Try increasing `k`. If the problem persists, raise an issue, this might be a case that "
                   "hasn't been presented before.";
        case RT_TRACK_LENGTH_MISMATCH:
            return "Track with `uid` %d has a length that do not match the sum of its segments lengths with the "
                   "provided tolerance `rtol`. Check whether this is an actual error or increase `rtol`.";
        case RT_TRACK_UNDEF_INTERSECTION: return "UndefVarError: `x_int1` not defined";
        case RT_TRACK_ITER_CAP: return "segmentize!: iteration cap reached while stepping by `tiny_step` (no progress).";
        default: return "unknown track status";
    }
}

int32_t rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

rt_mesh *rt_mesh_create(int32_t device, const double *x, const double *y, int32_t n_nodes,
                        const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                        const int32_t *node_cells_data, const double *bb) {
    if (!x || !y || !cell_nodes || !node_cells_ptrs || !node_cells_data || !bb || n_nodes <= 0 || n_cells <= 0) {
        set_error("rt_mesh_create: null pointer or empty mesh");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("rt_mesh_create: no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_error("rt_mesh_create: device %d out of range [0,%d)", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    rt_mesh *m = new rt_mesh();
    m->device = device;
    if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        delete m;
        return nullptr;
    }
    m->stream = m->own_stream;
    if (build_mesh(m, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb) != RT_SUCCESS) {
        free_mesh(m);
        return nullptr;
    }
    return m;
}

void rt_mesh_destroy(rt_mesh *mesh) {
    if (!mesh) return;
    (void)hipSetDevice(mesh->device);
    free_mesh(mesh);
}

int32_t rt_mesh_set_stream(rt_mesh *mesh, void *hip_stream) {
    if (!mesh) { set_error("null mesh"); return RT_ERR_INVALID; }
    mesh->stream = hip_stream ? (hipStream_t)hip_stream : mesh->own_stream;
    return RT_SUCCESS;
}
void *rt_mesh_get_stream(rt_mesh *mesh) { return mesh ? (void *)mesh->stream : nullptr; }

int32_t rt_set_option(rt_mesh *mesh, const char *name, int64_t value) {
    if (!mesh || !name) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!strcmp(name, "iter_cap")) { mesh->iter_cap = value > 0 ? value : 4000000; return RT_SUCCESS; }
    set_error("unknown option '%s'", name);
    return RT_ERR_INVALID;
}

rt_tracks *rt_tracks_create(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py,
                            const double *phi, const double *cos_phi, const double *sin_phi, const double *A,
                            const double *B, const double *C, const double *ell, const int32_t *azim_idx) {
    if (!mesh || n_tracks < 0 || n_tracks > 0x7fffffff ||
        (n_tracks > 0 && (!px || !py || !phi || !cos_phi || !sin_phi || !A || !B || !C || !ell || !azim_idx))) {
        set_error("rt_tracks_create: null pointer or bad track count");
        return nullptr;
    }
    if (hipSetDevice(mesh->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    rt_tracks *t = new rt_tracks();
    t->mesh = mesh;
    t->n = n_tracks;
    hipStream_t s = mesh->stream;
    const size_t n = (size_t)n_tracks;
    // march order: longest tracks first, so the waves that take longest start first and the
    // 64 lanes of a wave carry tracks with similar segment counts
    std::vector<int32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return ell[a] > ell[b]; });
    bool ok = upload(t->px, px, n, s) == 0 && upload(t->py, py, n, s) == 0 && upload(t->phi, phi, n, s) == 0 &&
              upload(t->cs, cos_phi, n, s) == 0 && upload(t->sn, sin_phi, n, s) == 0 && upload(t->A, A, n, s) == 0 &&
              upload(t->B, B, n, s) == 0 && upload(t->C, C, n, s) == 0 && upload(t->ell, ell, n, s) == 0 &&
              upload(t->azim, azim_idx, n, s) == 0 && upload(t->perm, perm.data(), n, s) == 0;
    for (auto &e : t->ev)
        if (ok && hipEventCreate(&e) != hipSuccess) ok = false;
    if (ok && hipStreamSynchronize(s) != hipSuccess) ok = false;
    if (!ok) {
        if (g_last_error.empty()) set_error("rt_tracks_create: upload failed");
        free_tracks(t);
        return nullptr;
    }
    rt::DTracks &d = t->d;
    d.px = t->px.p; d.py = t->py.p; d.phi = t->phi.p; d.cs = t->cs.p; d.sn = t->sn.p;
    d.A = t->A.p; d.B = t->B.p; d.C = t->C.p; d.ell = t->ell.p; d.azim = t->azim.p; d.perm = t->perm.p;
    d.n = n_tracks;
    return t;
}

void rt_tracks_destroy(rt_tracks *tracks) {
    if (!tracks) return;
    (void)hipSetDevice(tracks->mesh->device);
    free_tracks(tracks);
}

int64_t rt_segmentize(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s,
                      int32_t n_azim_2) {
    if (!t || !delta_s || n_azim_2 <= 0) { set_error("rt_segmentize: bad arguments"); return RT_ERR_INVALID; }
    rt_mesh *m = t->mesh;
    RT_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int64_t n = t->n;
    t->segmentized = false;
    for (double &v : t->ms) v = 0.0;

    rt::DParams prm;
    prm.tiny_step = tiny_step; prm.rtol = rtol; prm.k = k; prm.n_azim_2 = n_azim_2; prm.iter_cap = m->iter_cap;

    const int64_t n_tiles = (n + rt::kScanTile - 1) / rt::kScanTile;
    RT_HIP(t->counts.reserve(n + 1));
    RT_HIP(t->status.reserve(n + 1));
    RT_HIP(t->offsets.reserve(n + 1));
    RT_HIP(t->tile_sums.reserve(n_tiles + 1));
    RT_HIP(t->scalars.reserve(4));
    RT_HIP(t->fail_info.reserve(2));
    RT_HIP(t->volumes.reserve(m->n_cells));
    if (int rc = upload(t->delta_s, delta_s, (size_t)n_azim_2, s)) return rc;

    const unsigned long long fi0[2] = {0ull, ~0ull};
    RT_HIP(hipMemcpyAsync(t->fail_info.p, fi0, sizeof(fi0), hipMemcpyHostToDevice, s));
    RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));

    rt::DOut out{};
    out.volumes = t->volumes.p;
    out.delta_s = t->delta_s.p;

    RT_HIP(hipEventRecord(t->ev[0], s));
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (n > 0) {
        hipLaunchKernelGGL(rt::k_march<false>, dim3(grid), dim3(64), 0, s, m->d, t->d, prm, t->counts.p, t->status.p,
                           (const int64_t *)nullptr, out, t->fail_info.p);
    }
    RT_HIP(hipEventRecord(t->ev[1], s));
    // CSR offsets
    if (n > 0) {
        hipLaunchKernelGGL(rt::k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, n,
                           t->tile_sums.p);
        hipLaunchKernelGGL(rt::k_scan_tiles, dim3(1), dim3(1024), 0, s, t->tile_sums.p, n_tiles, t->scalars.p);
        hipLaunchKernelGGL(rt::k_scan_write, dim3((unsigned)n_tiles), dim3(rt::kScanBlock), 0, s, t->counts.p, n,
                           t->tile_sums.p, t->scalars.p, t->offsets.p);
    } else {
        RT_HIP(hipMemsetAsync(t->scalars.p, 0, sizeof(int64_t), s));
        RT_HIP(hipMemsetAsync(t->offsets.p, 0, sizeof(int64_t), s));
    }
    RT_HIP(hipEventRecord(t->ev[2], s));
    int64_t total = 0;
    unsigned long long fi[2] = {0, 0};
    RT_HIP(hipMemcpyAsync(&total, t->scalars.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RT_HIP(hipMemcpyAsync(fi, t->fail_info.p, sizeof(fi), hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    const size_t cap = (size_t)(total > 0 ? total : 1);
    RT_HIP(t->spx.reserve(cap)); RT_HIP(t->spy.reserve(cap)); RT_HIP(t->sqx.reserve(cap));
    RT_HIP(t->sqy.reserve(cap)); RT_HIP(t->sell.reserve(cap)); RT_HIP(t->element.reserve(cap));
    out.px = t->spx.p; out.py = t->spy.p; out.qx = t->sqx.p; out.qy = t->sqy.p; out.ell = t->sell.p;
    out.element = t->element.p;
    RT_HIP(hipEventRecord(t->ev[3], s));
    if (n > 0) {
        hipLaunchKernelGGL(rt::k_march<true>, dim3(grid), dim3(64), 0, s, m->d, t->d, prm, t->counts.p, t->status.p,
                           (const int64_t *)t->offsets.p, out, t->fail_info.p);
    }
    RT_HIP(hipEventRecord(t->ev[4], s));
    hipLaunchKernelGGL(rt::k_scale_volumes, dim3((unsigned)((m->n_cells + 255) / 256)), dim3(256), 0, s, t->volumes.p,
                       m->n_cells, (double)n_azim_2);
    RT_HIP(hipEventRecord(t->ev[5], s));
    RT_HIP(hipStreamSynchronize(s));
    RT_HIP(hipGetLastError());
    float f = 0;
    RT_HIP(hipEventElapsedTime(&f, t->ev[0], t->ev[5])); t->ms[0] = f;
    RT_HIP(hipEventElapsedTime(&f, t->ev[0], t->ev[1])); t->ms[2] = f;   // count march
    RT_HIP(hipEventElapsedTime(&f, t->ev[1], t->ev[2])); t->ms[3] = f;   // scan
    RT_HIP(hipEventElapsedTime(&f, t->ev[3], t->ev[4])); t->ms[4] = f;   // fill march
    RT_HIP(hipEventElapsedTime(&f, t->ev[4], t->ev[5])); t->ms[5] = f;   // volumes
    t->total = total;
    t->n_failed = (int64_t)fi[0];
    t->first_failed_uid = fi[0] ? (int64_t)fi[1] : 0;
    t->first_failed_status = 0;
    if (fi[0]) {
        int32_t st = 0;
        RT_HIP(hipMemcpy(&st, t->status.p + (fi[1] - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
        t->first_failed_status = st;
    }
    t->segmentized = true;
    return total;
}

int32_t rt_failed_tracks(rt_tracks *t, int64_t *n_failed, int64_t *first_uid, int32_t *first_status) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (n_failed) *n_failed = t->n_failed;
    if (first_uid) *first_uid = t->first_failed_uid;
    if (first_status) *first_status = t->first_failed_status;
    return RT_SUCCESS;
}

int32_t rt_fetch_offsets(rt_tracks *t, int64_t *seg_offsets, int32_t *status) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    RT_HIP(hipSetDevice(t->mesh->device));
    if (seg_offsets) RT_HIP(hipMemcpy(seg_offsets, t->offsets.p, sizeof(int64_t) * (t->n + 1), hipMemcpyDeviceToHost));
    if (status && t->n) RT_HIP(hipMemcpy(status, t->status.p, sizeof(int32_t) * t->n, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_fetch_segments(rt_tracks *t, double *px, double *py, double *qx, double *qy, double *ell,
                          int32_t *element) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    RT_HIP(hipSetDevice(t->mesh->device));
    const size_t nb = sizeof(double) * (size_t)t->total;
    if (t->total == 0) return RT_SUCCESS;
    if (px) RT_HIP(hipMemcpy(px, t->spx.p, nb, hipMemcpyDeviceToHost));
    if (py) RT_HIP(hipMemcpy(py, t->spy.p, nb, hipMemcpyDeviceToHost));
    if (qx) RT_HIP(hipMemcpy(qx, t->sqx.p, nb, hipMemcpyDeviceToHost));
    if (qy) RT_HIP(hipMemcpy(qy, t->sqy.p, nb, hipMemcpyDeviceToHost));
    if (ell) RT_HIP(hipMemcpy(ell, t->sell.p, nb, hipMemcpyDeviceToHost));
    if (element) RT_HIP(hipMemcpy(element, t->element.p, sizeof(int32_t) * (size_t)t->total, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_fetch_volumes(rt_tracks *t, double *volumes) {
    if (!t || !volumes) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    RT_HIP(hipSetDevice(t->mesh->device));
    RT_HIP(hipMemcpy(volumes, t->volumes.p, sizeof(double) * t->mesh->n_cells, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_device_pointers(rt_tracks *t, void **p) {
    if (!t || !p) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    p[0] = t->offsets.p; p[1] = t->status.p; p[2] = t->spx.p; p[3] = t->spy.p; p[4] = t->sqx.p;
    p[5] = t->sqy.p; p[6] = t->sell.p; p[7] = t->element.p; p[8] = t->volumes.p;
    return RT_SUCCESS;
}

int32_t rt_last_timing(rt_tracks *t, double *ms, int32_t n) {
    if (!t || !ms || n < 6) { set_error("bad argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    for (int i = 0; i < n && i < 8; ++i) ms[i] = t->ms[i];
    return RT_SUCCESS;
}

}  // extern "C"
