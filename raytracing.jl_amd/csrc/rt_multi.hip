// rt_multi.hip — several GPUs behind one call of the C ABI (include/rt_segmentize.h, rt_multi_*).
//
// The reference marches tracks_by_uid one after the other and every track writes only its own segments
// (src/trackgenerator.jl:362-364), so the path shards without any data-path exchange: the mesh is replicated,
// tracks_by_uid is cut into contiguous uid ranges balanced by Σℓ (segments ∝ ℓ), one range per device, and every
// device runs the single-device path (rt_segmentize) on its range from its own host thread.  The only reduction
// across tracks, fill_volumes (src/trackgenerator.jl:371-386), is a sum of the shards' volumes.  Reassembling the
// global segment list is a gather: to the host straight from every device (each over its own PCIe link), or onto
// every device with peer-to-peer copies — xGMI is point to point, so shard j goes to device i over the link (j, i)
// while all other pairs use theirs.  Built only on the public single-device entry points and the HIP runtime.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_segmentize.h"

namespace rthost {
void set_error(const char *fmt, ...);
}
using rthost::set_error;

struct rt_multi {
    int32_t n = 0;
    int64_t n_tracks = 0, total = 0;
    int32_t n_cells = 0;
    std::vector<int32_t> device;
    std::vector<rt_mesh *> mesh;
    std::vector<rt_tracks *> tracks;
    std::vector<int64_t> uid_begin;  // [n+1]
    std::vector<int64_t> seg_begin;  // [n+1], after rt_multi_segmentize
    bool segmentized = false;
    // rt_multi_allgather: global arrays on every shard's device
    struct Gathered { void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap = 0; hipStream_t s = nullptr; };
    std::vector<Gathered> gathered;
};

namespace {

void free_multi(rt_multi *m) {
    for (size_t i = 0; i < m->gathered.size(); ++i) {
        if (i < m->device.size()) (void)hipSetDevice(m->device[i]);
        for (void *&q : m->gathered[i].p) { if (q) (void)hipFree(q); q = nullptr; }
        if (m->gathered[i].s) (void)hipStreamDestroy(m->gathered[i].s);
    }
    for (rt_tracks *t : m->tracks) rt_tracks_destroy(t);
    for (rt_mesh *h : m->mesh) rt_mesh_destroy(h);
    delete m;
}

// Run f(i) for every shard on its own host thread; collects the first error text.
template <typename F>
bool for_each_shard(int n, F f, std::string &err) {
    std::vector<std::string> errs(n);
    std::vector<int> ok(n, 1);
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i]() {
            if (!f(i)) { ok[i] = 0; errs[i] = rt_last_error(); }  // rt_last_error is per thread: copy it here
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (!ok[i]) { err = "shard " + std::to_string(i) + ": " + errs[i]; return false; }
    return true;
}

}  // namespace

extern "C" {

rt_multi *rt_multi_create(const int32_t *device_ids, int32_t n_devices, const double *x, const double *y, int32_t n_nodes,
                          const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                          const int32_t *node_cells_data, const double *bb, int64_t n_tracks, const double *px,
                          const double *py, const double *phi, const double *cos_phi, const double *sin_phi, const double *A,
                          const double *B, const double *C, const double *ell, const int32_t *azim_idx) {
    if (!device_ids || n_devices <= 0 || n_devices > 64 || n_tracks < 0 ||
        (n_tracks > 0 && (!px || !py || !phi || !cos_phi || !sin_phi || !A || !B || !C || !ell || !azim_idx))) {
        set_error("rt_multi_create: bad arguments");
        return nullptr;
    }
    rt_multi *m = new rt_multi();
    m->n = n_devices;
    m->n_tracks = n_tracks;
    m->n_cells = n_cells;
    m->device.assign(device_ids, device_ids + n_devices);
    // contiguous uid ranges with ≈ equal Σℓ: cut r is the first uid at which the running sum reaches r/n of the total
    // (the same rule as the Python host's shard_ranges)
    m->uid_begin.assign(n_devices + 1, 0);
    {
        std::vector<double> cum((size_t)n_tracks + 1, 0.0);
        for (int64_t u = 0; u < n_tracks; ++u) cum[u + 1] = cum[u] + ell[u];
        int64_t prev = 0;
        for (int r = 1; r < n_devices; ++r) {
            const double target = cum[n_tracks] * (double)r / (double)n_devices;
            int64_t lo = 0, hi = n_tracks + 1;  // first index with cum[i] >= target
            while (lo < hi) {
                const int64_t mid = (lo + hi) / 2;
                if (cum[mid] < target) lo = mid + 1; else hi = mid;
            }
            int64_t cut = lo > n_tracks ? n_tracks : lo;
            if (cut < prev) cut = prev;
            m->uid_begin[r] = cut;
            prev = cut;
        }
        m->uid_begin[n_devices] = n_tracks;
    }
    m->mesh.assign(n_devices, nullptr);
    m->tracks.assign(n_devices, nullptr);
    std::string err;
    const bool ok = for_each_shard(n_devices, [&](int i) {
        m->mesh[i] = rt_mesh_create(device_ids[i], x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb);
        if (!m->mesh[i]) return false;
        const int64_t lo = m->uid_begin[i], cnt = m->uid_begin[i + 1] - lo;
        m->tracks[i] = rt_tracks_create(m->mesh[i], cnt, px + lo, py + lo, phi + lo, cos_phi + lo, sin_phi + lo, A + lo, B + lo,
                                        C + lo, ell + lo, azim_idx + lo);
        return m->tracks[i] != nullptr;
    }, err);
    if (!ok) {
        free_multi(m);
        set_error("rt_multi_create: %s", err.c_str());
        return nullptr;
    }
    return m;
}

void rt_multi_destroy(rt_multi *m) {
    if (m) free_multi(m);
}

int32_t rt_multi_set_option(rt_multi *m, const char *name, int64_t value) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    for (rt_mesh *h : m->mesh)
        if (int32_t rc = rt_set_option(h, name, value)) return rc;
    return RT_SUCCESS;
}

int64_t rt_multi_segmentize(rt_multi *m, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
    if (!m) { set_error("rt_multi_segmentize: null handle"); return RT_ERR_INVALID; }
    m->segmentized = false;
    std::vector<int64_t> tot(m->n, 0);
    std::string err;
    const bool ok = for_each_shard(m->n, [&](int i) {
        tot[i] = rt_segmentize(m->tracks[i], tiny_step, k, rtol, delta_s, n_azim_2);
        return tot[i] >= 0;
    }, err);
    if (!ok) {
        int64_t rc = RT_ERR_HIP;
        for (int64_t t : tot) if (t < 0) { rc = t; break; }
        set_error("rt_multi_segmentize: %s", err.c_str());
        return rc;
    }
    m->seg_begin.assign(m->n + 1, 0);
    for (int i = 0; i < m->n; ++i) m->seg_begin[i + 1] = m->seg_begin[i] + tot[i];
    m->total = m->seg_begin[m->n];
    m->segmentized = true;
    return m->total;
}

int32_t rt_multi_shards(rt_multi *m, int64_t *uid_begin, int64_t *seg_begin) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (uid_begin) memcpy(uid_begin, m->uid_begin.data(), sizeof(int64_t) * (m->n + 1));
    if (seg_begin) {
        if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
        memcpy(seg_begin, m->seg_begin.data(), sizeof(int64_t) * (m->n + 1));
    }
    return m->n;
}

rt_tracks *rt_multi_shard(rt_multi *m, int32_t i) {
    if (!m || i < 0 || i >= m->n) { set_error("rt_multi_shard: bad index"); return nullptr; }
    return m->tracks[i];
}

int32_t rt_multi_failed_tracks(rt_multi *m, int64_t *n_failed, int64_t *first_uid, int32_t *first_status) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    int64_t nf = 0, fu = 0;
    int32_t fs = 0;
    for (int i = 0; i < m->n; ++i) {
        int64_t n_i = 0, u_i = 0;
        int32_t s_i = 0;
        if (int32_t rc = rt_failed_tracks(m->tracks[i], &n_i, &u_i, &s_i)) return rc;
        if (n_i && !nf) { fu = m->uid_begin[i] + u_i; fs = s_i; }  // shards are in uid order: the first failing shard holds the first uid
        nf += n_i;
    }
    if (n_failed) *n_failed = nf;
    if (first_uid) *first_uid = fu;
    if (first_status) *first_status = fs;
    return RT_SUCCESS;
}

int32_t rt_multi_fetch_offsets(rt_multi *m, int64_t *seg_offsets, int32_t *status) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    std::string err;
    const bool ok = for_each_shard(m->n, [&](int i) {
        const int64_t lo = m->uid_begin[i], cnt = m->uid_begin[i + 1] - lo;
        std::vector<int64_t> off((size_t)cnt + 1);
        if (rt_fetch_offsets(m->tracks[i], seg_offsets ? off.data() : nullptr, status ? status + lo : nullptr)) return false;
        if (seg_offsets)
            for (int64_t u = 0; u < cnt; ++u) seg_offsets[lo + u] = m->seg_begin[i] + off[u];
        return true;
    }, err);
    if (!ok) { set_error("rt_multi_fetch_offsets: %s", err.c_str()); return RT_ERR_HIP; }
    if (seg_offsets) seg_offsets[m->n_tracks] = m->total;
    return RT_SUCCESS;
}

int32_t rt_multi_fetch_segments(rt_multi *m, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    std::string err;
    const bool ok = for_each_shard(m->n, [&](int i) {  // every device copies over its own PCIe link
        const int64_t s0 = m->seg_begin[i];
        return rt_fetch_segments(m->tracks[i], px ? px + s0 : nullptr, py ? py + s0 : nullptr, qx ? qx + s0 : nullptr,
                                 qy ? qy + s0 : nullptr, ell ? ell + s0 : nullptr, element ? element + s0 : nullptr) == RT_SUCCESS;
    }, err);
    if (!ok) { set_error("rt_multi_fetch_segments: %s", err.c_str()); return RT_ERR_HIP; }
    return RT_SUCCESS;
}

int32_t rt_multi_fetch_volumes(rt_multi *m, double *volumes) {
    if (!m || !volumes) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    std::vector<double> part((size_t)m->n_cells);
    for (int32_t c = 0; c < m->n_cells; ++c) volumes[c] = 0.0;
    for (int i = 0; i < m->n; ++i) {  // summed in shard order: deterministic
        if (int32_t rc = rt_fetch_volumes(m->tracks[i], part.data())) return rc;
        for (int32_t c = 0; c < m->n_cells; ++c) volumes[c] += part[c];
    }
    return RT_SUCCESS;
}

int32_t rt_multi_allgather(rt_multi *m, void **ptrs_dev, double *ms) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    const size_t total = (size_t)m->total;
    const size_t esz[6] = {8, 8, 8, 8, 8, 4};
    m->gathered.resize(m->n);
    // sources: every shard's device-resident records
    std::vector<void *> src((size_t)m->n * 9, nullptr);
    for (int j = 0; j < m->n; ++j)
        if (int32_t rc = rt_device_pointers(m->tracks[j], &src[(size_t)j * 9])) return rc;
    for (int i = 0; i < m->n; ++i) {
        if (hipSetDevice(m->device[i]) != hipSuccess) { set_error("hipSetDevice(%d) failed", m->device[i]); return RT_ERR_HIP; }
        rt_multi::Gathered &g = m->gathered[i];
        if (!g.s && hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return RT_ERR_HIP; }
        if (total > g.cap) {
            for (void *&q : g.p) { if (q) (void)hipFree(q); q = nullptr; }
            g.cap = 0;
            const size_t cap = total + total / 16 + 64;
            for (int a = 0; a < 6; ++a)
                if (hipMalloc(&g.p[a], cap * esz[a]) != hipSuccess) { set_error("rt_multi_allgather: hipMalloc of %zu bytes failed", cap * esz[a]); return RT_ERR_HIP; }
            g.cap = cap;
        }
        for (int j = 0; j < m->n; ++j)
            if (m->device[j] != m->device[i]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, m->device[i], m->device[j]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(m->device[j], 0);  // already enabled: an error we ignore
                (void)hipGetLastError();
            }
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < m->n; ++i) {
        (void)hipSetDevice(m->device[i]);
        for (int jj = 0; jj < m->n; ++jj) {
            const int j = (i + jj) % m->n;  // every destination starts with a different source: all links busy at once
            const size_t cnt = (size_t)(m->seg_begin[j + 1] - m->seg_begin[j]);
            if (!cnt) continue;
            for (int a = 0; a < 6; ++a) {
                char *dst = (char *)m->gathered[i].p[a] + (size_t)m->seg_begin[j] * esz[a];
                const hipError_t e = hipMemcpyPeerAsync(dst, m->device[i], src[(size_t)j * 9 + 2 + a], m->device[j], cnt * esz[a],
                                                        m->gathered[i].s);
                if (e != hipSuccess) { set_error("hipMemcpyPeerAsync %d -> %d failed: %s", m->device[j], m->device[i], hipGetErrorString(e)); return RT_ERR_HIP; }
            }
        }
    }
    for (int i = 0; i < m->n; ++i) {
        (void)hipSetDevice(m->device[i]);
        if (hipStreamSynchronize(m->gathered[i].s) != hipSuccess) { set_error("rt_multi_allgather: stream synchronize failed"); return RT_ERR_HIP; }
    }
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ptrs_dev)
        for (int i = 0; i < m->n; ++i)
            for (int a = 0; a < 6; ++a) ptrs_dev[(size_t)i * 6 + a] = m->gathered[i].p[a];
    return RT_SUCCESS;
}

}  // extern "C"
