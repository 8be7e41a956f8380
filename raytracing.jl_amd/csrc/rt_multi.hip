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
    // (one stream per source: copies into one destination from different sources then run concurrently, each over the
    //  xGMI link of its own pair — on a single stream they would use one link at a time)
    struct Gathered { void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap = 0; std::vector<hipStream_t> s; };
    std::vector<Gathered> gathered;
    std::vector<double> link_GBs;  // [n*n] achieved rate of the last all-gather per (destination, source) pair; 0: no copy
};

namespace {

void free_multi(rt_multi *m) {
    for (size_t i = 0; i < m->gathered.size(); ++i) {
        if (i < m->device.size()) (void)hipSetDevice(m->device[i]);
        for (void *&q : m->gathered[i].p) { if (q) (void)hipFree(q); q = nullptr; }
        for (hipStream_t st : m->gathered[i].s) if (st) (void)hipStreamDestroy(st);
    }
    for (rt_tracks *t : m->tracks) rt_tracks_destroy(t);
    for (rt_mesh *h : m->mesh) rt_mesh_destroy(h);
    delete m;
}

// Run f(i) for every shard on its own host thread; collects the first error text.  No exception leaves this function
// with a joinable thread behind (that would be std::terminate): a shard whose thread cannot be started, or whose f throws,
// is reported as failed after every started thread has been joined.
template <typename F>
bool for_each_shard(int n, F f, std::string &err) {
    std::vector<std::string> errs(n);
    std::vector<int> ok(n, 1);
    std::vector<std::thread> th;
    th.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        try {
            th.emplace_back([&, i]() {
                try {
                    if (!f(i)) { ok[i] = 0; errs[i] = rt_last_error(); }  // rt_last_error is per thread: copy it here
                } catch (const std::exception &e) {
                    ok[i] = 0; errs[i] = e.what();
                }
            });
        } catch (const std::exception &e) {  // std::system_error: no thread for this shard
            ok[i] = 0; errs[i] = std::string("could not start a host thread: ") + e.what();
        }
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (!ok[i]) { err = "shard " + std::to_string(i) + ": " + errs[i]; return false; }
    return true;
}

}  // namespace

namespace {

rt_multi *rt_multi_create_impl(const int32_t *device_ids, int32_t n_devices, const double *x, const double *y, int32_t n_nodes,
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
    struct Guard { rt_multi *p; ~Guard() { if (p) free_multi(p); } } guard{m};  // released on success (a throwing vector frees the handle)
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
        set_error("rt_multi_create: %s", err.c_str());
        return nullptr;
    }
    guard.p = nullptr;
    return m;
}


int32_t rt_multi_set_option_impl(rt_multi *m, const char *name, int64_t value) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    for (rt_mesh *h : m->mesh)
        if (int32_t rc = rt_set_option(h, name, value)) return rc;
    return RT_SUCCESS;
}

int64_t rt_multi_segmentize_impl(rt_multi *m, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
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

int32_t rt_multi_shards_impl(rt_multi *m, int64_t *uid_begin, int64_t *seg_begin) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (uid_begin) memcpy(uid_begin, m->uid_begin.data(), sizeof(int64_t) * (m->n + 1));
    if (seg_begin) {
        if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
        memcpy(seg_begin, m->seg_begin.data(), sizeof(int64_t) * (m->n + 1));
    }
    return m->n;
}

rt_tracks *rt_multi_shard_impl(rt_multi *m, int32_t i) {
    if (!m || i < 0 || i >= m->n) { set_error("rt_multi_shard: bad index"); return nullptr; }
    return m->tracks[i];
}

int32_t rt_multi_failed_tracks_impl(rt_multi *m, int64_t *n_failed, int64_t *first_uid, int32_t *first_status) {
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

int32_t rt_multi_fetch_offsets_impl(rt_multi *m, int64_t *seg_offsets, int32_t *status) {
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

int32_t rt_multi_fetch_segments_impl(rt_multi *m, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element) {
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

int32_t rt_multi_fetch_volumes_impl(rt_multi *m, double *volumes) {
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

int32_t rt_multi_allgather_impl(rt_multi *m, void **ptrs_dev, double *ms) {
    if (!m) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!m->segmentized) { set_error("rt_multi_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    const size_t total = (size_t)m->total;
    const size_t esz[6] = {8, 8, 8, 8, 8, 4};
    const int n = m->n;
    m->gathered.resize(n);
    m->link_GBs.assign((size_t)n * n, 0.0);
    // sources: every shard's device-resident records
    std::vector<void *> src((size_t)n * 9, nullptr);
    for (int j = 0; j < n; ++j)
        if (int32_t rc = rt_device_pointers(m->tracks[j], &src[(size_t)j * 9])) return rc;
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(m->device[i]) != hipSuccess) { set_error("hipSetDevice(%d) failed", m->device[i]); return RT_ERR_HIP; }
        rt_multi::Gathered &g = m->gathered[i];
        g.s.resize(n, nullptr);
        for (int j = 0; j < n; ++j)
            if (!g.s[j] && hipStreamCreateWithFlags(&g.s[j], hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return RT_ERR_HIP; }
        if (total > g.cap) {
            for (void *&q : g.p) { if (q) (void)hipFree(q); q = nullptr; }
            g.cap = 0;
            const size_t cap = total + total / 16 + 64;
            for (int a = 0; a < 6; ++a)
                if (hipMalloc(&g.p[a], cap * esz[a]) != hipSuccess) { set_error("rt_multi_allgather: hipMalloc of %zu bytes failed", cap * esz[a]); return RT_ERR_HIP; }
            g.cap = cap;
        }
        for (int j = 0; j < n; ++j)
            if (m->device[j] != m->device[i]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, m->device[i], m->device[j]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(m->device[j], 0);  // already enabled: an error we ignore
                (void)hipGetLastError();
            }
    }
    // One stream per (destination i, source j): the n - 1 blocks a device receives travel over n - 1 different xGMI links
    // at once (and the n - 1 it sends likewise) instead of one after the other on the destination's single stream.
    const auto t0 = std::chrono::steady_clock::now();
    for (int jj = 0; jj < n; ++jj)  // round jj: destination i pulls from source (i + jj) % n — every round uses n distinct links
        for (int i = 0; i < n; ++i) {
            const int j = (i + jj) % n;
            const size_t cnt = (size_t)(m->seg_begin[j + 1] - m->seg_begin[j]);
            if (!cnt) continue;
            (void)hipSetDevice(m->device[i]);
            for (int a = 0; a < 6; ++a) {
                char *dst = (char *)m->gathered[i].p[a] + (size_t)m->seg_begin[j] * esz[a];
                const hipError_t e = hipMemcpyPeerAsync(dst, m->device[i], src[(size_t)j * 9 + 2 + a], m->device[j], cnt * esz[a],
                                                        m->gathered[i].s[j]);
                if (e != hipSuccess) { set_error("hipMemcpyPeerAsync %d -> %d failed: %s", m->device[j], m->device[i], hipGetErrorString(e)); return RT_ERR_HIP; }
            }
        }
    // wait pair by pair, in issue order; the time at which a pair's stream drains gives its rate (a lower bound: the host
    // notices the end of a copy only when it gets to that stream)
    for (int jj = 0; jj < n; ++jj)
        for (int i = 0; i < n; ++i) {
            const int j = (i + jj) % n;
            const size_t cnt = (size_t)(m->seg_begin[j + 1] - m->seg_begin[j]);
            if (!cnt) continue;
            (void)hipSetDevice(m->device[i]);
            if (hipStreamSynchronize(m->gathered[i].s[j]) != hipSuccess) { set_error("rt_multi_allgather: stream synchronize failed"); return RT_ERR_HIP; }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            m->link_GBs[(size_t)i * n + j] = sec > 0 ? 44.0 * (double)cnt / sec / 1e9 : 0.0;
        }
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ptrs_dev)
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 6; ++a) ptrs_dev[(size_t)i * 6 + a] = m->gathered[i].p[a];
    return RT_SUCCESS;
}

int32_t rt_multi_link_rates_impl(rt_multi *m, double *GBs) {
    if (!m || !GBs) { set_error("null argument"); return RT_ERR_INVALID; }
    if (m->link_GBs.size() != (size_t)m->n * m->n) { set_error("rt_multi_allgather has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    memcpy(GBs, m->link_GBs.data(), sizeof(double) * m->link_GBs.size());
    return RT_SUCCESS;
}

}  // namespace

// No C++ exception may cross the C ABI (a Julia ccall or a ctypes caller would end in std::terminate): every entry point
// catches what the standard library throws (std::bad_alloc from the host vectors, std::system_error from std::thread)
// and reports it through rt_last_error.
#define RT_MULTI_GUARD(call, on_error)                         \
    try {                                                      \
        return call;                                           \
    } catch (const std::exception &e) {                        \
        set_error("%s: %s", __func__, e.what());               \
        return on_error;                                       \
    }

extern "C" {

rt_multi *rt_multi_create(const int32_t *device_ids, int32_t n_devices, const double *x, const double *y, int32_t n_nodes,
                          const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                          const int32_t *node_cells_data, const double *bb, int64_t n_tracks, const double *px,
                          const double *py, const double *phi, const double *cos_phi, const double *sin_phi, const double *A,
                          const double *B, const double *C, const double *ell, const int32_t *azim_idx) {
    RT_MULTI_GUARD(rt_multi_create_impl(device_ids, n_devices, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb,
                                        n_tracks, px, py, phi, cos_phi, sin_phi, A, B, C, ell, azim_idx), (rt_multi *)nullptr)
}
void rt_multi_destroy(rt_multi *m) {
    if (m) free_multi(m);
}
int32_t rt_multi_set_option(rt_multi *m, const char *name, int64_t value) { RT_MULTI_GUARD(rt_multi_set_option_impl(m, name, value), RT_ERR_INVALID) }
int64_t rt_multi_segmentize(rt_multi *m, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
    RT_MULTI_GUARD(rt_multi_segmentize_impl(m, tiny_step, k, rtol, delta_s, n_azim_2), (int64_t)RT_ERR_INVALID)
}
int32_t rt_multi_shards(rt_multi *m, int64_t *uid_begin, int64_t *seg_begin) { RT_MULTI_GUARD(rt_multi_shards_impl(m, uid_begin, seg_begin), RT_ERR_INVALID) }
rt_tracks *rt_multi_shard(rt_multi *m, int32_t i) { RT_MULTI_GUARD(rt_multi_shard_impl(m, i), (rt_tracks *)nullptr) }
int32_t rt_multi_failed_tracks(rt_multi *m, int64_t *n_failed, int64_t *first_uid, int32_t *first_status) {
    RT_MULTI_GUARD(rt_multi_failed_tracks_impl(m, n_failed, first_uid, first_status), RT_ERR_INVALID)
}
int32_t rt_multi_fetch_offsets(rt_multi *m, int64_t *seg_offsets, int32_t *status) { RT_MULTI_GUARD(rt_multi_fetch_offsets_impl(m, seg_offsets, status), RT_ERR_INVALID) }
int32_t rt_multi_fetch_segments(rt_multi *m, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element) {
    RT_MULTI_GUARD(rt_multi_fetch_segments_impl(m, px, py, qx, qy, ell, element), RT_ERR_INVALID)
}
int32_t rt_multi_fetch_volumes(rt_multi *m, double *volumes) { RT_MULTI_GUARD(rt_multi_fetch_volumes_impl(m, volumes), RT_ERR_INVALID) }
int32_t rt_multi_allgather(rt_multi *m, void **ptrs_dev, double *ms) { RT_MULTI_GUARD(rt_multi_allgather_impl(m, ptrs_dev, ms), RT_ERR_INVALID) }
int32_t rt_multi_link_rates(rt_multi *m, double *GBs) { RT_MULTI_GUARD(rt_multi_link_rates_impl(m, GBs), RT_ERR_INVALID) }

}  // extern "C"
