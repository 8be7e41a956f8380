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
#include "rt_internal.hpp"
#include "rt_hostpar.hpp"

#include <condition_variable>
#include <functional>
#include <unistd.h>

namespace rthost {
thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {  // shared with rt_host.cpp
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace rthost

namespace rtx {

// Wait for a stream the way a latency-bound caller wants it: hipStreamSynchronize may sleep on an interrupt and
// wake well after the last kernel ended.  Poll for the first milliseconds, then sleep.
hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) return hipStreamSynchronize(s);
    }
}

// Stream-ordered calls (option "async"): wait until the scan's host copy of the control block carries this call's sequence number.
// Falls back to waiting for the stream when the number does not arrive (a failed launch never writes it).
hipError_t wait_seq(const unsigned long long *h_res, unsigned long long seq, hipStream_t s) {
    const volatile unsigned long long *flag = h_res + rt::kCtlWords;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
            return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? hipSuccess : hipErrorUnknown;
        }
    }
}

// Every entry point that reads what a call produced first waits for a call that is still on the stream (option "async").
int finish_call(rt_tracks *t) {
    if (t && t->in_flight) {
        RT_HIP(hipSetDevice(t->mesh->device));
        RT_HIP(wait_stream(t->mesh->stream));
        t->in_flight = false;
    }
    return RT_SUCCESS;
}

}  // namespace rtx
using namespace rtx;

namespace {

inline size_t nw_all_early(size_t n) { return (n + 63) / 64; }

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
    // --- node grid, per-cell walk records and certificate margins (rt_mesh_prep.hpp)
    const double W = bb[2] - bb[0], H = bb[3] - bb[1];
    if (!(W > 0) || !(H > 0) || !std::isfinite(W) || !std::isfinite(H)) {  // also: inboundary() relies on a finite box
        set_error("empty or non-finite bounding box");
        return RT_ERR_INVALID;
    }
    const auto t_prep0 = std::chrono::steady_clock::now();
    rtprep::Prep P = rtprep::prepare(x, y, n_nodes, cn.data(), n_cells, bb);
    m->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep0).count();
    const std::vector<int32_t> &gstart = P.gstart, &gnode = P.gnode;
    const double gh = P.gh, ginv = P.ginv;
    const int gnx = P.gnx, gny = P.gny;
    hipStream_t s = m->stream;
    int rc;
    if ((rc = upload(m->x, x, n_nodes, s))) return rc;
    if ((rc = upload(m->y, y, n_nodes, s))) return rc;
    if ((rc = upload(m->cn, cn.data(), cn.size(), s))) return rc;
    if ((rc = upload(m->ncp, ncp.data(), ncp.size(), s))) return rc;
    if ((rc = upload(m->ncd, ncd.data(), (size_t)nnz, s))) return rc;
    if ((rc = upload(m->gstart, gstart.data(), gstart.size(), s))) return rc;
    if ((rc = upload(m->gnode, gnode.data(), (size_t)n_nodes, s))) return rc;
    if ((rc = upload(m->c3start, P.c3start.data(), P.c3start.size(), s))) return rc;
    if ((rc = upload(m->c3node, P.c3node.data(), P.c3node.size(), s))) return rc;
    if ((rc = upload(m->c3x, P.c3x.data(), P.c3x.size(), s))) return rc;
    if ((rc = upload(m->c3y, P.c3y.data(), P.c3y.size(), s))) return rc;
    std::vector<rt::FanEntry> fan((size_t)std::max(nnz, 1));
    for (int32_t i = 0; i < nnz; ++i) {  // node -> cells with the cells' vertices, in the table's own order
        const int32_t c = ncd[i];
        rt::FanEntry &e = fan[i];
        e.x1 = x[cn[3 * c]]; e.y1 = y[cn[3 * c]]; e.x2 = x[cn[3 * c + 1]]; e.y2 = y[cn[3 * c + 1]]; e.x3 = x[cn[3 * c + 2]]; e.y3 = y[cn[3 * c + 2]];
        e.cell = c;
        for (int q = 0; q < 3; ++q) e.adj[q] = P.adjr[(size_t)3 * c + q];
    }
    if ((rc = upload(m->fan, fan.data(), fan.size(), s))) return rc;
    if ((rc = upload(m->wrec, reinterpret_cast<const rt::WalkRec *>(P.wrec.data()), P.wrec.size(), s))) return rc;
    if ((rc = upload(m->adjr, P.adjr.data(), P.adjr.size(), s))) return rc;
    if ((rc = upload(m->trec, reinterpret_cast<const rt::TopoRec *>(P.trec.data()), P.trec.size(), s))) return rc;
    if ((rc = upload(m->etab, reinterpret_cast<const rt::EdgeABC *>(P.etab.data()), P.etab.size(), s))) return rc;
    RT_HIP(hipStreamSynchronize(s));  // host vectors die at return
    m->n_nodes = n_nodes;
    m->n_cells = n_cells;
    rt::DMesh &d = m->d;
    using rt::as_global;
    rt::DGeo g{};
    g.x = as_global(m->x.p); g.y = as_global(m->y.p); g.cn = as_global(m->cn.p); g.ncp = as_global(m->ncp.p);
    g.ncd = as_global(m->ncd.p); g.gstart = as_global(m->gstart.p); g.gnode = as_global(m->gnode.p);
    g.c3start = as_global(m->c3start.p); g.c3node = as_global(m->c3node.p); g.c3x = as_global(m->c3x.p); g.c3y = as_global(m->c3y.p);
    g.fan = as_global((const rt::FanEntry *)m->fan.p);
    g.gx0 = bb[0]; g.gy0 = bb[1]; g.gh = gh; g.ginv = ginv; g.gnx = gnx; g.gny = gny;
    g.n_nodes = n_nodes;
    if ((rc = upload(m->geo, &g, 1, s))) return rc;
    RT_HIP(hipStreamSynchronize(s));
    d.geo = (const RT_K rt::DGeo *)m->geo.p;
    d.bx0 = bb[0]; d.by0 = bb[1]; d.bx1 = bb[2]; d.by1 = bb[3];
    d.n_cells = n_cells;
    d.wrec = as_global(m->wrec.p); d.adjr = as_global(m->adjr.p); d.d_vertex = P.d_vertex; d.l_min = P.l_min;
    d.walk_ok = P.walk_ok ? 1 : 0;
    d.trec = as_global((const rt::TopoRec *)m->trec.p); d.etab = as_global((const rt::EdgeABC *)m->etab.p);
    m->topo_available = P.walk_ok && P.topo_ok;
    m->topo_tiny_max = P.topo_tiny_max; m->topo_rmax = P.topo_rmax; m->topo_end_err = P.topo_end_err; m->tally_a = P.tally_a; m->tally_b = P.tally_b;
    m->n_records_topo = P.n_records_topo;
    m->walk_available = P.walk_ok;
    m->kappa = P.kappa;
    m->prep_note = P.note;
    m->n_records = P.n_records; m->n_records_walk = P.n_records_walk;
    m->n_cells_fragile = P.n_cells_fragile; m->n_cells_wild = P.n_cells_wild;
    m->n_edges_nonmanifold = P.n_edges_nonmanifold; m->extras_max = P.extras_max;
    m->eps_min = P.eps_min; m->eps_max = P.eps_max;
    return RT_SUCCESS;
}

void free_mesh(rt_mesh *m) {
    m->x.release(); m->y.release(); m->cn.release(); m->ncp.release(); m->ncd.release();
    m->gstart.release(); m->gnode.release(); m->c3start.release(); m->c3node.release(); m->c3x.release(); m->c3y.release(); m->fan.release(); m->wrec.release(); m->adjr.release(); m->trec.release(); m->etab.release(); m->geo.release();
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    for (hipEvent_t &e : m->side_ev) if (e) (void)hipEventDestroy(e);
    delete m;
}

void pin_release_to_cache(rt_tracks *t);  // defined with rt_fetch_segments_pinned

void free_tracks(rt_tracks *t) {
    t->px.release(); t->py.release(); t->phi.release(); t->cs.release(); t->sn.release();
    t->A.release(); t->B.release(); t->C.release(); t->ell.release(); t->azim.release(); t->perm.release(); t->corder.release();
    t->in_arena.release(); t->cnt_slot.release(); t->off_slot.release(); t->w_slot.release();
    t->counts.release(); t->status.release(); t->element.release(); t->offsets.release();
    t->tile_sums.release(); t->tile_acc.release(); t->ctl.release(); t->vacc.release();
#ifdef RT_TIMING
    t->dbg.release();
#endif
    if (t->h_ctl) (void)hipHostFree(t->h_ctl);
    if (t->cq_started) (void)hipHostFree(t->cq_started);
    if (t->pin_off) (void)hipHostFree(t->pin_off);
    if (t->pin_st) (void)hipHostFree(t->pin_st);
    pin_release_to_cache(t);
    t->spx.release(); t->spy.release(); t->sqx.release(); t->sqy.release(); t->sell.release();
    t->volumes.release(); t->volumes_prev.release(); t->delta_s.release(); t->tau.release(); t->sigma_t.release();
    t->gpx.release(); t->gpy.release(); t->gqx.release(); t->gqy.release();
    t->gelement.release(); t->ctab.release(); t->cowner.release();
    t->sw_src.release(); t->sw_w.release(); t->sw_xs.release(); t->sw_psi_in.release(); t->sw_psi_out.release(); t->sw_phi.release();
    t->sw_ell.release(); t->sw_cell.release();
    t->side_px.release(); t->side_py.release(); t->side_qx.release(); t->side_qy.release(); t->side_el.release(); t->marg.release();
    t->vorder.release(); t->vw_wave.release(); t->vw_k.release(); t->w_base.release(); t->w_P.release();
    t->s_el.release(); t->s_eq.release(); t->p_count.release(); t->p_flags.release(); t->p_valid.release(); t->p_rel.release();
    t->s_px.release(); t->s_py.release(); t->s_qx.release(); t->s_qy.release(); t->s_ell.release(); t->p_sum.release();
    for (auto &e : t->ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

}  // namespace

// Page-locked staging blocks for the upload of a track set: kept process-wide, one per concurrent caller (rt_multi_create uploads
// its shards from several threads), of ONE fixed size.  A track set larger than half a block goes up in ranges through the block's
// two halves — the host writes one half while the other is in flight — so that no call pays for page-locking its whole input (0.08 ms
// per MB: 11 ms for a BWR assembly's 90 MB, the whole of round 3's upload again) and the page-locked memory a process holds is bounded.
namespace {
constexpr size_t kStageBytes = 32u << 20;
// A block belongs to the device that was current when it was made: its two events can only be recorded on that device's streams
// (an event of device A on a stream of device B is hipErrorInvalidHandle), so a caller gets a block of ITS device or a new one.
struct StagingBlock { void *p = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy = false; int device = -1; };
std::vector<StagingBlock> g_staging;
std::mutex g_staging_mutex;
bool staging_new_block(StagingBlock &b, int device) {
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    b.device = device;
    if (hipHostMalloc(&b.p, kStageBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return false; }
    for (auto &e : b.ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(b.p); b.p = nullptr; return false; }
    return true;
}
// The runtime creates its copy-engine queues lazily: the FIRST device-to-host copy that finds the engine it would have used busy
// is given another engine, whose queue is created on the spot — 6.5-7.5 ms inside that hipMemcpyAsync (and ≈4,100 page faults: a
// 16-MB ring).  A pipelined fetch keeps two copies in flight, so the first fetch of a process paid that in its second or third
// piece — `e2e.one_shot_ms` 17-22 ms instead of 10.5 in the driver's bench lines of rounds 4-6 (profiles/r06/exp_one_shot_spread.log).
// Here, on the thread that page-locks the process's first staging block beside rt_mesh_create: a burst of overlapping copies in
// both directions through the block, so that the queues exist before anybody waits for them.  (RT_NO_ENGINE_WARMUP=1: off.)
static void warm_copy_engines(const StagingBlock &b) {
    if (getenv("RT_NO_ENGINE_WARMUP")) return;
    void *d = nullptr;
    const size_t piece = (size_t)4 << 20, n = kStageBytes / piece;
    if (hipMalloc(&d, kStageBytes) != hipSuccess) { (void)hipGetLastError(); return; }
    hipStream_t st[2] = {nullptr, nullptr};
    bool ok = hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking) == hipSuccess;
    // host to device first (the device block gets defined contents), then device to host — what a fetch does —, each direction as
    // eight 4-MB copies on two streams at once, twice
    for (int round = 0; ok && round < 4; ++round) {
        for (size_t i = 0; ok && i < n; ++i) {
            if (round == 0) ok = hipMemcpyAsync((char *)d + i * piece, (const char *)b.p + i * piece, piece, hipMemcpyHostToDevice, st[i & 1]) == hipSuccess;
            else ok = hipMemcpyAsync((char *)b.p + i * piece, (const char *)d + i * piece, piece, hipMemcpyDeviceToHost, st[i & 1]) == hipSuccess;
        }
        for (hipStream_t q : st) if (q) (void)hipStreamSynchronize(q);
    }
    for (hipStream_t q : st) if (q) (void)hipStreamDestroy(q);
    (void)hipFree(d);
    (void)hipGetLastError();
}

static void warm_copy_engines_once(const StagingBlock &b) {  // once per device and process
    static std::mutex m;
    static std::vector<int> done;
    {
        std::lock_guard<std::mutex> lk(m);
        if (std::find(done.begin(), done.end(), b.device) != done.end()) return;
        done.push_back(b.device);
    }
    warm_copy_engines(b);
}

// The process's first block is page-locked (≈1.4 ms) by a thread that rt_mesh_create starts — a mesh always precedes its track sets,
// and its own preprocessing and upload take longer than that — so that the first rt_tracks_create does not wait for it.
struct StagingPrefetch {
    std::thread th;
    std::once_flag once;
    std::mutex join_m;  // (wait() before start() must not use up the join: the thread started later would never be joined)
    void start(int device) {
        std::call_once(once, [&] {
            std::lock_guard<std::mutex> lk(join_m);
            try {
                th = std::thread([device] {
                    StagingBlock b;
                    if (!staging_new_block(b, device)) return;
                    memset(b.p, 0, kStageBytes);  // (the host's first touch of its pages, here rather than in the first upload)
                    warm_copy_engines_once(b);
                    std::lock_guard<std::mutex> lk(g_staging_mutex);
                    g_staging.push_back(b);
                });
            } catch (...) {
            }
        });
    }
    // (rt_multi_create reaches this from several host threads at once: the join happens once, the others wait for it)
    void wait() {
        std::lock_guard<std::mutex> lk(join_m);
        if (th.joinable()) th.join();
    }
    ~StagingPrefetch() { wait(); }
} g_staging_prefetch;
int staging_acquire(StagingBlock *out, int device) {
    g_staging_prefetch.wait();
    {
        std::lock_guard<std::mutex> lk(g_staging_mutex);
        for (size_t i = 0; i < g_staging.size(); ++i)
            if (!g_staging[i].busy && g_staging[i].device == device) { g_staging[i].busy = true; *out = g_staging[i]; return (int)i; }
    }
    StagingBlock b;
    if (!staging_new_block(b, device)) return -1;  // (page-locking takes a millisecond: not under the lock)
    warm_copy_engines_once(b);  // (a device's first block that the prefetch thread did not make: rt_multi's other devices)
    b.busy = true;
    std::lock_guard<std::mutex> lk(g_staging_mutex);
    g_staging.push_back(b);
    *out = b;
    return (int)g_staging.size() - 1;
}
void staging_release(int slot) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_staging_mutex);
    g_staging[(size_t)slot].busy = false;
}
}  // namespace


// Device arrays into the caller's (pageable, often freshly allocated) host arrays at close to the PCIe rate: pieces through the two
// halves of a page-locked block — the copy engine fills one half while the host's threads move the other into place (and take the
// page faults of a fresh destination in parallel).  A plain hipMemcpy into pageable memory runs at a third of the link's rate (C3's
// 410 MB of records: 25-42 ms, C5's 5 GB: 330-600 ms), and page-locking the whole result first (rt_fetch_pinned's first call) costs
// as much.  dst[a] == NULL: skipped.
static int fetch_pipelined(rt_tracks *t, int n_arrays, const void *const *src, void *const *dst, const size_t *bytes,
                           rthostpar::ResultBlock *blk = nullptr) {
    hipStream_t s = t->mesh->stream;
    StagingBlock stage;
    const int slot = staging_acquire(&stage, t->mesh->device);
    // (an early return may leave a copy into the block in flight: the stream is drained before the block goes back to the pool)
    struct Rel { int slot; hipStream_t st; bool ok; ~Rel() { if (!ok && slot >= 0) (void)hipStreamSynchronize(st); staging_release(slot); } } rel{slot, s, false};
    if (slot < 0) {  // no page-locked block to be had
        for (int a = 0; a < n_arrays; ++a)
            if (dst[a] && bytes[a]) RT_HIP(hipMemcpy(dst[a], src[a], bytes[a], hipMemcpyDeviceToHost));
        rel.ok = true;
        return RT_SUCCESS;
    }
    const size_t half = kStageBytes / 2;
    // A destination the CALLER allocated (blk == nullptr) is usually fresh: the copying threads take its page faults.  Transparent
    // huge pages are asked for once per array (option "fetch_hugepages", default on: a no-op behind numpy, −40 % behind an allocator
    // that does not ask) — and taken back for what is still to come when a piece's copy stalls (below 2 GB/s: 2-MB faults that wait
    // for compaction; 4-KB faults in parallel are then the faster way).  A block of the library's own (blk) is faulted in by its
    // threads, in address order: a piece is copied once the front has passed it.
    const bool huge_hint = !blk && t->mesh->fetch_hugepages != 0;
    bool huge_stalled = false;
    struct Piece { char *d; size_t bytes; int h; char *rest; size_t rest_bytes; };
    Piece prev{nullptr, 0, 0, nullptr, 0};
    // development (RT_RESULT_TIMING=1): where the fetch's time goes — waiting for a piece to arrive, moving it into place
    static const bool ftime = getenv("RT_RESULT_TIMING") != nullptr;
    double f_wait = 0, f_copy = 0, f_wait_max = 0, f_copy_max = 0, f_enq = 0, f_enq_max = 0;
    int f_pieces = 0, f_enq_max_at = -1, f_enq_max_what = 0;
    auto fnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto drain = [&](const Piece &pc) -> int {  // the piece has arrived in its half: into place
        if (!pc.d) return RT_SUCCESS;
        const double tw0 = ftime ? fnow() : 0.0;
        RT_HIP(hipEventSynchronize(stage.ev[pc.h]));
        if (ftime) { const double d = fnow() - tw0; f_wait += d; f_wait_max = std::max(f_wait_max, d); ++f_pieces; }
        const char *hb = (const char *)stage.p + (size_t)pc.h * half;
        if (blk) blk->wait_front((size_t)(pc.d + pc.bytes - blk->base));
        const auto t0 = std::chrono::steady_clock::now();
        rthostpar::copy_into_place(pc.d, hb, pc.bytes);
        if (ftime) { const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); f_copy += d; f_copy_max = std::max(f_copy_max, d); }
        if (huge_hint && !huge_stalled && pc.bytes >= ((size_t)4 << 20)) {
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if ((double)pc.bytes / sec < 2.0e9) {
                huge_stalled = true;
                rthostpar::unhint_huge_pages(pc.rest, pc.rest_bytes);
            }
        }
        return RT_SUCCESS;
    };
    int k = 0;
    for (int a = 0; a < n_arrays; ++a) {
        if (!dst[a]) continue;
        if (huge_hint && !huge_stalled) rthostpar::hint_huge_pages((char *)dst[a], bytes[a]);  // (one madvise per destination array)
        for (size_t o = 0; o < bytes[a]; o += half, ++k) {
            const size_t nbp = std::min(half, bytes[a] - o);
            const int h = k & 1;
            // (half h was drained two pieces ago: `prev` is the piece in the OTHER half)
            const double te0 = ftime ? fnow() : 0.0;
            RT_HIP(hipMemcpyAsync((char *)stage.p + (size_t)h * half, (const char *)src[a] + o, nbp, hipMemcpyDeviceToHost, s));
            const double te1 = ftime ? fnow() : 0.0;
            RT_HIP(hipEventRecord(stage.ev[h], s));
            if (ftime) {
                const double te2 = fnow();
                f_enq += te2 - te0;
                if (te1 - te0 > f_enq_max) { f_enq_max = te1 - te0; f_enq_max_at = k; f_enq_max_what = 1; }
                if (te2 - te1 > f_enq_max) { f_enq_max = te2 - te1; f_enq_max_at = k; f_enq_max_what = 2; }
            }
            if (int rc = drain(prev)) return rc;
            if (huge_stalled && huge_hint) rthostpar::unhint_huge_pages((char *)dst[a] + o, bytes[a] - o);  // (a later array of a stalled fetch)
            prev = Piece{(char *)dst[a] + o, nbp, h, (char *)dst[a] + o + nbp, bytes[a] - o - nbp};
        }
    }
    const int rc = drain(prev);
    rel.ok = rc == RT_SUCCESS;
    if (ftime)
        fprintf(stderr, "[rt fetch] %d pieces: enqueue %.2f ms (longest call %.2f ms: %s of piece %d), waited for arrivals %.2f ms (longest %.2f), moved into place %.2f ms (longest %.2f)\n",
                f_pieces, f_enq, f_enq_max, f_enq_max_what == 1 ? "hipMemcpyAsync" : "hipEventRecord", f_enq_max_at, f_wait, f_wait_max, f_copy, f_copy_max);
    return rc;
}

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

static rt_mesh *mesh_create_impl(int32_t device, const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes,
                                 int32_t n_cells, const int32_t *node_cells_ptrs, const int32_t *node_cells_data, const double *bb);
static rt_tracks *tracks_create_impl(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py, const double *phi,
                                     const double *cos_phi, const double *sin_phi, const double *A, const double *B, const double *C,
                                     const double *ell, const int32_t *azim_idx);
static int64_t segmentize_impl(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2);

// No C++ exception may cross the C ABI (a Julia ccall or a ctypes caller would end in std::terminate): the entry points
// that allocate host memory catch what the standard library throws and report it through rt_last_error.
rt_mesh *rt_mesh_create(int32_t device, const double *x, const double *y, int32_t n_nodes,
                        const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                        const int32_t *node_cells_data, const double *bb) {
    try {
        return mesh_create_impl(device, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb);
    } catch (const std::exception &e) {
        set_error("rt_mesh_create: %s", e.what());
        return nullptr;
    }
}
rt_tracks *rt_tracks_create(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py, const double *phi,
                            const double *cos_phi, const double *sin_phi, const double *A, const double *B, const double *C,
                            const double *ell, const int32_t *azim_idx) {
    try {
        return tracks_create_impl(mesh, n_tracks, px, py, phi, cos_phi, sin_phi, A, B, C, ell, azim_idx);
    } catch (const std::exception &e) {
        set_error("rt_tracks_create: %s", e.what());
        return nullptr;
    }
}
int64_t rt_segmentize(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
    try {
        return segmentize_impl(t, tiny_step, k, rtol, delta_s, n_azim_2);
    } catch (const std::exception &e) {
        set_error("rt_segmentize: %s", e.what());
        return RT_ERR_INVALID;
    }
}

static rt_mesh *mesh_create_impl(int32_t device, const double *x, const double *y, int32_t n_nodes,
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
    g_staging_prefetch.start(device);  // (the page-locked block of this process's track uploads, beside the preprocessing below)
    rt_mesh *m = new rt_mesh();
    struct Guard { rt_mesh *p; ~Guard() { if (p) free_mesh(p); } } guard{m};  // released on success
    m->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->n_cus = cus;
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0) m->lds_per_block = lds;
    }
    if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        return nullptr;
    }
    m->stream = m->own_stream;
    // (the lean plan's k_serve runs beside k_cheap on a second stream; without it, behind k_cheap)
    if (hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking) != hipSuccess) m->side_stream = nullptr;
    for (hipEvent_t &e : m->side_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    if (!m->side_ev[0] || !m->side_ev[1]) { if (m->side_stream) (void)hipStreamDestroy(m->side_stream); m->side_stream = nullptr; }
    if (build_mesh(m, x, y, n_nodes, cell_nodes, n_cells, node_cells_ptrs, node_cells_data, bb) != RT_SUCCESS) return nullptr;
    // development knob: RT_OPTIONS="name=value,name=value" applies rt_set_option at creation
    if (const char *env = getenv("RT_OPTIONS")) {
        std::string e(env);
        size_t pos = 0;
        while (pos < e.size()) {
            size_t end = e.find(',', pos);
            if (end == std::string::npos) end = e.size();
            const std::string kv = e.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) (void)rt_set_option(m, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1));
            pos = end + 1;
        }
    }
    guard.p = nullptr;
    return m;
}

void rt_mesh_destroy(rt_mesh *mesh) {
    if (!mesh) return;
    (void)hipSetDevice(mesh->device);
    free_mesh(mesh);
}

int32_t rt_mesh_set_stream(rt_mesh *mesh, void *hip_stream) {
    if (!mesh) { set_error("null mesh"); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(mesh->device));
    RT_HIP(hipStreamSynchronize(mesh->stream));  // (a stream-ordered call, option "async", may still be on the old stream)
    mesh->stream = hip_stream ? (hipStream_t)hip_stream : mesh->own_stream;
    return RT_SUCCESS;
}
void *rt_mesh_get_stream(rt_mesh *mesh) { return mesh ? (void *)mesh->stream : nullptr; }

int32_t rt_mesh_set_enqueue_hook(rt_mesh *mesh, rt_enqueue_hook hook, void *user) {
    if (!mesh) { set_error("rt_mesh_set_enqueue_hook: null mesh"); return RT_ERR_INVALID; }
    mesh->enqueue_hook = hook;
    mesh->enqueue_hook_user = user;
    return RT_SUCCESS;
}

int32_t rt_set_option(rt_mesh *mesh, const char *name, int64_t value) {
    if (!mesh || !name) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!strcmp(name, "iter_cap")) { mesh->iter_cap = value > 0 ? value : 4000000; return RT_SUCCESS; }
#ifdef RT_EXPERIMENTAL
    // the two-pass march (count, scan, march again writing at the CSR offsets; fill_volumes with global atomics or as its own
    // pass): round 1's first correct path, 1.8x slower than the single pass — kept as an independent cross-check of the staging /
    // compaction machinery in builds with -DRT_EXPERIMENTAL (tests/test_gpu_experimental_build.py)
    if (!strcmp(name, "volumes_mode")) { mesh->volumes_mode = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "single_pass")) { mesh->single_pass = value != 0; return RT_SUCCESS; }
#else
    if (!strcmp(name, "volumes_mode") || !strcmp(name, "single_pass")) {
        set_error("option '%s' needs a library built with -DRT_EXPERIMENTAL", name);
        return RT_ERR_INVALID;
    }
#endif
    if (!strcmp(name, "split")) { mesh->split = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "fuse_volumes")) { mesh->fuse_volumes = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "compact")) { mesh->compact = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_gp")) { mesh->sweep_gp = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_ell")) { mesh->sweep_ell = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "fetch_hugepages")) { mesh->fetch_hugepages = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_rows")) { mesh->sweep_rows = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RT_SUCCESS; }
    if (!strcmp(name, "sweep_waves")) { mesh->sweep_waves = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "sweep_debug")) { mesh->sweep_debug = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "compact_debug")) { mesh->compact_debug = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "mat_kernel")) { mesh->mat_kernel = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "march_waves")) { mesh->march_waves = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "topo")) { mesh->topo = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RT_SUCCESS; }
    if (!strcmp(name, "record_order")) { mesh->record_order = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RT_SUCCESS; }
    if (!strcmp(name, "lean")) { mesh->lean = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RT_SUCCESS; }
    if (!strcmp(name, "serve_blocks")) { mesh->serve_blocks = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "cheap_per_cu")) { mesh->cheap_per_cu = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "timing")) { mesh->timing = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "async")) { mesh->async_calls = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "pool_chunks_hint")) { mesh->pool_chunks_hint = value; return RT_SUCCESS; }
    if (!strcmp(name, "test_out_records")) { mesh->test_out_records = value; return RT_SUCCESS; }
    if (!strcmp(name, "test_volumes_fallback")) { mesh->test_volumes_fallback = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "test_exact_sums")) { mesh->test_exact_sums = (int)value; return RT_SUCCESS; }
    if (!strcmp(name, "test_tally_tau")) { mesh->test_tally_tau = value; return RT_SUCCESS; }
    if (!strcmp(name, "fused_scan")) { mesh->fused_scan = value != 0; return RT_SUCCESS; }
    if (!strcmp(name, "test_reserved_pct")) { mesh->test_reserved_pct = value; return RT_SUCCESS; }  // read by rt_tracks_create
    if (!strcmp(name, "side_entries_hint")) { mesh->side_entries_hint = value; return RT_SUCCESS; }
    if (!strcmp(name, "sort_mode")) { mesh->sort_mode = (int)value; return RT_SUCCESS; }  // read by rt_tracks_create
    if (!strcmp(name, "walk")) {  // 0: generic step only (literal emulation), 1: certified walk step + generic fallback
        mesh->d.walk_ok = (value != 0 && mesh->walk_available) ? 1 : 0;
        return RT_SUCCESS;
    }
    set_error("unknown option '%s'", name);
    return RT_ERR_INVALID;
}

static rt_tracks *tracks_create_impl(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py,
                            const double *phi, const double *cos_phi, const double *sin_phi, const double *A,
                            const double *B, const double *C, const double *ell, const int32_t *azim_idx) {
    if (!mesh || n_tracks < 0 || n_tracks > 0x7fffffff ||
        (n_tracks > 0 && (!px || !py || !phi || !cos_phi || !sin_phi || !A || !B || !C || !ell || !azim_idx))) {
        set_error("rt_tracks_create: null pointer or bad track count");
        return nullptr;
    }
    if (hipSetDevice(mesh->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    rt_tracks *t = new rt_tracks();
    struct Guard { rt_tracks *p; ~Guard() { if (p) free_tracks(p); } } guard{t};  // released on success
    t->mesh = mesh;
    t->n = n_tracks;
    hipStream_t s = mesh->stream;
    const size_t n = (size_t)n_tracks;
    // RT_CREATE_TIMING=1: where the call's host time goes, to stderr (development)
    const bool ctime = getenv("RT_CREATE_TIMING") != nullptr;
    auto cnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double cstamp[6] = {cnow(), 0, 0, 0, 0, 0};
    // march order, reserved staging chunks, compaction order: host-only code (rt_hostpar.hpp — also built under the sanitizers)
    rthostpar::MarchPlan plan;
    rthostpar::plan_march_order(ell, n, mesh->sort_mode, mesh->kappa, mesh->test_reserved_pct, rt::kStaticRegions, rt::kChunkRows, plan);
    std::vector<int32_t> &perm = plan.perm;
    std::vector<int32_t> &h_corder = plan.corder;
    static_assert(rt::kStaticRegions <= (int)(sizeof(plan.reg_cap) / sizeof(plan.reg_cap[0])), "MarchPlan::reg_cap holds every region");
    for (int j = 0; j < rt::kStaticRegions; ++j) t->reg_cap[j] = plan.reg_cap[j];
    {
        // Σℓ in a fixed order (blocks of 4096 tracks, added in block order) whatever the number of threads; range of azim_idx
        const size_t nb = (n + 4095) / 4096;
        std::vector<double> bs(nb, 0.0);
        std::vector<int32_t> bmin(nb, 0x7fffffff), bmax(nb, (int32_t)0x80000000);
        rthostpar::par_ranges(nb, 16, [&](size_t b0, size_t b1) {
            for (size_t b = b0; b < b1; ++b) {
                double sl = 0.0;
                int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
                for (size_t i = b * 4096; i < std::min(n, b * 4096 + 4096); ++i) { sl += ell[i]; lo = std::min(lo, azim_idx[i]); hi = std::max(hi, azim_idx[i]); }
                bs[b] = sl; bmin[b] = lo; bmax[b] = hi;
            }
        });
        for (size_t b = 0; b < nb; ++b) { t->sum_ell += bs[b]; t->azim_min = std::min(t->azim_min, bmin[b]); t->azim_max = std::max(t->azim_max, bmax[b]); }
    }
    if (n > 0) {
        if (t->azim_min < 1) {  // δs[azim_idx] is read on the device (fill_volumes, src/trackgenerator.jl:379-382)
            set_error("rt_tracks_create: azim_idx must be 1-based (smallest value %d)", t->azim_min);
            return nullptr;
        }
    }
    // split plan: pieces per wave of 64 consecutive uids, canonical numbering, dispatch order
    std::vector<int32_t> h_vorder, h_vw_wave, h_vw_k, h_w_base, h_w_P;
    // Splitting pays when the batch has far too few waves to fill the chip (the march is then bound by its
    // longest dependent chain).  Pieces march with exact steps (k_march<SPLIT>), whole tracks with the two-phase march's cheap
    // steps since round 4 — and against THAT march the pieces win only below ≈150 waves (same box, ms per call whole / in pieces,
    // profiles/r05/exp_split_threshold.log: 7 waves 0.127 / 0.116, 26 waves 0.184 / 0.131, 103 waves (C2) 0.181 / 0.160;
    // 204 waves 0.167 / 0.218, 510 waves 0.161 / 0.272, 1,019 waves — C3 on two GPUs — 0.199 / 0.310, 1,530 waves 0.259 / 0.448).
    // Rounds 2-4 split every batch below 1,536 waves, a rule measured against the exact-step march of round 2.
    const size_t nw_all = (n + 63) / 64;
    const bool auto_split = mesh->split < 0 && nw_all < 160;
    const int p_auto = auto_split ? (int)std::min<size_t>(16, (2048 + nw_all - 1) / std::max<size_t>(1, nw_all)) : 1;
    if ((mesh->split > 0 || (auto_split && p_auto > 1)) && n > 0) {
        const size_t nw = (n + 63) / 64;
        h_w_base.resize(nw); h_w_P.resize(nw);
        std::vector<double> piece_len;
        int32_t nv = 0;
        for (size_t w = 0; w < nw; ++w) {
            double lmax = 0.0;
            for (size_t l = 0; l < 64 && w * 64 + l < n; ++l) lmax = std::max(lmax, ell[w * 64 + l]);
            const double est = lmax * mesh->kappa;  // expected segments of the longest track of the wave
            int32_t P = mesh->split > 0 ? std::min(8, (int32_t)std::ceil(est / (double)mesh->split))
                                        : std::min(p_auto, (int32_t)(est / 12.0));  // auto: pieces of >= ~12 segments
            P = std::max(1, P);
            if (est > 0.5 * rt::kMaxIter) P = 1;  // MAX_ITER (src/track.jl:104) counts whole tracks
            h_w_base[w] = nv; h_w_P[w] = P;
            for (int32_t k = 0; k < P; ++k) { h_vw_wave.push_back((int32_t)w); h_vw_k.push_back(k); piece_len.push_back(lmax / P); }
            nv += P;
        }
        h_vorder.resize(nv);
        std::iota(h_vorder.begin(), h_vorder.end(), 0);
        std::stable_sort(h_vorder.begin(), h_vorder.end(), [&](int32_t a, int32_t b) { return piece_len[a] > piece_len[b]; });
        t->n_vwaves = nv;
    }
    // One device allocation, one page-locked staging block (kept process-wide), one host-to-device copy: the eleven pageable
    // uploads into eleven allocations of round 3 were 31.7 ms of a C5 call whose kernels take 3.
    bool ok = true;
    cstamp[1] = cnow();
    {
        const size_t na = (n + 31) & ~(size_t)31;  // every array starts on a 256-B boundary
        const size_t ncord = (h_corder.size() + 63) & ~(size_t)63;
        // the arena: what goes up — nine double arrays, azim_idx, the march order, the materialise order — and behind it what a small
        // kernel derives on the device (the lines' coefficients, lengths, directions, start points, angles and azimuthal indices in march-slot order, the inverse
        // of the march order: 84 B per track that need not cross PCIe)
        const size_t up_bytes = 9 * na * sizeof(double) + 2 * na * sizeof(int32_t) + ncord * sizeof(int32_t);
        const size_t bytes = up_bytes + 9 * na * sizeof(double) + 2 * na * sizeof(int32_t) + 256;
        StagingBlock stage;
        const int slot = staging_acquire(&stage, mesh->device);
        // (a failed upload may leave a copy out of the block in flight: the stream is drained before the block goes back to the pool)
        struct Rel { int slot; hipStream_t st; const bool *ok; ~Rel() { if (!*ok && slot >= 0) (void)hipStreamSynchronize(st); staging_release(slot); } } rel{slot, s, &ok};
        ok = t->in_arena.reserve(bytes) == hipSuccess;
        unsigned char *db = t->in_arena.p;
        const double *src8[9] = {px, py, phi, cos_phi, sin_phi, A, B, C, ell};
        DevView<double> *dst8[9] = {&t->px, &t->py, &t->phi, &t->cs, &t->sn, &t->A, &t->B, &t->C, &t->ell};
        const size_t ints_off = 9 * na * sizeof(double);
        if (ok) {
            for (int a = 0; a < 9; ++a) dst8[a]->p = (double *)(db + (size_t)a * na * sizeof(double));
            t->azim.p = (int32_t *)(db + ints_off); t->perm.p = t->azim.p + na;
            t->corder.p = h_corder.empty() ? nullptr : t->perm.p + na;
            t->As.p = (double *)(db + up_bytes); t->Bs.p = t->As.p + na; t->Cs.p = t->Bs.p + na; t->Ls.p = t->Cs.p + na;
            t->Dxs.p = t->Ls.p + na; t->Dys.p = t->Dxs.p + na;
            t->Pxs.p = t->Dys.p + na; t->Pys.p = t->Pxs.p + na; t->Phis.p = t->Pys.p + na;
            t->iperm.p = (int32_t *)(t->Phis.p + na); t->Azs.p = t->iperm.p + na;
        }
        cstamp[2] = cnow();
        if (ok && n > 0) {
            if (slot >= 0 && up_bytes <= kStageBytes / 2) {
                // everything fits in half a block: the image of the uploaded part as it lies in the arena, ONE copy
                unsigned char *hb = (unsigned char *)stage.p;
                int32_t *h_az = (int32_t *)(hb + ints_off), *h_pm = h_az + na, *h_co = h_pm + na;
                rthostpar::pack_tracks_image(hb, na, 0, n, src8, azim_idx, perm.data());
                (void)h_az; (void)h_pm;
                if (!h_corder.empty()) memcpy(h_co, h_corder.data(), h_corder.size() * sizeof(int32_t));
                cstamp[3] = cnow();
                ok = hipMemcpyAsync(db, hb, up_bytes, hipMemcpyHostToDevice, s) == hipSuccess;
            } else if (slot >= 0) {
                // ranges of tracks through the block's two halves: eleven slices per range, written by the host threads while the
                // previous range's copies are in flight
                const size_t half = kStageBytes / 2, per_track = 9 * sizeof(double) + 2 * sizeof(int32_t);
                const size_t rcap = (half / per_track) & ~(size_t)63;
                int k = 0;
                for (size_t i0 = 0; i0 < n && ok; i0 += rcap, ++k) {
                    const size_t i1 = std::min(n, i0 + rcap), m = i1 - i0;
                    unsigned char *hb = (unsigned char *)stage.p + (size_t)(k & 1) * half;
                    if (k >= 2) ok = hipEventSynchronize(stage.ev[k & 1]) == hipSuccess;  // the half's previous range has left it
                    int32_t *h_az = (int32_t *)(hb + 9 * rcap * sizeof(double)), *h_pm = h_az + rcap;
                    rthostpar::pack_tracks_image(hb, rcap, i0, m, src8, azim_idx, perm.data());
                    for (int a = 0; a < 9 && ok; ++a)
                        ok = hipMemcpyAsync(dst8[a]->p + i0, hb + (size_t)a * rcap * sizeof(double), m * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
                    ok = ok && hipMemcpyAsync(t->azim.p + i0, h_az, m * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess &&
                         hipMemcpyAsync(t->perm.p + i0, h_pm, m * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess &&
                         hipEventRecord(stage.ev[k & 1], s) == hipSuccess;
                }
                cstamp[3] = cnow();
                if (ok && !h_corder.empty())
                    ok = hipMemcpyAsync(t->corder.p, h_corder.data(), h_corder.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess;
            } else {
                // no page-locked block to be had: from where the arrays lie
                cstamp[3] = cnow();
                for (int a = 0; a < 9 && ok; ++a) ok = hipMemcpyAsync(dst8[a]->p, src8[a], n * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
                ok = ok && hipMemcpyAsync(t->azim.p, azim_idx, n * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess &&
                     hipMemcpyAsync(t->perm.p, perm.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess;
                if (ok && !h_corder.empty())
                    ok = hipMemcpyAsync(t->corder.p, h_corder.data(), h_corder.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess;
            }
            if (ok) {
                rt::DTracks du{};  // (the uploaded arrays, uid order)
                du.px = rt::as_global(t->px.p); du.py = rt::as_global(t->py.p); du.phi = rt::as_global(t->phi.p); du.cs = rt::as_global(t->cs.p);
                du.sn = rt::as_global(t->sn.p); du.A = rt::as_global(t->A.p); du.B = rt::as_global(t->B.p); du.C = rt::as_global(t->C.p);
                du.ell = rt::as_global(t->ell.p); du.azim = rt::as_global(t->azim.p); du.perm = rt::as_global(t->perm.p);
                du.n = (int64_t)n;
                rtx::launch_slot_arrays(s, (int64_t)n, du, t->As.p, t->Bs.p, t->Cs.p, t->Ls.p, t->Dxs.p, t->Dys.p, t->Pxs.p, t->Pys.p, t->Phis.p, t->Azs.p, t->iperm.p);
            }
            ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            cstamp[4] = cnow();
        }
        ok = ok && t->cnt_slot.reserve(na + 64) == hipSuccess && t->off_slot.reserve(na + 64) == hipSuccess && t->w_slot.reserve(na + 64) == hipSuccess;
    }
    if (ok && t->n_vwaves > 0) {
        const size_t np = (size_t)t->n_vwaves * 64;
        ok = upload(t->vorder, h_vorder.data(), h_vorder.size(), s) == 0 && upload(t->vw_wave, h_vw_wave.data(), h_vw_wave.size(), s) == 0 &&
             upload(t->vw_k, h_vw_k.data(), h_vw_k.size(), s) == 0 && upload(t->w_base, h_w_base.data(), h_w_base.size(), s) == 0 &&
             upload(t->w_P, h_w_P.data(), h_w_P.size(), s) == 0 && t->s_el.reserve(np) == hipSuccess && t->s_eq.reserve(np) == hipSuccess &&
             t->p_count.reserve(np) == hipSuccess && t->p_flags.reserve(np) == hipSuccess && t->p_valid.reserve(np) == hipSuccess &&
             t->p_rel.reserve(np) == hipSuccess && t->s_px.reserve(np) == hipSuccess && t->s_py.reserve(np) == hipSuccess &&
             t->s_qx.reserve(np) == hipSuccess && t->s_qy.reserve(np) == hipSuccess && t->s_ell.reserve(np) == hipSuccess &&
             t->p_sum.reserve(np) == hipSuccess;
    }
    for (auto &e : t->ev)
        if (ok && hipEventCreate(&e) != hipSuccess) ok = false;
    if (ok && hipStreamSynchronize(s) != hipSuccess) ok = false;
    if (!ok) {
        if (g_last_error.empty()) set_error("rt_tracks_create: upload failed");
        return nullptr;
    }
    rt::DTracks &d = t->d;
    using rt::as_global;
    d.px = as_global(t->px.p); d.py = as_global(t->py.p); d.phi = as_global(t->phi.p); d.cs = as_global(t->cs.p);
    d.sn = as_global(t->sn.p); d.A = as_global(t->A.p); d.B = as_global(t->B.p); d.C = as_global(t->C.p);
    d.ell = as_global(t->ell.p); d.azim = as_global(t->azim.p); d.perm = as_global(t->perm.p);
    d.As = as_global(t->As.p); d.Bs = as_global(t->Bs.p); d.Cs = as_global(t->Cs.p); d.Ls = as_global(t->Ls.p); d.Dxs = as_global(t->Dxs.p); d.Dys = as_global(t->Dys.p); d.iperm = as_global(t->iperm.p);
    d.Pxs = as_global(t->Pxs.p); d.Pys = as_global(t->Pys.p); d.Phis = as_global(t->Phis.p); d.Azs = as_global(t->Azs.p);
    d.cnt_slot = as_global(t->cnt_slot.p); d.off_slot = as_global(t->off_slot.p); d.w_slot = as_global(t->w_slot.p);
    d.n = n_tracks;
    guard.p = nullptr;
    if (ctime)
        fprintf(stderr, "[rt_tracks_create] %lld tracks: order + plan %.3f ms, arena + staging %.3f, host image %.3f, copy + sync %.3f, rest %.3f\n", (long long)n_tracks,
                cstamp[1] - cstamp[0], cstamp[2] - cstamp[1], cstamp[3] - cstamp[2], cstamp[4] - cstamp[3], cnow() - cstamp[4]);
    return t;
}

void rt_tracks_destroy(rt_tracks *tracks) {
    if (!tracks) return;
    (void)hipSetDevice(tracks->mesh->device);
    (void)finish_call(tracks);
    free_tracks(tracks);
}

#ifdef RT_HOST_TIMING
static double g_ht[6];
static long g_hn;
static inline double ht_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif

namespace {

// One rt_segmentize call (segmentize!, src/trackgenerator.jl:357-369): its arguments, the plan chosen for it, the state of its
// attempts and what it leaves on the handle.  segmentize_impl (below) is the driver: begin -> choose_plan -> [single pass:
// estimate_pools, then per attempt grow_pools, bind_stage, enqueue_attempt, wait_attempt, after_attempt] -> finish.
struct SegmentizeCall {
    // ---- arguments
    rt_tracks *const t;
    const double tiny_step;
    const int32_t k;
    const double rtol;
    const double *const delta_s;
    const int32_t n_azim_2;
    SegmentizeCall(rt_tracks *t_, double tiny_, int32_t k_, double rtol_, const double *ds_, int32_t na_)
        : t(t_), tiny_step(tiny_), k(k_), rtol(rtol_), delta_s(ds_), n_azim_2(na_) {}

    // ---- begin(): handle, sizes, parameters, the call's control block
    rt_mesh *m = nullptr;
    hipStream_t s = nullptr;
    int64_t n = 0, n_tiles = 0, n_waves = 0;
    rt::DParams prm{};
    int cb = 0;  // calls alternate between two control blocks; the scan of a call resets the other one for the next call
    unsigned long long *d_ctl = nullptr, *d_ctl_other = nullptr, *d_fail = nullptr, *h_res = nullptr, *h_res_dev = nullptr;
    int64_t *d_total = nullptr;
    int32_t *d_cursor = nullptr;
    bool ctl_was_clean = false, vacc_was_clean = false;
    int64_t ctl_was_first = -1;
    rt::DOut out{};
    rt::DStage stg{};
    rt::DSplit sp{};
    // ---- choose_plan()
    bool widek = false;       // find_element's knn fallback beyond the in-register list: separate kernel instantiations
    bool fuse = false;        // fill_volumes in the march's LDS copy of `volumes`
    bool do_compact = true;   // the call writes the 44-B records
    bool split = false;       // track pieces (DSplit)
    bool topo = false;        // cheap steps: the two-phase march
    bool lean = false;        // ... in three kernels: k_first, k_cheap, k_serve (DLean)
    int cheap_waves = 4;      // waves per workgroup of k_cheap
    bool completion = false;  // the records in completion order, written beside the march (DStage::cq)
    uint32_t cq_epoch = 0;
    rt::DLean dl{};
    int fuse_waves = 4;
    size_t hist_bytes = 0, fuse_smem = 0;
    const int32_t *corder = nullptr;  // compaction order of the whole-track waves
    // ---- estimate_pools() / grow_pools()
    int64_t want = 0, side_want = 0, side_static = 0;
    // ---- one attempt
    int32_t first_chunk = 0, side_first = 0;
    int64_t reset_key = 0;
    bool need_reset = false;
    int32_t *tile_acc_cur = nullptr, *tile_acc_other = nullptr;  // (set for two-phase calls)
    rt::DStage stg_pieces{};
    rt::DTracks d_whole{};
    bool fused_volumes = false;  // fill_volumes rode along with the march (until a recovery path recomputes it from the records)
    bool volumes_pass = true;    // false: no separate pass, no ev[6]
    // ---- results
    int64_t total = 0;
    unsigned long long fi[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t cur[4] = {0, 0, 0, 0};  // pool cursor, pool overflow / argument mismatch, side-list cursor, side-list overflow
#ifdef RT_HOST_TIMING
    double ht0 = 0, ht1 = 0, ht2 = 0;
#endif
    enum After { kDone = 0, kRetry = 1, kRestartWhole = 2 };

    // HIP events between the kernels (rt_last_timing) only on request: each costs ≈4 µs of stream time
    int rec(int i) {
        if (m->timing) RT_HIP(hipEventRecord(t->ev[i], s));
        return RT_SUCCESS;
    }

    int begin() {
#ifdef RT_HOST_TIMING
        ht0 = ht_now();
#endif
        if (!t || !delta_s || n_azim_2 <= 0) { set_error("rt_segmentize: bad arguments"); return RT_ERR_INVALID; }
        if (k < 0) {  // knn(kdtree, x, k, ...) rejects a negative k (src/mesh.jl:123); any k >= 0 is honoured
            set_error("rt_segmentize: k = %d (must be >= 0)", k);
            return RT_ERR_INVALID;
        }
        if (t->n > 0 && t->azim_max > n_azim_2) {
            set_error("rt_segmentize: track azim_idx reaches %d but delta_s has n_azim_2 = %d entries", t->azim_max, n_azim_2);
            return RT_ERR_INVALID;
        }
        m = t->mesh;
        RT_HIP(hipSetDevice(m->device));
        s = m->stream;
        n = t->n;
        t->segmentized = false;
        t->tau_groups = 0;  // τ of the previous records is void
        for (double &v : t->ms) v = 0.0;

        prm.tiny_step = tiny_step; prm.rtol = rtol; prm.k = k; prm.n_azim_2 = n_azim_2; prm.iter_cap = m->iter_cap;
        prm.topo_tiny_max = m->topo_tiny_max; prm.topo_rmax = m->topo_rmax; prm.topo_end_err = m->topo_end_err;
        prm.topo_force = m->topo == 2 ? 1 : 0; prm.pad_ = 0;
        {
            // the chord of a cheap record is used for fill_volumes where its error bound a + b/D_x (rt_mesh_prep.hpp) is at most ε of the
            // chord: ε/8 for a, 7ε/8 for b.  "test_tally_tau" (tests, A/B): ε in 1e-12; < 0: every cheap record is tallied from its length
            const double eps = m->test_tally_tau > 0 ? 1e-12 * (double)m->test_tally_tau : 8e-11;
            prm.tally_c1 = m->test_tally_tau < 0 ? (double)INFINITY : m->tally_a / (0.125 * eps);
            prm.tally_c2 = m->test_tally_tau < 0 ? (double)INFINITY : m->tally_b / (0.875 * eps);
        }

        n_tiles = (n + rt::kScanTile - 1) / rt::kScanTile;
        n_waves = (n + 63) / 64;
        RT_HIP(t->counts.reserve(n + 1));
        RT_HIP(t->status.reserve(n + 1));
        RT_HIP(t->offsets.reserve(n + 1));
        RT_HIP(t->tile_sums.reserve(n_tiles + 1));
        if (t->tile_acc_tiles != n_tiles + 1 || !t->tile_acc.p) {  // (two zeroed halves: k_march<TOPO> adds to one, the scan clears the other)
            RT_HIP(t->tile_acc.reserve(2 * (size_t)(n_tiles + 1) * rt::kTileAccStride));
            RT_HIP(hipMemsetAsync(t->tile_acc.p, 0, 2 * (size_t)(n_tiles + 1) * rt::kTileAccStride * sizeof(int32_t), s));
            t->tile_acc_tiles = n_tiles + 1;
            t->tile_acc_clean[0] = t->tile_acc_clean[1] = true;
        }
        RT_HIP(t->ctl.reserve(2 * rt::kCtlWords));
        RT_HIP(t->vacc.reserve(m->n_cells));
        if (!t->h_ctl) {
            RT_HIP(hipHostMalloc((void **)&t->h_ctl, (2 * rt::kCtlWords + 8) * sizeof(unsigned long long), hipHostMallocDefault));
            for (int i = 0; i < 2 * rt::kCtlWords + 8; ++i) t->h_ctl[i] = 0;  // (h_res[kCtlWords]: the sequence number of the call it holds)
            t->h_ctl[1] = ~0ull;  // first failing uid: atomicMin target
        }
        // the call's control block: calls alternate between two, and the scan of a call resets the other one for the next call —
        // in the steady state no reset kernel runs in front of the march (its launch gap was 5 µs of every step)
        cb = t->ctl_idx;
        d_ctl = t->ctl.p + (size_t)cb * rt::kCtlWords;
        d_ctl_other = t->ctl.p + (size_t)(1 - cb) * rt::kCtlWords;
        ctl_was_clean = t->ctl_clean[cb];
        ctl_was_first = t->ctl_first_chunk[cb];
        vacc_was_clean = t->vacc_clean;
        t->ctl_clean[0] = t->ctl_clean[1] = false;  // (set again when this call has succeeded)
        t->vacc_clean = false;
        d_fail = d_ctl;
        d_total = reinterpret_cast<int64_t *>(d_ctl + 16);
        d_cursor = reinterpret_cast<int32_t *>(d_ctl + 18);
        h_res = t->h_ctl + rt::kCtlWords;
        // the same pinned block as the device sees it (k_scan_tile_sums writes it); looked up once per handle
        if (!t->h_res_dev) RT_HIP(hipHostGetDevicePointer((void **)&t->h_res_dev, h_res, 0));
        h_res_dev = t->h_res_dev;
        std::swap(t->volumes, t->volumes_prev);  // a consumer may still be all-reducing the previous call's volumes
        RT_HIP(t->volumes.reserve(m->n_cells));
        if (t->h_delta_s.size() != (size_t)n_azim_2 || memcmp(t->h_delta_s.data(), delta_s, sizeof(double) * n_azim_2) != 0) {
            t->h_delta_s.assign(delta_s, delta_s + n_azim_2);
            if (int rc = upload(t->delta_s, t->h_delta_s.data(), (size_t)n_azim_2, s)) return rc;
        }
        using rt::as_global;
        out.volumes = as_global(t->volumes.p);  // (single pass with fused fill_volumes: the accumulator `vacc`, see enqueue_attempt)
        out.delta_s = as_global(t->delta_s.p);
        out.fused_volumes = (m->volumes_mode == 1 && !m->single_pass) ? 1 : 0;
        out.dbg = m->compact_debug;
        t->compacted = false;
        t->completion_order = false;
        t->sw_ell_valid = false;
        t->sw_rowsc_valid = false;
        t->cplan = rt_tracks::CompactPlan{};
        t->last_split = 0;
        t->last_record_kernel = 0;
        return RT_SUCCESS;
    }

    // Which march this call runs: track pieces (DSplit) for every wave of a batch too small to fill the chip — a call that cannot
    // use the plan (once a track reached MAX_ITER segments) marches every track whole —, cheap steps (k_march<..., TOPO>) for
    // whole-track batches on meshes with cheap-step records, the usual k and fill_volumes fused.
    void choose_plan() {
        using rt::as_global;
        widek = k > rt::kMaxK;
        hist_bytes = (size_t)m->n_cells * sizeof(double);
        // The march fits three waves per SIMD (12 per CU): four-wave workgroups when three copies of the LDS histogram fit in the
        // CU's 160 KB, six-wave workgroups (two copies) for larger meshes (six-wave workgroups measured -10 % march time on a
        // batch that is resident at once, BWR-like C4, and +5 % on one that takes many rounds, C5 on one GPU).
        fuse_waves = (3 * (hist_bytes + 4 * rt::kMaxChunks * sizeof(int32_t)) <= 158 * 1024 || (n + 63) / 64 > 3072) ? 4 : 6;
        if (m->march_waves == 4 || m->march_waves == 6) fuse_waves = m->march_waves;  // (experiments)
        // fill_volumes fused into the march when an LDS copy of `volumes` (+ the chunk tables) leaves room for two workgroups per
        // CU; larger meshes use the separate k_volumes pass.  A wide k marches with the one-wave kernels only: fewer
        // instantiations of a rare case.
        fuse = m->volumes_mode == 2 && m->fuse_volumes && 2 * (hist_bytes + fuse_waves * rt::kMaxChunks * sizeof(int32_t)) <= 158 * 1024 && !widek;
        fuse_smem = hist_bytes + fuse_waves * rt::kMaxChunks * sizeof(int32_t);
        // Option "compact" = 0: stop after march + scan (a device-resident consumer, rt_sweep, reads the staged rows); the separate
        // volumes pass needs the compact records, so a call that cannot fuse fill_volumes compacts anyway.  Whole tracks only.
        do_compact = m->compact || !fuse || !m->single_pass;
        split = m->single_pass && t->n_vwaves > 0 && !t->force_unsplit && do_compact;  // pieces are marched in this call
        topo = m->single_pass && m->topo && m->topo_available && m->d.walk_ok && !split && !widek && n > 0 && fuse && tiny_step > 0 &&
               tiny_step <= m->topo_tiny_max && (m->topo == 2 || 10 * m->n_records_topo >= 9 * m->n_records_walk);
        t->last_topo = topo ? 1 : 0;
        // The lean plan: the cheap loop in a kernel of its own at four waves per SIMD (16 per CU) — as many workgroups per CU as
        // LDS copies of `volumes` fit: 4 x 4 waves, 2 x 8 or 1 x 16
        lean = false;
        if (topo && m->lean && !t->lean_gave_up) {
            for (int cw : {4, 8, 16}) {
                const size_t per = hist_bytes + (size_t)cw * rt::kMaxChunks * sizeof(int32_t);
                if ((size_t)(16 / cw) * per <= 158 * 1024) { lean = true; cheap_waves = cw; break; }
            }
        }
        t->last_lean = lean ? m->lean : 0;
        // Records in completion order: the record kernel beside the march, on the mesh's second stream.  Automatic ("record_order"
        // 1) for batches whose march workgroups are all resident at once — where the march's second half leaves the chip half idle
        // (profiles/r06/exp_fused_tail.log) —, "record_order" 2 for any two-phase call that writes records.  Not with events between
        // the kernels (option "timing"), not with calls that return early ("async").
        {
            const int64_t march_blocks = (n_waves + fuse_waves - 1) / fuse_waves;
            const int64_t resident = (int64_t)m->n_cus * (fuse_waves == 4 ? 2 : 1);  // (218 VGPRs: two waves per SIMD)
            completion = topo && !lean && do_compact && m->record_order > 0 && !t->completion_gave_up && m->side_stream && !m->timing &&
                         !m->async_calls && m->mat_kernel != 1 && n > 0 && march_blocks < (1 << 30) / (4 * fuse_waves) &&
                         (m->record_order == 2 || march_blocks <= resident);
        }
        if (split) {
            sp.vorder = as_global(t->vorder.p); sp.vw_wave = as_global(t->vw_wave.p); sp.vw_k = as_global(t->vw_k.p);
            sp.w_base = as_global(t->w_base.p); sp.w_P = as_global(t->w_P.p);
            sp.s_el = as_global(t->s_el.p); sp.s_eq = as_global(t->s_eq.p);
            sp.s_px = as_global(t->s_px.p); sp.s_py = as_global(t->s_py.p); sp.s_qx = as_global(t->s_qx.p);
            sp.s_qy = as_global(t->s_qy.p); sp.s_ell = as_global(t->s_ell.p);
            sp.p_count = as_global(t->p_count.p); sp.p_flags = as_global(t->p_flags.p); sp.p_sum = as_global(t->p_sum.p);
            sp.p_valid = as_global(t->p_valid.p); sp.p_rel = as_global(t->p_rel.p);
            sp.n_vwaves = t->n_vwaves;
        }
        // compaction order of the whole-track waves (only when every track marches whole with the full march order)
        corder = (t->corder.p && m->single_pass && !t->n_vwaves) ? (const int32_t *)t->corder.p : (const int32_t *)nullptr;
    }

    // copy_out: k_scan_tile_sums' last block also writes the control block to the pinned host copy; scale: k_scan_write also
    // applies volumes ./= n_azim_2 (fused fill_volumes only: `volumes` is final once the march has ended); reset_other: the
    // scan's last block also resets the OTHER control block for the next call (single-pass calls)
    // write_slots = false (records in completion order): off_slot holds the tracks' places in completion order — the scan leaves it alone
    int scan_counts(bool copy_out, bool scale, bool reset_other, bool slot_order = false, bool write_slots = true) {
        if (n > 0 && slot_order && tile_acc_cur) {  // (two-phase calls: the march has left the tile sums)
            ++t->call_seq;
            launch_scan_fused(s, t, n_tiles, d_ctl, tile_acc_cur, reset_other ? tile_acc_other : (int32_t *)nullptr,
                              reset_other ? d_ctl_other : (unsigned long long *)nullptr, first_chunk, side_first, write_slots);
            if (reset_other) t->tile_acc_clean[1 - cb] = true;
        } else if (n > 0) {
            launch_scan(s, t, n_tiles, d_ctl, copy_out ? h_res_dev : (unsigned long long *)nullptr,
                        reset_other ? d_ctl_other : (unsigned long long *)nullptr, first_chunk, side_first, ++t->call_seq,
                        scale ? t->volumes.p : (double *)nullptr, (double)n_azim_2, slot_order && write_slots);
        } else {
            RT_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), s));
            RT_HIP(hipMemsetAsync(t->offsets.p, 0, sizeof(int64_t), s));
        }
        return RT_SUCCESS;
    }

    // fill_volumes as its own pass over the compact records + volumes ./= n_azim_2
    int launch_volumes() {
        if (m->volumes_mode == 2 && n > 0 && !fused_volumes)
            if (int rc = launch_volumes_pass(s, t, m->single_pass ? (const int32_t *)(d_cursor + 1) : (const int32_t *)nullptr, out.cap)) return rc;
        if (!(fused_volumes && n > 0))  // the fused path scales inside k_scan_write
            launch_scale_volumes(s, t->volumes.p, m->n_cells, (double)n_azim_2);
        return RT_SUCCESS;
    }
    // the separate pass again, from the (now complete) records: a recovery path's last step
    int recompute_volumes() {
        RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
        return launch_volumes();
    }

    int march(int mode, int waves, bool pieces, bool wide, bool cheap, unsigned blocks, size_t smem, const rt::DTracks &tracks,
              const rt::DStage &stage, const int64_t *offsets = nullptr, int phase = 0, hipStream_t on = nullptr) {
        t->last_march_waves = waves; t->last_split = std::max(t->last_split, pieces ? 1 : 0); t->last_widek = wide ? 1 : 0;
        return launch_march(mode, waves, pieces, wide, cheap, blocks, smem, on ? on : s, m->d, tracks, prm, t->counts.p, t->status.p, offsets, out,
                            stage, d_fail, sp, phase, phase ? &dl : nullptr);
    }

    // The lean plan's three kernels (DLean): k_first on the call's stream, then k_cheap — and k_serve behind it on the same stream
    // ("lean" 1) or beside it on the mesh's second stream ("lean" 2: it starts when k_first has ended and ends when k_cheap has and
    // the queue is empty; the call's stream waits for it before the scan)
    int march_lean() {
        const unsigned blocks1 = (unsigned)((n_waves + fuse_waves - 1) / fuse_waves);
        if (int rc = march(rt::kStage, fuse_waves, false, false, true, blocks1, fuse_smem, d_whole, stg, nullptr, 1)) return rc;
        const bool beside = m->lean == 2 && m->side_stream;
        if (beside) {
            RT_HIP(hipEventRecord(m->side_ev[0], s));
            RT_HIP(hipStreamWaitEvent(m->side_stream, m->side_ev[0], 0));
        }
        size_t smem_c = hist_bytes + (size_t)cheap_waves * rt::kMaxChunks * sizeof(int32_t);
        {
            // A batch of one residency round or less: as many workgroups per CU as it takes to use every CU, not as many as fit —
            // the dispatcher fills a CU before it moves on, and 510 workgroups at four per CU leave half the chip idle.  (LDS is the
            // one resource a launch can ask more of than it needs.)
            const int fit = 16 / cheap_waves;
            int per_cu = (int)std::min<int64_t>(fit, (dl.n_cheap_wgs + m->n_cus - 1) / std::max(1, m->n_cus));
            if (m->cheap_per_cu > 0) per_cu = std::min(fit, m->cheap_per_cu);
            per_cu = std::max(1, per_cu);
            if (per_cu < fit) smem_c = std::max(smem_c, (size_t)(158 * 1024 / per_cu) & ~(size_t)255);
            smem_c = std::min<size_t>(smem_c, (size_t)m->lds_per_block);
        }
        if (int rc = launch_cheap(cheap_waves, (unsigned)dl.n_cheap_wgs, smem_c, s, m->d, d_whole, prm, t->counts.p, t->status.p, out, stg, d_ctl, dl)) return rc;
        const unsigned blocks2 = (unsigned)std::max(1, m->serve_blocks > 0 ? m->serve_blocks : 64);
        if (int rc = march(rt::kStage, fuse_waves, false, false, true, blocks2, fuse_smem, d_whole, stg, nullptr, 2, beside ? m->side_stream : s)) return rc;
        if (beside) {
            RT_HIP(hipEventRecord(m->side_ev[1], m->side_stream));
            RT_HIP(hipStreamWaitEvent(s, m->side_ev[1], 0));
        }
        return RT_SUCCESS;
    }

    // ---- staged single-pass march: the pool is sized from the Cauchy–Crofton estimate (or from what the previous call needed)
    //      and grown + re-run on overflow
    void estimate_pools() {
        const bool split_all = split;  // every wave in pieces (small batches, or "split" > 0)
        want = t->chunks_needed_last > 0
                   ? t->chunks_needed_last + t->chunks_needed_last / 16 + 16
                   : (int64_t)(1.3 * (m->kappa * t->sum_ell + (double)n) / (64.0 * rt::kChunkRows)) + 2 * ((split ? t->n_vwaves : 0) + (split_all ? 0 : n_waves)) + 64;
        if (!split) {  // (the reserved regions of the whole-track march and some room behind them)
            int64_t res = 0;
            for (int j = 0; j < rt::kStaticRegions; ++j) res += t->reg_cap[j];
            want = std::max(want, res + res / 8 + 64);
        }
        if (m->pool_chunks_hint > 0 && t->pool_chunks == 0) want = m->pool_chunks_hint;
        fused_volumes = fuse;
        // the side list of the two-phase march: one reserved entry per march slot (a track's first record) + the records the
        // generic step makes further on — estimated from the share of records without a walk certificate (or the last call's need)
        side_static = n_waves * 64;
        side_want = 0;
        if (topo) {
            const double unwalked = m->n_records > 0 ? 1.0 - (double)m->n_records_walk / (double)m->n_records : 1.0;
            const int64_t dyn = t->side_needed_last > 0 ? t->side_needed_last + t->side_needed_last / 8 + 1024
                                                        : (int64_t)(1.3 * unwalked * m->kappa * t->sum_ell) + n / 16 + 4096;
            side_want = side_static + ((m->side_entries_hint > 0 && t->side_cap == 0) ? m->side_entries_hint : dyn);
        }
    }

    int grow_pools() {
        RT_HIP(t->ctab.reserve((size_t)std::max<int64_t>(1, split ? t->n_vwaves : n_waves) * rt::kMaxChunks));
        if (want > t->pool_chunks || (!topo && t->gqx.cap < (size_t)t->pool_chunks * rt::kChunkRows * 64)) {
            want = std::max(want, t->pool_chunks);
            const size_t slots = (size_t)want * rt::kChunkRows * 64;
            // (q, ±cell) rows with sparse p for the exact march; the two-phase march stages one 4-B word per record
            if (!topo) {
                RT_HIP(t->gpx.reserve(slots)); RT_HIP(t->gpy.reserve(slots)); RT_HIP(t->gqx.reserve(slots)); RT_HIP(t->gqy.reserve(slots));
            }
            RT_HIP(t->gelement.reserve(slots));
            RT_HIP(t->cowner.reserve((size_t)want));
            t->pool_chunks = want;
        }
        if (lean) {
            const size_t ns = (size_t)n_waves * 64;
            if (t->lean_i.cap < 9 * ns + 1024) {
                RT_HIP(t->lean_i.reserve(9 * ns + 1024));
                RT_HIP(t->lean_d.reserve(8 * ns));
                t->lean_q_clean = false;
            }
        }
        if (topo && side_want > t->side_cap) {
            const size_t ne = (size_t)std::min<int64_t>(side_want, 0x7ffffff0);
            RT_HIP(t->side_px.reserve(ne)); RT_HIP(t->side_py.reserve(ne)); RT_HIP(t->side_qx.reserve(ne)); RT_HIP(t->side_qy.reserve(ne));
            RT_HIP(t->side_el.reserve(ne));
            t->side_cap = (int64_t)ne;
        }
        // The six output arrays are sized from the Cauchy–Crofton estimate of the record count (or from what the
        // previous call produced), not from the pool's slots: march -> scan -> compaction still run back to back
        // without a host sync — the compaction simply does not write beyond the capacity, and in the rare call
        // whose total exceeds it the host grows the arrays and compacts again (the staged rows are still there).
        const int64_t est_records = t->total_last > 0 ? t->total_last + t->total_last / 32 + 4096
                                                      : (int64_t)(1.08 * m->kappa * t->sum_ell) + 2 * n + 4096;
        if (do_compact)
            if (int rc = reserve_records(t, std::min<int64_t>(m->test_out_records > 0 && t->total_last == 0 ? m->test_out_records : est_records,
                                                              t->pool_chunks * rt::kChunkRows * 64), out)) return rc;
        return RT_SUCCESS;
    }

    // The staging pool, its reserved regions, the side list and the tile sums as the kernels see them (DStage), and the plan the
    // compaction may need later (CompactPlan: option "compact" = 0)
    int bind_stage(int attempt) {
        using rt::as_global;
        stg.px = as_global(t->gpx.p); stg.py = as_global(t->gpy.p); stg.qx = as_global(t->gqx.p);
        stg.qy = as_global(t->gqy.p); stg.element = as_global(t->gelement.p);
        stg.ctab = as_global(t->ctab.p); stg.cowner = as_global(t->cowner.p); stg.cursor = as_global(d_cursor);
        stg.pool_chunks = (int32_t)std::min<int64_t>(t->pool_chunks, 0x7fffffff);
        // reserved chunks of the whole-track march (DStage): region j holds chunk j of the first reg_cap[j] march waves
        int64_t reserved = 0;
        stg.n_regions = 0;
        if (!split) {
            for (int j = 0; j < rt::kStaticRegions && t->reg_cap[j] > 0; ++j) {
                if (reserved + t->reg_cap[j] >= stg.pool_chunks) break;
                stg.reg_cap[j] = t->reg_cap[j]; stg.reg_base[j] = (int32_t)reserved;
                reserved += t->reg_cap[j];
                stg.n_regions = j + 1;
            }
        }
        for (int j = stg.n_regions; j < rt::kStaticRegions; ++j) stg.reg_cap[j] = stg.reg_base[j] = 0;
        stg.tile_acc = nullptr;
        if (topo && m->fused_scan) {
            tile_acc_cur = t->tile_acc.p + (size_t)cb * (size_t)(n_tiles + 1) * rt::kTileAccStride;
            tile_acc_other = t->tile_acc.p + (size_t)(1 - cb) * (size_t)(n_tiles + 1) * rt::kTileAccStride;
            stg.tile_acc = as_global(tile_acc_cur);
        }
        if (topo) {
            stg.s_px = as_global(t->side_px.p); stg.s_py = as_global(t->side_py.p); stg.s_qx = as_global(t->side_qx.p);
            stg.s_qy = as_global(t->side_qy.p); stg.s_el = as_global(t->side_el.p);
            stg.side_cap = (int32_t)t->side_cap; stg.side_static = (int32_t)side_static;
        }
#ifdef RT_TIMING
        RT_HIP(t->dbg.reserve((size_t)std::max<int64_t>(1, n_waves) * 4));
        RT_HIP(hipMemsetAsync(t->dbg.p, 0, sizeof(unsigned long long) * 4 * std::max<int64_t>(1, n_waves), s));
        stg.dbg = t->dbg.p;
#endif
        if (lean) {
            const size_t ns = (size_t)n_waves * 64;
            int32_t *ip = t->lean_i.p;
            double *dp = t->lean_d.p;
            dl.pred = as_global(ip); dl.last = as_global(ip + ns); dl.i = as_global(ip + 2 * ns); dl.it = as_global(ip + 3 * ns);
            dl.fl = as_global(ip + 4 * ns); dl.word = as_global(ip + 5 * ns); dl.prev_el = as_global(ip + 6 * ns); dl.wk_last = as_global(ip + 7 * ns);
            dl.queue = as_global(ip + 8 * ns); dl.dump = as_global(ip + 9 * ns);
            dl.sp = as_global(dp); dl.sn = as_global(dp + ns); dl.ttP = as_global(dp + 2 * ns); dl.ttN = as_global(dp + 3 * ns);
            dl.ttp = as_global(dp + 4 * ns); dl.dprev = as_global(dp + 5 * ns); dl.lqx = as_global(dp + 6 * ns); dl.lqy = as_global(dp + 7 * ns);
            dl.qctl = as_global(reinterpret_cast<int32_t *>(d_ctl + rt::kLeanCtl));
            dl.n_cheap_wgs = (int32_t)((n_waves + cheap_waves - 1) / cheap_waves);
            dl.pad_ = 0;
            if (!t->lean_q_clean || attempt > 0) {  // (k_serve leaves every entry it consumed at -1: clean again after a complete call)
                RT_HIP(hipMemsetAsync(t->lean_i.p + 8 * ns, 0xff, ns * sizeof(int32_t), s));
                t->lean_q_clean = true;
            }
        }
        stg.cq = nullptr; stg.tab_off = nullptr; stg.cq_started = nullptr; stg.cq_epoch = 0; stg.cq_blocks = 0;
        if (completion) {
            const int64_t march_blocks = (n_waves + fuse_waves - 1) / fuse_waves;
            RT_HIP(t->tab_off.reserve((size_t)n + 1));
            if (t->cq.cap < 8 * (size_t)march_blocks) {
                RT_HIP(t->cq.reserve(8 * (size_t)march_blocks));
                RT_HIP(hipMemsetAsync(t->cq.p, 0, t->cq.cap * sizeof(unsigned long long), s));  // (epoch 0 is never used)
            }
            if (!t->cq_started) {
                RT_HIP(hipHostMalloc((void **)&t->cq_started, 8 * sizeof(unsigned long long), hipHostMallocDefault));
                for (int i = 0; i < 8; ++i) t->cq_started[i] = 0;
            }
            // the epoch of this attempt: entries and start flags of earlier calls / attempts never match it
            cq_epoch = (uint32_t)(++t->cq_epoch_last);
            if (cq_epoch == 0) cq_epoch = (uint32_t)(++t->cq_epoch_last);
            unsigned long long *started_dev = nullptr;
            RT_HIP(hipHostGetDevicePointer((void **)&started_dev, t->cq_started, 0));
            stg.cq = as_global(t->cq.p); stg.tab_off = as_global(t->tab_off.p); stg.cq_started = started_dev;
            stg.cq_epoch = cq_epoch; stg.cq_blocks = (int32_t)march_blocks;
        }
        stg_pieces = stg;
        d_whole = t->d;
        {
            rt_tracks::CompactPlan &c = t->cplan;
            c.stg = stg; c.stg_pieces = stg_pieces; c.d_whole = d_whole; c.sp = sp; c.corder = corder;
            c.n_whole_waves = n_waves; c.split = split; c.split_all = split; c.staged = false;
            c.codes = topo; c.rtol = rtol; c.march_waves = fuse_waves;
        }
        // fused fill_volumes accumulates into `vacc` (zero between calls: k_scan_write leaves it so); otherwise the separate
        // pass adds into `volumes`, zeroed by the prologue.  The reset kernel runs only when the control block or the accumulator
        // is not known to be clean: a handle's first call, a re-run after a pool overflow, a changed number of reserved chunks.
        first_chunk = (int32_t)reserved;
        side_first = topo ? (int32_t)side_static : 0;
        reset_key = (int64_t)first_chunk | ((int64_t)side_first << 32);
        if (fuse && n > 0) out.volumes = as_global(t->vacc.p);
        need_reset = attempt > 0 || !ctl_was_clean || ctl_was_first != reset_key || !(fuse && n > 0 && vacc_was_clean);
        return RT_SUCCESS;
    }

    // Everything one attempt puts on the stream, back to back.  (Capturing it once into a HIP graph and replaying it was tried:
    // the event-record nodes keep the ≈6-µs gaps between the kernels, and hipEventElapsedTime fails on events that were only
    // ever recorded inside a graph — EXPERIMENTS.md §A.)
    int enqueue_attempt() {
        if (need_reset)
            launch_prologue(s, d_ctl, (fuse && n > 0) ? t->vacc.p : t->volumes.p, m->n_cells, first_chunk, side_first);
        // (the tile sums' half of this control block: clean when the previous two-phase call's scan has cleared it — not after a
        //  void attempt, and not when a call without cheap steps came in between)
        if (tile_acc_cur && (need_reset || !t->tile_acc_clean[cb]))
            RT_HIP(hipMemsetAsync(tile_acc_cur, 0, (size_t)(n_tiles + 1) * rt::kTileAccStride * sizeof(int32_t), s));
        if (tile_acc_cur) t->tile_acc_clean[cb] = false;
        if (int rc = rec(1)) return rc;
        const size_t one_wave_smem = rt::kMaxChunks * sizeof(int32_t);
        if (n > 0 && split) {  // pieces (batches that leave the chip underfilled: four-wave workgroups whatever the mesh size)
            launch_seed(widek, (unsigned)t->n_vwaves, s, m->d, t->d, prm, sp);
            int rc;
            if (fuse) rc = march(rt::kStage, 4, true, false, false, (unsigned)((t->n_vwaves + 3) / 4), hist_bytes + 4 * rt::kMaxChunks * sizeof(int32_t), t->d, stg_pieces);
            else rc = march(rt::kStage, 1, true, widek, false, (unsigned)t->n_vwaves, one_wave_smem, t->d, stg_pieces);
            if (rc) return rc;
            launch_resolve((unsigned)((n + 255) / 256), s, t->d, prm, sp, t->counts.p, t->status.p, d_fail);
        }
        if (n > 0 && !split) {  // whole tracks
            int rc;
            if (topo && lean) rc = march_lean();
            else if (topo) rc = march(rt::kStage, fuse_waves, false, false, true, (unsigned)((n_waves + fuse_waves - 1) / fuse_waves), fuse_smem, d_whole, stg);
            else if (fuse) rc = march(rt::kStage, fuse_waves, false, false, false, (unsigned)((n_waves + fuse_waves - 1) / fuse_waves), fuse_smem, d_whole, stg);
            else rc = march(rt::kStage, 1, false, widek, false, (unsigned)n_waves, one_wave_smem, d_whole, stg);
            if (rc) return rc;
        }
        if (completion) return enqueue_completion();
        if (int rc = rec(2)) return rc;
        if (int rc = scan_counts(!topo, fuse && !topo, true, topo)) return rc;  // (two-phase: k_finish scales the volumes, behind k_materialise)
        if (int rc = rec(3)) return rc;  // every event record costs ≈4 µs of stream time: none is recorded twice
        if (topo) {
            // codes -> records (or, "compact" = 0, (ℓ, cell) rows) + Σℓ / status; k_finish completes them and copies the control
            // block to the host
            if (int rc = launch_materialise(t, out, s, do_compact, !do_compact, true, d_ctl)) return rc;
            launch_finish(t, out, s, !do_compact, fuse, (double)n_azim_2, d_ctl, h_res_dev, t->call_seq);
        } else if (do_compact) {
            launch_compaction(t, out, s);
        }
        if (int rc = rec(5)) return rc;
        if (int rc = launch_volumes()) return rc;
        volumes_pass = !(fuse && n > 0);
        if (volumes_pass) { if (int rc = rec(6)) return rc; }
        return RT_SUCCESS;
    }

    // Records in completion order: behind the march (already on the call's stream) the record kernel goes to the mesh's SECOND
    // stream — launched only when the march's last workgroups are seen to have started (DStage::cq_started: they then all have
    // their slots, or the batch's last residency round has begun; a record workgroup that waits for its march workgroup can no
    // longer keep one off the chip) —, the offsets' scan (CSR offsets for whoever asks for that layout later, the statistics, the
    // other control block's reset) behind the march on the call's stream, and k_finish behind both.
    int enqueue_completion() {
        hipStream_t s2 = m->side_stream;
        {
            const int nflags = (int)std::min<int64_t>(8, stg.cq_blocks);
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; ++spin) {
                bool all = true;
                for (int i = 0; i < nflags; ++i) all = all && __atomic_load_n(&t->cq_started[i], __ATOMIC_ACQUIRE) == (unsigned long long)cq_epoch;
                if (all) break;
                if ((spin & 1023u) == 1023u) {
                    if (hipStreamQuery(s) != hipErrorNotReady) break;  // (the march is over — or failed: the record kernel simply runs behind it)
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;
                }
            }
        }
        if (int rc = launch_materialise(t, out, s2, true, false, true, d_ctl, true)) return rc;
        if (int rc = scan_counts(false, false, true, true, false)) return rc;
        RT_HIP(hipEventRecord(m->side_ev[0], s));
        RT_HIP(hipStreamWaitEvent(s2, m->side_ev[0], 0));
        launch_finish(t, out, s2, false, fuse, (double)n_azim_2, d_ctl, h_res_dev, t->call_seq, true);
        RT_HIP(hipEventRecord(m->side_ev[1], s2));
        RT_HIP(hipStreamWaitEvent(s, m->side_ev[1], 0));
        volumes_pass = false;
        return RT_SUCCESS;
    }

    // The host's wait, and the control block's copy: total, failure summary, cursors
    int wait_attempt(int attempt) {
        if (n == 0) RT_HIP(hipMemcpyAsync(h_res, d_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        if (attempt == 0 && m->enqueue_hook) m->enqueue_hook(m->enqueue_hook_user);
        // option "async": back to the caller as soon as the scan's copy of the control block has arrived — total, failure
        // summary and pool cursor are final then, the compaction goes on behind the call (whole-track calls without events)
        const bool async_call = m->async_calls && !m->timing && n > 0 && !split;
        // two-phase calls: the control block's copy and the sequence number behind it are the LAST thing the call's last kernel
        // writes (k_finish's last block, after every other block of it has finished) — seeing the number in pinned memory is
        // seeing the call complete, a few microseconds before the stream reports it (hipStreamQuery); what is still to happen
        // on the stream is that kernel's retirement, which every later operation on the stream is ordered behind anyway
        const bool seq_done = topo && !m->timing && n > 0;
        if (async_call || seq_done) RT_HIP(wait_seq(h_res, t->call_seq, s));
        else RT_HIP(wait_stream(s));
        t->in_flight = async_call || seq_done;  // (accessors wait for the stream: immediate here)
        memcpy(fi, h_res, sizeof(fi));
        memcpy(&total, h_res + 16, sizeof(total));
        memcpy(cur, h_res + 18, sizeof(cur));
        t->chunks_needed_last = cur[0];
        if (topo) t->side_needed_last = std::max<int64_t>(0, (int64_t)cur[2] - side_static);
        return RT_SUCCESS;
    }

    // What an attempt may leave to repair — outputs that were too small, fused volumes that counted records twice — and whether
    // the call is complete (kDone), runs again with larger pools (kRetry) or with whole tracks (kRestartWhole); < 0: error
    int after_attempt(int attempt) {
        if (lean && reinterpret_cast<const int32_t *>(h_res + rt::kLeanCtl)[3] != 0) {
            // k_serve ended without its work (it never saw k_cheap finish): not a result — this handle marches with the one-kernel plan
            t->lean_gave_up = true; t->lean_q_clean = false; t->marg_clean = false;
            if (attempt >= 3) { set_error("rt_segmentize: the lean plan's queue was not served"); return RT_ERR_HIP; }
            RT_HIP(hipStreamSynchronize(s));
            choose_plan();
            return kRetry;
        }
        if (completion && n > 0 && !cur[1] && !cur[3] &&
            (h_res[rt::kCtlCq + 2] != 0 || h_res[rt::kCtlCq + 3] != (unsigned long long)stg.cq_blocks * 4ull * (unsigned long long)fuse_waves)) {
            // the record kernel beside the march waited in vain (the march never queued itself), or units were left untaken (an XCD
            // without a record workgroup): not a result — this handle keeps CSR order
            t->completion_gave_up = true; t->marg_clean = false;
            if (attempt >= 3) { set_error("rt_segmentize: the completion queue was not served"); return RT_ERR_HIP; }
            RT_HIP(hipStreamSynchronize(m->side_stream));
            RT_HIP(hipStreamSynchronize(s));
            choose_plan();
            return kRetry;
        }
        const bool pools_ok = !cur[1] && !cur[3];
        if (do_compact && pools_ok && total > out.cap) {
            // the estimate was short: grow the outputs and compact again (staging pool and offsets are still valid)
            if (int rc = reserve_records(t, total + total / 32 + 4096, out)) return rc;
            launch_compaction(t, out, s);
            if (topo && h_res[rt::kCtlDeferred] != 0) {
                // tracks whose exact Σℓ k_finish could not form from the truncated records: once more, from the complete ones
                launch_finish(t, out, s, false, false, (double)n_azim_2, d_ctl, h_res_dev, t->call_seq, completion);
                RT_HIP(hipStreamSynchronize(s));
                memcpy(fi, h_res, sizeof(fi));
            }
            if (!fused_volumes && m->volumes_mode == 2)  // the separate volumes pass read truncated records
                if (int rc = recompute_volumes()) return rc;
            RT_HIP(hipStreamSynchronize(s));
        }
        if (!cur[1] && split && fuse && (fi[7] != 0 || m->test_volumes_fallback)) {
            // some piece marched past the seed it should have stopped at: its surplus records were dropped by
            // k_resolve but had already been added to the fused volumes — recompute them from the kept records
            fused_volumes = false;
            if (int rc = recompute_volumes()) return rc;
            RT_HIP(hipStreamSynchronize(s));
        }
        if (pools_ok && topo && fuse && h_res[rt::kCtlRestarts] != 0) {
            // a track whose iteration bound reached the cap was marched again with exact steps: its cheap records had
            // already been added to the fused volumes — recompute them from the records
            if (!do_compact) {
                if (int rc = reserve_records(t, total, out)) return rc;
                launch_compaction(t, out, s);
            }
            fused_volumes = false;
            if (int rc = recompute_volumes()) return rc;
            RT_HIP(hipStreamSynchronize(s));
            t->compacted = true;
        }
        if (!cur[1] && split && h_res[21] != 0) {  // a track reached MAX_ITER segments (or the iteration cap) in pieces
            t->force_unsplit = true;
            return kRestartWhole;
        }
        if (pools_ok) {
            t->cplan.staged = true;
            if (do_compact) { t->compacted = !completion; t->completion_order = completion; }
            if (topo && !do_compact) t->sw_ell_valid = true;  // (k_materialise left the (ℓ, cell) rows)
            if (n > 0) {  // this call's scan has reset the other control block and (fused) left the accumulator zero
                t->ctl_clean[1 - cb] = true; t->ctl_first_chunk[1 - cb] = reset_key;
                t->vacc_clean = fuse;
                t->ctl_idx = 1 - cb;
            }
            return kDone;
        }
        t->marg_clean = false;  // (a void attempt may have left entries in the list of tracks to sum exactly)
        t->lean_q_clean = false;
        if (attempt >= 3) { set_error("staging pool / side list overflow persists (%d chunks, %d entries needed)", cur[0], cur[2]); return RT_ERR_HIP; }
        if (cur[1]) want = (int64_t)cur[0] + cur[0] / 8 + 64;  // the cursor kept counting: this is what the march needs
        if (cur[3]) side_want = (int64_t)cur[2] + cur[2] / 8 + 1024;
        return kRetry;
    }

    // The two-pass march (count, scan, march again writing at the CSR offsets): round 1's first correct path, kept as an
    // independent cross-check in builds with -DRT_EXPERIMENTAL
    int run_two_pass() {
#ifdef RT_EXPERIMENTAL
        if (int rc = rec(0)) return rc;
        RT_HIP(hipMemsetAsync(t->volumes.p, 0, sizeof(double) * m->n_cells, s));
        const unsigned grid = (unsigned)n_waves;
        RT_HIP(hipMemcpyAsync(d_ctl, t->h_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        if (int rc = rec(1)) return rc;
        if (n > 0)
            if (int rc = march(rt::kCount, 1, false, widek, false, grid, sizeof(int32_t), t->d, stg)) return rc;
        if (int rc = rec(2)) return rc;
        if (int rc = scan_counts(false, false, false)) return rc;
        if (int rc = rec(3)) return rc;
        RT_HIP(hipMemcpyAsync(h_res, d_ctl, rt::kCtlWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        if (m->enqueue_hook) m->enqueue_hook(m->enqueue_hook_user);  // (the count march is the longer half of this mode)
        RT_HIP(hipStreamSynchronize(s));
        memcpy(fi, h_res, sizeof(fi));
        memcpy(&total, h_res + 16, sizeof(total));
        if (int rc = reserve_records(t, total, out)) return rc;
        if (int rc = rec(4)) return rc;
        if (n > 0)
            if (int rc = march(rt::kFill, 1, false, widek, false, grid, sizeof(int32_t), t->d, stg, t->offsets.p)) return rc;
        if (int rc = rec(5)) return rc;
        if (int rc = launch_volumes()) return rc;
        if (int rc = rec(6)) return rc;
        RT_HIP(hipStreamSynchronize(s));
        t->compacted = true;
        return RT_SUCCESS;
#else
        set_error("the two-pass march needs a library built with -DRT_EXPERIMENTAL");
        return RT_ERR_INVALID;
#endif
    }

    // Timings (option "timing"), development prints, and the call's statistics and failure summary on the handle
    int finish() {
        RT_HIP(hipGetLastError());
        if (m->timing) {
            float f = 0;
            RT_HIP(hipEventElapsedTime(&f, t->ev[m->single_pass ? 1 : 0], t->ev[volumes_pass ? 6 : 5])); t->ms[0] = f;   // whole call, device side
            RT_HIP(hipEventElapsedTime(&f, t->ev[1], t->ev[2])); t->ms[2] = f;   // march (staged, or count)
            RT_HIP(hipEventElapsedTime(&f, t->ev[2], t->ev[3])); t->ms[3] = f;   // offsets scan (+ volumes ./= n_azim_2 when fused)
            RT_HIP(hipEventElapsedTime(&f, t->ev[m->single_pass ? 3 : 4], t->ev[5])); t->ms[4] = f;   // compaction (or fill march)
            if (volumes_pass) { RT_HIP(hipEventElapsedTime(&f, t->ev[5], t->ev[6])); t->ms[5] = f; }   // volumes as its own pass
        }
#ifdef RT_TIMING
        if (const char *path = getenv("RT_TIMING_DUMP")) {
            if (t->dbg.p) {
                std::vector<unsigned long long> h((size_t)((n + 63) / 64) * 4);
                RT_HIP(hipMemcpy(h.data(), t->dbg.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fp); fclose(fp); }
            }
        }
        fprintf(stderr, "[rt timing] per wave-iteration (lane-0 view, cycles): top+load %.0f | walk_step %.0f | emit %.0f | loop-back %.0f | iters/wave %.1f | loop cycles/wave %.0f\n",
                (double)fi[8] / fi[12], (double)fi[9] / fi[12], (double)fi[10] / fi[12], (double)fi[11] / fi[12], (double)fi[12] / fi[14], (double)fi[13] / fi[14]);
#endif
#ifdef RT_STATS_DISTINCT
        {
            fprintf(stderr, "[rt distinct] wave-iterations %llu, cheap lanes per iteration %.1f\n  distinct successor records 1..8+:", h_res[63], (double)h_res[62] / (double)std::max<unsigned long long>(1, h_res[63]));
            for (int b = 1; b <= 8; ++b) fprintf(stderr, " %.3f", (double)h_res[44 + b] / (double)std::max<unsigned long long>(1, h_res[63]));
            fprintf(stderr, "\n  distinct exit edges 1..8+:");
            for (int b = 1; b <= 8; ++b) fprintf(stderr, " %.3f", (double)h_res[53 + b] / (double)std::max<unsigned long long>(1, h_res[63]));
            fprintf(stderr, "\n");
        }
#endif
#ifdef RT_STATS
        if (split)
            fprintf(stderr, "[rt stats] split plan: %llu tracks, %llu of %llu seeds alive, %llu pieces kept, %llu records marched by pieces, %llu dropped\n",
                    h_res[22], h_res[23], h_res[24], h_res[26], h_res[25], fi[7]);
        fprintf(stderr, "[rt stats] walk: generic=%llu skip=%llu emit=%llu | wave-iterations=%llu with-generic-lane=%llu | chunks=%lld pool=%lld\n",
                fi[2], fi[3], fi[4], fi[5], fi[6], (long long)t->chunks_needed_last, (long long)t->pool_chunks);
#endif
        t->total = total;
        t->total_last = total;
        t->n_generic_records = (int64_t)fi[15];
        t->n_exact_walk_records = topo ? (int64_t)fi[14] : 0;
        for (int b = 0; b < 9; ++b) t->refusals[b] = m->single_pass ? (int64_t)h_res[rt::kCtlRefusal + b] : 0;
        t->n_near_rtol = (int64_t)h_res[rt::kCtlNearRtol];
        t->n_exact_tally = (int64_t)h_res[rt::kCtlExactTally];
        t->n_restarts = m->single_pass ? (int64_t)h_res[rt::kCtlRestarts] : 0;
        t->last_completion = completion ? 1 : 0;
        t->n_lean_queued = lean ? (int64_t)reinterpret_cast<const int32_t *>(h_res + rt::kLeanCtl)[0] : 0;
        t->n_failed = (int64_t)fi[0];
        t->first_failed_uid = fi[0] ? (int64_t)fi[1] : 0;
        t->first_failed_status = 0;
        if (fi[0]) {
            // (on the call's own stream, behind its last kernel: a two-phase call returns on the sequence number k_finish wrote, and a
            //  copy on the null stream is not ordered against a kernel of this non-blocking stream)
            int32_t stt = 0;
            RT_HIP(hipMemcpyAsync(&stt, t->status.p + (fi[1] - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RT_HIP(rtx::wait_stream(s));
            t->first_failed_status = stt;
        }
        t->segmentized = true;
        return RT_SUCCESS;
    }
};

}  // namespace

static int64_t segmentize_impl(rt_tracks *t, double tiny_step, int32_t k, double rtol, const double *delta_s, int32_t n_azim_2) {
    SegmentizeCall c(t, tiny_step, k, rtol, delta_s, n_azim_2);
    if (int rc = c.begin()) return rc;
    c.choose_plan();
    if (!c.m->single_pass) {
        if (int rc = c.run_two_pass()) return rc;
    } else {
        c.estimate_pools();
        for (int attempt = 0;; ++attempt) {
            if (int rc = c.grow_pools()) return rc;
            if (int rc = c.bind_stage(attempt)) return rc;
#ifdef RT_HOST_TIMING
            c.ht1 = ht_now();
#endif
            if (int rc = c.enqueue_attempt()) return rc;
#ifdef RT_HOST_TIMING
            c.ht2 = ht_now();
#endif
            if (int rc = c.wait_attempt(attempt)) return rc;
#ifdef RT_HOST_TIMING
            {
                const double ht3 = ht_now();
                g_ht[0] += c.ht1 - c.ht0; g_ht[1] += c.ht2 - c.ht1; g_ht[2] += ht3 - c.ht2; ++g_hn;
                if (g_hn % 50 == 0) fprintf(stderr, "[rt host] per call: before enqueue %.1f us, enqueue %.1f us, wait %.1f us\n", g_ht[0] / g_hn, g_ht[1] / g_hn, g_ht[2] / g_hn);
            }
#endif
            const int r = c.after_attempt(attempt);
            t->last_attempts = attempt + 1;
            if (r < 0) return r;
            if (r == SegmentizeCall::kRestartWhole) return segmentize_impl(t, tiny_step, k, rtol, delta_s, n_azim_2);
            if (r == SegmentizeCall::kDone) break;
        }
    }
    if (int rc = c.finish()) return rc;
    return c.total;
}

int32_t rt_wait(rt_tracks *t) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    return finish_call(t);
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
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    const void *src[2] = {t->offsets.p, t->status.p};
    void *dst[2] = {seg_offsets, t->n ? status : nullptr};
    const size_t bytes[2] = {sizeof(int64_t) * (size_t)(t->n + 1), sizeof(int32_t) * (size_t)t->n};
    return fetch_pipelined(t, 2, src, dst, bytes);
}

// ---- a host block owned by the library for everything a fetch returns (rt_hostpar.hpp, ResultBlock)
struct rt_result {
    rthostpar::ResultBlock b;
    int64_t total = 0;
};

rt_result *rt_result_alloc(rt_mesh *mesh, int64_t n_tracks, double sum_ell, int64_t n_records_hint) {
    if (!mesh || n_tracks < 0) { set_error("rt_result_alloc: bad arguments"); return nullptr; }
    // the Cauchy–Crofton estimate of the record count (as the staging pool's): κ·Σℓ + a few per track
    const int64_t est = n_records_hint > 0 ? n_records_hint : (int64_t)(1.08 * mesh->kappa * std::max(sum_ell, 0.0)) + 2 * n_tracks + 4096;
    rt_result *r = new (std::nothrow) rt_result();
    if (!r || !r->b.map_for(n_tracks, est, mesh->fetch_hugepages != 0)) {
        set_error("rt_result_alloc: no memory for %lld records", (long long)est);
        delete r;
        return nullptr;
    }
    r->b.prefault_start(std::min(12u, std::max(2u, std::thread::hardware_concurrency() / 2)));  // (in the background: returns at once)
    return r;
}

void rt_result_free(rt_result *r) { delete r; }

int32_t rt_result_fetch(rt_tracks *t, rt_result *r, void **host_ptrs, int64_t *total) {
    if (!t || !r) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    if (r->b.n_tracks != t->n || r->b.cap_records < t->total) {
        // the estimate was short (or the block belongs to another track set): a block of the right size, faulted in here
        if (!r->b.map_for(t->n, t->total + t->total / 64 + 64, t->mesh->fetch_hugepages != 0)) { set_error("rt_result_fetch: no memory"); return RT_ERR_INVALID; }
        r->b.prefault_start(std::min(16u, std::max(2u, std::thread::hardware_concurrency())));
    }
    rthostpar::ResultBlock &b = r->b;
    const void *src[8] = {t->offsets.p, t->status.p, t->spx.p, t->spy.p, t->sqx.p, t->sqy.p, t->sell.p, t->element.p};
    void *dst[8];
    for (int a = 0; a < 8; ++a) dst[a] = b.base + b.off[a];
    const size_t nt = (size_t)t->n, nr = (size_t)t->total;
    const size_t bytes[8] = {8 * (nt + 1), 4 * nt, 8 * nr, 8 * nr, 8 * nr, 8 * nr, 8 * nr, 4 * nr};
    const auto tf0 = std::chrono::steady_clock::now();
    if (int rc = fetch_pipelined(t, 8, src, dst, bytes, &b)) return rc;
    if (getenv("RT_RESULT_TIMING")) {  // development: where a slow fetch into the library's block spent its time
        std::lock_guard<std::mutex> lk(b.m);
        fprintf(stderr, "[rt result] %zu units of 2 MB; fetch started %.2f ms after the block was mapped and took %.2f ms; last unit faulted at %.2f ms; slowest first "
                        "touch %.2f ms, %d above 0.5 ms, %ld units on 4-KB pages; the copy waited %.2f ms for the front\n",
                b.n_units, std::chrono::duration<double, std::milli>(tf0 - b.t_map).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count(), b.done_at_ms, b.max_unit_ms, b.slow_units,
                b.small_units, b.waited_ms);
    }
    r->total = t->total;
    if (host_ptrs) for (int a = 0; a < 8; ++a) host_ptrs[a] = dst[a];
    if (total) *total = t->total;
    return RT_SUCCESS;
}

int32_t rt_fetch_segments(rt_tracks *t, double *px, double *py, double *qx, double *qy, double *ell,
                          int32_t *element) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    if (t->total == 0) return RT_SUCCESS;
    const void *src[6] = {t->spx.p, t->spy.p, t->sqx.p, t->sqy.p, t->sell.p, t->element.p};
    void *dst[6] = {px, py, qx, qy, ell, element};
    const size_t bytes[6] = {8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total,
                             4 * (size_t)t->total};
    return fetch_pipelined(t, 6, src, dst, bytes);
}

// Page-locked host buffers are expensive to create (≈45 ms for C3's 410 MB) and cheap to keep: one set is kept
// process-wide when a handle dies, so that a host that creates a fresh handle per call (the Julia shim) pays
// for pinning once.
namespace {
struct PinSet { void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap = 0; };
PinSet g_pin_cache;
std::mutex g_pin_mutex;
void pin_free(PinSet &ps) {
    for (void *&q : ps.p) { if (q) (void)hipHostFree(q); q = nullptr; }
    ps.cap = 0;
}
}  // namespace

namespace {
void pin_release_to_cache(rt_tracks *t) {
    if (!t->pin_cap) return;
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    PinSet mine;
    for (int a = 0; a < 6; ++a) { mine.p[a] = t->pin[a]; t->pin[a] = nullptr; }
    mine.cap = t->pin_cap; t->pin_cap = 0;
    if (mine.cap > g_pin_cache.cap) std::swap(mine, g_pin_cache);
    pin_free(mine);
}
}  // namespace

int32_t rt_fetch_segments_pinned(rt_tracks *t, void **host_ptrs) {
    if (!t || !host_ptrs) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    const size_t n = (size_t)t->total;
    if (n > t->pin_cap) {
        pin_release_to_cache(t);
        {
            std::lock_guard<std::mutex> lk(g_pin_mutex);
            if (g_pin_cache.cap >= n) {
                for (int a = 0; a < 6; ++a) { t->pin[a] = g_pin_cache.p[a]; g_pin_cache.p[a] = nullptr; }
                t->pin_cap = g_pin_cache.cap; g_pin_cache.cap = 0;
            }
        }
        if (n > t->pin_cap) {
            const size_t cap = n + n / 8 + 64;
            for (int a = 0; a < 6; ++a) RT_HIP(hipHostMalloc(&t->pin[a], cap * (a < 5 ? sizeof(double) : sizeof(int32_t)), hipHostMallocDefault));
            t->pin_cap = cap;
        }
    }
    hipStream_t s = t->mesh->stream;
    const void *src[6] = {t->spx.p, t->spy.p, t->sqx.p, t->sqy.p, t->sell.p, t->element.p};
    for (int a = 0; a < 6 && n > 0; ++a)
        RT_HIP(hipMemcpyAsync(t->pin[a], src[a], n * (a < 5 ? sizeof(double) : sizeof(int32_t)), hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    for (int a = 0; a < 6; ++a) host_ptrs[a] = t->pin[a];
    return RT_SUCCESS;
}

int32_t rt_fetch_pinned(rt_tracks *t, void **host_ptrs) {
    if (!t || !host_ptrs) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (!t->pin_off) RT_HIP(hipHostMalloc((void **)&t->pin_off, sizeof(int64_t) * (size_t)(t->n + 1), hipHostMallocDefault));
    if (!t->pin_st) RT_HIP(hipHostMalloc((void **)&t->pin_st, sizeof(int32_t) * (size_t)std::max<int64_t>(t->n, 1), hipHostMallocDefault));
    hipStream_t s = t->mesh->stream;
    // (queued in front of the records' copies: one synchronisation for all eight arrays)
    RT_HIP(hipMemcpyAsync(t->pin_off, t->offsets.p, sizeof(int64_t) * (size_t)(t->n + 1), hipMemcpyDeviceToHost, s));
    if (t->n) RT_HIP(hipMemcpyAsync(t->pin_st, t->status.p, sizeof(int32_t) * (size_t)t->n, hipMemcpyDeviceToHost, s));
    if (int32_t rc = rt_fetch_segments_pinned(t, host_ptrs + 2)) return rc;
    host_ptrs[0] = t->pin_off;
    host_ptrs[1] = t->pin_st;
    return RT_SUCCESS;
}

int32_t rt_fetch_volumes(rt_tracks *t, double *volumes) {
    if (!t || !volumes) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    RT_HIP(hipMemcpy(volumes, t->volumes.p, sizeof(double) * t->mesh->n_cells, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_fill_tau(rt_tracks *t, const double *sigma_t, int32_t n_groups, void **tau_dev, double *ms) {
    if (!t || !sigma_t || n_groups <= 0) { set_error("rt_fill_tau: bad arguments"); return RT_ERR_INVALID; }
    if (n_groups > 1024) { set_error("rt_fill_tau: at most 1024 groups"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    rt_mesh *m = t->mesh;
    RT_HIP(hipSetDevice(m->device));
    if (int rc = ensure_compacted(t)) return rc;
    hipStream_t s = m->stream;
    const size_t n = (size_t)t->total * (size_t)n_groups;
    t->tau_groups = 0;  // (set again once the kernel has been enqueued: a failure below leaves no τ to fetch)
    RT_HIP(t->tau.reserve(n > 0 ? n : 1));
    if (int rc = upload(t->sigma_t, sigma_t, (size_t)m->n_cells * n_groups, s)) return rc;
    RT_HIP(hipEventRecord(t->ev[0], s));
    if (n > 0) {
        launch_fill_tau(s, t, n_groups);
    }
    RT_HIP(hipEventRecord(t->ev[7], s));
    RT_HIP(hipStreamSynchronize(s));
    RT_HIP(hipGetLastError());
    t->tau_groups = n_groups;
    if (ms) { float f = 0; RT_HIP(hipEventElapsedTime(&f, t->ev[0], t->ev[7])); *ms = f; }
    if (tau_dev) *tau_dev = t->tau.p;
    return RT_SUCCESS;
}

int32_t rt_fetch_tau(rt_tracks *t, double *tau) {
    if (!t || !tau) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized || t->tau_groups <= 0) { set_error("rt_fill_tau has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    const size_t n = (size_t)t->total * (size_t)t->tau_groups;
    if (n) RT_HIP(hipMemcpy(tau, t->tau.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

int32_t rt_device_pointers(rt_tracks *t, void **p) {
    if (!t || !p) { set_error("null argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (int rc = ensure_compacted(t)) return rc;
    p[0] = t->offsets.p; p[1] = t->status.p; p[2] = t->spx.p; p[3] = t->spy.p; p[4] = t->sqx.p;
    p[5] = t->sqy.p; p[6] = t->sell.p; p[7] = t->element.p; p[8] = t->volumes.p;
    return RT_SUCCESS;
}

// ---- records in completion order: the per-track table (include/rt_segmentize.h)
int32_t rt_record_order(rt_tracks *t) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    return t->completion_order ? 1 : 0;
}

// the handle's records as they lie: in completion order, or — a call that left staged rows only — made now, in CSR order
static int records_as_stored(rt_tracks *t) {
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    RT_HIP(hipSetDevice(t->mesh->device));
    if (!t->completion_order)
        if (int rc = ensure_compacted(t)) return rc;
    return RT_SUCCESS;
}

int32_t rt_device_table(rt_tracks *t, void **p) {
    if (!t || !p) { set_error("null argument"); return RT_ERR_INVALID; }
    if (int rc = records_as_stored(t)) return rc;
    p[0] = t->completion_order ? t->tab_off.p : t->offsets.p; p[1] = t->counts.p; p[2] = t->status.p;
    p[3] = t->spx.p; p[4] = t->spy.p; p[5] = t->sqx.p; p[6] = t->sqy.p; p[7] = t->sell.p; p[8] = t->element.p; p[9] = t->volumes.p;
    return RT_SUCCESS;
}

int32_t rt_fetch_table(rt_tracks *t, int64_t *seg_begin, int32_t *seg_count, int32_t *status) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (int rc = records_as_stored(t)) return rc;
    if (t->n == 0) return RT_SUCCESS;
    const void *src[3] = {t->completion_order ? t->tab_off.p : t->offsets.p, t->counts.p, t->status.p};
    void *dst[3] = {seg_begin, seg_count, status};
    const size_t bytes[3] = {sizeof(int64_t) * (size_t)t->n, sizeof(int32_t) * (size_t)t->n, sizeof(int32_t) * (size_t)t->n};
    return fetch_pipelined(t, 3, src, dst, bytes);
}

int32_t rt_fetch_records(rt_tracks *t, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element) {
    if (!t) { set_error("null handle"); return RT_ERR_INVALID; }
    if (int rc = records_as_stored(t)) return rc;
    if (t->total == 0) return RT_SUCCESS;
    const void *src[6] = {t->spx.p, t->spy.p, t->sqx.p, t->sqy.p, t->sell.p, t->element.p};
    void *dst[6] = {px, py, qx, qy, ell, element};
    const size_t bytes[6] = {8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total, 8 * (size_t)t->total,
                             4 * (size_t)t->total};
    return fetch_pipelined(t, 6, src, dst, bytes);
}

int32_t rt_mesh_info(rt_mesh *m, double *info, int32_t n_info, char *note, int32_t note_cap) {
    if (!m || (n_info > 0 && !info) || n_info < 0 || note_cap < 0) { set_error("rt_mesh_info: bad argument"); return RT_ERR_INVALID; }
    const double v[RT_MESH_INFO_COUNT] = {
        (double)(m->d.walk_ok ? 1 : 0), (double)m->n_records, (double)m->n_records_walk, m->eps_min, m->eps_max,
        m->d.d_vertex, m->d.l_min, (double)m->n_cells_fragile, (double)m->n_cells_wild, (double)m->n_edges_nonmanifold,
        (double)m->extras_max, m->prep_ms, m->kappa, (double)(m->walk_available ? 1 : 0),
        (double)(m->topo_available ? m->n_records_topo : 0), m->topo_tiny_max};
    for (int i = 0; i < n_info && i < RT_MESH_INFO_COUNT; ++i) info[i] = v[i];
    if (note && note_cap > 0) {
        strncpy(note, m->prep_note.c_str(), (size_t)note_cap - 1);
        note[note_cap - 1] = 0;
    }
    return RT_SUCCESS;
}

int32_t rt_last_stats(rt_tracks *t, int64_t *stats, int32_t n) {
    if (!t || !stats || n < 4) { set_error("rt_last_stats: bad argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    stats[0] = t->total;
    stats[1] = t->n_generic_records;
    stats[2] = t->chunks_needed_last;
    stats[3] = t->pool_chunks;
    if (n > 4) stats[4] = t->last_march_waves;
    if (n > 5) stats[5] = t->last_split;  // 0 whole tracks, 1 pieces
    if (n > 6) stats[6] = t->last_widek;
    if (n > 8) stats[8] = t->last_topo ? t->total - t->n_generic_records - t->n_exact_walk_records : 0;  // records made by cheap steps
    for (int b = 0; b < 9 && 9 + b < n; ++b) stats[9 + b] = t->refusals[b];
    if (n > 18) stats[18] = t->n_near_rtol;
    if (n > 19) stats[19] = t->n_restarts;
    if (n > 20) stats[20] = t->last_topo ? t->n_exact_tally : 0;
    if (n > 21) stats[21] = t->last_lean;      // the lean plan of the last call (0: the march in one kernel)
    if (n > 22) stats[22] = t->n_lean_queued;  // ... and the lanes k_serve finished
    if (n > 27) stats[27] = t->last_attempts;   // attempts of the last call (> 1: a staging pool / side list that was too small, a fall-back)
    if (n > 26) stats[26] = t->side_cap;        // side-list entries allocated (one reserved per march slot + the dynamic part)
    if (n > 25) stats[25] = t->side_needed_last;  // ... and used beyond the reserved ones
    if (n > 24) stats[24] = t->last_completion;  // 1: the call wrote its records beside the march, in completion order (option "record_order")
    if (n > 23) stats[23] = t->last_record_kernel;  // 1 k_compact3, 2 k_materialise, 3 k_materialise_lin, 4 k_materialise writing (ℓ, cell) rows only
    if (n > 7) {  // device memory held by this handle: inputs, staging pools, tables, results
        auto b = [](const auto &d) { return (int64_t)(d.cap * sizeof(*d.p)); };
        stats[7] = b(t->in_arena) + b(t->cnt_slot) + b(t->off_slot) + b(t->w_slot) +
                   b(t->counts) + b(t->status) + b(t->element) + b(t->offsets) + b(t->tile_sums) + b(t->tile_acc) + b(t->ctl) + b(t->spx) +
                   b(t->spy) + b(t->sqx) + b(t->sqy) + b(t->sell) + b(t->volumes) + b(t->volumes_prev) + b(t->delta_s) + b(t->gpx) + b(t->gpy) +
                   b(t->gqx) + b(t->gqy) + b(t->gelement) + b(t->ctab) + b(t->cowner) + b(t->vorder) + b(t->vw_wave) + b(t->vw_k) +
                   b(t->w_base) + b(t->w_P) + b(t->s_el) + b(t->s_eq) + b(t->p_count) + b(t->p_flags) + b(t->p_valid) + b(t->p_rel) + b(t->s_px) +
                   b(t->s_py) + b(t->s_qx) + b(t->s_qy) + b(t->s_ell) + b(t->p_sum) + b(t->vacc) + b(t->tau) +
                   b(t->sigma_t) + b(t->sw_src) + b(t->sw_w) + b(t->sw_xs) + b(t->sw_psi_in) + b(t->sw_psi_out) + b(t->sw_phi) + b(t->sw_ell) +
                   b(t->sw_cell) + b(t->side_px) + b(t->side_py) + b(t->side_qx) + b(t->side_qy) + b(t->side_el) + b(t->marg);
    }
    return RT_SUCCESS;
}

int32_t rt_last_timing(rt_tracks *t, double *ms, int32_t n) {
    if (!t || !ms || n < 6) { set_error("bad argument"); return RT_ERR_INVALID; }
    if (!t->segmentized) { set_error("rt_segmentize has not run"); return RT_ERR_NOT_SEGMENTIZED; }
    if (int rc = finish_call(t)) return rc;
    for (int i = 0; i < n && i < 8; ++i) ms[i] = t->ms[i];
    return RT_SUCCESS;
}

}  // extern "C"
