// rt_hostpar.hpp — the host-side, device-free parts of rt_tracks_create and of the pipelined fetch (librt_segmentize.so): the team of
// host threads, the march order / reserved-chunk / compaction-order plan of a track set, the packing of a track set's image into a
// page-locked block, the copy of a fetched piece into the caller's arrays.  No HIP call in here: tests/sanitize/hostpar_san.cpp
// builds this header with AddressSanitizer + UBSan and with ThreadSanitizer (tests/sanitize/run.sh), with the device copies of the
// callers replaced by memcpy.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>
#include <unistd.h>
#if defined(__linux__)
#include <sys/mman.h>
#endif
#if defined(__x86_64__) && defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

namespace rthostpar {

// A few host threads that stay: starting a thread costs ≈30 µs and rt_tracks_create makes four passes over the track arrays — with
// threads per pass that was a third of the call at the headline configuration.  One job at a time; a caller that finds the team
// busy (rt_multi_create uploads its shards from several threads) does its work alone.
// Round 6: the workers stay HOT between the jobs of a burst.  A fetch hands the team one job per 16-MB piece, every ≈0.3 ms; a
// worker asleep on a condition variable wakes in 20-60 µs on a quiet box and in milliseconds on a busy one (one straggler per piece
// is the piece's time: the C3 fetch's 30 pieces lost 2 ms to wake-ups on a good run and 3-8 ms on a bad one).  A worker that has
// finished a part now polls the job generation for kHotUs before it sleeps, and the caller polls the pending count before it
// sleeps: inside a burst nobody sleeps, outside the team costs nothing.
inline void cpu_relax() {
#if (defined(__x86_64__) || defined(__i386__)) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
}
class WorkerTeam {
  public:
    static constexpr int kHotUs = 600;
    ~WorkerTeam() {
        if (pid_ != getpid()) return;  // (a forked child never had the threads)
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; stop_a_.store(true, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : *th_) t.join();
    }
    // f(k) for k in [0, parts): part 0 on the caller's thread; `parts` comes back as the number that run (the team may not get all
    // its threads) before the first one starts; false if the team is busy (nothing was run)
    template <typename F>
    bool run(unsigned &parts, F &f, std::vector<std::exception_ptr> &err) {
        std::unique_lock<std::mutex> job(job_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (pid_ != getpid()) { th_ = new std::vector<std::thread>; pid_ = getpid(); }  // after a fork: the parent's threads are not here
            while (th_->size() + 1 < parts) {
                const unsigned id = (unsigned)th_->size() + 1;
                try { th_->emplace_back([this, id] { loop(id); }); } catch (...) { break; }
            }
            parts = std::min<unsigned>(parts, (unsigned)th_->size() + 1);
            call_ = [&f, &err](unsigned k) { try { f(k); } catch (...) { err[k] = std::current_exception(); } };
            parts_ = parts;
            pending_.store(parts - 1, std::memory_order_relaxed);
            ++gen_;
            gen_a_.store(gen_, std::memory_order_release);  // (what the hot workers poll; the job itself is read under the mutex)
        }
        cv_.notify_all();
        call_(0);
        // the caller waits the same way: a poll first (the parts are of equal size: the others end within microseconds), then the
        // condition variable
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
            cpu_relax();
            if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) {
                std::unique_lock<std::mutex> lk(m_);
                done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
                break;
            }
        }
        return true;
    }
  private:
    void loop(unsigned id) {
        unsigned long seen = 0;
        for (;;) {
            // hot: poll for the next job of a burst
            bool hot = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; ++spin) {
                if (gen_a_.load(std::memory_order_acquire) != seen) { hot = true; break; }
                if (stop_a_.load(std::memory_order_acquire)) return;
                cpu_relax();
                if ((spin & 127u) == 127u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kHotUs)) break;
            }
            unsigned my_parts;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (!hot) cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                my_parts = parts_;
            }
            if (id >= my_parts) continue;
            call_(id);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);  // (the caller may be about to sleep on done_: the notify must not fall between its check and its wait)
                done_.notify_one();
            }
        }
    }
    std::mutex m_, job_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> *th_ = new std::vector<std::thread>;
    pid_t pid_ = getpid();
    std::function<void(unsigned)> call_;
    unsigned parts_ = 0;
    std::atomic<unsigned> pending_{0};
    unsigned long gen_ = 0;
    std::atomic<unsigned long> gen_a_{0};
    bool stop_ = false;
    std::atomic<bool> stop_a_{false};
};
inline WorkerTeam g_team;

// f(i0, i1) over [0, n) on a few host threads (results must not depend on the split); what a worker throws is rethrown here
template <typename F>
inline void par_ranges(size_t n, size_t grain, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)nt, (size_t)16, n / std::max<size_t>(grain, 1) + 1}));
    if (nt == 1) { f((size_t)0, n); return; }
    std::vector<std::exception_ptr> err(nt);
    unsigned parts = nt;  // (the team may run fewer: part k is [n·k/parts, n·(k+1)/parts) of however many do)
    auto body = [&](unsigned k) { f(n * k / parts, n * (k + 1) / parts); };
    const bool ran = g_team.run(parts, body, err);
    if (!ran) { f((size_t)0, n); return; }
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

// What rt_tracks_create derives from the tracks' lengths before anything is uploaded.
struct MarchPlan {
    std::vector<int32_t> perm;     // march slot -> uid
    int32_t reg_cap[16] = {};      // DStage: the leading march waves that get a reserved j-th chunk
    std::vector<int32_t> corder;   // large batches: march waves in the order of their output addresses (else empty)
};
inline void plan_march_order(const double *ell, size_t n, int sort_mode, double kappa, int64_t test_reserved_pct, int n_regions, int chunk_rows,
                             MarchPlan &out) {
    // march order (default 2): waves of 64 CONSECUTIVE uids, longest wave first.  Neighbouring
    // tracks of one angle cross the same cells at the same time (shared walk records, coherent
    // branches) and have nearly equal lengths; sorting individual tracks by length measured 20 %
    // slower because it scatters the lanes of a wave over the whole mesh.
    // (one call is what the reference makes, src/trackgenerator.jl:357-369: the host's share of it — wave maxima, the fills,
    //  the copy into the staging block — runs on a few threads; the sorts are over waves, not tracks)
    std::vector<int32_t> &perm = out.perm;
    perm.assign(n, 0);
    if (sort_mode == 1) {
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return ell[a] > ell[b]; });
    } else if (sort_mode == 2) {
        const size_t nw = (n + 63) / 64;
        std::vector<double> wmax(nw, 0.0);
        par_ranges(nw, 512, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                double mx = 0.0;
                for (size_t i = w * 64; i < std::min(n, w * 64 + 64); ++i) mx = std::max(mx, ell[i]);
                wmax[w] = mx;
            }
        });
        // (keys and indices side by side: the comparator of a sort over indices alone jumps through wmax)
        std::vector<std::pair<double, int32_t>> wkey(nw);
        for (size_t w = 0; w < nw; ++w) wkey[w] = {wmax[w], (int32_t)w};
        std::stable_sort(wkey.begin(), wkey.end(), [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first > b.first; });
        std::vector<int32_t> worder(nw);
        for (size_t w = 0; w < nw; ++w) worder[w] = wkey[w].second;
        // the batch's last wave of uids may be partial: the slots behind it are packed (no padding), so its position shifts them
        std::vector<size_t> first(nw + 1, 0);
        for (size_t w = 0; w < nw; ++w) first[w + 1] = first[w] + std::min<size_t>(64, n - (size_t)worder[w] * 64);
        par_ranges(nw, 512, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                size_t k2 = first[w];
                for (size_t l = 0; l < 64 && (size_t)worder[w] * 64 + l < n; ++l) perm[k2++] = (int32_t)(worder[w] * 64 + l);
            }
        });
    } else {
        std::iota(perm.begin(), perm.end(), 0);
    }
    {
        // Reserved staging chunks (DStage): march wave w (64 slots of the march order) is expected to need
        // ceil((1.15·κ·ℓ_max + 12) / 32) chunks — κ·ℓ is the Cauchy–Crofton mean, a wave through a denser part of the mesh takes
        // its further chunks from the cursor — and region j serves the leading waves that need a j-th chunk
        const size_t nw = (n + 63) / 64;
        std::vector<int32_t> need(nw, 1);
        par_ranges(nw, 512, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                double mx = 0.0;
                for (size_t i = w * 64; i < std::min(n, w * 64 + 64); ++i) mx = std::max(mx, ell[perm[i]]);
                const double est = (1.15 * kappa * mx + 12.0) * (test_reserved_pct >= 0 ? 0.01 * (double)test_reserved_pct : 1.0);
                need[w] = (int32_t)std::min<double>((double)n_regions, std::max(1.0, std::ceil(est / (double)chunk_rows)));
            }
        });
        // region j = the waves [0, reg_cap[j]): the estimates do not fall monotonically along the march order in every sort mode
        // (nor behind a partial wave of uids packed into the middle), so a wave reserves what any wave behind it needs
        for (size_t w = nw; w-- > 1;) need[w - 1] = std::max(need[w - 1], need[w]);
        for (int j = 0; j < n_regions; ++j) {
            size_t c = 0;
            while (c < nw && need[c] > j) ++c;
            out.reg_cap[j] = (int32_t)c;
        }
    }
    std::vector<int32_t> &h_corder = out.corder;
    h_corder.clear();
    if (((n + 63) / 64) > 4096) {  // batches of many rounds: compaction in output order (measured -8 % at 16 k waves, +1.5 % at 2 k)
        const size_t nw = (n + 63) / 64;
        // march waves (64 slots each) in the order of the uid of their first track.  Every march wave starts in another wave of
        // uids (the slots behind a partial last uid-wave are packed, so a march wave may straddle two of them — its first slot
        // still lies in one no other march wave starts in): a counting sort over uid-waves
        std::vector<int32_t> at(nw, -1);
        for (size_t a = 0; a < nw; ++a) {
            int32_t &slot_of = at[(size_t)perm[a * 64] >> 6];
            if (slot_of >= 0) { at.clear(); break; }  // (not expected: fall back to the sort)
            slot_of = (int32_t)a;
        }
        if (!at.empty()) {
            h_corder.reserve(nw);
            for (size_t w = 0; w < nw; ++w)
                if (at[w] >= 0) h_corder.push_back(at[w]);
        }
        if (h_corder.size() != nw) {
            h_corder.resize(nw);
            std::iota(h_corder.begin(), h_corder.end(), 0);
            std::stable_sort(h_corder.begin(), h_corder.end(), [&](int32_t a, int32_t b) { return perm[(size_t)a * 64] < perm[(size_t)b * 64]; });
        }
    }
}

// The image of tracks [i0, i0 + m) as it lies in the arena / in a range's half of the staging block: nine double arrays of `stride`
// entries each, then azim_idx and the march order (int32, `stride` entries each).  Written by the team's threads.
inline void pack_tracks_image(unsigned char *hb, size_t stride, size_t i0, size_t m, const double *const src8[9], const int32_t *azim_idx,
                              const int32_t *perm) {
    int32_t *h_az = (int32_t *)(hb + 9 * stride * sizeof(double)), *h_pm = h_az + stride;
    par_ranges(m, 16384, [&](size_t j0, size_t j1) {
        for (int a = 0; a < 9; ++a) memcpy((double *)(hb + (size_t)a * stride * sizeof(double)) + j0, src8[a] + i0 + j0, (j1 - j0) * sizeof(double));
        memcpy(h_az + j0, azim_idx + i0 + j0, (j1 - j0) * sizeof(int32_t));
        memcpy(h_pm + j0, perm + i0 + j0, (j1 - j0) * sizeof(int32_t));
    });
}

// The destination of a fetched piece — usually arrays the caller has just allocated — is asked for as transparent huge pages
// before the host copies into it (512 times fewer faults; a no-op where the allocator has asked already, as numpy does; Julia's and
// malloc's large blocks have not).  Measured on 410 MB into fresh arrays (profiles/r05/exp_fetch_fresh_modes.log): 9.5-9.6 ms with the
// hint against 18-24 ms with 4-KB pages.  Populating the pages ahead of the copy (MADV_POPULATE_WRITE by the worker team, round 5's
// first version) was SLOWER than letting the copy take the faults: 11-18 ms.  Best effort: errors are ignored.
inline void hint_huge_pages(char *dst, size_t bytes) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
    const uintptr_t page = 4096;
    const uintptr_t a = ((uintptr_t)dst + page - 1) & ~(page - 1), b = ((uintptr_t)dst + bytes) & ~(page - 1);
    if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
#else
    (void)dst; (void)bytes;
#endif
}

// ---- a host block owned by the library for everything a fetch returns (rt_result_alloc, round 6) -----------------------------
// One call of segmentize! ends with its records on the host (src/trackgenerator.jl:357-369: `t.tracks_by_uid[i].segments`), and at the
// headline configuration 98 % of such a call is the copy into 410 MB of FRESH host memory: 7.3 ms of PCIe — and 2-19 ms more of page
// faults, by the state of the box's huge pages (a destination the caller allocated can only be hinted at: rounds 4-5).  Here the
// library owns the destination: an anonymous mapping aligned to 2 MB, transparent huge pages asked for BEFORE its first touch, and
// faulted in by threads of its own in the BACKGROUND, in address order, while the caller uploads its tracks and the kernels run —
// the fetch then copies behind the front of what is already there.  Where a 2-MB fault stalls (memory that must be compacted first)
// the unit is re-advised to 4-KB pages and faulted by 512 small faults instead: independent of the box's state.
struct ResultBlock {
    static constexpr size_t kUnit = (size_t)2 << 20;
    char *map = nullptr; size_t map_bytes = 0;   // the mapping as mmap returned it
    char *base = nullptr; size_t bytes = 0;      // its 2-MB aligned part: the eight arrays
    size_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // seg_offsets (i64), status (i32), px, py, qx, qy, ell (f64), element (i32)
    int64_t n_tracks = 0, cap_records = 0;
    std::vector<std::thread> th;
    std::vector<unsigned char> done;             // per 2-MB unit: faulted in
    std::mutex m;
    std::condition_variable cv;
    size_t front_units = 0, next_unit = 0, n_units = 0;
    bool stop = false;
    long small_units = 0;                        // units that fell back to 4-KB pages
    // development (RT_RESULT_TIMING=1: rt_result_fetch prints them): the slowest first touch of a unit, when the last unit was done,
    // how long the fetch waited for the front
    double max_unit_ms = 0.0, done_at_ms = 0.0, waited_ms = 0.0;
    int slow_units = 0;                          // first touches above 0.5 ms
    std::chrono::steady_clock::time_point t_map;
    double stall_ms = 2.0;                       // a 2-MB unit whose first touch takes longer than this is a stall
    ~ResultBlock() { release(); }
    static size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }
    bool map_for(int64_t n_tracks_, int64_t cap_records_, bool huge) {
#if defined(__linux__)
        release();
        n_tracks = n_tracks_; cap_records = cap_records_;
        const size_t nt = (size_t)std::max<int64_t>(n_tracks, 0), nr = (size_t)std::max<int64_t>(cap_records, 1);
        const size_t len[8] = {8 * (nt + 1), 4 * std::max<size_t>(nt, 1), 8 * nr, 8 * nr, 8 * nr, 8 * nr, 8 * nr, 4 * nr};
        size_t o = 0;
        for (int a = 0; a < 8; ++a) { off[a] = o; o = up(o + len[a], a < 2 ? 4096 : kUnit); }  // (every record array starts a 2-MB unit)
        bytes = up(o, kUnit);
        map_bytes = bytes + kUnit;
        void *p = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { map = nullptr; map_bytes = bytes = 0; return false; }
        map = (char *)p;
        base = (char *)up((size_t)(uintptr_t)map, kUnit);
#if defined(MADV_HUGEPAGE)
        if (huge) (void)madvise(base, bytes, MADV_HUGEPAGE);
#endif
        n_units = bytes / kUnit;
        done.assign(n_units, 0);
        front_units = next_unit = 0; stop = false; small_units = 0;
        max_unit_ms = done_at_ms = waited_ms = 0.0; slow_units = 0; t_map = std::chrono::steady_clock::now();
        return true;
#else
        (void)n_tracks_; (void)cap_records_; (void)huge;
        return false;
#endif
    }
    // one unit: its first byte — a 2-MB fault where the kernel has a huge page at hand; if that took long (compaction), or no huge
    // page came, the other 511 pages by small faults
    void fault_unit(size_t u) {
        volatile char *p = (volatile char *)(base + u * kUnit);
        const auto t0 = std::chrono::steady_clock::now();
        p[0] = 0;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(m);
            if (ms > max_unit_ms) max_unit_ms = ms;
            if (ms > 0.5) ++slow_units;
        }
#if defined(__linux__) && defined(MADV_NOHUGEPAGE)
        if (ms > stall_ms) {
            // (the fault had to wait for a huge page: the units still to come take 4-KB pages — many small faults in parallel beat a
            //  few that each wait for compaction)
            std::lock_guard<std::mutex> lk(m);
            if (u + 1 < n_units) (void)madvise(base + (u + 1) * kUnit, (n_units - u - 1) * kUnit, MADV_NOHUGEPAGE);
            ++small_units;
        }
#else
        (void)ms;
#endif
        for (size_t o = 4096; o < kUnit; o += 4096) p[o] = 0;  // (already present behind a huge page: 511 stores)
    }
    void worker() {
        for (;;) {
            size_t u;
            {
                std::lock_guard<std::mutex> lk(m);
                if (stop || next_unit >= n_units) return;
                u = next_unit++;
            }
            fault_unit(u);
            {
                std::lock_guard<std::mutex> lk(m);
                done[u] = 1;
                while (front_units < n_units && done[front_units]) ++front_units;
                if (front_units == n_units) done_at_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_map).count();
            }
            cv.notify_all();
        }
    }
    void prefault_start(unsigned n_threads) {
        if (!base) return;
        n_threads = std::max(1u, std::min(n_threads, 16u));
        // (ONE thread is started here — ≈30 µs of the caller's time —, it starts the others and works itself)
        try {
            th.emplace_back([this, n_threads] {
                std::vector<std::thread> more;
                for (unsigned k = 1; k < n_threads; ++k) {
                    try { more.emplace_back([this] { worker(); }); } catch (...) { break; }
                }
                worker();
                for (auto &t : more) t.join();
            });
        } catch (...) {
            worker();  // (no thread to be had: here)
        }
    }
    // until [0, upto) of the block is faulted in (the fetch copies behind this front)
    void wait_front(size_t upto) {
        const size_t need = std::min(n_units, (upto + kUnit - 1) / kUnit);
        std::unique_lock<std::mutex> lk(m);
        if (front_units >= need) return;
        const auto t0 = std::chrono::steady_clock::now();
        cv.wait(lk, [&] { return front_units >= need; });
        waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void join() {
        for (auto &t : th) if (t.joinable()) t.join();
        th.clear();
    }
    void release() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        join();
#if defined(__linux__)
        if (map) (void)munmap(map, map_bytes);
#endif
        map = base = nullptr; map_bytes = bytes = 0; n_units = 0; done.clear();
    }
    template <typename T> T *array(int a) const { return reinterpret_cast<T *>(base + off[a]); }
};

inline void unhint_huge_pages(char *dst, size_t bytes) {
#if defined(__linux__) && defined(MADV_NOHUGEPAGE)
    const uintptr_t page = 4096;
    const uintptr_t a = ((uintptr_t)dst + page - 1) & ~(page - 1), b = ((uintptr_t)dst + bytes) & ~(page - 1);
    if (dst && b > a) (void)madvise((void *)a, b - a, MADV_NOHUGEPAGE);
#else
    (void)dst; (void)bytes;
#endif
}

// A fetched piece from its half of the staging block into the caller's array (the threads also take the page faults of a fresh
// destination in parallel).
// (Streaming stores: the destination is hundreds of megabytes that nobody reads before the fetch is over — written around the
//  caches, a piece costs its 16 MB of reads and 16 MB of writes, not a third 16 MB of read-for-ownership.)
inline void stream_copy(char *dst, const char *src, size_t n) {
#if defined(__x86_64__) && defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
    if (n >= ((size_t)64 << 10)) {
        const size_t head = (size_t)(-(intptr_t)dst) & 15;
        if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; ++i) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + 64 * i)), b = _mm_loadu_si128((const __m128i *)(src + 64 * i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + 64 * i + 32)), d = _mm_loadu_si128((const __m128i *)(src + 64 * i + 48));
            _mm_stream_si128((__m128i *)(dst + 64 * i), a);
            _mm_stream_si128((__m128i *)(dst + 64 * i + 16), b);
            _mm_stream_si128((__m128i *)(dst + 64 * i + 32), c);
            _mm_stream_si128((__m128i *)(dst + 64 * i + 48), d);
        }
        _mm_sfence();
        const size_t done = blocks * 64;
        if (n > done) memcpy(dst + done, src + done, n - done);
        return;
    }
#endif
    memcpy(dst, src, n);
}
inline void copy_into_place(char *dst, const char *src, size_t bytes) {
    par_ranges(bytes, (size_t)1 << 20, [&](size_t b0, size_t b1) { stream_copy(dst + b0, src + b0, b1 - b0); });
}

}  // namespace rthostpar
