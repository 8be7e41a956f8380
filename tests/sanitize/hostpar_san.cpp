// Sanitizer driver for raytracing.jl_amd/csrc/rt_hostpar.hpp — the host threading of rt_tracks_create and of the pipelined fetch,
// with the device replaced by memcpy (test infrastructure; tests/sanitize/run.sh builds it twice: -fsanitize=address,undefined
// and -fsanitize=thread).  What it exercises:
//   * par_ranges from one caller and from several at once (rt_multi_create uploads its shards from several host threads: a
//     caller that finds the team busy does its work alone), with ranges that do not divide evenly, and an exception in a worker;
//   * plan_march_order on track sets whose size is and is not a multiple of 64, in the three sort modes: the march order is a
//     permutation, reserved-chunk prefixes shrink with the chunk index, the compaction order is a permutation of the waves;
//   * the upload's two forms — one image, and ranges through the two halves of a staging block with a "copy engine" thread that
//     drains a half while the host fills the other (the events of the real code are a mutex + condition variable here);
//   * the fetch's drain: pieces through two halves into a fresh destination.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>

#include "../../raytracing.jl_amd/csrc/rt_hostpar.hpp"

using namespace rthostpar;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "hostpar_san: CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); ++g_fail; } } while (0)

static void test_par_ranges() {
    for (size_t n : {(size_t)0, (size_t)1, (size_t)63, (size_t)1000, (size_t)100003}) {
        std::vector<int> hit(n, 0);
        par_ranges(n, 7, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) ++hit[i]; });
        for (size_t i = 0; i < n; ++i) CHECK(hit[i] == 1);
    }
    // several callers at once: every range of every caller exactly once
    std::vector<std::thread> th;
    std::atomic<long> total{0};
    for (int c = 0; c < 6; ++c)
        th.emplace_back([&, c] {
            for (int rep = 0; rep < 50; ++rep) {
                const size_t n = 5000 + 977 * (size_t)c + (size_t)rep;
                std::vector<unsigned char> hit(n, 0);
                par_ranges(n, 64, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) ++hit[i]; });
                long s = 0;
                for (unsigned char h : hit) s += h;
                CHECK(s == (long)n);
                total += s;
            }
        });
    for (auto &t : th) t.join();
    CHECK(total.load() > 0);
    // what a worker throws reaches the caller
    bool caught = false;
    try {
        par_ranges(100000, 10, [&](size_t a, size_t) { if (a != 0) throw std::runtime_error("worker"); });
    } catch (const std::runtime_error &) { caught = true; }
    unsigned hw = std::thread::hardware_concurrency();
    CHECK(caught || hw <= 1);
}

static void test_plan(size_t n, int sort_mode, std::mt19937 &rng) {
    std::vector<double> ell(n);
    std::uniform_real_distribution<double> U(0.01, 3.0);
    for (auto &v : ell) v = U(rng);
    MarchPlan p;
    plan_march_order(ell.data(), n, sort_mode, /*kappa*/ 45.0, /*test_reserved_pct*/ -1, /*n_regions*/ 12, /*chunk_rows*/ 32, p);
    CHECK(p.perm.size() == n);
    std::vector<char> seen(n, 0);
    for (int32_t u : p.perm) { CHECK(u >= 0 && (size_t)u < n); if (u >= 0 && (size_t)u < n) { CHECK(!seen[u]); seen[u] = 1; } }
    const size_t nw = (n + 63) / 64;
    for (int j = 0; j + 1 < 12; ++j) CHECK(p.reg_cap[j] >= p.reg_cap[j + 1]);
    CHECK(n == 0 || (size_t)p.reg_cap[0] == nw);  // every wave has a first chunk reserved
    if (!p.corder.empty()) {
        CHECK(p.corder.size() == nw);
        std::vector<char> w(nw, 0);
        for (int32_t a : p.corder) { CHECK(a >= 0 && (size_t)a < nw); if (a >= 0 && (size_t)a < nw) { CHECK(!w[a]); w[a] = 1; } }
        for (size_t k = 1; k < nw; ++k) CHECK(p.perm[(size_t)p.corder[k - 1] * 64] < p.perm[(size_t)p.corder[k] * 64]);
    }
    if (sort_mode == 2)  // waves of consecutive uids, longest wave first
        for (size_t s = 0; s + 1 < n; ++s)
            if ((p.perm[s] & 63) != 63 && (size_t)p.perm[s] + 1 < n) CHECK(p.perm[s + 1] == p.perm[s] + 1);
}

// The upload as rt_tracks_create makes it, against a "device" that is host memory: returns the arena image
static void test_upload(size_t n, size_t block_bytes, std::mt19937 &rng) {
    std::vector<std::vector<double>> src(9, std::vector<double>(n));
    std::vector<int32_t> azim(n), perm(n);
    for (auto &a : src) for (auto &v : a) v = (double)rng();
    for (auto &v : azim) v = (int32_t)(rng() % 64) + 1;
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    const double *src8[9];
    for (int a = 0; a < 9; ++a) src8[a] = src[a].data();
    const size_t na = (n + 31) & ~(size_t)31;
    std::vector<unsigned char> dev(9 * na * 8 + 2 * na * 4, 0xee), block(block_bytes);
    const size_t up_bytes = 9 * na * 8 + 2 * na * 4;
    if (up_bytes <= block_bytes / 2) {
        pack_tracks_image(block.data(), na, 0, n, src8, azim.data(), perm.data());
        memcpy(dev.data(), block.data(), up_bytes);  // (hipMemcpyAsync + stream sync)
    } else {
        // ranges through the two halves; the "copy engine" drains a half while the host packs the other
        const size_t half = block_bytes / 2, per_track = 9 * 8 + 2 * 4;
        const size_t rcap = (half / per_track) & ~(size_t)63;
        CHECK(rcap > 0);
        struct Job { size_t i0, m; int h; };
        std::mutex mu;
        std::condition_variable cv;
        std::vector<Job> queue;
        bool done = false;
        int drained[2] = {0, 0};  // ranges of each half the engine has finished (what hipEventSynchronize waits for)
        std::thread engine([&] {
            for (;;) {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return done || !queue.empty(); });
                    if (queue.empty()) return;
                    j = queue.front(); queue.erase(queue.begin());
                }
                const unsigned char *hb = block.data() + (size_t)j.h * half;
                for (int a = 0; a < 9; ++a) memcpy(dev.data() + (size_t)a * na * 8 + j.i0 * 8, hb + (size_t)a * rcap * 8, j.m * 8);
                memcpy(dev.data() + 9 * na * 8 + j.i0 * 4, hb + 9 * rcap * 8, j.m * 4);
                memcpy(dev.data() + 9 * na * 8 + na * 4 + j.i0 * 4, hb + 9 * rcap * 8 + rcap * 4, j.m * 4);
                { std::lock_guard<std::mutex> lk(mu); ++drained[j.h]; }
                cv.notify_all();
            }
        });
        int k = 0, issued[2] = {0, 0};
        for (size_t i0 = 0; i0 < n; i0 += rcap, ++k) {
            const size_t m = std::min(n - i0, rcap);
            const int h = k & 1;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return drained[h] == issued[h]; }); }  // the half's previous range has left it
            pack_tracks_image(block.data() + (size_t)h * half, rcap, i0, m, src8, azim.data(), perm.data());
            { std::lock_guard<std::mutex> lk(mu); queue.push_back(Job{i0, m, h}); ++issued[h]; }
            cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(mu); done = true; }
        cv.notify_all();
        engine.join();
    }
    for (int a = 0; a < 9; ++a) CHECK(memcmp(dev.data() + (size_t)a * na * 8, src[a].data(), n * 8) == 0);
    CHECK(memcmp(dev.data() + 9 * na * 8, azim.data(), n * 4) == 0);
    CHECK(memcmp(dev.data() + 9 * na * 8 + na * 4, perm.data(), n * 4) == 0);
}

// fetch_pipelined's shape: the engine fills half k & 1 with piece k, the host drains piece k - 1 meanwhile
static void test_fetch(size_t bytes, size_t block_bytes, std::mt19937 &rng) {
    std::vector<unsigned char> dev(bytes), block(block_bytes);
    for (auto &v : dev) v = (unsigned char)rng();
    unsigned char *dst = (unsigned char *)malloc(bytes ? bytes : 1);  // (fresh pages, as a caller's new array)
    const size_t half = block_bytes / 2;
    std::mutex mu;
    std::condition_variable cv;
    int arrived[2] = {0, 0}, asked[2] = {0, 0};
    struct Req { size_t o, nb; int h; };
    std::vector<Req> queue;
    bool done = false;
    std::thread engine([&] {
        for (;;) {
            Req r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                r = queue.front(); queue.erase(queue.begin());
            }
            memcpy(block.data() + (size_t)r.h * half, dev.data() + r.o, r.nb);
            { std::lock_guard<std::mutex> lk(mu); ++arrived[r.h]; }
            cv.notify_all();
        }
    });
    struct Piece { unsigned char *d; size_t nb; int h; int seq; };
    Piece prev{nullptr, 0, 0, 0};
    auto drain = [&](const Piece &pc) {
        if (!pc.d) return;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived[pc.h] >= pc.seq; }); }
        copy_into_place((char *)pc.d, (const char *)block.data() + (size_t)pc.h * half, pc.nb);
    };
    int k = 0;
    for (size_t o = 0; o < bytes; o += half, ++k) {
        const size_t nb = std::min(half, bytes - o);
        const int h = k & 1;
        { std::lock_guard<std::mutex> lk(mu); queue.push_back(Req{o, nb, h}); ++asked[h]; }
        cv.notify_all();
        const int seq = asked[h];
        hint_huge_pages((char *)dst + o, nb);  // (the destination of the piece in flight, as fetch_pipelined does)
        drain(prev);  // (half h was drained two pieces ago: `prev` is the piece in the OTHER half)
        prev = Piece{dst + o, nb, h, seq};
    }
    drain(prev);
    { std::lock_guard<std::mutex> lk(mu); done = true; }
    cv.notify_all();
    engine.join();
    CHECK(bytes == 0 || memcmp(dst, dev.data(), bytes) == 0);
    free(dst);
}

// The library-owned result block (rt_result_alloc): mapped, faulted in by its background threads in address order, a "fetch" that
// copies behind the front (as fetch_pipelined does: wait_front, then the team's copy_into_place), a block replaced by a larger one
// while its threads still run, a block released while they run.
static void fail(const char *what) { fprintf(stderr, "hostpar_san: %s\n", what); ++g_fail; }
static void test_result_block(int64_t n_tracks, int64_t n_records, bool huge) {
    rthostpar::ResultBlock b;
    if (!b.map_for(n_tracks, n_records, huge)) { fail("result block: map_for"); return; }
    b.stall_ms = 0.05;  // (exercise the 4-KB fallback's madvise on some units)
    b.prefault_start(4);
    const size_t len[8] = {8 * (size_t)(n_tracks + 1), 4 * (size_t)n_tracks, 8 * (size_t)n_records, 8 * (size_t)n_records, 8 * (size_t)n_records,
                           8 * (size_t)n_records, 8 * (size_t)n_records, 4 * (size_t)n_records};
    std::vector<char> src((size_t)3 << 20);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (char)(i * 131u + 7u);
    for (int a = 0; a < 8; ++a) {
        char *dst = b.base + b.off[a];
        if (a + 1 < 8 && b.off[a] + len[a] > b.off[a + 1]) fail("result block: arrays overlap");
        for (size_t o = 0; o < len[a]; o += src.size()) {
            const size_t nb = std::min(src.size(), len[a] - o);
            b.wait_front((size_t)(dst + o + nb - b.base));
            rthostpar::copy_into_place(dst + o, src.data(), nb);
        }
    }
    for (int a = 0; a < 8; ++a)
        for (size_t o = 0; o < len[a]; o += 4099)
            if (b.base[b.off[a] + o] != src[o % src.size()]) { fail("result block: a copied byte was overwritten by the faulting threads"); break; }
    // a larger block while the threads of the first may still be running, then release while running
    if (!b.map_for(n_tracks, 2 * n_records + 1, huge)) { fail("result block: second map_for"); return; }
    b.prefault_start(3);
    b.wait_front(b.bytes / 2);
    b.release();
    if (b.base) fail("result block: release");
}

// the streaming copy of a fetched piece (head to 16-B alignment, 64-B blocks of non-temporal stores, tail): every alignment of the
// destination and the source, sizes around its thresholds — against memcpy, with guard bytes on either side
static void test_stream_copy() {
    std::vector<unsigned char> src((2u << 20) + 256), dst((2u << 20) + 512), ref((2u << 20) + 512);
    std::mt19937 r(99);
    for (auto &v : src) v = (unsigned char)r();
    for (size_t n : {(size_t)0, (size_t)1, (size_t)65535, (size_t)65536, (size_t)65537, (size_t)70001, ((size_t)1 << 20) + 13, (size_t)2 << 20})
        for (size_t da = 0; da < 33; da += (n > (1u << 20) ? 7 : 1))
            for (size_t sa : {(size_t)0, (size_t)1, (size_t)8, (size_t)15}) {
                std::fill(dst.begin(), dst.end(), (unsigned char)0xa5);
                std::fill(ref.begin(), ref.end(), (unsigned char)0xa5);
                rthostpar::stream_copy((char *)dst.data() + 64 + da, (const char *)src.data() + sa, n);
                memcpy(ref.data() + 64 + da, src.data() + sa, n);
                if (dst != ref) { fail("stream_copy differs from memcpy (or wrote outside its range)"); return; }
            }
}

int main() {
    std::mt19937 rng(20261004);
    test_stream_copy();
    test_result_block(1000, 300000, true);
    test_result_block(1, 1, false);
    test_result_block(130456, 1200000, true);
    test_par_ranges();
    for (size_t n : {(size_t)1, (size_t)63, (size_t)64, (size_t)130456, (size_t)300000, (size_t)65 * 4200 + 17})
        for (int mode = 0; mode < 3; ++mode) test_plan(n, mode, rng);
    test_upload(1000, 1u << 20, rng);        // one image
    test_upload(130456, 4u << 20, rng);      // ranges through the halves
    test_upload(70001, 1u << 20, rng);
    test_fetch(0, 1u << 20, rng);
    test_fetch(12345, 1u << 20, rng);
    test_fetch(37u << 20, 4u << 20, rng);
    // concurrent uploads (rt_multi_create): the team serves one, the others work alone
    {
        std::vector<std::thread> th;
        for (int c = 0; c < 4; ++c) th.emplace_back([c] { std::mt19937 r2(77 + c); test_upload(90000 + 1000 * (size_t)c, 2u << 20, r2); test_fetch((5u << 20) + c, 1u << 20, r2); });
        for (auto &t : th) t.join();
    }
    printf("hostpar_san: %s\n", g_fail ? "FAILED" : "par_ranges, march plans (18 sets x 3 modes), uploads (image / ranges / concurrent), fetch drains, result blocks: ok");
    return g_fail ? 1 : 0;
}
