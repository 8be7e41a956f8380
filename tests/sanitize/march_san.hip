// march_san.hip — the device march's per-lane logic (csrc/rt_device.hpp through tests/host_march.hip) on the host under
// AddressSanitizer + UndefinedBehaviorSanitizer: GPU sanitizers are not available on the pool, so the geometry code is
// sanitized where it can run.  For every mesh file: trace! two quadratures (csrc/rt_host.cpp), march every track with the
// walk step on and off (several k) and require identical records; the checker (oracle/rt_oracle.c, same flags) must
// agree too.  TEST INFRASTRUCTURE (tests/sanitize/run.sh).
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

namespace rthost {
thread_local std::string g_last_error;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace rthost

#include "../../raytracing.jl_amd/csrc/rt_host.cpp"
#include "../host_march.hip"

extern "C" {
void *orc_mesh_create(const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                      const int32_t *nc_ptrs, const int32_t *nc_data, const double *bb);
void orc_mesh_destroy(void *mv);
int64_t orc_segmentize(void *mv, int64_t n_tracks, const double *px, const double *py, const double *phi, const double *cosphi,
                       const double *sinphi, const double *A, const double *B, const double *C, const double *ell, double tiny_step,
                       int32_t k, double rtol, int64_t iter_cap, int32_t n_threads, int64_t *seg_offsets, int32_t *status,
                       int64_t *n_iters);
int64_t orc_fetch(void *mv, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element);
}

int main(int argc, char **argv) {
    long long segs = 0, walk = 0, cheap = 0, chain_dec = 0, chain_marg = 0;
    int bad = 0, n_mesh = 0;
    for (int a = 1; a < argc; ++a) {
        rt_msh *M = rt_msh_load(argv[a]);
        if (!M) continue;
        ++n_mesh;
        int32_t nn, nc, nnz;
        rt_msh_sizes(M, &nn, &nc, &nnz);
        std::vector<double> x(nn), y(nn);
        std::vector<int32_t> cells(3 * (size_t)nc), ptrs(nn + 1), data(nnz);
        double bb[4];
        rt_msh_fetch(M, x.data(), y.data(), cells.data(), ptrs.data(), data.data(), bb);
        rt_msh_free(M);
        for (int q = 0; q < 2; ++q) {
            const int n_azim = q ? 16 : 4, k = (a + q) % 3 == 0 ? 12 : (q ? 5 : 2);
            const double delta = (q ? 0.03 : 0.11) * std::min(bb[2] - bb[0], bb[3] - bb[1]);
            std::vector<int64_t> ntx(n_azim / 2), nty(n_azim / 2);
            const int64_t n = rt_trace_counts(bb[2] - bb[0], bb[3] - bb[1], n_azim, delta, ntx.data(), nty.data());
            if (n < 0) continue;
            const int32_t bcs[4] = {1, 1, 1, 1};
            std::vector<double> ph(n_azim / 2), ds(n_azim / 2), om(n_azim / 2);
            std::vector<int32_t> az(n), ti(n);
            std::vector<std::vector<double>> D(11, std::vector<double>(n));
            std::vector<std::vector<int8_t>> B8(4, std::vector<int8_t>(n));
            std::vector<int64_t> nf(n), nb(n);
            if (rt_trace(bb, n_azim, ntx.data(), nty.data(), bcs, ph.data(), ds.data(), om.data(), az.data(), ti.data(), D[0].data(), D[1].data(),
                         D[2].data(), D[3].data(), D[4].data(), D[5].data(), D[6].data(), D[7].data(), D[8].data(), D[9].data(), D[10].data(),
                         B8[0].data(), B8[1].data(), B8[2].data(), B8[3].data(), nf.data(), nb.data())) continue;
            std::vector<Rec> recs[3];
            std::vector<int64_t> offs[3];
            std::vector<int32_t> st[3];
            for (int w = 0; w < 3; ++w) {  // exact walk steps, walk off, cheap steps
                hostmarch_run(x.data(), y.data(), nn, cells.data(), nc, ptrs.data(), data.data(), bb, n, D[0].data(), D[1].data(), D[4].data(),
                              D[5].data(), D[6].data(), D[8].data(), D[9].data(), D[10].data(), D[7].data(), 1e-8, k, 1.4901161193847656e-8,
                              200000, w == 0 ? 1 : (w == 1 ? 0 : 2), 2, nullptr);
                recs[w] = g_res.recs; offs[w] = g_res.offsets; st[w] = g_res.status;
                if (w == 0) walk += g_res.stats[0];
                if (w == 2) cheap += g_res.stats[5];
            }
            {   // the Σℓ chain of k_materialise_lin (rt_device.hpp chain_*) on the cheap-step march's records, against the
                // left-to-right check, wherever it decides (tests/test_sum_chain_cpu.py has the tuned tolerances)
                std::vector<int32_t> cst(n);
                std::vector<double> cS(n), cE(n);
                const double cmax = std::max(std::max(fabs(bb[0]), fabs(bb[2])), std::max(fabs(bb[1]), fabs(bb[3])));
                for (const double rtol : {1.4901161193847656e-8, 1e-6, 1e-11}) {
                    hostmarch_chain(n, D[5].data(), D[6].data(), D[7].data(), rtol, cmax, cst.data(), cS.data(), cE.data());
                    for (int64_t u = 0; u < n; ++u) {
                        if (cst[u] == 2) { ++chain_marg; continue; }
                        ++chain_dec;
                        if ((cst[u] == 1) == rt::isapprox_s(D[7][u], cE[u], rtol)) { ++bad; printf("CHAIN MISMATCH %s track %lld rtol %g\n", argv[a], (long long)u, rtol); }
                    }
                }
            }
            void *orc = orc_mesh_create(x.data(), y.data(), nn, cells.data(), nc, ptrs.data(), data.data(), bb);
            std::vector<int64_t> ooff(n + 1);
            std::vector<int32_t> ost(n);
            const int64_t tot = orc_segmentize(orc, n, D[0].data(), D[1].data(), D[4].data(), D[5].data(), D[6].data(), D[8].data(), D[9].data(),
                                               D[10].data(), D[7].data(), 1e-8, k, 1.4901161193847656e-8, 200000, 1, ooff.data(), ost.data(), nullptr);
            std::vector<double> o[5];
            for (auto &v : o) v.resize(tot > 0 ? tot : 1);
            std::vector<int32_t> oel(tot > 0 ? tot : 1);
            orc_fetch(orc, o[0].data(), o[1].data(), o[2].data(), o[3].data(), o[4].data(), oel.data());
            orc_mesh_destroy(orc);
            bool ok = (int64_t)recs[0].size() == tot && recs[1].size() == recs[0].size() && recs[2].size() == recs[0].size() && offs[0] == offs[1] &&
                      offs[0] == offs[2] && st[0] == st[1] && st[0] == st[2] && offs[0] == ooff && st[0] == ost;
            for (int64_t i = 0; ok && i < tot; ++i) {
                const Rec &r0 = recs[0][i], &r1 = recs[1][i], &r2 = recs[2][i];
                ok = r0.px == r2.px && r0.py == r2.py && r0.qx == r2.qx && r0.qy == r2.qy && r0.ell == r2.ell && r0.element == r2.element &&
                     r0.px == r1.px && r0.py == r1.py && r0.qx == r1.qx && r0.qy == r1.qy && r0.ell == r1.ell && r0.element == r1.element &&
                     r0.px == o[0][i] && r0.py == o[1][i] && r0.qx == o[2][i] && r0.qy == o[3][i] && r0.ell == o[4][i] && r0.element == oel[i];
            }
            if (!ok) { ++bad; printf("MISMATCH %s nφ=%d k=%d\n", argv[a], n_azim, k); }
            segs += tot;
        }
    }
    printf("march_san: %d meshes, %lld segments (%lld by the walk step, %lld by cheap steps), exact walk steps == walk off == cheap steps == checker: %s; "
           "Σℓ chain: %lld decisions equal the left-to-right check's, %lld left to the exact sum\n",
           n_mesh, segs, walk, cheap, bad ? "MISMATCH" : "yes", chain_dec, chain_marg);
    return bad ? 1 : 0;
}
