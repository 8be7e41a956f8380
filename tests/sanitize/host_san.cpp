// host_san.cpp — the host-side C++ of the library (csrc/rt_host.cpp: trace!, mesh ingest; csrc/rt_mesh_prep.hpp: node
// grid, walk records and their certificates) and the CPU checker (oracle/rt_oracle.c), compiled with
// -fsanitize=address,undefined and driven over mesh files given on the command line.  TEST INFRASTRUCTURE
// (tests/sanitize/run.sh builds and runs it; the log is kept under profiles/).  GPU sanitizers are not available on
// the pool, so the device code is covered by the same header compiled for the host (tests/host_march.hip) under
// the same flags.
//
// usage: host_san <mesh file>...      every file is loaded with rt_msh_load; well-formed ones go through prepare(),
//                                     rt_trace_counts / rt_trace (two quadratures) and a short checker run;
//                                     malformed ones must be refused with a message, never crash.
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
#include "../../raytracing.jl_amd/csrc/rt_mesh_prep.hpp"

extern "C" {
void *orc_mesh_create(const double *x, const double *y, int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                      const int32_t *nc_ptrs, const int32_t *nc_data, const double *bb);
void orc_mesh_destroy(void *mv);
int64_t orc_segmentize(void *mv, int64_t n_tracks, const double *px, const double *py, const double *phi, const double *cosphi,
                       const double *sinphi, const double *A, const double *B, const double *C, const double *ell, double tiny_step,
                       int32_t k, double rtol, int64_t iter_cap, int32_t n_threads, int64_t *seg_offsets, int32_t *status,
                       int64_t *n_iters);
}

int main(int argc, char **argv) {
    int loaded = 0, refused = 0;
    long long records = 0, walkable = 0, segments = 0;
    for (int a = 1; a < argc; ++a) {
        rt_msh *M = rt_msh_load(argv[a]);
        if (!M) { ++refused; printf("refused  %s: %s\n", argv[a], rthost::g_last_error.c_str()); continue; }
        ++loaded;
        int32_t nn, nc, nnz;
        rt_msh_sizes(M, &nn, &nc, &nnz);
        std::vector<double> x(nn), y(nn);
        std::vector<int32_t> cells(3 * (size_t)nc), ptrs(nn + 1), data(nnz);
        double bb[4];
        rt_msh_fetch(M, x.data(), y.data(), cells.data(), ptrs.data(), data.data(), bb);
        rt_msh_free(M);
        std::vector<int32_t> cn0(cells);
        for (auto &v : cn0) v -= 1;
        rtprep::Prep P = rtprep::prepare(x.data(), y.data(), nn, cn0.data(), nc, bb);
        records += P.n_records; walkable += P.n_records_walk;
        for (int q = 0; q < 2; ++q) {
            const int n_azim = q ? 16 : 4;
            const double delta = (q ? 0.03 : 0.11) * std::min(bb[2] - bb[0], bb[3] - bb[1]);
            std::vector<int64_t> ntx(n_azim / 2), nty(n_azim / 2);
            const int64_t n = rt_trace_counts(bb[2] - bb[0], bb[3] - bb[1], n_azim, delta, ntx.data(), nty.data());
            if (n < 0) { printf("trace_counts refused %s: %s\n", argv[a], rthost::g_last_error.c_str()); continue; }
            const int32_t bcs[4] = {1, 0, 2, 2};
            std::vector<double> ph(n_azim / 2), ds(n_azim / 2), om(n_azim / 2);
            std::vector<int32_t> az(n), ti(n);
            std::vector<std::vector<double>> D(11, std::vector<double>(n));
            std::vector<std::vector<int8_t>> B8(4, std::vector<int8_t>(n));
            std::vector<int64_t> nf(n), nb(n);
            const int rc = rt_trace(bb, n_azim, ntx.data(), nty.data(), bcs, ph.data(), ds.data(), om.data(), az.data(), ti.data(), D[0].data(),
                                    D[1].data(), D[2].data(), D[3].data(), D[4].data(), D[5].data(), D[6].data(), D[7].data(), D[8].data(),
                                    D[9].data(), D[10].data(), B8[0].data(), B8[1].data(), B8[2].data(), B8[3].data(), nf.data(), nb.data());
            if (rc) { printf("trace refused %s: %s\n", argv[a], rthost::g_last_error.c_str()); continue; }
            void *orc = orc_mesh_create(x.data(), y.data(), nn, cells.data(), nc, ptrs.data(), data.data(), bb);
            std::vector<int64_t> off(n + 1);
            std::vector<int32_t> st(n);
            segments += orc_segmentize(orc, n, D[0].data(), D[1].data(), D[4].data(), D[5].data(), D[6].data(), D[8].data(), D[9].data(),
                                       D[10].data(), D[7].data(), 1e-8, q ? 12 : 5, 1.4901161193847656e-8, 200000, 1, off.data(), st.data(), nullptr);
            orc_mesh_destroy(orc);
        }
    }
    // argument checks of the host entry points
    int64_t c[2];
    if (rt_trace_counts(1, 1, 6, 0.1, c, c) >= 0 || rt_trace_counts(1, 1, 4, -1, c, c) >= 0 || rt_trace_counts(1, 1, 4, 0.1, nullptr, c) >= 0) return 3;
    if (rt_trace(nullptr, 4, c, c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == 0) return 3;
    if (rt_msh_load(nullptr) || rt_msh_load("/nonexistent/file.msh")) return 3;
    printf("host_san: %d files loaded, %d refused; %lld walk records (%lld walkable); %lld checker segments\n", loaded, refused, records,
           walkable, segments);
    return 0;
}
