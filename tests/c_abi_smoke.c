/*
 * c_abi_smoke.c — a non-Python caller of the C ABI (include/rt_segmentize.h), in plain C.
 *
 * Makes the call sequence of the Julia shim (julia/RayTracingAMD.jl: segmentize_amd!) with the shim's argument
 * types: dlopen of the library by path (what Julia's `ccall((:sym, LIB), ...)` does), 1-based CSR `ptrs` as
 * Gridap's Table holds them, rt_mesh_create -> rt_tracks_create -> rt_segmentize -> rt_failed_tracks (+ the
 * "%d" -> uid substitution into rt_status_message's text) -> rt_fetch_pinned (offsets, status and the six record arrays in
 * page-locked buffers; cross-checked against rt_fetch_offsets) -> rt_fetch_volumes -> rt_sweep_set_links -> rt_sweep ->
 * rt_sweep_fetch (one transport sweep over the cyclic tracks on the device) -> destroy.  Inputs come from the library's own host rows (rt_msh_load, rt_trace_counts,
 * rt_trace), so no Julia or Python is involved.  Prints one JSON line with order-sensitive checksums of every
 * result array; tests/test_gpu_c_abi.py compares them with the checker's arrays.
 *
 * Build: gcc -O1 -std=c11 -o c_abi_smoke c_abi_smoke.c -ldl     (no link against the library: it is dlopen'ed)
 * Run:   c_abi_smoke <librt_segmentize.so> <mesh.msh> <n_azim> <delta> [fail_uid [n_shards]]
 *        n_shards > 0: the same through rt_multi_* with device_ids = {0, 0, ...} (julia/RayTracingAMD.jl: segmentize_amd_multi!)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_segmentize.h" /* types, constants and the prototypes the pointers below must match */

#define LOAD(name)                                                          \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name);      \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

static uint64_t sum_bits64(const void *a, int64_t n) {  /* Σ (i+1)·bits_i mod 2^64: order-sensitive, cheap to mirror in numpy */
    const uint64_t *w = (const uint64_t *)a;
    uint64_t s = 0;
    for (int64_t i = 0; i < n; ++i) s += (uint64_t)(i + 1) * w[i];
    return s;
}
static uint64_t sum_bits32(const int32_t *a, int64_t n) {
    uint64_t s = 0;
    for (int64_t i = 0; i < n; ++i) s += (uint64_t)(i + 1) * (uint64_t)(uint32_t)a[i];
    return s;
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s lib mesh.msh n_azim delta [fail_uid]\n", argv[0]); return 2; }
    const int32_t n_azim = atoi(argv[3]);
    const double delta = atof(argv[4]);
    const int64_t fail_uid = argc > 5 ? atoll(argv[5]) : 0; /* 1-based: spoil this track's length so that its Σℓ check fails */
    const int32_t n_shards = argc > 6 ? atoi(argv[6]) : 0;
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(rt_abi_version) LOAD(rt_last_error) LOAD(rt_status_message) LOAD(rt_device_count)
    LOAD(rt_mesh_create) LOAD(rt_mesh_destroy) LOAD(rt_mesh_info) LOAD(rt_tracks_create) LOAD(rt_tracks_destroy)
    LOAD(rt_segmentize) LOAD(rt_failed_tracks) LOAD(rt_fetch_offsets) LOAD(rt_fetch_segments_pinned) LOAD(rt_fetch_pinned) LOAD(rt_fetch_volumes)
    LOAD(rt_result_alloc) LOAD(rt_result_fetch) LOAD(rt_result_free)
    LOAD(rt_set_option) LOAD(rt_record_order) LOAD(rt_fetch_table) LOAD(rt_fetch_records)
    LOAD(rt_multi_create) LOAD(rt_multi_destroy) LOAD(rt_multi_segmentize) LOAD(rt_multi_failed_tracks) LOAD(rt_multi_fetch_offsets)
    LOAD(rt_multi_fetch_segments) LOAD(rt_multi_fetch_volumes) LOAD(rt_multi_shards)
    LOAD(rt_sweep_set_links) LOAD(rt_sweep) LOAD(rt_sweep_fetch) LOAD(rt_sweep_info)
    LOAD(rt_msh_load) LOAD(rt_msh_sizes) LOAD(rt_msh_fetch) LOAD(rt_msh_free) LOAD(rt_trace_counts) LOAD(rt_trace)
    if (p_rt_abi_version() != RT_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    if (p_rt_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }

    /* ---- Mesh(model): the arrays the shim flattens out of Gridap (src/mesh.jl:10-31) */
    rt_msh *msh = p_rt_msh_load(argv[2]);
    if (!msh) { fprintf(stderr, "rt_msh_load: %s\n", p_rt_last_error()); return 1; }
    int32_t n_nodes = 0, n_cells = 0, nnz = 0;
    p_rt_msh_sizes(msh, &n_nodes, &n_cells, &nnz);
    double *x = malloc(sizeof(double) * n_nodes), *y = malloc(sizeof(double) * n_nodes), bb[4];
    int32_t *cell_nodes = malloc(sizeof(int32_t) * 3 * (size_t)n_cells);
    int32_t *nc_ptrs = malloc(sizeof(int32_t) * ((size_t)n_nodes + 1)), *nc_data = malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (p_rt_msh_fetch(msh, x, y, cell_nodes, nc_ptrs, nc_data, bb)) { fprintf(stderr, "rt_msh_fetch: %s\n", p_rt_last_error()); return 1; }
    p_rt_msh_free(msh);
    for (int32_t i = 0; i <= n_nodes; ++i) nc_ptrs[i] += 1; /* Gridap's Table.ptrs is 1-based; the shim passes it as is */

    /* ---- TrackGenerator ctor + trace! (host rows of the library) */
    const int32_t n2 = n_azim / 2;
    int64_t *ntx = malloc(sizeof(int64_t) * n2), *nty = malloc(sizeof(int64_t) * n2);
    const int64_t n = p_rt_trace_counts(bb[2] - bb[0], bb[3] - bb[1], n_azim, delta, ntx, nty);
    if (n < 0) { fprintf(stderr, "rt_trace_counts: %s\n", p_rt_last_error()); return 1; }
    const int32_t bcs[4] = {1, 1, 1, 1}; /* all Reflective, as test/runtests.jl:8 */
    double *phis = malloc(sizeof(double) * n2), *delta_s = malloc(sizeof(double) * n2), *omega = malloc(sizeof(double) * n2);
    int32_t *azim = malloc(sizeof(int32_t) * n), *tidx = malloc(sizeof(int32_t) * n);
    double *D[11];
    for (int a = 0; a < 11; ++a) D[a] = malloc(sizeof(double) * (n > 0 ? n : 1));
    double *px = D[0], *py = D[1], *qx_t = D[2], *qy_t = D[3], *phi = D[4], *cs = D[5], *sn = D[6], *ell = D[7], *A = D[8], *B = D[9], *C = D[10];
    int8_t *b8[4];
    for (int a = 0; a < 4; ++a) b8[a] = malloc((size_t)(n > 0 ? n : 1));
    int64_t *nf = malloc(sizeof(int64_t) * n), *nb = malloc(sizeof(int64_t) * n);
    if (p_rt_trace(bb, n_azim, ntx, nty, bcs, phis, delta_s, omega, azim, tidx, px, py, qx_t, qy_t, phi, cs, sn, ell, A, B, C, b8[0],
                   b8[1], b8[2], b8[3], nf, nb)) { fprintf(stderr, "rt_trace: %s\n", p_rt_last_error()); return 1; }
    if (fail_uid >= 1 && fail_uid <= n) ell[fail_uid - 1] *= 1.0 + 1e-6;

    const double rtol = 1.4901161193847656e-8; /* Base.rtoldefault(Float64) */
    int64_t total, n_failed = 0, first_uid = 0;
    int32_t first_status = 0, walk_enabled = -1;
    int64_t *offs = malloc(sizeof(int64_t) * ((size_t)n + 1));
    int32_t *status = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    double *volumes = malloc(sizeof(double) * n_cells);
    void *hp[6];
    double sweep_phi = 0.0, sweep_psi = 0.0;
    int32_t sweep_input = 0, table_order = -1, table_ok = -1;
    rt_mesh *hm = NULL;
    rt_tracks *ht = NULL;
    rt_multi *mm = NULL;
    if (n_shards > 0) {
        /* ---- several devices behind one call (here: the one device, n_shards times) */
        int32_t ids[64];
        for (int i = 0; i < n_shards && i < 64; ++i) ids[i] = 0;
        mm = p_rt_multi_create(ids, n_shards, x, y, n_nodes, cell_nodes, n_cells, nc_ptrs, nc_data, bb, n, px, py, phi, cs, sn, A, B, C, ell, azim);
        if (!mm) { fprintf(stderr, "rt_multi_create: %s\n", p_rt_last_error()); return 1; }
        total = p_rt_multi_segmentize(mm, 1e-8, 5, rtol, delta_s, n2);
        if (total < 0) { fprintf(stderr, "rt_multi_segmentize: %s\n", p_rt_last_error()); return 1; }
        p_rt_multi_failed_tracks(mm, &n_failed, &first_uid, &first_status);
        if (p_rt_multi_fetch_offsets(mm, offs, status)) { fprintf(stderr, "rt_multi_fetch_offsets: %s\n", p_rt_last_error()); return 1; }
        for (int a = 0; a < 6; ++a) hp[a] = malloc((size_t)(total > 0 ? total : 1) * (a < 5 ? sizeof(double) : sizeof(int32_t)));
        if (p_rt_multi_fetch_segments(mm, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5])) { fprintf(stderr, "rt_multi_fetch_segments: %s\n", p_rt_last_error()); return 1; }
        if (p_rt_multi_fetch_volumes(mm, volumes)) { fprintf(stderr, "rt_multi_fetch_volumes: %s\n", p_rt_last_error()); return 1; }
        int64_t ub[65], sb[65];
        if (p_rt_multi_shards(mm, ub, sb) != n_shards || ub[n_shards] != n || sb[n_shards] != total) { fprintf(stderr, "rt_multi_shards: inconsistent\n"); return 1; }
    } else {
    /* ---- the shim's sequence */
    hm = p_rt_mesh_create(0, x, y, n_nodes, cell_nodes, n_cells, nc_ptrs, nc_data, bb);
    if (!hm) { fprintf(stderr, "rt_mesh_create: %s\n", p_rt_last_error()); return 1; }
    double info[RT_MESH_INFO_COUNT];
    char note[128];
    p_rt_mesh_info(hm, info, RT_MESH_INFO_COUNT, note, sizeof note);
    walk_enabled = (int32_t)info[RT_MESH_INFO_WALK_ENABLED];
    /* the destination first (round 6): a host block of the library's, faulted in in the background from here on */
    double sum_ell = 0.0;
    for (int64_t u = 0; u < n; ++u) sum_ell += ell[u];
    rt_result *hr = p_rt_result_alloc(hm, n, sum_ell, 0);
    if (!hr) { fprintf(stderr, "rt_result_alloc: %s\n", p_rt_last_error()); return 1; }
    ht = p_rt_tracks_create(hm, n, px, py, phi, cs, sn, A, B, C, ell, azim);
    if (!ht) { fprintf(stderr, "rt_tracks_create: %s\n", p_rt_last_error()); return 1; }
    total = p_rt_segmentize(ht, 1e-8, 5, rtol, delta_s, n2);
    if (total < 0) { fprintf(stderr, "rt_segmentize: %s\n", p_rt_last_error()); return 1; }
    p_rt_failed_tracks(ht, &n_failed, &first_uid, &first_status);
    /* offsets, status and the six record arrays in page-locked buffers of the handle: one call, one synchronisation */
    void *hp8[8];
    if (p_rt_fetch_pinned(ht, hp8)) { fprintf(stderr, "rt_fetch_pinned: %s\n", p_rt_last_error()); return 1; }
    /* (and the older pair of calls gives the same bytes) */
    if (p_rt_fetch_offsets(ht, offs, status)) { fprintf(stderr, "rt_fetch_offsets: %s\n", p_rt_last_error()); return 1; }
    if (memcmp(offs, hp8[0], sizeof(int64_t) * ((size_t)n + 1)) || memcmp(status, hp8[1], sizeof(int32_t) * (size_t)n)) {
        fprintf(stderr, "rt_fetch_pinned: offsets / status differ from rt_fetch_offsets\n");
        return 1;
    }
    for (int a = 0; a < 6; ++a) hp[a] = hp8[2 + a];
    {   /* ... and the library's own host block holds the same bytes (what the shim's eager rebuild reads) */
        void *hb8[8];
        int64_t tot2 = -1;
        if (p_rt_result_fetch(ht, hr, hb8, &tot2)) { fprintf(stderr, "rt_result_fetch: %s\n", p_rt_last_error()); return 1; }
        int same = tot2 == total && !memcmp(offs, hb8[0], sizeof(int64_t) * ((size_t)n + 1)) && !memcmp(status, hb8[1], sizeof(int32_t) * (size_t)n);
        for (int a = 0; a < 6 && same; ++a) same = !memcmp(hp8[2 + a], hb8[2 + a], (size_t)total * (a < 5 ? sizeof(double) : sizeof(int32_t)));
        if (!same) { fprintf(stderr, "rt_result_fetch: the block differs from rt_fetch_pinned\n"); return 1; }
        p_rt_result_free(hr);
    }
    if (p_rt_fetch_volumes(ht, volumes)) { fprintf(stderr, "rt_fetch_volumes: %s\n", p_rt_last_error()); return 1; }
    {   /* ---- the same call with the records in COMPLETION order (option "record_order"; whole tracks: "split" 0): a second mesh
         *      handle, and every track's records through the per-track table against the CSR arrays above, byte for byte */
        rt_mesh *hm2 = p_rt_mesh_create(0, x, y, n_nodes, cell_nodes, n_cells, nc_ptrs, nc_data, bb);
        if (!hm2) { fprintf(stderr, "rt_mesh_create (2): %s\n", p_rt_last_error()); return 1; }
        p_rt_set_option(hm2, "record_order", 2);
        p_rt_set_option(hm2, "split", 0);
        rt_tracks *ht2 = p_rt_tracks_create(hm2, n, px, py, phi, cs, sn, A, B, C, ell, azim);
        if (!ht2) { fprintf(stderr, "rt_tracks_create (2): %s\n", p_rt_last_error()); return 1; }
        const int64_t total2 = p_rt_segmentize(ht2, 1e-8, 5, rtol, delta_s, n2);
        table_order = p_rt_record_order(ht2);
        int64_t *beg = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        int32_t *cnt = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)), *st2 = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        void *rp[6];
        for (int a = 0; a < 6; ++a) rp[a] = malloc((size_t)(total2 > 0 ? total2 : 1) * (a < 5 ? sizeof(double) : sizeof(int32_t)));
        if (p_rt_fetch_table(ht2, beg, cnt, st2)) { fprintf(stderr, "rt_fetch_table: %s\n", p_rt_last_error()); return 1; }
        if (p_rt_fetch_records(ht2, rp[0], rp[1], rp[2], rp[3], rp[4], rp[5])) { fprintf(stderr, "rt_fetch_records: %s\n", p_rt_last_error()); return 1; }
        table_ok = total2 == total && !memcmp(st2, status, sizeof(int32_t) * (size_t)n);
        for (int64_t u = 0; u < n && table_ok; ++u) {
            table_ok = cnt[u] == offs[u + 1] - offs[u] && beg[u] >= 0 && beg[u] + cnt[u] <= total;
            for (int a = 0; a < 6 && table_ok; ++a) {
                const size_t w = a < 5 ? sizeof(double) : sizeof(int32_t);
                table_ok = !memcmp((const char *)rp[a] + (size_t)beg[u] * w, (const char *)hp[a] + (size_t)offs[u] * w, (size_t)cnt[u] * w);
            }
        }
        p_rt_tracks_destroy(ht2);
        p_rt_mesh_destroy(hm2);
    }
    /* ---- a consumer that stays on the device: one transport sweep over the cyclic tracks with the linking rt_trace produced
     *      (next_track_fwd / next_track_bwd, dir_next_track_*, bc_*: src/trackgenerator.jl:231-348), two groups */
    {
        const int32_t G = 2;
        const size_t ncg = (size_t)n_cells * G;
        double *sig = malloc(sizeof(double) * ncg), *src = malloc(sizeof(double) * ncg), *phi_t = malloc(sizeof(double) * ncg);
        double *psi_in = malloc(sizeof(double) * 2 * (size_t)n * G), *psi_out = malloc(sizeof(double) * 2 * (size_t)n * G);
        for (size_t i = 0; i < ncg; ++i) { sig[i] = 0.2 + 1.4 * (double)i / (double)(ncg - 1); src[i] = (double)i / (double)(ncg - 1); }
        for (size_t i = 0; i < 2 * (size_t)n * G; ++i) psi_in[i] = 1.0;
        if (p_rt_sweep_set_links(ht, nf, nb, b8[2], b8[3], b8[0], b8[1])) { fprintf(stderr, "rt_sweep_set_links: %s\n", p_rt_last_error()); return 1; }
        double ms = 0.0;
        if (p_rt_sweep(ht, G, sig, src, NULL, psi_in, 0, &ms)) { fprintf(stderr, "rt_sweep: %s\n", p_rt_last_error()); return 1; }
        if (p_rt_sweep_fetch(ht, phi_t, psi_out, NULL)) { fprintf(stderr, "rt_sweep_fetch: %s\n", p_rt_last_error()); return 1; }
        int32_t info[4] = {0, 0, 0, 0};
        p_rt_sweep_info(ht, NULL, info);
        for (size_t i = 0; i < ncg; ++i) sweep_phi += phi_t[i];
        for (size_t i = 0; i < 2 * (size_t)n * G; ++i) sweep_psi += psi_out[i];
        sweep_input = info[0];
    }
    }
    char message[512] = "";
    if (n_failed > 0) { /* error(replace(msg, "%d" => string(uid))) */
        const char *msg = p_rt_status_message(first_status);
        const char *at = strstr(msg, "%d");
        if (at) snprintf(message, sizeof message, "%.*s%" PRId64 "%s", (int)(at - msg), msg, first_uid, at + 2);
        else snprintf(message, sizeof message, "%s", msg);
    }
    double vsum = 0.0;
    for (int32_t c = 0; c < n_cells; ++c) vsum += volumes[c];
    /* per-track rebuild as the shim does it: walk the CSR ranges once (here: just check they tile the arrays) */
    int64_t walked = 0;
    for (int64_t u = 0; u < n; ++u) walked += offs[u + 1] - offs[u];
    printf("{\"n_tracks\": %" PRId64 ", \"total\": %" PRId64 ", \"walked\": %" PRId64 ", \"n_failed\": %" PRId64
           ", \"first_uid\": %" PRId64 ", \"first_status\": %d, \"message\": \"%s\", \"walk_enabled\": %d, "
           "\"sum_offsets\": %" PRIu64 ", \"sum_status\": %" PRIu64 ", \"px\": %" PRIu64 ", \"py\": %" PRIu64 ", \"qx\": %" PRIu64
           ", \"qy\": %" PRIu64 ", \"ell\": %" PRIu64 ", \"element\": %" PRIu64 ", \"volumes_sum\": %.17g, \"tracks_px\": %" PRIu64
           ", \"sweep_phi_sum\": %.17g, \"sweep_psi_out_sum\": %.17g, \"sweep_input\": %d, \"table_order\": %d, \"table_ok\": %d}\n",
           n, total, walked, n_failed, first_uid, first_status, message, (int)walk_enabled,
           sum_bits64(offs, n + 1), sum_bits32(status, n), sum_bits64(hp[0], total), sum_bits64(hp[1], total), sum_bits64(hp[2], total),
           sum_bits64(hp[3], total), sum_bits64(hp[4], total), sum_bits32((const int32_t *)hp[5], total), vsum, sum_bits64(px, n),
           sweep_phi, sweep_psi, (int)sweep_input, (int)table_order, (int)table_ok);
    if (mm) p_rt_multi_destroy(mm);
    if (ht) p_rt_tracks_destroy(ht);
    if (hm) p_rt_mesh_destroy(hm);
    return 0;
}
