/*
 * rt_segmentize.h — C ABI of the MI355X-native segmentize! path.
 *
 * The reference (RayTracing.jl, pure Julia) has no FFI seam; its boundary for this path is
 * the exported Julia function
 *     segmentize!(t::TrackGenerator{T}; k::Int=5, rtol::Real=Base.rtoldefault(T)) -> t
 *                                                   (src/trackgenerator.jl:357-369)
 * which runs _segmentize_track! (src/track.jl:106-178) over `tracks_by_uid` and then
 * fill_volumes (src/trackgenerator.jl:371-386).  A Julia shim binds the entry points below
 * with `ccall` (julia/RayTracingAMD.jl); the Python host mirror binds the
 * same symbols with ctypes (raytracing.jl_amd/_capi.py).
 *
 * Conventions: plain pointers and sizes only; every array argument is caller-owned host
 * memory unless its name ends in `_dev`; the library copies what it needs and owns only the
 * device memory behind its handles.  Ids are 1-based Int32 exactly as the reference holds
 * them (cell_nodes, node_cells, segment.element).  All real data is IEEE double.
 * Functions returning int32_t return RT_SUCCESS (0) or a negative RT_ERR_* code; the text
 * of the last failure on the calling thread is rt_last_error().  One host thread per
 * handle; work is enqueued on the handle's HIP stream.
 */
#ifndef RT_SEGMENTIZE_H
#define RT_SEGMENTIZE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

/* return codes */
#define RT_SUCCESS 0
#define RT_ERR_INVALID (-1)         /* bad argument                                      */
#define RT_ERR_HIP (-2)             /* a HIP runtime call failed (see rt_last_error)     */
#define RT_ERR_NO_DEVICE (-3)       /* no usable GPU                                     */
#define RT_ERR_NOT_SEGMENTIZED (-4) /* results requested before rt_segmentize            */

/* per-track status codes (what the reference signals by throwing) */
#define RT_TRACK_OK 0
#define RT_TRACK_LOCATE_FAILED 1      /* error("Try increasing `k`. ...")      src/track.jl:141 */
#define RT_TRACK_LENGTH_MISMATCH 2    /* error("Track with `uid` ... ")        src/track.jl:172 */
#define RT_TRACK_UNDEF_INTERSECTION 3 /* UndefVarError in intersections  src/intersection.jl:82-94 */
#define RT_TRACK_ITER_CAP 4           /* this library's guard on the reference's unbounded
                                         `continue` paths (src/track.jl:126-130,147-150,156-159) */

typedef struct rt_mesh rt_mesh;     /* device-resident flattened mesh + acceleration tables */
typedef struct rt_tracks rt_tracks; /* device-resident track set + the results of segmentize */

int32_t rt_abi_version(void);
const char *rt_last_error(void);
/* The message the reference throws for a per-track status (text of src/track.jl:141 / :172);
 * for RT_TRACK_LENGTH_MISMATCH the caller substitutes the uid for "%d". */
const char *rt_status_message(int32_t status);
/* Number of visible HIP devices (0 when there is none; never fails). */
int32_t rt_device_count(void);

/*
 * Mesh(model) — src/mesh.jl:24-31 (fields :10-17), consumed by find_element / inboundary
 * (src/mesh.jl:91-146) and intersections (src/intersection.jl:34-119).
 *   x, y[n_nodes]            get_node_coordinates(get_grid(model)), split into SoA
 *   cell_nodes[3*n_cells]    get_cell_node_ids(grid), 1-based, reference order per cell
 *   node_cells_ptrs[n_nodes+1], node_cells_data[ptrs[n_nodes]]
 *                            get_faces(topology, 0, 2) as CSR; ptrs may be 0- or 1-based
 *                            (Gridap's Table is 1-based), data are 1-based cell ids
 *   bb[4]                    bb_min.x, bb_min.y, bb_max.x, bb_max.y (src/mesh.jl:53-69)
 * The kd-tree (src/mesh.jl:38-42) is replaced by an exact nearest-node structure built
 * here.  Returns NULL on failure.
 */
rt_mesh *rt_mesh_create(int32_t device, const double *x, const double *y, int32_t n_nodes,
                        const int32_t *cell_nodes, int32_t n_cells, const int32_t *node_cells_ptrs,
                        const int32_t *node_cells_data, const double *bb);
void rt_mesh_destroy(rt_mesh *mesh);
/*
 * Diagnostics of the mesh preprocessing: which regime the march of this mesh runs in.  The device march takes,
 * per iteration, either the literal step (find_element + intersections, src/mesh.jl:103-146,
 * src/intersection.jl:34-119) or a "walk step" that predicts the next cell through the edge adjacency and proves
 * with per-record certificates that the reference's procedure gives the same answer; both give
 * bit-identical records.  info[i], i < n_info (RT_MESH_INFO_*): see the index names below.  `note` (may be NULL)
 * receives a 0-terminated remark of the preprocessing (why records were switched off), at most note_cap bytes.
 */
#define RT_MESH_INFO_WALK_ENABLED 0      /* 1: the walk step is in use (available and not switched off by "walk"=0) */
#define RT_MESH_INFO_RECORDS 1           /* (cell, entry edge) records = 3 * n_cells */
#define RT_MESH_INFO_RECORDS_WALK 2      /* records on which the walk step's certificates can hold */
#define RT_MESH_INFO_EPS_MIN 3           /* smallest / largest barycentric isolation margin among those records */
#define RT_MESH_INFO_EPS_MAX 4
#define RT_MESH_INFO_D_VERTEX 5          /* clearance of the track line from a cell's vertices required by the walk step */
#define RT_MESH_INFO_L_MIN 6             /* shortest chord the walk step handles */
#define RT_MESH_INFO_CELLS_FRAGILE 7     /* cells whose own barycentric test is too noisy at the sqrt(eps) level */
#define RT_MESH_INFO_CELLS_DEGENERATE 8  /* cells of (numerically) zero area */
#define RT_MESH_INFO_EDGES_NONMANIFOLD 9 /* edges shared by more than two cells */
#define RT_MESH_INFO_EXTRAS_MAX 10       /* largest scan-rank bound among the walk records */
#define RT_MESH_INFO_PREP_MS 11          /* host time of the preprocessing inside rt_mesh_create */
#define RT_MESH_INFO_KAPPA 12            /* expected segments per unit track length (Cauchy-Crofton) */
#define RT_MESH_INFO_WALK_AVAILABLE 13   /* 1: at least one record can be walked */
#define RT_MESH_INFO_RECORDS_CHEAP 14    /* records on which the cheap step's certificates can hold (subset of RECORDS_WALK) */
#define RT_MESH_INFO_TINY_MAX 15         /* ... for tiny_step <= this value (larger tiny_step: exact walk steps only) */
#define RT_MESH_INFO_COUNT 16
int32_t rt_mesh_info(rt_mesh *mesh, double *info, int32_t n_info, char *note, int32_t note_cap);

/* Enqueue all later work of this mesh's track sets on an existing hipStream_t (NULL = the
 * library's own stream). */
int32_t rt_mesh_set_stream(rt_mesh *mesh, void *hip_stream);
void *rt_mesh_get_stream(rt_mesh *mesh);
/* Optional hook: called by rt_segmentize on the calling thread once all kernels of the call have been
 * enqueued and before it waits for them, so that a host can put other work beside the march — e.g. the
 * RCCL all-reduce of the previous batch's volumes on another stream (bench.py, multi-GPU).  Not called
 * again if the call has to re-run (staging pool growth).  NULL removes the hook. */
typedef void (*rt_enqueue_hook)(void *user);
int32_t rt_mesh_set_enqueue_hook(rt_mesh *mesh, rt_enqueue_hook hook, void *user);

/*
 * The per-track inputs of _segmentize_track! (src/track.jl:106-108: track.p, ϕ, ℓ, ABC),
 * in uid order, as produced by trace! (src/trackgenerator.jl:179-273).  cos_phi / sin_phi
 * are cos(ϕ), sin(ϕ) evaluated by the host (advance_step, src/point.jl:43, evaluates them
 * with the host libm; the device never calls trig functions).  azim_idx is the 1-based
 * track.azim_idx that fill_volumes uses (src/trackgenerator.jl:379).
 */
rt_tracks *rt_tracks_create(rt_mesh *mesh, int64_t n_tracks, const double *px, const double *py,
                            const double *phi, const double *cos_phi, const double *sin_phi,
                            const double *A, const double *B, const double *C, const double *ell,
                            const int32_t *azim_idx);
void rt_tracks_destroy(rt_tracks *tracks);

/*
 * segmentize!(t; k, rtol) — src/trackgenerator.jl:357-369, including fill_volumes
 * (:371-386; delta_s[n_azim_2] = azimuthal_quadrature.δs).  tiny_step = t.tiny_step.
 * Runs every track's march on the device and leaves, behind `tracks`:
 *   seg_offsets[n_tracks+1]  CSR offsets: track u owns segments [off[u], off[u+1]) in march order
 *   status[n_tracks]         RT_TRACK_* per track
 *   px,py,qx,qy,ell[total]   segment.p, segment.q, segment.ℓ   (src/segment.jl:23-33)
 *   element[total]           segment.element, 1-based Int32
 *   volumes[n_cells]         t.volumes
 * `k` is find_element's knn width (src/mesh.jl:123): any k >= 0 is honoured (k < 0: RT_ERR_INVALID, where
 * NearestNeighbors throws); n_azim_2 must cover every track's azim_idx (else RT_ERR_INVALID).
 * Returns the total number of segments (>= 0) or a negative RT_ERR_* code.  A non-OK track
 * status is not an error of this call: the caller decides (the shims throw the reference's
 * message for the first failing uid).
 */
int64_t rt_segmentize(rt_tracks *tracks, double tiny_step, int32_t k, double rtol,
                      const double *delta_s, int32_t n_azim_2);

/* Number of tracks with status != RT_TRACK_OK after the last rt_segmentize, and the 1-based
 * uid of the first one (0 if none). */
int32_t rt_failed_tracks(rt_tracks *tracks, int64_t *n_failed, int64_t *first_uid, int32_t *first_status);
/* Completion.  A whole-track call with cheap steps (the default regime) ends with a one-workgroup kernel that copies the call's
 * control block (total, failure summary, cursors, statistics) to page-locked host memory and stores the call's sequence number behind
 * it; rt_segmentize returns when the host sees that number — everything is complete then (segmentize! semantics), a few microseconds
 * before the stream itself reports idle.  Stream-ordered calls — rt_set_option(mesh, "async", 1): for calls that march with exact
 * steps only, rt_segmentize returns as soon as the host knows the total and the failure summary (the scan's copy), while the
 * compaction may still be running on the mesh's stream; rt_sweep under the option queues its kernels and returns (ms = 0), so that
 * consecutive sweeps — with the source updated through rt_sweep_xs_pointer on the same stream — run without a host turnaround.
 * Every entry point that reads results (rt_fetch_*, rt_device_pointers, rt_fill_tau, rt_sweep*, rt_last_timing,
 * rt_tracks_destroy) first waits; a consumer with its own stream orders against rt_mesh_get_stream or calls rt_wait.  Default 0.
 * Calls that march track pieces, or with the "timing" option on, always complete before they return. */
int32_t rt_wait(rt_tracks *tracks);

/* Copy results into caller-allocated host buffers (any pointer may be NULL to skip it).  rt_fetch_segments moves the records in
 * pieces through a page-locked block the library keeps — the copy engine fills one half while host threads move the other into
 * place — so that pageable, freshly allocated destinations are filled at close to the PCIe rate (C3's 410 MB: 10 ms, C5's 5 GB:
 * 114 ms; a plain copy into pageable memory took 25-42 / 330-600 ms): what a caller that runs segmentize! ONCE should use. */
int32_t rt_fetch_offsets(rt_tracks *tracks, int64_t *seg_offsets, int32_t *status);
int32_t rt_fetch_segments(rt_tracks *tracks, double *px, double *py, double *qx, double *qy,
                          double *ell, int32_t *element);
int32_t rt_fetch_volumes(rt_tracks *tracks, double *volumes);
/* Like rt_fetch_segments, but into page-locked host buffers owned by the handle (allocated on first use and
 * reused): host_ptrs[6] receives px, py, qx, qy, ell (double *) and element (int32_t *), `total` entries
 * each.  These buffers take the records at the PCIe rate (C3's 410 MB: 7.3 ms) and hand them out without a copy —
 * for a caller that fetches REPEATEDLY: page-locking itself is slow (≈0.08 ms per MB: 25-40 ms for 410 MB, 260-550 ms
 * for 5 GB, on the first call), so one set of buffers survives its handle in a process-wide cache and serves the next handle.
 * The pointers stay valid until the next rt_segmentize, rt_fetch_segments_pinned or rt_tracks_destroy on
 * this handle. */
int32_t rt_fetch_segments_pinned(rt_tracks *tracks, void **host_ptrs);
/* Everything a host rebuilds track.segments from, in ONE call and one synchronisation: host_ptrs[8] receives seg_offsets
 * (int64_t *, n_tracks + 1), status (int32_t *, n_tracks) and the six record arrays as rt_fetch_segments_pinned returns
 * them — all page-locked and owned by the handle, same lifetime.  (rt_fetch_offsets into fresh pageable arrays costs two
 * synchronous copies: 6 ms for 1.5 MB at 130 k tracks, as long as the 410 MB of records take through pinned buffers.) */
int32_t rt_fetch_pinned(rt_tracks *tracks, void **host_ptrs);

/*
 * The same eight arrays in a host block that the LIBRARY owns (round 6).  One segmentize! ends with track.segments on the host
 * (src/trackgenerator.jl:357-369), and into fresh host memory that copy is bound by page faults whose cost depends on the state of
 * the box's huge pages.  rt_result_alloc maps an anonymous block aligned to 2 MB — seg_offsets[n_tracks + 1], status[n_tracks] and six
 * record arrays of the estimated record count (n_records_hint <= 0: the Cauchy–Crofton estimate from sum_ell = Σ track.ℓ) —, asks for
 * transparent huge pages BEFORE its first touch and returns at once: threads of the library fault the block in, in address order, in the
 * background (a unit whose 2-MB fault stalls switches what is still to come to 4-KB pages).  Call it BEFORE rt_tracks_create /
 * rt_segmentize: the upload and the kernels then run beside the page faults.  rt_result_fetch copies offsets, status and records of the
 * last rt_segmentize into the block behind that front (a block that turns out too small is replaced) and returns the eight host pointers
 * (host_ptrs[8], order as rt_fetch_pinned) and the record count; they stay valid until rt_result_free — independent of the track
 * handle's lifetime: a Julia host wraps them with unsafe_wrap(Array, ptr, n; own = false) and frees the block in a finalizer.
 */
typedef struct rt_result rt_result;
rt_result *rt_result_alloc(rt_mesh *mesh, int64_t n_tracks, double sum_ell, int64_t n_records_hint);
int32_t rt_result_fetch(rt_tracks *tracks, rt_result *result, void **host_ptrs, int64_t *total);
void rt_result_free(rt_result *result);

/*
 * Device-resident results for consumers that stay on the GPU (RCCL all-gather of shards,
 * a device-side transport sweep).  ptrs_dev[9] receives, in this order: seg_offsets (i64),
 * status (i32), px, py, qx, qy, ell (f64), element (i32), volumes (f64).  The pointers stay
 * valid until the next rt_segmentize / rt_tracks_destroy on this handle — except `volumes`, which
 * alternates between two buffers from call to call and stays valid (and untouched) during the next
 * call too, so that a host can still be reducing one batch's volumes across GPUs while the next
 * batch runs (bench.py).
 */
int32_t rt_device_pointers(rt_tracks *tracks, void **ptrs_dev);

/*
 * Records in COMPLETION order (round 6; rt_set_option(mesh, "record_order", 1 | 2)).  The reference keeps a Vector{Segment} per
 * track (track.segments, src/track.jl:18) and no order between tracks; a call under this option leaves every track's records
 * contiguous and in march order, as always, but the TRACKS in the order in which the march's workgroups ended: a workgroup that has
 * finished its tracks takes their span of the six arrays from an atomic cursor, and the record-writing kernel runs beside the rest
 * of the march instead of behind it (C3: -20 % per step).  What describes the layout is the per-track table
 *   seg_begin[n_tracks]   first record of track u        seg_count[n_tracks]   its number of records
 * rt_record_order: 1 if the handle's records currently lie in completion order, 0 if in CSR order (uid order).
 * rt_device_table: ptrs_dev[10] = seg_begin (i64), seg_count (i32), status (i32), px, py, qx, qy, ell (f64), element (i32),
 *   volumes (f64) — valid in either order, never rewrites anything (in CSR order seg_begin is seg_offsets).
 * rt_fetch_table / rt_fetch_records: the table and the six arrays as they lie on the device, to the host.
 * Every other entry point that hands out records (rt_fetch_offsets / _segments / _pinned, rt_result_fetch, rt_device_pointers,
 * rt_fill_tau, rt_sweep over the compact records, rt_multi_*) promises the CSR layout: on a handle in completion order it first
 * rewrites the records in uid order — once, by the record kernel's second run over the staged words (0.12 ms at C3) — and the
 * handle is in CSR order from then on.
 */
int32_t rt_record_order(rt_tracks *tracks);
int32_t rt_device_table(rt_tracks *tracks, void **ptrs_dev);
int32_t rt_fetch_table(rt_tracks *tracks, int64_t *seg_begin, int32_t *seg_count, int32_t *status);
int32_t rt_fetch_records(rt_tracks *tracks, double *px, double *py, double *qx, double *qy, double *ell, int32_t *element);

/*
 * A consumer of the device-resident records (the reference's consumption pattern is
 * "for track in tg.tracks_by_uid, for segment in track.segments: segment.ℓ, segment.element", README.md:127-135, and
 * Segment.τ is its "storage for transport-related data (e.g., optical thickness)", src/segment.jl:14,28): for every
 * segment s of the last rt_segmentize and every energy group g,
 *     τ[s * n_groups + g] = sigma_t[(element[s] - 1) * n_groups + g] * ℓ[s]
 * computed on the device from the records where they lie.  sigma_t: host array [n_cells * n_groups] (total cross
 * section per cell and group).  *tau_dev (may be NULL) receives the device pointer of τ (total * n_groups doubles,
 * owned by the handle, valid until the next rt_fill_tau / rt_segmentize / rt_tracks_destroy), *ms (may be NULL) the
 * kernel's HIP-event duration.  rt_fetch_tau copies τ to a caller-allocated host buffer.
 */
int32_t rt_fill_tau(rt_tracks *tracks, const double *sigma_t, int32_t n_groups, void **tau_dev, double *ms);
int32_t rt_fetch_tau(rt_tracks *tracks, double *tau);

/*
 * A transport sweep over the cyclic tracks, on the device — the consumer the reference's layout exists
 * for: "for track in tg.tracks_by_uid, for segment in track.segments" (README.md:127-135) with Segment.τ as its storage
 * (src/segment.jl:14,28), the tracks chained into closed loops by next_track_fwd / next_track_bwd and dir_next_track_fwd /
 * dir_next_track_bwd (src/track.jl:42-77; walked like this in demo/makie.jl:103-133).
 *
 * rt_sweep_set_links: the linking that trace! / next_tracks produced (src/trackgenerator.jl:231-348), in uid order, exactly
 * the arrays rt_trace returns: 1-based uids of next_track_fwd / next_track_bwd, dir_next_track_* (0 Forward, 1 Backward),
 * bc_fwd / bc_bwd (0 Vacuum, 1 Reflective, 2 Periodic).  A next uid of 0 means "not in this track set": nothing is handed on
 * locally (a uid shard of a multi-GPU run: the host sends psi_out of such traversals to the owner of the linked track, which
 * writes it into its psi_in — both reachable through rt_sweep_info; raytracing.jl_amd/distributed.py, ShardedSweep).
 *
 * rt_sweep: one method-of-characteristics sweep over the records of the last rt_segmentize.  Every track u is traversed
 * forward (its segments in march order, starting from psi_in[0][u][:]) and backward (reversed, from psi_in[1][u][:]); along
 * a segment of length ℓ in cell e, for every group g,
 *     τ = sigma_t[e][g]·ℓ,   Δ = (ψ − source[e][g] / sigma_t[e][g]) · (−expm1(−τ)),   ψ ← ψ − Δ,   φ[e][g] += w[u]·Δ
 * i.e. ψ_out = ψ_in·e^{−τ} + (q/Σt)(1 − e^{−τ}); a cell with sigma_t = 0 leaves ψ unchanged.  The flux a track ends with is
 * handed to the entry of the linked track, in the linked direction, as that entry's incoming flux for the NEXT sweep — 0
 * behind a Vacuum boundary — so repeated calls with psi_in = NULL iterate on the device (Jacobi over the boundary fluxes).
 *   n_groups            G >= 1; changing it resets the boundary fluxes to 0.  A "group" is any independent flux component:
 *                       a solver with polar angles θ_p passes G = groups x polar angles with sigma_t[e][(g, p)] = Σt_g / sin θ_p and
 *                       source[e][(g, p)] = q_g / sin θ_p (the ratio q/Σt is unchanged, τ is the 3-D optical length) and applies
 *                       the polar weights to φ when it folds the components
 *   sigma_t, source     [n_cells * G] total cross section and source per cell and group; NULL, NULL: those of the previous
 *                       call (source alone may be NULL: zero source)
 *   track_weight        [n_tracks] w[u]; NULL: the previous call's, or δs[azim_idx[u]] — the weight fill_volumes gives a
 *                       segment (src/trackgenerator.jl:379-382) — if none was ever given
 *   psi_in              [2][n_tracks][G] incoming boundary flux; NULL: what the previous sweep handed on (0 at first)
 *   input               1: the compact CSR records (ℓ and element — the reference's layout).  Round 6: they are swept as coalesced
 *                       (ℓ, cell) ROWS too — the staging's while the handle has them, else rows transposed ONCE per segmentation from
 *                       the records (rt_sweep_rows_kind = 2); rt_set_option "sweep_rows" 0: where they lie (12 B per segment and
 *                       direction, a wave-load touches 64 lines: 2.4x slower); 2: rows in the
 *                       march's staging layout, coalesced — (ℓ, cell) rows, 12 B, which a two-phase call with "compact" 0 writes
 *                       instead of the records and which the first sweep after any other two-phase call makes from the staged
 *                       words (once per segmentation); after a call with exact steps only: its (q, cell) rows, 20 B, ℓ = ‖p − q‖
 *                       rebuilt with the Segment constructor's expression, bit-identical — what makes
 *                       rt_set_option(mesh, "compact", 0) a complete step: march + offsets scan + sweep; 0 (recommended): rows
 *                       whenever the last call left them (whole-track calls do), else the compact records
 *   ms                  (may be NULL) HIP-event duration of the sweep's kernels
 * rt_sweep_fetch copies out (any may be NULL) phi[n_cells * G], psi_out[2][n_tracks][G] (the flux every traversal ended with)
 * and psi_next[2][n_tracks][G] (the boundary flux of the next sweep).  rt_sweep_info: the device pointers of those three
 * (ptrs_dev[3], may be NULL) and info[4] = {input used (1 / 2), groups per pass (0: global atomics), passes, n_groups}.
 * Results agree with a sequential evaluation to rounding: device expm1 and the order of the tallies' additions differ — and the
 * attenuation factor of an optically thin segment takes one of two forms (2 ulp apart) by what the other 63 lanes of its wave hold, so
 * the fluxes are not bitwise invariant across march orders / shardings of one problem (rt_set_option "sweep_debug" 4: one form always).
 */
int32_t rt_sweep_set_links(rt_tracks *tracks, const int64_t *next_fwd, const int64_t *next_bwd, const int8_t *dir_fwd,
                           const int8_t *dir_bwd, const int8_t *bc_fwd, const int8_t *bc_bwd);
int32_t rt_sweep(rt_tracks *tracks, int32_t n_groups, const double *sigma_t, const double *source,
                 const double *track_weight, const double *psi_in, int32_t input, double *ms);
int32_t rt_sweep_fetch(rt_tracks *tracks, double *phi, double *psi_out, double *psi_next);
int32_t rt_sweep_info(rt_tracks *tracks, void **ptrs_dev, int32_t *info);
/* How the last rt_sweep read its records: 0 where they lie (compact records, or the exact march's 20-B staging rows on their first
 * pass), 1 (ℓ, cell) rows in the staging's layout, 2 (ℓ, cell) rows made from the compact records; < 0: RT_ERR_*. */
int32_t rt_sweep_rows_kind(rt_tracks *tracks);
/* The device copy of the cross sections as rt_sweep reads them: [n_cells * n_groups][2] doubles = {sigma_t, source / sigma_t}
 * (0 for the second where sigma_t = 0).  A solver that updates its source on the device writes the second components there —
 * ordered against rt_mesh_get_stream — and calls rt_sweep with sigma_t = source = NULL ("those of the previous call"): no
 * host round trip between sweeps.  Valid until the group count changes. */
int32_t rt_sweep_xs_pointer(rt_tracks *tracks, void **xs_dev);

/*
 * HIP-event timings (milliseconds) of the last rt_segmentize on this handle, measured on
 * the stream the kernels ran on: ms[0] whole call (device side), ms[1] plan (track
 * binning), ms[2] march (the dominant kernel), ms[3] offsets scan, ms[4] compaction /
 * fill, ms[5] volumes.  Unused slots are 0.  n = capacity of ms (>= 6).
 * The events are recorded only while rt_set_option(mesh, "timing", 1) is in force (default 0: every event
 * record costs ≈4 µs of stream time between two kernels); without it all slots are 0.
 */
int32_t rt_last_timing(rt_tracks *tracks, double *ms, int32_t n);

/* Counters of the last rt_segmentize on this handle: stats[0] segment records, stats[1] records produced by the
 * literal step (the walk step produced the rest; track pieces' seeds in split mode count as neither), stats[2]
 * staging chunks used, stats[3] staging chunks allocated; if n allows: stats[4] waves per workgroup of the march kernel
 * the call launched, stats[5] 1 if it marched track pieces (split mode; 2: only the longest waves), stats[6] 1 for the wide-k instantiation,
 * stats[7] bytes of device memory this handle holds (inputs, staging pools, tables, results), stats[8] records decided by
 * cheap steps (a subset of the walk step's: the decision from the vertices' signed distances to the track line alone,
 * option "topo"; 0 when the call did not use them); stats[9..17] cheap steps REFUSED in the call, by the certificate term
 * that failed (a refusal may fail several): 9 no predicted record, 10 node-scan window (extras > k), 11 |s2| < d_vertex,
 * 12 entry edge not crossed, 13 m < E·D + g1 (isolation / border margin), 14 D < k2, 15 Dx < k2 (rounding of the entry / exit
 * point), 16 D·c1 < dtf (bound on tiny steps), 17 |s_v| < lc·lcf (chord length / order guard) — DESIGN.md §2; stats[18] tracks
 * whose Σℓ check (src/track.jl:171) lies within summation-order noise (64 ulp·n) of its rtol threshold: their status could
 * differ under Julia's pairwise / @simd `sum`; stats[19] tracks marched again with exact steps because the cheap steps'
 * iteration bound reached the iteration cap; stats[20] cheap records whose fill_volumes term was added from the record's own
 * length by the second kernel of the two-phase march (a short chord or a shallow crossing: the chord from the vertices' distances
 * would not be within 4e-11 of it — DESIGN.md §2).
 * stats[21] the lean plan of the call's march (option "lean"; 0: one kernel), stats[22] the lanes its k_serve finished; stats[23] the
 * kernel that wrote the call's records (1 k_compact3, 2 k_materialise, 3 k_materialise_lin, 4 k_materialise writing (ℓ, cell) rows);
 * stats[24] 1 if the call wrote its records beside the march, in completion order (option "record_order"); stats[25] side-list entries
 * the call used beyond the one reserved per track, stats[26] side-list entries allocated, stats[27] attempts the call took (> 1: a staging
 * pool, side list or result array that was too small on the first one, or a fall-back to another plan).
 * n = capacity of stats (>= 4). */
int32_t rt_last_stats(rt_tracks *tracks, int64_t *stats, int32_t n);

/* ---------------------------------------------------------------------------------------
 * Several GPUs behind one call.  segmentize! marches tracks_by_uid one after the other and every track writes
 * only its own segments (src/trackgenerator.jl:362-364): the path shards without a data-path exchange.  The mesh
 * is replicated on every device of device_ids[n_devices] (a device may be named more than once), tracks_by_uid is
 * cut into contiguous uid ranges of ≈ equal Σℓ, one per entry of device_ids, and every range runs the
 * single-device path from its own host thread.  Arguments as for rt_mesh_create + rt_tracks_create.
 * --------------------------------------------------------------------------------------- */
typedef struct rt_multi rt_multi;
rt_multi *rt_multi_create(const int32_t *device_ids, int32_t n_devices, const double *x, const double *y,
                          int32_t n_nodes, const int32_t *cell_nodes, int32_t n_cells,
                          const int32_t *node_cells_ptrs, const int32_t *node_cells_data, const double *bb,
                          int64_t n_tracks, const double *px, const double *py, const double *phi,
                          const double *cos_phi, const double *sin_phi, const double *A, const double *B,
                          const double *C, const double *ell, const int32_t *azim_idx);
void rt_multi_destroy(rt_multi *multi);
/* rt_set_option on every replica of the mesh. */
int32_t rt_multi_set_option(rt_multi *multi, const char *name, int64_t value);
/* segmentize!(t; k, rtol) over all shards at once (arguments as rt_segmentize).  Returns the global number of
 * segments or a negative RT_ERR_* code. */
int64_t rt_multi_segmentize(rt_multi *multi, double tiny_step, int32_t k, double rtol, const double *delta_s,
                            int32_t n_azim_2);
/* The partition: shard i owns 0-based uids [uid_begin[i], uid_begin[i+1]) and, after rt_multi_segmentize, the global
 * segment positions [seg_begin[i], seg_begin[i+1]); both arrays hold n_devices + 1 entries (either may be NULL).
 * Returns n_devices. */
int32_t rt_multi_shards(rt_multi *multi, int64_t *uid_begin, int64_t *seg_begin);
/* Shard i's own handle (owned by `multi`): rt_device_pointers / rt_last_timing / rt_last_stats per device. */
rt_tracks *rt_multi_shard(rt_multi *multi, int32_t i);
/* As rt_failed_tracks, with the GLOBAL 1-based uid of the first failing track. */
int32_t rt_multi_failed_tracks(rt_multi *multi, int64_t *n_failed, int64_t *first_uid, int32_t *first_status);
/* The global results in caller-allocated host buffers, exactly what rt_fetch_* of an unsharded run would hold:
 * seg_offsets[n_tracks+1], status[n_tracks], the six segment arrays [total] (every device copies its shard over its
 * own PCIe link, concurrently), volumes[n_cells] = the sum of the shards' volumes (fill_volumes is the only
 * reduction across tracks, src/trackgenerator.jl:371-386).  Any pointer may be NULL to skip it. */
int32_t rt_multi_fetch_offsets(rt_multi *multi, int64_t *seg_offsets, int32_t *status);
int32_t rt_multi_fetch_segments(rt_multi *multi, double *px, double *py, double *qx, double *qy, double *ell,
                                int32_t *element);
int32_t rt_multi_fetch_volumes(rt_multi *multi, double *volumes);
/* Reassemble the global segment list ON every shard's device with peer-to-peer copies (shard j travels to device i
 * over the xGMI link of that pair; all pairs at once).  ptrs_dev[n_devices*6] (may be NULL) receives, per shard i, the
 * device pointers of the global px, py, qx, qy, ell (f64) and element (i32) arrays on device_ids[i], valid until the
 * next rt_multi_allgather / rt_multi_destroy; *ms (may be NULL) the wall time of the copies. */
int32_t rt_multi_allgather(rt_multi *multi, void **ptrs_dev, double *ms);
/* Achieved rate of the last rt_multi_allgather per (destination i, source j) pair, GBs[i * n_devices + j] in GB/s of
 * 44-B records (0 where nothing was copied): every pair runs on its own stream, i.e. over its own xGMI link. */
int32_t rt_multi_link_rates(rt_multi *multi, double *GBs);

/* ---------------------------------------------------------------------------------------
 * Host-side rows around the hot path: they run on the CPU, like in the
 * reference, but natively — no Julia or Python needed to produce the device path's inputs.
 * --------------------------------------------------------------------------------------- */

/* TrackGenerator ctor, src/trackgenerator.jl:96-110 (+ argument checks of AzimuthalQuadrature,
 * src/azimuthal_quad.jl:21-25): per-angle track counts n_tracks_x / n_tracks_y [n_azim/2] for a
 * width x height domain.  Returns n_total_tracks, or RT_ERR_INVALID (DomainError text in
 * rt_last_error). */
int64_t rt_trace_counts(double width, double height, int32_t n_azim, double delta, int64_t *n_tracks_x,
                        int64_t *n_tracks_y);

/* trace!(t) + next_tracks, src/trackgenerator.jl:134-348.  bb = bb_min.x, bb_min.y, bb_max.x,
 * bb_max.y; bcs = {top, bottom, right, left} with 0 Vacuum, 1 Reflective, 2 Periodic
 * (src/boundary.jl:12-16).  Per-angle outputs [n_azim/2]: phis (ϕs), delta_s (δs), omega (ωₐ).
 * Per-track outputs in uid order [n_total_tracks]: azim_idx, track_idx (1-based), p, q, ϕ, cos ϕ,
 * sin ϕ, ℓ, ABC, bc_fwd / bc_bwd, dir_next_track_fwd / _bwd (0 Forward, 1 Backward), and the
 * 1-based uids of next_track_fwd / next_track_bwd.  Errors reproduce the reference's:
 * DomainError("could not found track exit point."), "Boundaries do not match!". */
int32_t rt_trace(const double *bb, int32_t n_azim, const int64_t *n_tracks_x, const int64_t *n_tracks_y,
                 const int32_t *bcs, double *phis, double *delta_s, double *omega, int32_t *azim_idx,
                 int32_t *track_idx, double *px, double *py, double *qx, double *qy, double *phi, double *cos_phi,
                 double *sin_phi, double *ell, double *A, double *B, double *C, int8_t *bc_fwd, int8_t *bc_bwd,
                 int8_t *dir_fwd, int8_t *dir_bwd, int64_t *next_fwd, int64_t *next_bwd);

/* Mesh ingest, replaces GmshDiscreteModel / DiscreteModelFromFile + Mesh(model) (src/mesh.jl:24-69) for
 * gmsh 4.1 ASCII files and for Gridap JSON models (a file that starts with '{': "grid" ->
 * "node_coordinates", "cell_node_ids"; what test/runtests.jl:5-6 and demo/pincell.jl:6-7 load): node
 * coordinates, cell->nodes (1-based; gmsh: ascending per cell as Gridap's oriented grid stores them,
 * JSON: as stored), node->cells CSR (0-based ptrs, 1-based ascending cell ids) and the bounding box —
 * exactly the arrays rt_mesh_create takes.  Returns NULL on failure. */
typedef struct rt_msh rt_msh;
rt_msh *rt_msh_load(const char *path);
int32_t rt_msh_sizes(rt_msh *msh, int32_t *n_nodes, int32_t *n_cells, int32_t *nnz);
int32_t rt_msh_fetch(rt_msh *msh, double *x, double *y, int32_t *cell_nodes, int32_t *node_cells_ptrs,
                     int32_t *node_cells_data, double *bb);
void rt_msh_free(rt_msh *msh);

/* Tunables (name/value); unknown names return RT_ERR_INVALID.  See DESIGN.md.  The ones a caller may want:
 *   "timing"   1: record HIP events for rt_last_timing (default 0)
 *   "async"    1: stream-ordered calls, see rt_wait (default 0)
 *   "sweep_ell" 0: every pass of rt_sweep over the staged rows derives ℓ from the exit points; default 1: the first pass after an
 *                 rt_segmentize keeps ℓ per row (8 B per staging slot) and every later pass and sweep reads (ℓ, cell) rows
 *   "compact"  0: rt_segmentize does not write the 44-B records — offsets, status and volumes are final, the records stay staged
 *                 (4-B words of the two-phase march, or 20-B rows) and are only produced when somebody asks for them
 *                 (rt_fetch_segments*, rt_device_pointers, rt_fill_tau); rt_sweep reads rows in the staging layout (default 1)
 *   "topo"     0: exact walk steps only, 1: cheap steps where >= 90 % of the walkable records carry a cheap certificate
 *                 (default), 2: forced — wherever a record carries one, and waves never hand back to exact steps
 *   "walk"     0: literal step only (find_element + intersections every iteration)
 *   "iter_cap" guard on the reference's unbounded `continue` paths (default 4,000,000 iterations per track)
 *   "split"    (read by rt_tracks_create) 0: never march track pieces; L > 0: pieces of about L records; default −1: pieces for
 *                 batches far below the chip's capacity (< 160 march waves; above that the two-phase march of whole tracks is faster)
 * Development and test knobs ("march_waves", "pool_chunks_hint", "side_entries_hint", "test_*", "sweep_*", "sort_mode") are
 * listed in DESIGN.md / tools/README.md; "single_pass" and "volumes_mode" exist only in a library built with -DRT_EXPERIMENTAL. */
int32_t rt_set_option(rt_mesh *mesh, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* RT_SEGMENTIZE_H */
