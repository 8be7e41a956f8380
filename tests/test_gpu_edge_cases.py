"""Edge cases of the HIP path against the oracle (GPU only): parameters the reference exposes
(k, rtol, tiny_step), shifted / anisotropic domains, structured meshes whose tracks run through
vertices and along edges, tiny batches, and the library's internal modes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("px", "py", "qx", "qy", "ell")


def _oracle(orc, tg, **kw):
    om = orc.OracleMesh.from_mesh(tg.mesh)
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                      tiny_step=tg.tiny_step, iter_cap=4000000, **kw)
    aq = tg.azimuthal_quadrature
    r["volumes"] = om.fill_volumes(r["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    return r


def _same(tg, ref, check_volumes=True):
    s = tg.segments
    assert np.array_equal(tg.track_status, ref["status"]), "per-track status differs"
    assert np.array_equal(s.offsets, ref["offsets"]), "segment counts differ"
    assert np.array_equal(s.element, ref["element"]), "element ids differ"
    for k in FIELDS:
        a, b = getattr(s, k), ref[k]
        assert np.all(np.abs(a - b) <= 1e-10 * np.maximum(np.abs(b), 1e-300)), k
    if check_volumes:
        assert np.allclose(tg.volumes, ref["volumes"], rtol=1e-10, atol=1e-300)


@pytest.mark.parametrize("k", [0, 1, 2, 3, 8, 12, 40])
def test_knn_width(rt, orc, traced, k):
    tg = traced(16, 0.02)
    rt.segmentize(tg, k=k, check=False)
    _same(tg, _oracle(orc, tg, k=k))


@pytest.mark.parametrize("k", [2, 5, 12, 40])
def test_knn_width_on_a_distorted_mesh(rt, orc, k):
    """find_element(mesh, x, k) takes any k (src/mesh.jl:123); on this random Delaunay mesh the reference fails to
    locate on many tracks with k = 2 and on fewer with a wider search — the device must follow for every k (beyond
    the 8-entry in-register list the node list is streamed), with the walk step on and off."""
    import meshgen

    model = meshgen.random_model(rt, 18, 568)
    tg = rt.TrackGenerator(model, 16, 0.01)
    rt.trace(tg)
    ref = _oracle(orc, tg, k=k)
    for walk in (True, False):
        rt.segmentize(tg, k=k, check=False, walk=walk)
        _same(tg, ref)
    print(f"k={k}: {int(np.count_nonzero(ref['status'] == 1))} of {tg.n_total_tracks} tracks fail to locate; "
          f"regime {tg.device_mesh.info()}")


@pytest.mark.parametrize("tiny", [1e-7, 1e-9])
def test_tiny_step(rt, orc, pincell, tiny):
    tg = rt.TrackGenerator(pincell, 8, 0.03, tiny_step=tiny)
    rt.trace(tg)
    rt.segmentize(tg, check=False)
    _same(tg, _oracle(orc, tg))


@pytest.mark.parametrize("tiny", [1e-6, 5e-8, 1e-8, 1e-10])
def test_tiny_step_with_and_without_cheap_steps(rt, orc, pincell, tiny):
    """The cheap step's certificates hold up to RT_MESH_INFO_TINY_MAX (pincell: 5.1e-8); a larger tiny_step marches with
    exact walk steps, a smaller one bounds more tiny steps per record.  Whole tracks (split = 0), against the checker."""
    from raytracing_jl_amd import _capi

    tg = rt.TrackGenerator(pincell, 16, 0.01, tiny_step=tiny)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("split", 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    off, st = dt.fetch_offsets()
    s = dt.fetch_segments()
    assert np.array_equal(st, ref["status"]) and np.array_equal(off, ref["offsets"]) and np.array_equal(s["element"], ref["element"])
    for k in FIELDS:
        assert np.array_equal(s[k], ref[k]), k
    info, stats = dm.info(), dt.stats()
    assert (stats["cheap_records"] > 0) == (tiny <= info["cheap_tiny_max"]), (info["cheap_tiny_max"], stats)
    print(f"tiny_step={tiny:g}: tiny_max {info['cheap_tiny_max']:.2e}, {stats['cheap_records']} of {stats['records']} records by cheap steps")


def test_rtol_controls_length_check(rt, orc, pincell):
    tg = rt.TrackGenerator(pincell, 8, 0.05)
    rt.trace(tg)
    tg.ell = tg.ell.copy()
    tg.ell[7] *= 1.0 + 1e-6
    with pytest.raises(RuntimeError, match="Track with `uid` 8 has a length"):
        rt.segmentize(tg)
    rt.segmentize(tg, rtol=1e-5)
    rt.segmentize(tg, check=False)
    assert tg.track_status[7] == 2 and np.count_nonzero(tg.track_status) == 1


def test_length_check_at_the_threshold_is_the_left_to_right_sum(rt, orc, traced):
    """isapprox(track.ℓ, Σℓ; rtol) (src/track.jl:171) for tracks whose ℓ sits within a few ulp of the threshold: the two-phase
    march adds a track's lengths in another order than the reference's left-to-right sum, decides by margin, and hands the
    tracks inside the margin to k_finish's left-to-right sum — statuses equal the checker's, and rt_last_stats counts them."""
    from raytracing_jl_amd import _capi

    tg0 = traced(16, 1e-2)
    om = orc.OracleMesh.from_mesh(tg0.mesh, omp=True)
    kw = dict(cos_phi=tg0.cos_phi, sin_phi=tg0.sin_phi, tiny_step=tg0.tiny_step, n_threads=0)
    base = om.segmentize(tg0.px, tg0.py, tg0.phi, tg0.A, tg0.B, tg0.C, tg0.ell, **kw)
    rtol = 1e-6
    ell = tg0.ell.copy()
    picked = list(range(0, len(ell), 7))[:200]
    for n, u in enumerate(picked):
        seg = base["ell"][base["offsets"][u]:base["offsets"][u + 1]]
        S = 0.0
        for v in seg:  # left to right, as the reference's check (and the checker) add
            S += float(v)
        L = S / (1.0 - rtol)  # |L − S| = rtol·L up to rounding
        ell[u] = np.nextafter(L, np.inf if n % 2 else -np.inf) if n % 3 else L
        for _ in range(n % 5):
            ell[u] = np.nextafter(ell[u], np.inf if n % 2 else -np.inf)
    ref = om.segmentize(tg0.px, tg0.py, tg0.phi, tg0.A, tg0.B, tg0.C, ell, rtol=rtol, **kw)
    assert 0 < np.count_nonzero(ref["status"][picked] == 2) < len(picked)  # some fail, some pass: the threshold is hit from both sides
    for opts in (dict(split=0), dict(split=0, compact=0), dict(split=0, topo=0), dict()):
        dm = _capi.DeviceMesh(tg0.mesh, 0)
        for k, v in opts.items():
            dm.set_option(k, v)
        dt = _capi.DeviceTracks(dm, tg0.px, tg0.py, tg0.phi, tg0.cos_phi, tg0.sin_phi, tg0.A, tg0.B, tg0.C, ell, tg0.azim_idx)
        aq = tg0.azimuthal_quadrature
        assert dt.segmentize(tg0.tiny_step, 5, rtol, aq.delta_s, aq.n_azim_2) == ref["total"]
        off, st = dt.fetch_offsets()
        assert dt.stats()["tracks_near_rtol"] >= len(picked) // 2, dt.stats()
        if opts:
            assert np.array_equal(st, ref["status"]), opts
            n_fail, first, first_status = dt.failed()
            assert n_fail == np.count_nonzero(ref["status"]) and first == np.flatnonzero(ref["status"])[0] + 1 and first_status == 2
        else:
            # (a batch this small marches in pieces by default: Σℓ is then the sum of the pieces' sums — the statistic flags the
            #  tracks whose status that can change, and only those differ)
            differ = np.flatnonzero(st != ref["status"])
            assert set(differ.tolist()) <= set(picked), differ
        dt.close(); dm.close()


def test_shifted_anisotropic_domain(rt, orc, pincell):
    xy = pincell.node_coordinates * np.array([1.7, 0.6]) + np.array([3.25, -2.5])
    model = rt.DiscreteModel(xy, pincell.cell_node_ids)
    tg = rt.TrackGenerator(model, 16, 0.02)
    rt.trace(tg)
    rt.segmentize(tg, check=False)
    ref = _oracle(orc, tg)
    _same(tg, ref)
    assert abs(tg.volumes.sum() - tg.mesh.width() * tg.mesh.height()) < 1e-9


@pytest.mark.parametrize("n_azim,delta,flip", [(4, 0.25, False), (4, 0.5, True), (8, 0.13, False), (16, 0.07, True),
                                               (4, 0.8, False), (4, 0.8, True), (8, 0.8, False)])
def test_structured_grid_vertex_and_edge_grazing(rt, orc, grid_model, n_azim, delta, flip):
    """Right-triangle grid: 45-degree tracks run along diagonals and through vertices, so the
    n_int ∈ {0,1,3}, parallel-edge and vertex-skip branches of the reference all fire.  Whatever
    the reference's procedure does (including failing a track), the device must do the same."""
    model = grid_model(8, 8, hx=0.5, hy=0.5, flip=flip)
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    rt.segmentize(tg, check=False)
    ref = _oracle(orc, tg)
    _same(tg, ref, check_volumes=False)
    ok = ref["status"] == 0
    print(f"grid nφ={n_azim} δ={delta} flip={flip}: {ok.sum()}/{len(ok)} tracks ok, {ref['total']} segments")


def test_single_track_and_empty_batch(rt, orc, traced):
    from raytracing_jl_amd import _capi

    tg = traced(8, 0.02)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    aq = tg.azimuthal_quadrature
    one = _capi.DeviceTracks(dm, tg.px[:1], tg.py[:1], tg.phi[:1], tg.cos_phi[:1], tg.sin_phi[:1], tg.A[:1], tg.B[:1],
                             tg.C[:1], tg.ell[:1], tg.azim_idx[:1])
    n = one.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    om = orc.OracleMesh.from_mesh(tg.mesh)
    ref = om.segmentize(tg.px[:1], tg.py[:1], tg.phi[:1], tg.A[:1], tg.B[:1], tg.C[:1], tg.ell[:1])
    assert n == ref["total"] and np.array_equal(one.fetch_segments()["element"], ref["element"])
    empty = _capi.DeviceTracks(dm, *([np.zeros(0)] * 9), np.zeros(0, np.int32))
    assert empty.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == 0
    off, st = empty.fetch_offsets()
    assert off.tolist() == [0] and len(st) == 0
    assert np.all(empty.fetch_volumes() == 0)


@pytest.mark.parametrize("opts", [dict(walk=0),
                                  dict(sort_mode=0), dict(sort_mode=1), dict(fuse_volumes=0), dict(split=48), dict(split=8), dict(split=20, walk=0),
                                  dict(split=24, test_volumes_fallback=1), dict(split=24, fuse_volumes=0),
                                  dict(split=0), dict(split=0, topo=0), dict(split=0, pool_chunks_hint=8), dict(split=0, sort_mode=0),
                                  dict(split=0, test_out_records=20000),
                                  # the two-phase march (whole tracks, cheap steps: codes + k_materialise): its side list overflowing, every
                                  # track's Σℓ check by k_finish's left-to-right sum (also with truncated records: the deferred second pass),
                                  # every cheap record's fill_volumes term / none of them added by k_materialise, codes on a mesh-order march
                                  dict(split=0, side_entries_hint=16), dict(split=0, test_exact_sums=1),
                                  dict(split=0, test_exact_sums=1, test_out_records=20000), dict(split=0, test_tally_tau=-1),
                                  dict(split=0, test_tally_tau=1), dict(split=0, test_tally_tau=20), dict(split=0, compact=0),
                                  dict(split=0, compact=0, test_exact_sums=1), dict(split=0, topo=2, sort_mode=1),
                                  # the two-launch scan under the two-phase march (default: the march leaves the tile sums, one scan kernel);
                                  # the fused scan with a pool that overflows first (the tile sums of the void attempt are cleared)
                                  dict(split=0, fused_scan=0), dict(split=0, fused_scan=1, pool_chunks_hint=8), dict(split=0, sort_mode=0, fused_scan=1)])
def test_internal_modes_give_identical_results(rt, traced, oracle_run, opts):
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    off, st = dt.fetch_offsets()
    s = dt.fetch_segments()
    assert np.array_equal(off, ref["offsets"]) and st.max() == 0
    assert np.array_equal(s["element"], ref["element"])
    for k in FIELDS:
        assert np.array_equal(s[k], ref[k]), k
    assert np.allclose(dt.fetch_volumes(), ref["volumes"], rtol=1e-10, atol=0)


def test_two_phase_march_regime_and_memory(rt, traced, oracle_run):
    """Whole-track batches on a mesh with cheap-step records march in two phases: k_march stages one 4-B word per record (and the
    generic step's records in a side list), k_materialise computes the 44-B records.  The handle then holds the word pool, not
    (q, ±cell) rows with sparse entry points: 4 instead of 36 B per staging slot."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    held = {}
    for topo in (1, 0):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("split", 0)
        dm.set_option("topo", topo)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        aq = tg.azimuthal_quadrature
        assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
        st = dt.stats()
        assert (st["cheap_records"] > 0.9 * ref["total"]) == (topo == 1)
        # fill_volumes of cheap records: the chord from the vertices' distances where its error bound allows, the record's own
        # length (k_materialise) for the short chords and shallow crossings — some, not most
        if topo == 1:
            assert 0 < st["records_tallied_from_lengths"] < 0.25 * st["cheap_records"], st
        else:
            assert st["records_tallied_from_lengths"] == 0
        held[topo] = st["device_bytes"]
        assert np.array_equal(dt.fetch_segments()["ell"], ref["ell"])
        dt.close(); dm.close()
    slots = st["chunks_allocated"] * 32 * 64
    assert held[0] - held[1] > 24 * slots, (held, slots)  # 36 B -> 4 B per slot, minus the side list (40 B per track + ...)
    print(f"device bytes held: two-phase {held[1]}, exact-step staging {held[0]} ({slots} staging slots)")


def test_staging_pool_overflow_is_recovered(rt, traced, oracle_run):
    """Force the first pool to be far too small: the call must grow it and re-run."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("pool_chunks_hint", 8)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    assert np.array_equal(dt.fetch_segments()["element"], ref["element"])


@pytest.mark.parametrize("pct", [0, 35, 100, 300])
def test_reserved_and_cursor_chunks_mix(rt, traced, oracle_run, pct):
    """The staging pool reserves chunks per chunk index from the waves' expected record counts and hands out the rest from its
    cursor.  With the reservation scaled down (0: only every wave's first chunk; 35 %: most waves need chunks from
    the cursor behind their reserved ones) or up, the records are the same."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    aq = tg.azimuthal_quadrature
    for topo in (1, 0):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("split", 0); dm.set_option("topo", topo); dm.set_option("test_reserved_pct", pct)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        for _ in range(2):  # (the second call sizes the pool from the first one's need)
            assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
            seg = dt.fetch_segments()
            for k in ("px", "py", "qx", "qy", "ell", "element"):
                assert np.array_equal(seg[k], ref[k]), (pct, topo, k)
            np.testing.assert_allclose(dt.fetch_volumes(), ref["volumes"], rtol=1e-10)
        dt.close(); dm.close()


def test_one_handle_through_changing_regimes(rt, traced, oracle_run):
    """Calls on ONE track-set handle while the mesh's options change between them: cheap steps on and off (the two-phase march leaves
    tile sums for its one-kernel scan in one of two buffers, which a call without cheap steps neither uses nor clears), records
    written or left staged, the two-launch scan forced.  Every call returns the reference's offsets, records and volumes."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    aq = tg.azimuthal_quadrature
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("split", 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    seq = [dict(topo=1), dict(topo=0), dict(topo=1), dict(topo=1), dict(topo=0), dict(topo=0), dict(topo=1, compact=0), dict(topo=1, compact=1),
           dict(fused_scan=0), dict(fused_scan=1), dict(topo=0), dict(topo=2), dict(topo=1)]
    for i, opts in enumerate(seq):
        for k, v in opts.items():
            dm.set_option(k, v)
        assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"], (i, opts)
        off, st = dt.fetch_offsets()
        assert np.array_equal(off, ref["offsets"]) and np.array_equal(st, ref["status"]), (i, opts)
        seg = dt.fetch_segments()
        for k in ("px", "py", "qx", "qy", "ell", "element"):
            assert np.array_equal(seg[k], ref[k]), (i, opts, k)
        np.testing.assert_allclose(dt.fetch_volumes(), ref["volumes"], rtol=1e-10)
        s = dt.stats()
        state = {"topo": 1}
        for d in seq[:i + 1]:
            state.update(d)
        assert s["generic_records"] > 0 and (s["cheap_records"] > 0) == (state["topo"] > 0), (i, s)
    dt.close(); dm.close()


def test_enqueue_hook_runs_once_per_call_beside_the_kernels(rt, traced, oracle_run):
    """rt_mesh_set_enqueue_hook: called after the kernels are enqueued and before the wait; not on removal."""
    from raytracing_jl_amd import _capi

    tg = traced(8, 2e-2)
    ref = oracle_run(tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    calls = []
    dm.set_enqueue_hook(lambda: calls.append(len(calls)))
    for _ in range(3):
        assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    assert calls == [0, 1, 2]
    dm.set_enqueue_hook(None)
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    assert calls == [0, 1, 2]
    assert np.array_equal(dt.fetch_segments()["element"], ref["element"])


def test_handles_release_their_device_memory(rt, traced):
    """Create / segmentize / destroy in a loop: free device memory must come back (no leak in the handles)."""
    import ctypes as C
    import importlib.util
    import os

    from raytracing_jl_amd import _capi

    # hipMemGetInfo of the HIP runtime the library runs on (torch's bundled copy when torch is installed, see
    # _capi._share_hip_runtime_with_torch) — without importing torch, which can take minutes on a cold box
    _capi.lib()
    spec = importlib.util.find_spec("torch")
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
    hip = C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    tg = traced(32, 5e-3)
    aq = tg.azimuthal_quadrature

    def cycle():
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dt.close()
        dm.close()

    cycle()
    free0 = free_bytes()
    for _ in range(20):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)  # allocator granularity, not 20 pools (≈ 20 x 100 MB)


def test_bad_arguments_are_refused_not_launched(rt, traced):
    from raytracing_jl_amd import _capi

    tg = traced(8, 2e-2)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    with pytest.raises(_capi.RtError):
        dm.set_option("no_such_option", 1)
    L = _capi.lib()
    assert L.rt_segmentize(None, 1e-8, 5, 1e-8, None, 4) < 0 and b"bad arguments" in L.rt_last_error()
    # k < 0: NearestNeighbors' knn throws; azim_idx must index delta_s (fill_volumes reads it on the device)
    aq = tg.azimuthal_quadrature
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    with pytest.raises(_capi.RtError, match="k = -1"):
        dt.segmentize(tg.tiny_step, -1, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    with pytest.raises(_capi.RtError, match="azim_idx reaches"):
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s[:1], 1)
    with pytest.raises(_capi.RtError, match="1-based"):
        _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx - 1)
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) > 0  # the handle is still good
    cells = np.asarray(tg.mesh.cell_nodes).copy()
    cells[0, 0] = tg.mesh.num_nodes + 5  # out-of-range node id
    bad = rt.Mesh.__new__(rt.Mesh)
    bad.__dict__.update(tg.mesh.__dict__)
    bad.cell_nodes = cells
    with pytest.raises(_capi.RtError, match="out of range"):
        _capi.DeviceMesh(bad, 0)
    # the march's boundary test assumes a finite box (rt_device.hpp, inboundary)
    unbounded = rt.Mesh.__new__(rt.Mesh)
    unbounded.__dict__.update(tg.mesh.__dict__)
    unbounded.bb_max = (float("inf"), tg.mesh.bb_max[1])
    with pytest.raises(_capi.RtError, match="bounding box"):
        _capi.DeviceMesh(unbounded, 0)


def test_pinned_fetch_equals_plain_fetch(rt, traced):
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    aq = tg.azimuthal_quadrature
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    a = dt.fetch_segments()
    b = dt.fetch_segments_pinned()
    for k in a:
        assert b[k].dtype == a[k].dtype and np.array_equal(a[k], b[k]), k
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    c = dt.fetch_segments_pinned()  # buffers are reused
    for k in a:
        assert np.array_equal(a[k], c[k]), k


def test_max_iter_counts_whole_tracks_even_when_marched_in_pieces(rt, orc):
    """MAX_ITER = 10000 segments per track (src/track.jl:104,119): the reference stops there and its Σℓ check then
    fails.  A band of 48000 needle cells crossed lengthwise by the shallowest tracks gives such tracks, while the
    mesh-wide estimate (few cells elsewhere) says a track has a few hundred segments — so the batch is marched in
    pieces, each far below the limit.  The device must still report what the reference reports."""
    from scipy.spatial import Delaunay

    n_band, h = 24000, 0.016
    xs = np.linspace(0.0, 1.0, n_band + 1)
    pts = [(x, 0.500) for x in xs[1:-1]] + [(x, 0.500 + h) for x in xs[1:-1]]
    g = np.linspace(0.0, 1.0, 11)
    pts += [(x, y) for x in g for y in g if not (0.45 < y < 0.56)] + [(0.0, 0.5), (1.0, 0.5), (0.0, 0.5 + h), (1.0, 0.5 + h)]
    xy = np.asarray(pts)
    cells = np.sort(Delaunay(xy).simplices.astype(np.int32) + 1, axis=1)
    a, b, c = xy[cells[:, 0] - 1], xy[cells[:, 1] - 1], xy[cells[:, 2] - 1]
    area2 = np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1]))
    model = rt.DiscreteModel(xy, cells[area2 > 1e-14])
    tg = rt.TrackGenerator(model, 256, 0.05)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    counts = np.diff(ref["offsets"])
    assert counts.max() == 10000 and np.any((ref["status"] == 2) & (counts == 10000)), (counts.max(), np.bincount(ref["status"]))
    rt.segmentize(tg, check=False)  # default options: this small batch is split into pieces
    assert tg.device_tracks.stats()["split"] == 0, "the call must have fallen back to whole tracks"
    _same(tg, ref, check_volumes=False)


@pytest.mark.parametrize("opts", [dict(), dict(fuse_volumes=0), dict(split=24)])
def test_short_output_estimate_is_recovered(rt, traced, oracle_run, opts):
    """The six output arrays are sized from an estimate of the record count; a call that produces more must grow them
    and compact again without marching again (forced here through the test knob)."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("test_out_records", 1000)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    for _ in range(2):
        assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
        s = dt.fetch_segments()
        assert np.array_equal(dt.fetch_offsets()[0], ref["offsets"]) and np.array_equal(s["element"], ref["element"])
        for k in FIELDS:
            assert np.array_equal(s[k], ref[k]), k
        assert np.allclose(dt.fetch_volumes(), ref["volumes"], rtol=1e-10, atol=0)


def test_iteration_guard_counts_whole_tracks_when_marched_in_pieces(rt, orc):
    """RT_TRACK_ITER_CAP (the library's guard on the reference's unbounded `continue` paths; the checker has the same
    guard) counts the iterations of a whole track.  Found by tools/fuzz_many.py (seed 7444): a lattice 1000 units from the
    origin, where a track creeps for more than 4 M tiny steps — marched in pieces, no piece reached the limit and the
    track ended as a length mismatch instead.  Such a track now sends the call to the whole-track march."""
    import sys, os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cpu

    kind, model, n_azim, delta, k = fuzz_cpu.case(7444)
    tg = rt.TrackGenerator(model, 256, delta)
    rt.trace(tg)
    ref = _oracle(orc, tg, k=k)
    assert np.count_nonzero(ref["status"] == 4) >= 1
    rt.segmentize(tg, k=k, check=False)  # default options: a small batch, marched in pieces first
    _same(tg, ref, check_volumes=False)
    assert tg.device_tracks.stats()["split"] == 0


@pytest.mark.parametrize("iter_cap", [40, 300])
def test_cheap_steps_keep_the_iteration_guard_exact(rt, orc, traced, iter_cap):
    """Cheap steps bound the reference's tiny steps between two records instead of replaying them, so the iteration counter
    is an upper bound after them; a track whose bound reaches the cap is marched again with exact steps only.  cap = 300:
    no track of this batch really reaches it (the bound does); cap = 40: most do.  Status, counts and records equal the
    checker's at the same cap."""
    from raytracing_jl_amd import _capi

    tg = traced(8, 2e-2)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, iter_cap=iter_cap, n_threads=0, k=5)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("split", 0)
    dm.set_option("iter_cap", iter_cap)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"]
    off, st = dt.fetch_offsets()
    s = dt.fetch_segments()
    assert np.array_equal(st, ref["status"]) and np.array_equal(off, ref["offsets"]) and np.array_equal(s["element"], ref["element"])
    for k in FIELDS:
        assert np.array_equal(s[k], ref[k]), k
    assert (np.count_nonzero(ref["status"] == 4) > 0) == (iter_cap == 40)
    assert dm.info()["records_cheap"] > 0
    # the restarted tracks' cheap records had already been added to the fused volumes: the call recomputes them from the
    # records (the oracle's fill_volumes over ITS records is the reference here, restart or not)
    stats = dt.stats()
    assert stats["tracks_restarted"] > 0
    vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    assert np.allclose(dt.fetch_volumes(), vol, rtol=1e-10, atol=0)
    print(f"iter_cap={iter_cap}: {stats}, failing {int(np.count_nonzero(st))}")


