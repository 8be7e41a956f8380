"""Records in COMPLETION order (GPU only; mesh option "record_order", default off — DESIGN.md §4 "Round 6, second session").

The reference keeps a `Vector{Segment}` per track (`track.segments`, src/track.jl:18; filled by `_segmentize_track!`,
src/track.jl:106-178) and no order between tracks.  Under the option a march workgroup that has ended takes the span of its
tracks' records from an atomic cursor and queues itself on its XCD, and the record kernel runs beside the rest of the march:
every track's records stay contiguous and in march order, the TRACKS lie in the order in which the workgroups ended, and a
per-track table (first record, count) describes the layout.  Checked here: every track's records through the table are the
checker's, bit for bit; the spans tile [0, total); status, counts and volumes are the CSR call's; every entry point that
promises the CSR layout produces it on demand — bit for bit the checker's arrays; the next call starts clean."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("px", "py", "qx", "qy", "ell", "element")


def _bits(a):
    return a.view(np.int64) if a.dtype == np.float64 else a


def _check_table(dt, ref, vol, what):
    """The handle's records as they lie, through the table, against the checker's CSR arrays."""
    total = int(ref["total"])
    order = dt.record_order()
    beg, cnt, st = dt.fetch_table()
    recs = dt.fetch_records()
    off = np.asarray(ref["offsets"], np.int64)
    assert np.array_equal(cnt, np.diff(off)), what
    assert np.array_equal(st, ref["status"]), what
    live = cnt > 0
    if total:
        b, e = np.sort(beg[live]), np.sort((beg + cnt)[live])
        assert b[0] == 0 and e[-1] == total and np.array_equal(b[1:], e[:-1]), (what, "the tracks' spans tile [0, total)")
    idx = np.repeat(beg - off[:-1], cnt) + np.arange(total)  # record r of the CSR layout lies at idx[r]
    for k in KEYS:
        assert np.array_equal(_bits(recs[k][idx]), _bits(np.asarray(ref[k]))), (k, what)
    assert np.allclose(dt.fetch_volumes(), vol, rtol=1e-10, atol=1e-300), what
    return order


def _check_csr(dt, ref, what):
    off, st = dt.fetch_offsets()
    seg = dt.fetch_segments()
    assert dt.record_order() == 0, what  # rewritten once; CSR order from then on
    assert np.array_equal(off, ref["offsets"]) and np.array_equal(st, ref["status"]), what
    for k in KEYS:
        assert np.array_equal(_bits(seg[k]), _bits(np.asarray(ref[k]))), (k, what)


def _handles(rt, tg, opts):
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    return dm, dt


def _seg(rt, tg, dt):
    aq = tg.azimuthal_quadrature
    return dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)


def test_completion_order_on_the_pincell(rt, traced, oracle_run):
    """C2 marched whole ("split" 0: a batch this small is cut into pieces by default, and pieces keep CSR order), three calls in a
    row: table, CSR on demand, table again."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    for opts in (dict(split=0, record_order=2), dict(split=0, record_order=2, topo=2), dict(split=0, record_order=1)):
        dm, dt = _handles(rt, tg, opts)
        for call in range(3):
            assert _seg(rt, tg, dt) == ref["total"]
            s = dt.stats()
            assert s["completion_order"] == 1 and s["record_kernel"] == "rt::k_materialise_lin<true>", (opts, s)
            assert _check_table(dt, ref, ref["volumes"], (opts, call)) == 1
            if call == 1:
                _check_csr(dt, ref, (opts, call))
                assert _check_table(dt, ref, ref["volumes"], (opts, call, "table of a handle in CSR order")) == 0
        dt.close(); dm.close()


def test_completion_order_keeps_out_of_plans_it_cannot_serve(rt, traced, oracle_run):
    """Pieces (the default for C2), exact steps only ("topo" 0), rows instead of records ("compact" 0), events between the kernels
    ("timing"): CSR order, whatever the option says — and the table of such a handle is its CSR offsets."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    for opts in (dict(record_order=2), dict(split=0, record_order=2, topo=0), dict(split=0, record_order=2, timing=1),
                 dict(split=0, record_order=2, mat_kernel=1), dict(split=0, record_order=0)):
        dm, dt = _handles(rt, tg, opts)
        assert _seg(rt, tg, dt) == ref["total"]
        assert dt.stats()["completion_order"] == 0, opts
        assert _check_table(dt, ref, ref["volumes"], opts) == 0
        _check_csr(dt, ref, opts)
        dt.close(); dm.close()


def test_completion_order_on_fuzz_meshes(rt, orc):
    """Random meshes with cheap steps forced (many refusals: side-list records inside tracks, exact steps between cheap ones), a
    staging pool, a side list and result arrays that are too small on the first attempt (the re-run must start clean, in
    completion order again), every Σℓ check by k_finish's exact sum over the records where they lie."""
    from meshgen import random_model

    for seed, n_int, kw in ((21, 500, {}), (22, 900, dict(cluster=True)), (23, 2500, dict(nb=40))):
        model = random_model(rt, seed, n_int, **kw)
        tg = rt.TrackGenerator(model, 16, 0.004)
        rt.trace(tg)
        om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
        ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step,
                            iter_cap=4000000, n_threads=0)
        aq = tg.azimuthal_quadrature
        vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
        for opts in (dict(split=0, record_order=2, topo=2), dict(split=0, record_order=2, topo=2, pool_chunks_hint=8, side_entries_hint=4),
                     dict(split=0, record_order=2, topo=2, test_out_records=1000), dict(split=0, record_order=2, topo=2, test_exact_sums=1)):
            dm, dt = _handles(rt, tg, opts)
            for call in range(2):
                assert _seg(rt, tg, dt) == ref["total"], (seed, opts)
                assert dt.stats()["completion_order"] == 1, (seed, opts, dt.stats())
                assert _check_table(dt, ref, vol, (seed, opts, call)) == 1
            _check_csr(dt, ref, (seed, opts))
            dt.close(); dm.close()


def test_completion_order_consumers_see_the_csr_layout(rt, traced, oracle_run):
    """rt_device_pointers, rt_fill_tau and the sweep over the compact records promise CSR order: each of them, called first on a
    handle in completion order, rewrites the records once."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    nc = tg.mesh.num_cells
    sig = np.linspace(0.2, 1.6, nc * 2).reshape(nc, 2)
    for first in ("pointers", "tau", "sweep"):
        dm, dt = _handles(rt, tg, dict(split=0, record_order=2))
        assert _seg(rt, tg, dt) == ref["total"] and dt.record_order() == 1
        if first == "pointers":
            assert dt.device_pointers()["ell"] != 0
        elif first == "tau":
            tau, _, _ = dt.fill_tau(sig)
            assert np.array_equal(tau, np.asarray(ref["ell"])[:, None] * sig[np.asarray(ref["element"]) - 1])
        else:
            dt.sweep_set_links(tg)
            a = dt.sweep(2, sig, np.linspace(0.0, 1.0, nc * 2).reshape(nc, 2), None, np.ones((2, tg.n_total_tracks, 2)), input="compact")
            dm2, dt2 = _handles(rt, tg, dict(split=0, record_order=0))
            _seg(rt, tg, dt2)
            dt2.sweep_set_links(tg)
            b = dt2.sweep(2, sig, np.linspace(0.0, 1.0, nc * 2).reshape(nc, 2), None, np.ones((2, tg.n_total_tracks, 2)), input="compact")
            assert np.allclose(a["phi"], b["phi"], rtol=1e-12, atol=0) and np.allclose(a["psi_out"], b["psi_out"], rtol=1e-12, atol=0)
            dt2.close(); dm2.close()
        if first != "sweep":
            assert dt.record_order() == 0
        _check_csr(dt, ref, first)
        dt.close(); dm.close()


def test_completion_order_at_the_headline_configuration(rt, traced, oracle_run):
    """C3: 130,456 tracks, 9.3 M records, 510 march workgroups resident at once — the configuration the automatic rule ("record_order"
    1) takes."""
    tg = traced(128, 1e-3)
    ref = oracle_run(tg)
    dm, dt = _handles(rt, tg, dict(record_order=1))
    for call in range(2):
        assert _seg(rt, tg, dt) == ref["total"]
        assert dt.stats()["completion_order"] == 1
        assert _check_table(dt, ref, ref["volumes"], ("C3", call)) == 1
    _check_csr(dt, ref, "C3")
    dt.close(); dm.close()


def test_completion_order_gives_up_and_falls_back(rt, traced, oracle_run):
    """The exit every waiting record workgroup reaches: march workgroups that never queue themselves ("compact_debug" 64, a test switch)
    leave the record kernel beside the march waiting — it gives up on its own bound (≈0.3 s), the attempt is void, and the call is
    made again in CSR order: same results, and the handle stays with CSR order from then on."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    dm, dt = _handles(rt, tg, dict(split=0, record_order=2, compact_debug=64))
    for call in range(2):
        assert _seg(rt, tg, dt) == ref["total"]
        assert dt.stats()["completion_order"] == 0 and dt.record_order() == 0
        assert _check_table(dt, ref, ref["volumes"], ("gave up", call)) == 0
        _check_csr(dt, ref, ("gave up", call))
    dt.close(); dm.close()


def test_completion_order_overflow_while_the_record_kernel_runs(rt, traced, oracle_run):
    """A staging pool / side list that runs out in the MIDDLE of a large batch: the overflow flag appears while the record kernel
    beside the march is serving units (fuzz seed 830802, round 6: 256 threads of a record workgroup read the flag for themselves,
    disagreed, and met different barriers — a memory access fault).  One lane decides per unit now; the attempt is void, the call
    runs again with larger pools and ends in completion order with the checker's records."""
    tg = traced(128, 1e-3)
    ref = oracle_run(tg)
    for opts in (dict(record_order=2, topo=2, side_entries_hint=100), dict(record_order=2, pool_chunks_hint=3000),
                 dict(record_order=2, topo=2, side_entries_hint=200, pool_chunks_hint=4000)):
        for rep in range(2):
            dm, dt = _handles(rt, tg, opts)
            assert _seg(rt, tg, dt) == ref["total"]
            s = dt.stats()
            assert s["completion_order"] == 1 and s["attempts"] >= 2, (opts, s)
            assert _check_table(dt, ref, ref["volumes"], (opts, rep)) == 1
            dt.close(); dm.close()
