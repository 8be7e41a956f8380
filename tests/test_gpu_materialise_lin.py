"""k_materialise_lin (round 5: a two-phase call's records in output order, raytracing.jl_amd/csrc/rt_materialise.hip) against the
kernel it replaces (option "mat_kernel" = 1: k_materialise, rt_records.hip) and against the oracle — GPU only.

What the kernel does differently from the chunk-shaped one and what is therefore aimed at here:
  * linear slots in RUN GROUPS: tracks whose records are not neighbours in memory (march orders 0 and 1, the unit that straddles the
    packed partial wave of uids when n % 64 != 0) start groups of their own, pairs at a group's ends are written as half pairs;
  * ROUNDS of 256 rows: tracks with more records take several rounds, the first row of a round starts where the last one ended;
  * records that keep their own end points in the MIDDLE of a track (generic steps behind refused cheap steps: forced cheap steps
    on meshes where a quarter of the certificates are missing) and at its END;
  * Σℓ from a track's first and last points (src/track.jl:171-175 is decided by margin; k_finish sums what is marginal): the
    per-track status must equal the oracle's whatever rtol is.
Records, offsets and status bit for bit; volumes to 1e-10 (north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("px", "py", "qx", "qy", "ell", "element")


def _oracle(orc, tg, rtol=None):
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    kw = {} if rtol is None else {"rtol": rtol}
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                      tiny_step=tg.tiny_step, iter_cap=4000000, n_threads=0, **kw)
    aq = tg.azimuthal_quadrature
    r["volumes"] = om.fill_volumes(r["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    r["total"] = int(r["offsets"][-1])
    return r


def _run(rt, tg, opts, rtol=None):
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT if rtol is None else rtol, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets()
    seg = dt.fetch_segments()
    vol = dt.fetch_volumes()
    stats = dt.stats()
    dt.close(); dm.close()
    return dict(total=total, offsets=off, status=st, volumes=vol, stats=stats, **{k: seg[k] for k in FIELDS})


def _equal(a, b, what, volumes_rtol=1e-10):
    assert a["total"] == b["total"], what
    assert np.array_equal(a["offsets"], b["offsets"]), what
    assert np.array_equal(a["status"], b["status"]), (what, np.nonzero(a["status"] != b["status"])[0][:10])
    for k in FIELDS:
        assert np.array_equal(a[k], b[k]), (what, k, np.nonzero(a[k] != b[k])[0][:10])
    np.testing.assert_allclose(a["volumes"], b["volumes"], rtol=volumes_rtol, atol=0, err_msg=str(what))


@pytest.mark.parametrize("sort_mode", [0, 1, 2])
@pytest.mark.parametrize("topo", [1, 2])
def test_march_orders_and_partial_waves(rt, orc, traced, sort_mode, topo):
    """pincell, nφ=32, δ=5e-3: 6,548 tracks (6548 % 64 = 20: the partial wave of uids is packed into the middle of the march order, the
    units behind it straddle two waves of uids); march orders 0 (uid), 1 (every track on its own: 16 run groups per unit), 2."""
    tg = traced(32, 5e-3)
    ref = _oracle(orc, tg)
    base = dict(split=0, sort_mode=sort_mode, topo=topo)
    new = _run(rt, tg, dict(base, mat_kernel=0))
    old = _run(rt, tg, dict(base, mat_kernel=1))
    assert new["stats"]["cheap_records"] > 0  # the two-phase march ran (k_materialise_lin writes its records)
    _equal(new, old, ("new vs old", sort_mode, topo))
    _equal(new, ref, ("new vs oracle", sort_mode, topo))


@pytest.mark.parametrize("cls,seed", [("random", 11), ("random", 12), ("cluster", 13), ("sliver", 14), ("lattice", 15), ("near_vertex", 16)])
def test_generic_records_inside_and_at_the_end_of_tracks(rt, orc, cls, seed):
    """Fuzz-class meshes with cheap steps FORCED ("topo" = 2): 5-25 % of the records come from the generic step behind a refused
    cheap step — records with their own p in the middle of a track (the gaps of the Σℓ chain) and at its end (the chain's end from
    the side list)."""
    import meshgen

    if cls == "random":
        model = meshgen.random_model(rt, seed, 900)
    elif cls == "cluster":
        model = meshgen.random_model(rt, seed, 900, cluster=True)
    elif cls == "sliver":
        model = meshgen.sliver_model(rt, seed, 24, 24)
    elif cls == "near_vertex" and hasattr(meshgen, "near_vertex_model"):
        model = meshgen.near_vertex_model(rt, seed, 600, 16, 0.01)
    else:
        model = meshgen.lattice_model(rt, seed, 30, 30)
    tg = rt.TrackGenerator(model, 16, 0.01)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    new = _run(rt, tg, dict(split=0, topo=2, mat_kernel=0))
    old = _run(rt, tg, dict(split=0, topo=2, mat_kernel=1))
    st = new["stats"]
    print(f"{cls} {seed}: {st['records']} records, {st['cheap_records']} by cheap steps, {st['generic_records']} by the generic step "
          f"({st['generic_records'] - tg.n_total_tracks} not a track's first)")
    _equal(new, old, ("new vs old", cls, seed))
    _equal(new, ref, ("new vs oracle", cls, seed))


def test_tracks_longer_than_a_round(rt, orc):
    """A fine random mesh (60,000 interior points): tracks of 300-500 records — two rounds of 256 rows per unit, the first row of the
    second round starts at the exit point the first round's last row left in LDS; the first chunks come with the header, the later
    ones from the table of chunks."""
    import meshgen

    model = meshgen.random_model(rt, 21, 60000, nb=48)
    tg = rt.TrackGenerator(model, 8, 0.02)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    counts = np.diff(ref["offsets"])
    assert counts.max() > 300, counts.max()
    for topo in (1, 2):
        new = _run(rt, tg, dict(split=0, topo=topo, mat_kernel=0))
        if new["stats"]["cheap_records"] == 0:
            continue  # (the mesh's certificates do not carry cheap steps as gated: the exact-step path ran, not this kernel)
        old = _run(rt, tg, dict(split=0, topo=topo, mat_kernel=1))
        _equal(new, old, ("new vs old", topo))
        _equal(new, ref, ("new vs oracle", topo))
    print(f"longest track {counts.max()} records, {int((counts > 256).sum())} of {len(counts)} tracks beyond one round")


@pytest.mark.parametrize("seed", [520017, 520021, 520332, 520050, 520139, 520159, 520479, 520979])
def test_fuzz_cases_that_found_the_chain_errors(rt, orc, seed):
    """Cases of tools/fuzz_many.py (round 5, seeds 520000 ...) on which the first version of the Σℓ chain failed.  (i) Tracks of
    258-344 records whose second round begins with a generic step's record: the gap in front of it is measured from the LAST ROW OF
    THE ROUND BEFORE (kept in LDS), not from the slot before it.  (ii) `steep` tracks (hand-made, within 1e-5 ... 1e-8 of ϕ = π/2):
    records that begin BEHIND the exit point before them — the gap is signed along the march, an overlap adds to Σℓ — and (seed
    520979) a generic step's record whose own two points are in the wrong order (order_intersection_points compares x coordinates,
    equal to the last bit near π/2): it walks backwards, the chain loses its length twice.  Forced and gated cheap
    steps; status, records and volumes against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cpu
    import meshgen

    kind, model, n_azim, delta, k = fuzz_cpu.case(seed)
    tg = rt.TrackGenerator(model, min(n_azim, 256), delta)
    rt.trace(tg)
    if kind == "steep":
        meshgen.steep_tracks(rt, tg, seed)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step,
                        k=k, iter_cap=4000000, n_threads=0)
    from raytracing_jl_amd import _capi
    for topo in (1, 2):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        for kk, v in dict(walk=1, split=0, topo=topo, mat_kernel=0).items():
            dm.set_option(kk, v)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        aq = tg.azimuthal_quadrature
        assert dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == int(ref["offsets"][-1])
        off, st = dt.fetch_offsets()
        seg = dt.fetch_segments()
        assert np.array_equal(off, ref["offsets"])
        assert np.array_equal(st, ref["status"]), (seed, kind, topo, np.nonzero(st != ref["status"])[0][:10])
        for f in FIELDS:
            assert np.array_equal(seg[f], ref[f]), (seed, f)
        dt.close(); dm.close()


@pytest.mark.parametrize("rtol", [1e-13, 3e-12, 1e-9])
def test_length_check_by_first_and_last_point(rt, orc, traced, rtol):
    """`isapprox(track.ℓ, Σℓ; rtol)` (src/track.jl:171-175) with tolerances at which many, some and no tracks fail: the kernel forms Σℓ
    from the chain first point -> last point (minus the gaps in front of records with their own p) and hands every track within
    the margin of a differently ordered sum to k_finish, which adds up left to right as the reference does — the status must be
    the oracle's, track for track."""
    tg = traced(32, 5e-3)
    ref = _oracle(orc, tg, rtol=rtol)
    new = _run(rt, tg, dict(split=0, mat_kernel=0), rtol=rtol)
    old = _run(rt, tg, dict(split=0, mat_kernel=1), rtol=rtol)
    n_fail = int((ref["status"] != 0).sum())
    print(f"rtol {rtol}: {n_fail} of {len(ref['status'])} tracks fail the check")
    assert np.array_equal(new["status"], ref["status"]) and np.array_equal(old["status"], ref["status"])
    for k in FIELDS:
        assert np.array_equal(new[k], ref[k]), k


@pytest.mark.parametrize("n_azim,delta,pieces", [(32, 5e-3, True), (64, 5e-3, False), (32, 1e-3, False)])
def test_default_plan_by_batch_size(rt, orc, traced, n_azim, delta, pieces):
    """Which march a call takes with DEFAULT options (round 5, profiles/r05/exp_split_threshold.log): below 160 march waves the tracks
    are marched in pieces (exact steps, k_compact3), from there on whole with the two-phase march (k_materialise_lin) — 103 waves (C2),
    204 waves and 511 waves (a shard of C3 on four GPUs) here; records, offsets and status against the oracle in either regime."""
    tg = traced(n_azim, delta)
    n_waves = (len(tg.px) + 63) // 64
    assert (n_waves < 160) == pieces, n_waves
    ref = _oracle(orc, tg)
    new = _run(rt, tg, {})
    st = new["stats"]
    assert (st["split"] == 1) == pieces, st
    assert (st["cheap_records"] > 0) == (not pieces), st
    assert new["total"] == ref["total"] and np.array_equal(new["offsets"], ref["offsets"]) and np.array_equal(new["status"], ref["status"])
    for k in FIELDS:
        assert np.array_equal(new[k], ref[k]), k
    np.testing.assert_allclose(new["volumes"], ref["volumes"], rtol=1e-10, atol=0)
