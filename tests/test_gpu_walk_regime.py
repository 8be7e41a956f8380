"""The certified walk step on the real kernels, regime visible (GPU only).

56 fixed seeds of meshes on which the walk step is enabled — jittered lattices (what a mesh generator produces),
lattices far from the origin, needle bands (gap 1e-2 … 1e-7 of the spacing), random Delaunay clouds and clusters.
For every mesh the C-ABI path runs with the walk step on (exact steps only, and with the cheap steps of the default) and
off; all must equal the checker bit for bit, and the test asserts through rt_mesh_info / rt_last_stats that the runs really
took walk steps / cheap steps / none."""
import numpy as np
import pytest

import meshgen

pytestmark = pytest.mark.gpu

FIELDS = ("px", "py", "qx", "qy", "ell")
CASES = [("lattice", s) for s in range(20)] + [("lattice_far", s) for s in range(6)] + [("sliver", s) for s in range(10)] + \
        [("sliver_fine", s) for s in range(10)] + [("random", s) for s in range(5)] + [("cluster", s) for s in range(5)] + \
        [("near_vertex", s) for s in range(6)] + [("aligned", s) for s in range(6)] + [("lattice_mid", s) for s in range(4)] + \
        [("steep", s) for s in range(4)]
# the last four classes aim at the thresholds of the certificates (tests/meshgen.py): nodes on / next to track lines, lattice rows
# within 1e-7 … 3e-3 rad of a track direction, lattices 10 – 60 units from the origin, hand-made tracks within 1e-5 … 1e-8 of
# ϕ = 0, π/2, π


def _make(rt, kind, seed):
    if kind == "lattice":
        n = 8 + 2 * seed
        return meshgen.lattice_model(rt, 1000 + seed, n, n, jitter=(0.1, 0.25, 0.4, 0.3)[seed % 4],
                                     w=(1.0, 2.5, 0.3)[seed % 3], h=(1.0, 0.4)[seed % 2]), min((1.0, 2.5, 0.3)[seed % 3], (1.0, 0.4)[seed % 2])
    if kind == "lattice_far":
        return meshgen.lattice_model(rt, 1100 + seed, 14, 14, jitter=0.3, x0=(100.0, -1000.0, 11.0)[seed % 3], y0=(50.0, 2000.0)[seed % 2]), 1.0
    if kind == "sliver":
        return meshgen.sliver_model(rt, 1200 + seed, 9 + 2 * seed, 9 + 2 * seed, gap=(1e-2, 1e-3, 1e-4)[seed % 3]), 1.0
    if kind == "sliver_fine":
        return meshgen.sliver_model(rt, 1300 + seed, 8 + 2 * seed, 8 + 2 * seed, gap=(1e-5, 1e-6, 1e-7)[seed % 3], x0=-3.25, y0=2.5), 1.0
    if kind == "random":
        return meshgen.random_model(rt, 1400 + seed, 150 + 300 * seed), 1.0
    if kind == "near_vertex":
        return meshgen.near_vertex_model(rt, 1600 + seed, 200 + 150 * seed, (8, 16, 32, 4, 64)[seed % 5], 0.008), 1.0
    if kind == "aligned":
        return meshgen.aligned_model(rt, 1700 + seed, 8 + 3 * seed, (8, 16, 32, 4, 64)[seed % 5], 0.008), 1.0
    if kind == "lattice_mid":
        return meshgen.lattice_model(rt, 1800 + seed, 12 + 4 * seed, 12 + 4 * seed, jitter=0.3, x0=(10.0, 30.0, -60.0, 30.0)[seed], y0=(5.0, -40.0, 25.0, 30.0)[seed]), 1.0
    if kind == "steep":
        return (meshgen.lattice_model(rt, 1900 + seed, 10 + 4 * seed, 10 + 4 * seed, jitter=(0.0, 0.2)[seed % 2]) if seed < 2 else
                meshgen.random_model(rt, 1900 + seed, 200 * seed)), 1.0
    return meshgen.random_model(rt, 1500 + seed, 300 + 250 * seed, cluster=True), 1.0


def _run(rt, tg, walk, k=5, split=0, cheap=0):
    """split=0: whole tracks per lane, so that rt_last_stats' counts are exact (pieces' seeds count as neither
    kind); split=-1: the library's default, which marches small batches like these in pieces.  cheap=1: the walk step's
    decision from the vertices' signed distances (option "topo", the default for whole-track batches), 0: exact walk steps."""
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("walk", 1 if walk else 0)
    dm.set_option("split", split)
    dm.set_option("topo", cheap)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    total = dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets()
    r = dict(total=total, offsets=off, status=st, info=dm.info(), stats=dt.stats(), volumes=dt.fetch_volumes(), **dt.fetch_segments())
    dt.close()
    dm.close()
    return r


def _same(r, ref, what):
    assert r["total"] == ref["total"], what
    assert np.array_equal(r["status"], ref["status"]), ("per-track status", what)
    assert np.array_equal(r["offsets"], ref["offsets"]), ("segment counts", what)
    assert np.array_equal(r["element"], ref["element"]), ("element ids", what)
    for f in FIELDS:
        assert np.array_equal(r[f], ref[f]), (f, what)
    assert np.allclose(r["volumes"], ref["volumes"], rtol=1e-10, atol=1e-300), what


@pytest.mark.parametrize("kind,seed", CASES)
def test_walk_on_off_checker(rt, orc, kind, seed):
    model, scale = _make(rt, kind, seed)
    n_azim = (8, 16, 32, 4, 64)[seed % 5]
    tg = rt.TrackGenerator(model, n_azim, 0.008 * scale)
    rt.trace(tg)
    if kind == "steep":
        meshgen.steep_tracks(rt, tg, seed)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, iter_cap=4000000, n_threads=0)
    aq = tg.azimuthal_quadrature
    ref["volumes"] = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    on, off, cheap = _run(rt, tg, True), _run(rt, tg, False), _run(rt, tg, True, cheap=1)
    _same(on, ref, "walk on")
    _same(off, ref, "walk off")
    _same(cheap, ref, "walk on, cheap steps")
    _same(_run(rt, tg, True, split=-1, cheap=1), ref, "walk on, default splitting")
    # forced cheap steps (option "topo" = 2): no 90 % gate, no hand-back of often-refused waves — every record that carries
    # a cheap certificate is decided by it, on every mesh class
    forced = _run(rt, tg, True, cheap=2)
    _same(forced, ref, "walk on, cheap steps forced")
    info = on["info"]
    if info["records_cheap"] > 0 and info["walk_enabled"] and tg.tiny_step <= info["cheap_tiny_max"]:
        assert forced["stats"]["cheap_records"] >= cheap["stats"]["cheap_records"], (forced["stats"], cheap["stats"])
        assert forced["stats"]["cheap_records"] > 0 or ref["total"] < 200, (info, forced["stats"])
    assert all(v >= 0 for v in forced["stats"]["cheap_refusals"].values())
    assert off["info"]["walk_enabled"] == 0 and off["stats"]["walk_records"] == 0
    assert on["stats"]["cheap_records"] == 0 and off["stats"]["cheap_records"] == 0
    assert cheap["stats"]["cheap_records"] <= cheap["stats"]["walk_records"] == on["stats"]["walk_records"]
    # (cheap steps are used when at least 90 % of the walkable records carry their certificates; a wave that is refused
    #  in more than one iteration out of eight goes on with exact steps)
    if 10 * info["records_cheap"] >= 9 * info["records_walk"] > 0 and tg.tiny_step <= info["cheap_tiny_max"]:
        assert cheap["stats"]["cheap_records"] > 0, (info, cheap["stats"])
    else:
        assert cheap["stats"]["cheap_records"] == 0
    frac = on["stats"]["walk_records"] / max(ref["total"], 1)
    if info["cells_fragile"] == model.num_cells:
        # thousands of units from the origin the reference's own barycentric test is rounding noise at the √eps level
        # (its closed form is not translation invariant): no cell can be certified, every step is the literal one
        assert kind == "lattice_far" and info["walk_available"] == 0 and on["stats"]["walk_records"] == 0, info
    else:
        assert info["walk_available"] == 1 and info["walk_enabled"] == 1, info
        assert on["stats"]["walk_records"] > 0
    if kind == "lattice":
        assert info["records_walk"] >= 0.85 * info["records"] and frac > 0.7, (info, on["stats"])
    print(f"{kind} seed {seed}: {model.num_cells} cells nφ={n_azim} {ref['total']} segments | regime: walk on, "
          f"{info['records_walk']}/{info['records']} records walkable, eps ≤ {info['eps_max']:.1e}, fragile {info['cells_fragile']}, "
          f"{frac:.1%} of the records by the walk step, {cheap['stats']['cheap_records'] / max(ref['total'], 1):.1%} by cheap steps "
          f"({info['records_cheap']} records; forced: {forced['stats']['cheap_records'] / max(ref['total'], 1):.1%}, refusals "
          f"{ {k: v for k, v in forced['stats']['cheap_refusals'].items() if v} }) | "
          f"{int(np.count_nonzero(ref['status']))} tracks on which the reference throws")
