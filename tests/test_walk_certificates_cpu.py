"""The walk step's certificates without a GPU: the device march's per-lane logic and the mesh preprocessing,
compiled for the host (tests/host_march.hip — test infrastructure built from the product's own headers), against
the CPU checker.  Walk step on == walk step off == cheap steps == checker, bit for bit, on every mesh class, and the
preprocessing's per-record switches behave as documented.  The kernels themselves are tested on the GPU
(tests/test_gpu_*.py); tools/fuzz_cpu.py runs the same comparison over hundreds of seeds."""
import os

import numpy as np
import pytest

import hostmarch as hm
import meshgen

FIELDS = ("px", "py", "qx", "qy", "ell")


def _oracle(orc, tg, **kw):
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    return om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                         tiny_step=tg.tiny_step, iter_cap=4000000, n_threads=0, **kw)


def _same(r, ref, what):
    assert r["total"] == ref["total"], what
    assert np.array_equal(r["status"], ref["status"]), ("per-track status", what)
    assert np.array_equal(r["offsets"], ref["offsets"]), ("segment counts", what)
    assert np.array_equal(r["element"], ref["element"]), ("element ids", what)
    for k in FIELDS:
        assert np.array_equal(r[k], ref[k]), (k, what)


def _both_modes(rt, orc, model, n_azim, delta, k=5, steep_seed=None):
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    if steep_seed is not None:
        meshgen.steep_tracks(rt, tg, steep_seed)
    ref = _oracle(orc, tg, k=k)
    on = hm.run(tg, k=k, walk=True)
    off = hm.run(tg, k=k, walk=False)
    cheap = hm.run(tg, k=k, walk="topo")  # cheap steps (decision from signed distances) + exact steps where they refuse
    _same(on, ref, "walk on")
    _same(off, ref, "walk off")
    _same(cheap, ref, "cheap steps")
    assert off["stats"]["walk_emits"] == 0 and off["stats"]["cheap_emits"] == 0 and on["stats"]["cheap_emits"] == 0
    on["cheap_stats"] = cheap["stats"]
    return tg, ref, on


@pytest.mark.parametrize("n_azim,delta", [(8, 2e-2), (32, 5e-3)])
def test_pincell_walk_on_off_checker(rt, orc, pincell, n_azim, delta):
    tg, ref, on = _both_modes(rt, orc, pincell, n_azim, delta)
    s, info = on["stats"], on["info"]
    assert info["walk_ok"] == 1 and info["records_walk"] == info["records"] - 160  # all but the 160 boundary-entry records
    assert info["cells_fragile"] == 0 and info["cells_degenerate"] == 0
    assert s["walk_emits"] >= 0.97 * ref["total"] - tg.n_total_tracks  # every track's first step is literal
    cs = on["cheap_stats"]
    assert cs["cheap_emits"] >= 0.999 * s["walk_emits"] and cs["cheap_restarts"] == 0  # nearly every walk step is a cheap one here
    print(f"pincell nφ={n_azim}: {ref['total']} segments, {s}; cheap steps: {cs}")


CASES = [("lattice", s) for s in range(4)] + [("lattice_far", s) for s in range(2)] + [("sliver", s) for s in range(4)] + \
        [("sliver_fine", s) for s in range(4)] + [("random", s) for s in range(3)] + [("cluster", s) for s in range(3)] + \
        [("near_vertex", s) for s in range(3)] + [("aligned", s) for s in range(3)] + [("lattice_mid", s) for s in range(2)] + \
        [("steep", s) for s in range(2)]  # the last four: aimed at the certificates' thresholds (tests/meshgen.py)


def _make(rt, kind, seed):
    if kind == "near_vertex":
        return meshgen.near_vertex_model(rt, 160 + seed, 250 + 200 * seed, (8, 16, 32, 4)[seed % 4], 0.01), 1.0
    if kind == "aligned":
        return meshgen.aligned_model(rt, 170 + seed, 9 + 4 * seed, (8, 16, 32, 4)[seed % 4], 0.01), 1.0
    if kind == "lattice_mid":
        return meshgen.lattice_model(rt, 180 + seed, 14, 14, jitter=0.3, x0=(10.0, 30.0)[seed], y0=(5.0, -40.0)[seed]), 1.0
    if kind == "steep":
        return (meshgen.lattice_model(rt, 190 + seed, 12, 12, jitter=0.2) if seed == 0 else meshgen.random_model(rt, 190 + seed, 300)), 1.0
    if kind == "lattice":
        return meshgen.lattice_model(rt, 100 + seed, 10 + 3 * seed, 10 + 3 * seed, jitter=(0.1, 0.25, 0.4, 0.3)[seed]), 1.0
    if kind == "lattice_far":
        return meshgen.lattice_model(rt, 110 + seed, 12, 12, jitter=0.3, w=2.5, h=0.4, x0=(100.0, -1000.0)[seed], y0=50.0), 0.4
    if kind == "sliver":
        return meshgen.sliver_model(rt, 120 + seed, 9 + 2 * seed, 9 + 2 * seed, gap=(1e-2, 1e-3, 1e-4, 1e-4)[seed]), 1.0
    if kind == "sliver_fine":
        return meshgen.sliver_model(rt, 130 + seed, 8 + 2 * seed, 8 + 2 * seed, gap=(1e-5, 1e-6, 1e-7, 1e-6)[seed], x0=-3.25, y0=2.5), 1.0
    if kind == "random":
        return meshgen.random_model(rt, 140 + seed, 150 + 200 * seed), 1.0
    return meshgen.random_model(rt, 150 + seed, 300 + 150 * seed, cluster=True), 1.0


@pytest.mark.parametrize("kind,seed", CASES)
def test_mesh_classes_walk_on_off_checker(rt, orc, kind, seed):
    model, scale = _make(rt, kind, seed)
    n_azim = (8, 16, 32, 4)[seed % 4]
    tg, ref, on = _both_modes(rt, orc, model, n_azim, 0.01 * scale, steep_seed=seed if kind == "steep" else None)
    s, info = on["stats"], on["info"]
    assert info["walk_ok"] == 1, "one bad cell must not switch the walk step off mesh-wide"
    assert s["walk_emits"] > 0
    frac = s["walk_emits"] / max(ref["total"], 1)
    if kind == "lattice":
        assert info["records_walk"] >= 0.85 * info["records"] and frac > 0.8, (info, s)
    print(f"{kind} seed {seed}: {model.num_cells} cells, {ref['total']} segments, {frac:.1%} by the walk step, "
          f"{int(info['records_walk'])}/{int(info['records'])} records walkable, eps ≤ {info['eps_max']:.1e}, "
          f"{int(np.count_nonzero(ref['status']))} tracks on which the reference itself throws")


@pytest.mark.parametrize("k", [1, 2, 8, 12, 40])
def test_knn_width_any_k(rt, orc, k):
    """find_element(mesh, x, k) takes any k (src/mesh.jl:123): wider than the in-register list (8) the node list is
    streamed; on a sliver mesh the fallback is really used."""
    model = meshgen.sliver_model(rt, 7, 10, 10, gap=1e-6)
    tg, ref, on = _both_modes(rt, orc, model, 16, 0.01, k=k)
    print(f"k={k}: {int(np.count_nonzero(ref['status'] == 1))} tracks fail to locate")


def test_wider_k_locates_more(rt, orc):
    """On a distorted mesh a wider k must rescue tracks (the reference's own advice, src/track.jl:141)."""
    model = meshgen.random_model(rt, 18, 568)
    tg = rt.TrackGenerator(model, 16, 0.01)
    rt.trace(tg)
    fails = {}
    for k in (2, 5, 12, 40):
        ref = _oracle(orc, tg, k=k)
        _same(hm.run(tg, k=k, walk=True), ref, f"k={k}")
        fails[k] = int(np.count_nonzero(ref["status"] == 1))
    print("locate failures by k:", fails)
    assert fails[40] <= fails[12] <= fails[5] <= fails[2]
    assert fails[12] < fails[2]


def test_nonmanifold_edge_is_not_crossed_but_walk_stays_on(rt, orc):
    """An edge shared by three cells (a cell stacked on another): only those records are switched off."""
    model = meshgen.lattice_model(rt, 3, 8, 8)
    xy = model.node_coordinates
    cells = np.asarray(model.cell_node_ids)
    # an interior edge (a, b) of cell 40: shared with exactly one other cell
    a, b, c = next((u, v, w) for u, v, w in (cells[40], cells[40][[1, 2, 0]], cells[40][[2, 0, 1]])
                   if sum(1 for cc in cells if u in cc and v in cc) == 2)
    # a third cell on edge (a, b): its apex is the centroid of cell 40 (it overlaps cell 40)
    xy2 = np.vstack([xy, xy[[a - 1, b - 1, c - 1]].mean(axis=0)])
    extra = np.sort(np.array([a, b, len(xy2)], np.int32))
    model2 = rt.DiscreteModel(xy2, np.vstack([cells, extra]))
    p = hm.prep(rt.Mesh(model2))
    assert p["info"]["walk_ok"] == 1 and "more than two cells" in p["note"]
    assert (p["epscode"][40] == -1).all(), "the overlapped cell must not be walked into"
    tg = rt.TrackGenerator(model2, 8, 0.02)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    _same(hm.run(tg, walk=True), ref, "walk on")
    _same(hm.run(tg, walk=False), ref, "walk off")


def test_degenerate_cell_switches_off_its_neighbourhood_only(rt, orc):
    model = meshgen.lattice_model(rt, 5, 10, 10)
    xy = model.node_coordinates.copy()
    cells = np.asarray(model.cell_node_ids)
    a, b, c = cells[55]
    xy[c - 1] = 0.5 * (xy[a - 1] + xy[b - 1])  # collapse cell 55 to a segment (its neighbours become distorted)
    model2 = rt.DiscreteModel(xy, cells)
    mesh = rt.Mesh(model2)
    p = hm.prep(mesh)
    assert p["info"]["cells_degenerate"] >= 1 and p["cls"][55] == 2
    assert p["info"]["walk_ok"] == 1 and 0 < p["info"]["records_walk"] < p["info"]["records"]
    off = (p["epscode"] == -1).all(axis=1)
    assert off[55] and off.sum() < 0.5 * len(off), "only the neighbourhood of the degenerate cell is switched off"
    tg = rt.TrackGenerator(model2, 8, 0.02)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    _same(hm.run(tg, walk=True), ref, "walk on")


def test_record_limit_switches_walk_off(rt, pincell, monkeypatch):
    """3·n_cells + 1 must fit the packed successor fields (2^27); mocked through the test knob."""
    monkeypatch.setenv("RT_TEST_WALK_RECORD_LIMIT", "1000")
    p = hm.prep(rt.Mesh(pincell))
    assert p["info"]["walk_ok"] == 0 and p["info"]["records_walk"] == 0 and "too many cells" in p["note"]
    monkeypatch.delenv("RT_TEST_WALK_RECORD_LIMIT")
    assert hm.prep(rt.Mesh(pincell))["info"]["walk_ok"] == 1


def test_host_code_under_sanitizers(tmp_path):
    """csrc/rt_host.cpp, csrc/rt_mesh_prep.hpp, the device geometry header on the host and the checker under
    AddressSanitizer + UBSan over fixtures, seeded fuzz meshes (degenerate cell, non-manifold edge included) and
    malformed mesh files; the host threading of rt_tracks_create and of the pipelined fetch (csrc/rt_hostpar.hpp) under ASan + UBSan and
    under ThreadSanitizer (tests/sanitize/run.sh; the 200-mesh log is kept as profiles/r05/host_sanitizers.log)."""
    import subprocess

    log = tmp_path / "san.log"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["bash", os.path.join(root, "tests", "sanitize", "run.sh"), "15", str(log)], stdout=subprocess.DEVNULL)
    text = log.read_text()
    assert text.count("exit code: 0") == 4 and "WARNING: ThreadSanitizer" not in text and "ERROR" not in text and "runtime error" not in text, text[-2000:]
    assert "12 refused" in text and "exact walk steps == walk off == cheap steps == checker: yes" in text


@pytest.mark.parametrize("tiny,shuffle", [(1e-8, True), (1e-5, False), (1e-6, True), (1e-10, False)])
def test_reseeding_distance_and_cell_node_order(rt, orc, tiny, shuffle):
    """tiny_step far from the reference's default, and cells whose three nodes come in random order (both edge
    orientations on either side of an edge; Gridap lists them ascending, the C ABI takes any order)."""
    model = meshgen.sliver_model(rt, 11, 12, 12, gap=1e-4)
    if shuffle:
        rs = np.random.default_rng(3)
        cells = np.asarray(model.cell_node_ids).copy()
        for c in range(len(cells)):
            cells[c] = cells[c][rs.permutation(3)]
        model = rt.DiscreteModel(model.node_coordinates, cells)
    tg = rt.TrackGenerator(model, 16, 0.01, tiny_step=tiny)
    rt.trace(tg)
    ref = _oracle(orc, tg)
    on = hm.run(tg, walk=True)
    _same(on, ref, "walk on")
    _same(hm.run(tg, walk=False), ref, "walk off")
    assert on["stats"]["walk_emits"] > 0.5 * ref["total"]


@pytest.mark.parametrize("iter_cap", [40, 300, 100000])
def test_cheap_steps_iteration_bound(rt, orc, pincell, iter_cap):
    """After cheap steps the iteration counter is an upper bound of the reference's (the tiny steps it takes while it still
    locates the cell it just left are bounded, not replayed).  A track whose bound reaches the cap is marched again with
    exact steps only: the status at the cap — and everything else — equals the checker's at any cap."""
    tg = rt.TrackGenerator(pincell, 8, 2e-2)
    rt.trace(tg)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, iter_cap=iter_cap, n_threads=0, k=5)
    r = hm.run(tg, k=5, walk="topo", iter_cap=iter_cap)
    _same(r, ref, "cheap steps, iter_cap=%d" % iter_cap)
    s = r["stats"]
    if iter_cap == 40:
        assert np.count_nonzero(ref["status"] == 4) > 0 and s["cheap_restarts"] > 0  # tracks that do hit the cap
    if iter_cap == 300:
        assert np.count_nonzero(ref["status"]) == 0 and s["cheap_restarts"] > 0      # only the bound reached it
    if iter_cap == 100000:
        assert s["cheap_restarts"] == 0
    print(f"iter_cap={iter_cap}: restarts {s['cheap_restarts']}, cheap emits {s['cheap_emits']}, failing {np.count_nonzero(ref['status'])}")


def test_bf16_up_is_an_upper_bound_and_monotone():
    """The cheap step's per-record constants g1, k2, dtf, lc are stored as bfloat16 rounded UP (rt_mesh_prep.hpp, bf16_up): the
    decoded value must never be below the double it came from, must be within one bfloat16 ulp (2^-7 relative) of it, and
    rounding must preserve order."""
    rng = np.random.default_rng(1)
    v = np.concatenate([10.0 ** rng.uniform(-300, 38, 500000), rng.uniform(0, 4, 400000), 2.0 ** rng.integers(-120, 120, 50000),
                        np.nextafter(2.0 ** rng.integers(-120, 120, 50000).astype(np.float64), np.inf)])
    pat, val = hm.bf16_up(v)
    fin = pat < 0x7F80
    assert np.all(val[fin] >= v[fin]), "bf16_up must round up"
    big = v > 1e-37  # (normal range of bfloat16 / float)
    assert np.all(val[fin & big] <= v[fin & big] * (1 + 2.0 ** -7))
    assert np.all(v[~fin] > 3.38e38), "only values beyond the bfloat16 range may map to inf"
    o = np.argsort(v)
    assert np.all(np.diff(val[o][fin[o]]) >= 0), "monotone"
    assert np.all(np.diff(pat[o].astype(np.int64)) >= 0)
    # exactly representable values map to themselves
    exact = (np.arange(128, 256, dtype=np.float64) / 128.0)[None, :] * (2.0 ** np.arange(-20, 20, dtype=np.float64))[:, None]
    _, ve = hm.bf16_up(exact.ravel())
    assert np.array_equal(ve, exact.ravel())
    assert hm.bf16_up(np.array([0.0, -1.0]))[0].tolist() == [0, 0]


@pytest.mark.parametrize("kind,seed", [("lattice", 1), ("sliver", 2), ("random", 1), ("cluster", 2), ("near_vertex", 1), ("aligned", 2),
                                       ("lattice_mid", 0), ("lattice_mid", 1), ("sliver_fine", 1)])
def test_cheap_step_constants_dominate_their_derivation(rt, kind, seed):
    """Every per-record constant of the cheap step, as the device decodes it, is at least the expression DESIGN.md §2 derives it
    from, recomputed here in numpy from the mesh alone (independent of rt_mesh_prep.hpp's arithmetic): g1 >= g·√eps·dtf,
    k2 >= κ0·l_max·g / (0.3·√eps), dtf >= 1.5·|dT| + 0.375·|dT'|, lc >= (l_min + 1.2·√eps·h_min) / (2·tan(γ/2)),
    E >= isolation margin (the walk record's) + 0.375·√eps + g·tiny_max, and a record whose g·δ_add exceeds 0.075·√eps has none."""
    model, _ = _make(rt, kind, seed)
    mesh = rt.Mesh(model)
    T = hm.topo_records(mesh)
    P = hm.prep(mesh)
    tol, ulp = 1.4901161193847656e-8, 1.1102230246251565e-16
    x, y = np.asarray(mesh.x), np.asarray(mesh.y)
    cn = np.asarray(mesh.cell_nodes).reshape(-1, 3) - 1
    bb = np.asarray(mesh.bb)
    corner = np.hypot(max(abs(bb[0]), abs(bb[2])), max(abs(bb[1]), abs(bb[3])))
    vx, vy = x[cn], y[cn]
    area2 = np.abs((vx[:, 1] - vx[:, 0]) * (vy[:, 2] - vy[:, 0]) - (vx[:, 2] - vx[:, 0]) * (vy[:, 1] - vy[:, 0]))
    elen = np.hypot(vx - np.roll(vx, -1, axis=1), vy - np.roll(vy, -1, axis=1))
    lmax = elen.max(axis=1)
    hmin = area2 / lmax
    g = 1.0 / hmin
    # neighbour across every edge
    owner = {}
    for c in range(len(cn)):
        for e in range(3):
            owner.setdefault(tuple(sorted((cn[c, e], cn[c, (e + 1) % 3]))), []).append(c)
    kappa0, dadd = 20 * ulp * corner, 3 * 4096 * ulp * corner
    assert T["tiny_max"] >= max(1e-6 * lmax.max(), 4e-8) * (1 - 1e-15)
    n_on = 0
    for c in range(len(cn)):
        for e in range(3):
            if T["extras"][c, e] >= 15:
                continue
            n_on += 1
            nb = [u for u in owner[tuple(sorted((cn[c, e], cn[c, (e + 1) % 3])))] if u != c]
            assert len(nb) == 1, "a cheap record needs exactly one cell behind its entry edge"
            assert g[c] * dadd <= 0.075 * tol * (1 + 1e-12)
            dtf = 1.5 * area2[nb[0]] + 0.375 * area2[c]
            assert T["dtf"][c, e] >= dtf
            assert T["g1"][c, e] >= g[c] * tol * dtf
            assert T["k2"][c, e] >= kappa0 * lmax[c] * g[c] / (0.3 * tol)
            i0, i1, i2 = e, (e + 1) % 3, (e + 2) % 3

            def half_tan(iv, ia, ib):
                ux, uy, wx, wy = vx[c, ia] - vx[c, iv], vy[c, ia] - vy[c, iv], vx[c, ib] - vx[c, iv], vy[c, ib] - vy[c, iv]
                return np.tan(0.5 * np.arctan2(abs(ux * wy - uy * wx), ux * wx + uy * wy))
            cv = 2.0 * min(half_tan(i0, i1, i2), half_tan(i1, i2, i0))
            assert T["lc"][c, e] >= (T["l_min"] + 1.2 * tol * hmin[c]) / cv
            iso = 2.0 ** (P["epscode"][c, e] - 20.0)  # the walk record's isolation margin
            assert P["epscode"][c, e] >= 0
            assert T["E"][c, e] >= iso + 0.375 * tol + g[c] * T["tiny_max"]
            # border clearance: along the reference's path inboundary(xp, tiny) stays false (every side of the box)
            for q, (b0, b1, b2) in enumerate(((vx[c, i0] - bb[0], vx[c, i1] - bb[0], vx[c, i2] - bb[0]), (bb[2] - vx[c, i0], bb[2] - vx[c, i1], bb[2] - vx[c, i2]),
                                              (vy[c, i0] - bb[1], vy[c, i1] - bb[1], vy[c, i2] - bb[1]), (bb[3] - vy[c, i0], bb[3] - vy[c, i1], bb[3] - vy[c, i2]))):
                assert b0 + b1 > 0
                assert T["E"][c, e] >= (T["tiny_max"] + 0.375 * tol * abs(b2)) / (b0 + b1)
    print(f"{kind} seed {seed}: {len(cn)} cells, {n_on} cheap records checked, tiny_max {T['tiny_max']:.1e}")
    if kind in ("lattice", "near_vertex", "aligned"):
        assert n_on > len(cn)
