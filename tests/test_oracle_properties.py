"""Pins the oracle (oracle/rt_oracle.c) with everything the reference's tests hold for the
segmentize! path (test/runtests.jl:10-12 no-throw, :30-35 entry/exit points, :37-43 Σℓ)
plus cross-checks that do not depend on the oracle's own logic (brute-force containment of
segment midpoints, contiguity, Σ volumes = domain area, brute-force nearest node)."""
import numpy as np
import pytest

RTOL = 1.4901161193847656e-8


def _isapprox_v2(px, py, qx, qy):
    d = np.hypot(px - qx, py - qy)
    return d <= RTOL * np.maximum(np.hypot(px, py), np.hypot(qx, qy))


def _containing_cells(mesh, x, y, chunk=2048):
    """Brute force: for each point the set of cells whose closed triangle contains it
    (orientation-free sign test with a small absolute slack)."""
    cn = mesh.cell_nodes - 1
    x1, y1 = mesh.x[cn[:, 0]], mesh.y[cn[:, 0]]
    x2, y2 = mesh.x[cn[:, 1]], mesh.y[cn[:, 1]]
    x3, y3 = mesh.x[cn[:, 2]], mesh.y[cn[:, 2]]
    out = []
    for s in range(0, len(x), chunk):
        px, py = x[s:s + chunk, None], y[s:s + chunk, None]
        d1 = (px - x2) * (y1 - y2) - (x1 - x2) * (py - y2)
        d2 = (px - x3) * (y2 - y3) - (x2 - x3) * (py - y3)
        d3 = (px - x1) * (y3 - y1) - (x3 - x1) * (py - y1)
        eps = 1e-12
        neg = (d1 < -eps) | (d2 < -eps) | (d3 < -eps)
        pos = (d1 > eps) | (d2 > eps) | (d3 > eps)
        out.append(~(neg & pos))
    return np.concatenate(out, axis=0)


@pytest.fixture(scope="module")
def c1(traced, oracle_run):
    tg = traced(8, 0.02)
    return tg, oracle_run(tg)


def test_no_track_fails(c1):  # test/runtests.jl:10-12 — segmentize! must not throw
    tg, r = c1
    assert r["status"].max() == 0
    assert r["total"] == r["offsets"][-1] == len(r["ell"])
    assert np.all(np.diff(r["offsets"]) >= 1)


def test_entry_and_exit_points(c1):  # test/runtests.jl:30-35
    tg, r = c1
    first, last = r["offsets"][:-1], r["offsets"][1:] - 1
    assert np.all(_isapprox_v2(tg.px, tg.py, r["px"][first], r["py"][first]))
    assert np.all(_isapprox_v2(tg.qx, tg.qy, r["qx"][last], r["qy"][last]))


def test_track_length(c1):  # test/runtests.jl:37-43
    tg, r = c1
    sums = np.add.reduceat(r["ell"], r["offsets"][:-1])
    assert np.all(np.abs(tg.ell - sums) <= RTOL * np.maximum(np.abs(tg.ell), np.abs(sums)))


def test_segment_lengths_and_contiguity(c1):
    tg, r = c1
    assert np.array_equal(r["ell"], np.sqrt((r["px"] - r["qx"]) ** 2 + (r["py"] - r["qy"]) ** 2))
    assert np.all(r["ell"] > 0)
    inner = np.ones(r["total"], bool)
    inner[r["offsets"][:-1]] = False  # first segment of each track has no predecessor
    gap = np.hypot(r["px"][1:] - r["qx"][:-1], r["py"][1:] - r["qy"][:-1])[inner[1:]]
    assert gap.max() <= 5e-8  # a handful of tiny_step hops at vertices


def test_midpoints_lie_in_their_element(c1):
    tg, r = c1
    mx, my = 0.5 * (r["px"] + r["qx"]), 0.5 * (r["py"] + r["qy"])
    inside = _containing_cells(tg.mesh, mx, my)
    assert inside[np.arange(r["total"]), r["element"] - 1].all()


def test_volumes_sum_to_domain_area(c1):
    tg, r = c1
    area = tg.mesh.width() * tg.mesh.height()
    assert abs(r["volumes"].sum() - area) < 1e-9 * area
    # ray-traced cell volumes approximate the triangle areas
    cn = tg.mesh.cell_nodes - 1
    x, y = tg.mesh.x, tg.mesh.y
    tri = 0.5 * np.abs((x[cn[:, 1]] - x[cn[:, 0]]) * (y[cn[:, 2]] - y[cn[:, 0]])
                       - (x[cn[:, 2]] - x[cn[:, 0]]) * (y[cn[:, 1]] - y[cn[:, 0]]))
    assert np.median(np.abs(r["volumes"] - tri) / tri) < 0.15  # coarse: δ=0.02 vs cell size ≈0.03


def test_kdtree_matches_bruteforce(rt, orc, pincell):
    mesh = rt.Mesh(pincell)
    om = orc.OracleMesh.from_mesh(mesh)
    rng = np.random.default_rng(7)
    pts = np.column_stack((rng.uniform(-0.05, 1.65, 4000), rng.uniform(-0.05, 1.65, 4000)))
    xy = pincell.node_coordinates
    for (x, y) in pts:
        d2 = (xy[:, 0] - x) ** 2 + (xy[:, 1] - y) ** 2
        order = np.argsort(d2, kind="stable")
        assert om.nn(x, y) == order[0] + 1
        got = om.knn(x, y, 5, skip=order[0] + 1)
        assert got.tolist() == (order[1:6] + 1).tolist()


def test_find_element_matches_bruteforce(rt, orc, pincell):
    mesh = rt.Mesh(pincell)
    om = orc.OracleMesh.from_mesh(mesh)
    rng = np.random.default_rng(11)
    x, y = rng.uniform(0.001, 1.599, 3000), rng.uniform(0.001, 1.599, 3000)
    inside = _containing_cells(mesh, x, y)
    for i in range(len(x)):
        e = om.find_element(x[i], y[i], 5)
        assert e >= 1 and inside[i, e - 1]


def test_bruteforce_and_kdtree_marches_agree(orc, traced):
    tg = traced(8, 0.02)
    om = orc.OracleMesh.from_mesh(tg.mesh)
    a = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell)
    om.set_bruteforce(True)
    b = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell)
    for k in ("offsets", "element", "px", "py", "qx", "qy", "ell"):
        assert np.array_equal(a[k], b[k])


def test_host_trig_and_libm_trig_agree(orc, traced):
    """The product receives cos ϕ / sin ϕ from the host; the reference's advance_step calls
    libm each time.  Same libm here, so both oracle modes must give identical records."""
    tg = traced(8, 0.02)
    om = orc.OracleMesh.from_mesh(tg.mesh)
    a = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell)
    b = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi)
    for k in ("offsets", "element", "px", "qy", "ell"):
        assert np.array_equal(a[k], b[k])


def test_rtol_and_length_mismatch_status(orc, traced):
    tg = traced(8, 0.02)
    om = orc.OracleMesh.from_mesh(tg.mesh)
    bad = tg.ell.copy()
    bad[3] *= 1.001
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, bad)
    assert r["status"][3] == 2 and (np.delete(r["status"], 3) == 0).all()
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, bad, rtol=0.01)
    assert r["status"].max() == 0


def test_headline_mesh_config2_properties(traced, oracle_run):
    """C2 (nφ=32, δ=5e-3): the same invariants at ≈4.7e5 segments."""
    tg = traced(32, 5e-3)
    r = oracle_run(tg)
    assert r["status"].max() == 0 and tg.n_total_tracks == 6548
    sums = np.add.reduceat(r["ell"], r["offsets"][:-1])
    assert np.all(np.abs(tg.ell - sums) <= RTOL * np.maximum(tg.ell, sums))
    assert abs(r["volumes"].sum() - 2.56) < 1e-9
    cn = tg.mesh.cell_nodes - 1
    x, y = tg.mesh.x, tg.mesh.y
    tri = 0.5 * np.abs((x[cn[:, 1]] - x[cn[:, 0]]) * (y[cn[:, 2]] - y[cn[:, 0]])
                       - (x[cn[:, 2]] - x[cn[:, 0]]) * (y[cn[:, 1]] - y[cn[:, 0]]))
    assert np.median(np.abs(r["volumes"] - tri) / tri) < 0.03  # converges to the triangle areas as δ→0
