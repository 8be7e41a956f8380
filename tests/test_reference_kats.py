"""Known answers held by the reference's own test-suite (test/runtests.jl), extracted into
tests/golden/reference_kats.json by tools/extract_reference_kats.py.  Both the product's
host-side trace! mirror and the oracle's restatement must reproduce them."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
BC = {"Vacuum": 0, "Reflective": 1, "Periodic": 2}
DIR = {"Forward": 0, "Backward": 1}


def test_tracing_counts_host(rt, pincell):  # test/runtests.jl:14-19
    k = KATS["main"]
    tg = rt.TrackGenerator(pincell, k["n_azim"], k["delta"])
    assert tg.n_total_tracks == k["n_total_tracks"]
    assert tg.n_tracks_x.tolist() == k["n_tracks_x"]
    assert tg.n_tracks_y.tolist() == k["n_tracks_y"]
    assert tg.n_tracks.tolist() == k["n_tracks"]


def test_tracing_counts_oracle(rt, orc, pincell):
    k = KATS["main"]
    mesh = rt.Mesh(pincell)
    total, ntx, nty = orc.track_counts(mesh.width(), mesh.height(), k["n_azim"], k["delta"])
    assert total == k["n_total_tracks"]
    assert ntx.tolist() == k["n_tracks_x"] and nty.tolist() == k["n_tracks_y"]


def test_azimuthal_quadrature(rt, orc, traced):  # test/runtests.jl:21-28
    k = KATS["main"]
    tg = traced(k["n_azim"], k["delta"])
    aq = tg.azimuthal_quadrature
    assert (aq.nazim(), aq.nazim2(), aq.nazim4()) == (8, 4, 2)
    assert np.isclose(aq.delta, k["delta"])
    assert np.allclose(aq.delta_s, k["delta_s"], rtol=1.5e-8, atol=0)
    assert np.allclose(aq.phis, k["phis"], rtol=1.5e-8, atol=0)
    o = orc.trace(tg.mesh.bb, k["n_azim"], k["delta"])
    assert np.allclose(o["delta_s"], k["delta_s"], rtol=1.5e-8, atol=0)
    assert np.allclose(o["phis"], k["phis"], rtol=1.5e-8, atol=0)
    # the two restatements agree to the last bit on every per-track input
    for name_o, name_h in (("px", "px"), ("py", "py"), ("qx", "qx"), ("qy", "qy"), ("phi", "phi"),
                           ("ell", "ell"), ("A", "A"), ("B", "B"), ("C", "C")):
        assert np.array_equal(o[name_o], getattr(tg, name_h)), name_o
    assert np.array_equal(o["azim_idx"], tg.azim_idx) and np.array_equal(o["track_idx"], tg.track_idx)
    assert np.allclose(o["omega"], aq.omega_a, rtol=0, atol=0)


@pytest.mark.parametrize("case", KATS["reflection"], ids=lambda c: "nphi%d" % c["n_azim"])
def test_reflection_linking(rt, orc, pincell, case):  # test/runtests.jl:46-334
    b = case["bcs"]
    bcs = rt.BoundaryConditions(**{side: getattr(rt, b[side]) for side in ("top", "bottom", "left", "right")})
    tg = rt.TrackGenerator(pincell, case["n_azim"], case["delta"], bcs=bcs)
    rt.trace(tg)
    o = orc.trace(tg.mesh.bb, case["n_azim"], case["delta"],
                  bcs=(BC[b["top"]], BC[b["bottom"]], BC[b["right"]], BC[b["left"]]))
    assert tg.n_total_tracks == len(case["tracks"]) == o["n_total_tracks"]
    for k in case["tracks"]:
        tr = tg.tracks_by_uid[k["uid"] - 1]
        assert rt.bc_fwd(tr).name == k["bc_fwd"] and rt.bc_bwd(tr).name == k["bc_bwd"]
        assert tr.next_track_fwd.uid == k["next_fwd_uid"]
        assert tr.next_track_bwd.uid == k["next_bwd_uid"]
        assert rt.dir_next_track_fwd(tr) == DIR[k["dir_fwd"]]
        assert rt.dir_next_track_bwd(tr) == DIR[k["dir_bwd"]]
        u = k["uid"] - 1
        assert o["bc_fwd"][u] == BC[k["bc_fwd"]] and o["bc_bwd"][u] == BC[k["bc_bwd"]]
        assert o["next_fwd"][u] == k["next_fwd_uid"] and o["next_bwd"][u] == k["next_bwd_uid"]
        assert o["dir_fwd"][u] == DIR[k["dir_fwd"]] and o["dir_bwd"][u] == DIR[k["dir_bwd"]]


def test_quadrature_argument_validation(rt, pincell):  # src/azimuthal_quad.jl:21-25
    for n_azim, delta in ((0, 0.1), (6, 0.1), (8, 0.0), (8, -1.0)):
        with pytest.raises(ValueError):
            rt.TrackGenerator(pincell, n_azim, delta)
