"""Native host rows of the C-ABI library: rt_trace_counts / rt_trace against the
reference's known answers, the oracle's restatement and the numpy mirror; rt_msh_load against the
Python loader.  CPU only — these entry points do not need a GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
BC = {"Vacuum": 0, "Reflective": 1, "Periodic": 2}
DIR = {"Forward": 0, "Backward": 1}


@pytest.fixture(scope="module")
def capi():
    from raytracing_jl_amd import _capi

    _capi.build()
    return _capi


def test_counts_known_answers(capi):  # test/runtests.jl:14-19
    k = KATS["main"]
    total, ntx, nty = capi.native_trace_counts(1.6, 1.6, k["n_azim"], k["delta"])
    assert total == k["n_total_tracks"]
    assert ntx.tolist() == k["n_tracks_x"] and nty.tolist() == k["n_tracks_y"]
    for bad in ((0, 0.1), (6, 0.1), (8, 0.0)):
        with pytest.raises(ValueError, match="DomainError"):
            capi.native_trace_counts(1.6, 1.6, *bad)


@pytest.mark.parametrize("n_azim,delta", [(8, 0.02), (32, 5e-3), (4, 0.8), (128, 1e-3)])
def test_native_numpy_and_oracle_agree_bitwise(rt, orc, pincell, n_azim, delta):
    a = rt.TrackGenerator(pincell, n_azim, delta)
    b = rt.TrackGenerator(pincell, n_azim, delta)
    rt.trace(a, backend="native")
    rt.trace(b, backend="numpy")
    o = orc.trace(a.mesh.bb, n_azim, delta)
    for name in ("px", "py", "qx", "qy", "phi", "cos_phi", "sin_phi", "ell", "A", "B", "C", "azim_idx", "track_idx",
                 "bc_fwd", "bc_bwd", "dir_next_fwd", "dir_next_bwd", "next_fwd_uid", "next_bwd_uid"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    for no, na in (("px", "px"), ("qy", "qy"), ("ell", "ell"), ("A", "A"), ("C", "C"), ("next_fwd", "next_fwd_uid")):
        assert np.array_equal(o[no], getattr(a, na)), no
    for name in ("phis", "delta_s", "omega_a"):
        assert np.array_equal(getattr(a.azimuthal_quadrature, name), getattr(b.azimuthal_quadrature, name))


def test_quadrature_known_answers(rt, pincell):  # test/runtests.jl:21-28
    k = KATS["main"]
    tg = rt.TrackGenerator(pincell, k["n_azim"], k["delta"])
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    assert np.allclose(aq.delta_s, k["delta_s"], rtol=1.5e-8, atol=0)
    assert np.allclose(aq.phis, k["phis"], rtol=1.5e-8, atol=0)


@pytest.mark.parametrize("case", KATS["reflection"], ids=lambda c: "nphi%d" % c["n_azim"])
def test_reflection_linking_native(rt, pincell, case):  # test/runtests.jl:46-334
    b = case["bcs"]
    bcs = rt.BoundaryConditions(**{side: getattr(rt, b[side]) for side in ("top", "bottom", "left", "right")})
    tg = rt.TrackGenerator(pincell, case["n_azim"], case["delta"], bcs=bcs)
    rt.trace(tg, backend="native")
    for k in case["tracks"]:
        u = k["uid"] - 1
        assert tg.bc_fwd[u] == BC[k["bc_fwd"]] and tg.bc_bwd[u] == BC[k["bc_bwd"]]
        assert tg.next_fwd_uid[u] == k["next_fwd_uid"] and tg.next_bwd_uid[u] == k["next_bwd_uid"]
        assert tg.dir_next_fwd[u] == DIR[k["dir_fwd"]] and tg.dir_next_bwd[u] == DIR[k["dir_bwd"]]


@pytest.mark.parametrize("bcs_kw", [dict(top="Periodic", bottom="Periodic", left="Periodic", right="Periodic"),
                                    dict(top="Vacuum", bottom="Reflective", left="Periodic", right="Periodic")])
def test_periodic_linking_matches_numpy(rt, pincell, bcs_kw):
    bcs = rt.BoundaryConditions(**{k: getattr(rt, v) for k, v in bcs_kw.items()})
    a = rt.TrackGenerator(pincell, 8, 0.1, bcs=bcs)
    b = rt.TrackGenerator(pincell, 8, 0.1, bcs=bcs)
    rt.trace(a, backend="native")
    rt.trace(b, backend="numpy")
    for name in ("next_fwd_uid", "next_bwd_uid", "dir_next_fwd", "dir_next_bwd", "bc_fwd", "bc_bwd"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


@pytest.mark.parametrize("name", ["pincell.msh", "bwr_like.msh"])
def test_native_msh_ingest_matches_python_loader(rt, name):
    path = rt.data_path(name)
    py = rt.Mesh(rt.GmshDiscreteModel(path))
    nat = rt.Mesh.from_msh(path)
    assert np.array_equal(py.x, nat.x) and np.array_equal(py.y, nat.y)
    assert np.array_equal(py.cell_nodes, nat.cell_nodes)
    assert np.array_equal(py.node_cells_ptrs, nat.node_cells_ptrs)
    assert np.array_equal(py.node_cells_data, nat.node_cells_data)
    assert py.bb_min == nat.bb_min and py.bb_max == nat.bb_max


def test_native_gridap_json_ingest_matches_python_loader(rt):
    """The reference's tests load demo/pincell.json (test/runtests.jl:5-6): native reader vs the Python one."""
    path = rt.data_path("pincell.json")
    py = rt.Mesh(rt.DiscreteModelFromFile(path))
    nat = rt.Mesh.from_file(path)
    assert np.array_equal(nat.x, py.x) and np.array_equal(nat.y, py.y)
    assert np.array_equal(nat.cell_nodes, py.cell_nodes)
    assert np.array_equal(nat.node_cells_ptrs, py.node_cells_ptrs)
    assert np.array_equal(nat.node_cells_data, py.node_cells_data)
    assert np.array_equal(nat.bb, py.bb)
    # and it is the same mesh as the gmsh file it was made from
    msh = rt.Mesh.from_file(rt.data_path("pincell.msh"))
    assert np.array_equal(nat.x, msh.x) and np.array_equal(nat.cell_nodes, msh.cell_nodes)


def test_native_json_errors(capi, tmp_path):
    bad = tmp_path / "bad.json"
    bad.write_text('{"grid": {"node_coordinates": [0, 0, 1, 0, 0, 1], "cell_node_ids": {"ptrs": [1, 5], "data": [1, 2, 3, 1]}}}')
    with pytest.raises(capi.RtError, match="triangular"):
        capi.native_load_msh(str(bad))
    bad.write_text('{"nothing": 1}')
    with pytest.raises(capi.RtError, match="grid"):
        capi.native_load_msh(str(bad))


def test_native_msh_errors(capi, tmp_path):
    with pytest.raises(capi.RtError, match="cannot open"):
        capi.native_load_msh(str(tmp_path / "missing.msh"))
    bad = tmp_path / "bad.msh"
    bad.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n")
    with pytest.raises(capi.RtError, match="4.1"):
        capi.native_load_msh(str(bad))


def test_native_msh_with_more_entity_blocks_than_nodes(rt, capi, tmp_path):
    """gmsh writes a block for every entity, also those without nodes ("1 1 0 0": a curve with no interior node), so a coarse
    mesh has more entity blocks than nodes — a 5-node, 4-triangle square with 9 blocks loads like it does in the Python reader."""
    msh = tmp_path / "coarse.msh"
    msh.write_text(
        "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n"
        "$Nodes\n9 5 1 5\n"
        "0 1 0 1\n1\n0 0 0\n0 2 0 1\n2\n1 0 0\n0 3 0 1\n3\n1 1 0\n0 4 0 1\n4\n0 1 0\n"
        "1 1 0 0\n1 2 0 0\n1 3 0 0\n1 4 0 0\n"
        "2 1 0 1\n5\n0.5 0.5 0\n"
        "$EndNodes\n"
        "$Elements\n5 8 1 8\n"
        "1 1 1 1\n1 1 2\n1 2 1 1\n2 2 3\n1 3 1 1\n3 3 4\n1 4 1 1\n4 4 1\n"
        "2 1 2 4\n5 1 2 5\n6 2 3 5\n7 3 4 5\n8 4 1 5\n"
        "$EndElements\n")
    x, y, cells, ptrs, data, bb = capi.native_load_msh(str(msh))
    py = rt.Mesh(rt.GmshDiscreteModel(str(msh)))
    assert len(x) == 5 and len(cells) == 4
    assert np.array_equal(x, py.x) and np.array_equal(y, py.y) and np.array_equal(cells, py.cell_nodes)
    assert np.array_equal(ptrs, py.node_cells_ptrs) and np.array_equal(data, py.node_cells_data)
