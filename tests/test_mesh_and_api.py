"""Host-side mirror of the reference's interface: mesh ingest, TrackGenerator, trace!,
Track/Segment views, error behaviour (src/mesh.jl:24-83, src/trackgenerator.jl:80-125,357-361)."""
import numpy as np
import pytest


def test_msh_and_json_loaders_agree(rt):
    a = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
    b = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    assert a.num_nodes == b.num_nodes == 2036 and a.num_cells == b.num_cells == 3910
    assert np.array_equal(a.cell_node_ids, b.cell_node_ids)
    assert np.array_equal(a.node_coordinates, b.node_coordinates)


def test_mesh_tables(rt, pincell):
    m = rt.Mesh(pincell)
    assert m.bb_min == (0.0, 0.0) and m.bb_max == (1.6, 1.6)
    assert m.width() == 1.6 and m.height() == 1.6
    assert m.node_cells_ptrs[0] == 0 and m.node_cells_ptrs[-1] == 3 * m.num_cells
    for n in (0, 17, 2035):
        cells = m.node_cells_data[m.node_cells_ptrs[n]:m.node_cells_ptrs[n + 1]]
        assert np.all(np.diff(cells) > 0)  # ascending cell ids per node
        assert all((n + 1) in m.cell_nodes[c - 1] for c in cells)
    val = np.diff(m.node_cells_ptrs)
    assert val.max() == 7 and abs(val.mean() - 5.76) < 0.01


def test_rejects_non_triangles(rt):
    with pytest.raises(ValueError):
        rt.DiscreteModel(np.zeros((4, 2)), np.array([[1, 2, 3, 4]], dtype=np.int32))


def test_segmentize_requires_trace(rt, pincell):  # src/trackgenerator.jl:360-361
    tg = rt.TrackGenerator(pincell, 8, 0.02)
    with pytest.raises(RuntimeError, match="Segmentation is intended after tracing"):
        rt.segmentize(tg)


def test_track_views(rt, traced):
    tg = traced(8, 0.02)
    tracks = tg.tracks_by_uid
    assert len(tracks) == 420 and len(tg.tracks) == 4 and [len(t) for t in tg.tracks] == [105] * 4
    t1 = tracks[0]
    assert t1.uid == 1 and t1.azim_idx == 1 and t1.track_idx == 1
    assert t1.p[1] == 0.0 and abs(np.hypot(t1.p[0] - t1.q[0], t1.p[1] - t1.q[1]) - t1.ell) < 1e-15
    assert t1.ℓ == t1.ell and t1.ϕ == t1.phi  # NFKC-normalised spellings of the reference's field names
    assert len(t1.segments) == 0  # "if it is zero, run segmentize!" (src/track.jl:94)
    assert t1.next_track_fwd.uid >= 1 and "Azimuthal angle" in repr(t1)
    A, B, C = t1.ABC
    assert abs(A * t1.p[0] + B * t1.p[1] + C) < 1e-15 and abs(A * t1.q[0] + B * t1.q[1] + C) < 1e-15
    assert abs(A * A + B * B + C * C - 1) < 1e-15


def test_cyclic_linking_is_closed(rt, pincell):
    """Following next_track_fwd with reflective BCs returns to the start (cyclic tracking)."""
    bcs = rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)
    tg = rt.TrackGenerator(pincell, 8, 0.1, bcs=bcs)
    rt.trace(tg)
    for start in (0, 5, tg.n_total_tracks - 1):
        u, d, seen = start, rt.Forward, 0
        while True:
            if d == rt.Forward:
                nu, nd = int(tg.next_fwd_uid[u]) - 1, int(tg.dir_next_fwd[u])
            else:
                nu, nd = int(tg.next_bwd_uid[u]) - 1, int(tg.dir_next_bwd[u])
            u, d, seen = nu, nd, seen + 1
            assert seen <= 4 * tg.n_total_tracks
            if u == start and d == rt.Forward:
                break


def test_shard_ranges_balance(rt, traced):
    from raytracing_jl_amd.distributed import shard_ranges

    tg = traced(32, 5e-3)
    for w in (1, 2, 3, 8):
        r = shard_ranges(tg.ell, w)
        assert r[0][0] == 0 and r[-1][1] == tg.n_total_tracks
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        loads = np.array([tg.ell[a:b].sum() for a, b in r])
        assert loads.max() / loads.mean() < 1.01
    assert shard_ranges(np.array([]), 2) == [(0, 0), (0, 0)]
