"""Random unstructured meshes against the oracle (GPU only).

scipy Delaunay triangulations of random point clouds in a rectangle (boundary points on the exact
sides, interior points uniformly random — so slivers, very small and very obtuse triangles occur),
cells kept in the generator's order with ascending node ids like Gridap's oriented grids.  The HIP
path must reproduce the oracle bit for bit whatever mix of walk steps, tiny-step runs and generic
steps a mesh provokes, in both staging modes and with track splitting.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_model(rt, seed, n_interior, **kw):
    from meshgen import random_model

    return random_model(rt, seed, n_interior, **kw)


def _oracle(orc, tg):
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                      tiny_step=tg.tiny_step, iter_cap=4000000, n_threads=0)
    aq = tg.azimuthal_quadrature
    r["volumes"] = om.fill_volumes(r["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)  # of the run just made
    return r


def _run(rt, tg, opts):
    """segmentize through the C ABI with internal options set before the track handle exists."""
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets()
    seg = dt.fetch_segments()
    vol = dt.fetch_volumes()
    dt.close()
    return total, off, st, seg, vol


@pytest.mark.parametrize("seed,n_interior,kw", [
    (1, 200, {}),
    (2, 1500, {}),
    (3, 600, dict(cluster=True)),
    (4, 400, dict(w=3.0, h=0.7, x0=-1.5, y0=10.0)),
    (5, 3000, dict(nb=40)),
])
def test_random_delaunay_mesh_matches_oracle(rt, orc, seed, n_interior, kw):
    model = _random_model(rt, seed, n_interior, **kw)
    w, h = kw.get("w", 1.0), kw.get("h", 1.0)
    tg = rt.TrackGenerator(model, 16, 0.004 * min(w, h))
    rt.trace(tg)
    ref = _oracle(orc, tg)
    ref_vol = ref["volumes"]
    for opts in (dict(), dict(split=24), dict(walk=0), dict(split=0)):
        total, off, st, seg, vol = _run(rt, tg, opts)
        assert total == ref["total"], opts
        assert np.array_equal(st, ref["status"]), ("per-track status differs", opts)
        assert np.array_equal(off, ref["offsets"]), ("segment counts differ", opts)
        assert np.array_equal(seg["element"], ref["element"]), ("element ids differ", opts)
        for k in ("px", "py", "qx", "qy", "ell"):
            assert np.array_equal(seg[k], ref[k]), (k, opts)  # bit-identical, beyond the 1e-10 bar
        assert np.allclose(vol, ref_vol, rtol=1e-10, atol=1e-300), opts
    print(f"seed {seed}: {model.num_cells} cells, {tg.n_total_tracks} tracks, {int(ref['total'])} segments, "
          f"{int(np.count_nonzero(ref['status']))} tracks on which the reference itself throws (status codes compared)")
