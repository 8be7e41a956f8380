"""Stream-ordered calls (rt_set_option "async" = 1): rt_segmentize returns after march + scan while the compaction is still on the
stream; every accessor waits by itself.  Results must be those of synchronous calls, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handles(rt, tg, **opts):
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    return dm, dt


def _all(dt):
    off, st = dt.fetch_offsets()
    recs = dt.fetch_segments()
    return [off, st] + [recs[k] for k in ("px", "py", "qx", "qy", "ell", "element")] + [dt.fetch_volumes()]


@pytest.mark.parametrize("mesh,na,d", [("pincell.msh", 32, 5e-3), ("pincell.msh", 128, 1e-3), ("bwr_like.msh", 16, 0.02)])
def test_async_calls_give_the_synchronous_results(rt, mesh, na, d):
    model = rt.GmshDiscreteModel(rt.data_path(mesh))
    tg = rt.TrackGenerator(model, na, d)
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    seg = lambda h: h.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    dm0, dt0 = _handles(rt, tg, split=0)
    total0 = seg(dt0)
    want = _all(dt0)
    dm1, dt1 = _handles(rt, tg, split=0, **{"async": 1})
    for _ in range(5):  # back to back: each call's march queues behind the previous call's compaction
        total1 = seg(dt1)
    assert total1 == total0
    assert dt1.failed() == dt0.failed()
    got = _all(dt1)  # (every accessor waits for the call)
    for a, b in zip(got[:-1], want[:-1]):
        assert np.array_equal(a, b)
    assert np.allclose(got[-1], want[-1], rtol=1e-12, atol=0)
    # explicit wait, then the device pointers are safe on any stream; the pinned fetch agrees too
    seg(dt1)
    dt1.wait()
    p_off, p_st, p_rec = dt1.fetch_pinned()
    assert np.array_equal(np.asarray(p_off), want[0]) and np.array_equal(np.asarray(p_st), want[1])
    assert np.array_equal(np.asarray(p_rec["element"]), want[7]) and np.array_equal(np.asarray(p_rec["qx"]), want[4])
    # back to synchronous calls on the same handle
    dm1.set_option("async", 0)
    assert seg(dt1) == total0
    for a, b in zip(_all(dt1)[:-1], want[:-1]):
        assert np.array_equal(a, b)
    for h in (dt0, dt1, dm0, dm1):
        h.close()


def test_async_call_then_sweep_over_the_staged_rows(rt):
    """compact = 0 + async: march + scan, then rt_sweep on the same stream; equal to the synchronous pipeline."""
    model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    B = rt.BoundaryConditions
    tg = rt.TrackGenerator(model, 32, 5e-3, bcs=B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective))
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    G, nc = 3, tg.mesh.num_cells
    sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
    src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
    psi = np.ones((2, tg.n_total_tracks, G))
    out = []
    for opts in ({"compact": 0}, {"compact": 0, "async": 1}, {"compact": 1, "async": 1}):
        dm, dt = _handles(rt, tg, split=0, **opts)
        for _ in range(2):
            dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dt.sweep_set_links(tg)
        out.append(dt.sweep(G, sig, src, None, psi))
        dt.close(); dm.close()
    for r in out[1:]:
        assert np.array_equal(r["psi_out"], out[0]["psi_out"])
        assert np.allclose(r["phi"], out[0]["phi"], rtol=1e-13, atol=0)


def test_async_sweeps_back_to_back(rt):
    """Under "async" rt_sweep queues its kernels and returns; consecutive sweeps iterate on the device; fetching waits."""
    model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    B = rt.BoundaryConditions
    tg = rt.TrackGenerator(model, 32, 5e-3, bcs=B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Periodic, right=rt.Periodic))
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    G, nc = 5, tg.mesh.num_cells
    sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
    src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
    psi = np.ones((2, tg.n_total_tracks, G))
    out = []
    for a in (0, 1):
        dm, dt = _handles(rt, tg, split=0, compact=0, **{"async": a})
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dt.sweep_set_links(tg)
        dt.sweep(G, sig, src, None, psi, fetch=False)
        for _ in range(4):
            dt.sweep(G, fetch=False)      # four more iterations over the boundary fluxes, queued back to back when async
        out.append(dt.sweep(G))           # (fetching waits)
        dt.close(); dm.close()
    assert np.array_equal(out[0]["psi_out"], out[1]["psi_out"]) and np.array_equal(out[0]["psi_next"], out[1]["psi_next"])
    scale = float(np.abs(out[0]["phi"]).max())
    assert float(np.abs(out[0]["phi"] - out[1]["phi"]).max()) <= 1e-12 * scale
