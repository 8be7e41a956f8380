"""development: what the first and later fetches of a handle's records cost (page-locked buffers owned by the handle vs the caller's pageable arrays)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
for mesh, na, delta in (("pincell.msh", 128, 1e-3), ("bwr_like.msh", 128, 5e-4)):
    tg = rt.TrackGenerator(rt.GmshDiscreteModel(rt.data_path(mesh)), na, delta); rt.trace(tg)
    aq = tg.azimuthal_quadrature
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for rep in range(2):
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        out = []
        for name, f in (("fetch_pinned (1st)", lambda: dt.fetch_pinned()), ("fetch_pinned (2nd)", lambda: dt.fetch_pinned()),
                        ("fetch_segments (1st, fresh arrays)", lambda: dt.fetch_segments()), ("fetch_segments (2nd)", lambda: dt.fetch_segments())):
            t0 = time.perf_counter(); r = f(); out.append(f"{name} {(time.perf_counter() - t0) * 1e3:.1f} ms")
        print(f"{mesh} handle {rep}: {total} records, {total * 44 / 1e6:.0f} MB | " + " | ".join(out), flush=True)
        dt.close()
    dm.close()
