"""development: host-side breakdown of rt_tracks_create (RT_CREATE_TIMING=1 makes the library print it) at C3 and C5"""
import os, sys, time
os.environ["RT_CREATE_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
for mesh, na, delta in (("pincell.msh", 128, 1e-3), ("bwr_like.msh", 128, 5e-4)):
    tg = rt.TrackGenerator(rt.GmshDiscreteModel(rt.data_path(mesh)), na, delta); rt.trace(tg)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for rep in range(4):
        t0 = time.perf_counter()
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        print(f"{mesh} rep {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms (python clock)", flush=True)
        dt.close()
    dm.close()
