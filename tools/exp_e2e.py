"""Development: end-to-end wall times of the drop-in call (mesh handle, track upload, segmentize, fetch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
for mesh, na, d in [("pincell.json", 128, 1e-3), ("bwr_like.msh", 64, 2e-3)]:
    model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
    t0 = time.perf_counter(); tg = rt.TrackGenerator(model, na, d); rt.trace(tg); t1 = time.perf_counter()
    for rep in range(3):
        a = time.perf_counter(); dm = _capi.DeviceMesh(tg.mesh, 0); b = time.perf_counter()
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx); c = time.perf_counter()
        aq = tg.azimuthal_quadrature
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2); e = time.perf_counter()
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2); f = time.perf_counter()
        s = dt.fetch_segments_pinned(); g = time.perf_counter()
        print(f"{mesh}: trace {1e3*(t1-t0):.1f} ms | rt_mesh_create {1e3*(b-a):.1f} ms | rt_tracks_create {1e3*(c-b):.1f} ms | first segmentize {1e3*(e-c):.1f} ms | second {1e3*(f-e):.2f} ms | pinned fetch {1e3*(g-f):.1f} ms", flush=True)
        dt.close(); dm.close()
