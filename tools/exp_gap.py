"""Development: wall time per call against the device time of its kernels (what the host adds between calls)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(model, 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
for _ in range(5):
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
for rep in range(3):
    dev = 0.0
    t0 = time.perf_counter()
    for _ in range(50):
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dev += dt.timing()["total"]
    el = (time.perf_counter() - t0) / 50 * 1e3
    print("wall %.4f ms/call, device %.4f, host adds %.1f us" % (el, dev / 50, (el - dev / 50) * 1e3))
