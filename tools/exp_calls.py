"""Development: a few plain rt_segmentize calls (default options, no HIP events between the kernels) for rocprofv3 traces.
usage: python tools/exp_calls.py [mesh nφ δ [name=value ...]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
model = rt.GmshDiscreteModel(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg); aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); dm.set_option(k, int(v))
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
import time
for _ in range(4):
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
t0 = time.perf_counter()
for _ in range(int(os.environ.get("CALLS", "100"))):
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
print("segments", total, "ms per step %.4f" % ((time.perf_counter() - t0) / int(os.environ.get("CALLS", "100")) * 1e3))
