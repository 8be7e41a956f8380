"""Development: per-wave march statistics from an RT_TIMING=1 build (RT_TIMING_DUMP=<file>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh, na, d = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
path = "/tmp/rt_wavedump.bin"
os.environ["RT_TIMING_DUMP"] = path
for _ in range(3):
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
print(dt.timing())
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
cyc, its, gen, em = a.T
print("waves", len(a), "cycles: mean %.0f max %.0f" % (cyc.mean(), cyc.max()), "its: mean %.1f max %.0f" % (its.mean(), its.max()),
      "generic-its: mean %.2f max %.0f" % (gen.mean(), gen.max()))
order = np.argsort(-cyc)[:12]
for w in order:
    print("  wave %5d cycles %9.0f its %4.0f generic-its %4.0f lane0-emits %4.0f  cycles/it %.0f" % (w, cyc[w], its[w], gen[w], em[w], cyc[w] / max(its[w], 1)))
print("cycles/it overall: %.0f ; for waves with generic-its<=1: %.0f" % (cyc.sum() / its.sum(), cyc[gen <= 1].sum() / its[gen <= 1].sum()))
