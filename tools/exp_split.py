"""Experiment: device timings vs the `split` option (pieces of about `split` expected segments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.json"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
splits = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,200,140,100,80,64,48").split(",")]
model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for sp in splits:
    dm.set_option("split", sp)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    acc = None
    for rep in range(6):
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tm = dt.timing()
        if rep >= 2:
            acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
    acc = {k: round(v / 4 * 1e3, 1) for k, v in acc.items()}
    print("split", sp, "segs", total, acc, flush=True)
    dt.close()
