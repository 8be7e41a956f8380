"""Development: a few rt_sweep calls on C3 (for rocprofv3 passes: tools/pmc_cmd.sh <tag> "<counters>" tools/exp_sweep_one.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
tg = rt.TrackGenerator(model, 128, 1e-3, bcs=rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective))
rt.trace(tg); aq = tg.azimuthal_quadrature; nc = tg.mesh.num_cells; G = 7
sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G); src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
dm = _capi.DeviceMesh(tg.mesh, 0); dm.set_option("compact", 0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); dm.set_option(k, int(v))
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2); dt.sweep_set_links(tg)
dt.sweep(G, sig, src, None, np.ones((2, tg.n_total_tracks, G)), input="staged", fetch=False)
for _ in range(3):
    print(dt.sweep(G, input="staged", fetch=False))
