"""Development: one line of rt_sweep timings (C3, staged rows and compact records, G groups) for the library named by
RT_SEGMENTIZE_LIB — used for same-box A/B of sweep kernel variants.   usage: python tools/exp_sweep_ab.py [G [mesh nφ δ]]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
G = int(sys.argv[1]) if len(sys.argv) > 1 else 7
mesh = sys.argv[2] if len(sys.argv) > 2 else "pincell.msh"
na = int(sys.argv[3]) if len(sys.argv) > 3 else 128
d = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
model = rt.GmshDiscreteModel(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d, bcs=rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective))
rt.trace(tg); aq = tg.azimuthal_quadrature
nc = tg.mesh.num_cells
sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G); src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
out = []
for inp in ("staged", "compact"):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("compact", 0 if inp == "staged" else 1)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    dt.sweep_set_links(tg)
    r = dt.sweep(G, sig, src, None, np.ones((2, tg.n_total_tracks, G)), input=inp)
    ms = sorted(dt.sweep(G, input=inp, fetch=False)["ms"] for _ in range(15))
    out.append(f"{inp} {ms[0]:.4f} (median {ms[7]:.4f}) psi_out sha {hashlib.sha256(np.ascontiguousarray(r['psi_out']).tobytes()).hexdigest()[:10]} phi sum {float(r['phi'].sum()):.15e}")
    dt.close(); dm.close()
print(os.path.basename(os.environ.get("RT_SEGMENTIZE_LIB", "in-tree")), "|", " | ".join(out), flush=True)
