"""Development: rt_sweep timings (C3 by default) for both inputs and the group-slab / workgroup-shape options.
usage (GPU box): python tools/sweep_time.py [mesh nφ δ G]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
G = int(sys.argv[4]) if len(sys.argv) > 4 else 7
model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d, bcs=rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective))
rt.trace(tg)
aq = tg.azimuthal_quadrature
nc = tg.mesh.num_cells
sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
ref = None
for inp in ("staged", "compact"):
    for gp in (0, 2, 1):
        for waves in (0, 8, 4):
            dm = _capi.DeviceMesh(tg.mesh, 0)
            dm.set_option("compact", 0 if inp == "staged" else 1)
            dm.set_option("sweep_gp", gp); dm.set_option("sweep_waves", waves)
            dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
            total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
            dt.sweep_set_links(tg)
            r = dt.sweep(G, sig, src, None, np.ones((2, tg.n_total_tracks, G)), input=inp)
            if ref is None:
                ref = r
            err = max(float(np.abs(r[k] - ref[k]).max() / np.abs(ref[k]).max()) for k in ("phi", "psi_out"))
            ms = sorted(dt.sweep(G, input=inp, fetch=False)["ms"] for _ in range(7))
            print(f"{inp:8s} gp={r['groups_per_pass']} passes={r['passes']} waves={waves or 'auto'}: {ms[0]:.3f} ms (median {ms[3]:.3f}) "
                  f"= {total * 2 * G / ms[0] / 1e6:.1f} G updates/s, vs first variant {err:.1e}", flush=True)
            dt.close(); dm.close()
