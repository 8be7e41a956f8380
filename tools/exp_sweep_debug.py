"""Development: what the sweep's time is made of (option "sweep_debug": 1 no tallies, 2 no DPP pre-reduction) and the tally variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
model = rt.GmshDiscreteModel(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d, bcs=rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective))
rt.trace(tg); aq = tg.azimuthal_quadrature; nc = tg.mesh.num_cells; G = 7
sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G); src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
ref = None
for gp, dbg, waves in ((0, 0, 16), (0, 0, 8), (0, 2, 16), (0, 1, 16), (1, 0, 16), (2, 0, 16), (8, 0, 16)):  # debug: 2 = no pre-reduction of the tallies, 1 = no tallies; gp 8 = tallies straight to HBM
    dm = _capi.DeviceMesh(tg.mesh, 0); dm.set_option("compact", 0); dm.set_option("sweep_debug", dbg); dm.set_option("sweep_waves", waves); dm.set_option("sweep_gp", gp)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2); dt.sweep_set_links(tg)
    r = dt.sweep(G, sig, src, None, np.ones((2, tg.n_total_tracks, G)), input="staged")
    if ref is None:
        ref = r
    err = float(np.abs(r["phi"] - ref["phi"]).max() / np.abs(ref["phi"]).max())
    ms = sorted(dt.sweep(G, input="staged", fetch=False)["ms"] for _ in range(7))
    print(f"gp {r['groups_per_pass']} (opt {gp}) passes {r['passes']} debug {dbg} waves {waves}: {ms[0]:.3f} ms (median {ms[3]:.3f}), phi vs first {err:.1e}", flush=True)
