"""rt_fill_tau at C3: kernel time (HIP events) by group count, and the result against ℓ·Σt on the host (development)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
refl = rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)
tg = rt.TrackGenerator(model, 128, 1e-3, bcs=refl)
rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
recs = dt.fetch_segments()
nc = tg.mesh.num_cells
for G in (1, 2, 3, 7, 8, 16):
    sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
    for rep in (0, 1):
        tau, _, _ = dt.fill_tau(sig)
        ok = np.array_equal(tau, sig[recs["element"] - 1] * recs["ell"][:, None])
        ms = sorted(dt.fill_tau(sig, fetch=False)[2] for _ in range(15))
        gb = total * (12 + 8 * G) / 1e9
        print(f"G={G}: bit-equal {ok}, best {ms[0]:.4f} ms median {ms[7]:.4f} ms = {gb / ms[7] * 1e3:.0f} GB/s ({gb / ms[7] * 1e3 / 8000:.3f} of 8 TB/s)", flush=True)
