"""Development: rt_segmentize with and without k_first (option "first") on one configuration; prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
mesh = sys.argv[1] if len(sys.argv) > 1 else "bwr_like.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-3
model = rt.GmshDiscreteModel(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg); aq = tg.azimuthal_quadrature
res = {}
for first in (0, 1):
    dm = _capi.DeviceMesh(tg.mesh, 0); dm.set_option("first", first)
    for kv in sys.argv[4:]:
        k, v = kv.split("="); dm.set_option(k, int(v))
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets(); seg = dt.fetch_segments()
    res[first] = (total, off.copy(), st.copy(), seg, dt.stats())
    print("first", first, "total", total, "failing", int(np.count_nonzero(st)), dt.stats()["march_waves"], flush=True)
a, b = res[0], res[1]
bad = np.nonzero((a[2] != b[2]) | (np.diff(a[1]) != np.diff(b[1])))[0]
print("tracks that differ:", len(bad), bad[:20], bad[-5:] if len(bad) else "")
for u in bad[:6]:
    print("uid", u + 1, "azim", tg.azim_idx[u], "p", tg.px[u], tg.py[u], "q", tg.qx[u], tg.qy[u], "phi", tg.phi[u], "ell", tg.ell[u],
          "| counts", np.diff(a[1])[u], np.diff(b[1])[u], "status", a[2][u], b[2][u])
    for r, nm in ((a, "first=0"), (b, "first=1")):
        s0, s1 = r[1][u], r[1][u + 1]
        print("   ", nm, [(int(r[3]["element"][s]), float(r[3]["px"][s]), float(r[3]["py"][s]), float(r[3]["qx"][s]), float(r[3]["qy"][s])) for s in range(s0, min(s1, s0 + 3))])
