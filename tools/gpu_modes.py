"""Development A/B: device timings of rt_segmentize under different internal options (C3 by default)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.json"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
opts = [a for a in sys.argv[4:]]  # name=v1,v2,...
model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dm.set_option("timing", 1)  # HIP events between the kernels (off by default)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
names = [o.split("=")[0] for o in opts]
vals = [[int(v) for v in o.split("=")[1].split(",")] for o in opts]
ref_vol = None
for combo in itertools.product(*vals) if opts else [()]:
    for n, v in zip(names, combo):
        dm.set_option(n, v)
    acc = None
    for rep in range(6):
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tm = dt.timing()
        if rep >= 2:
            acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
    acc = {k: round(v / 4, 4) for k, v in acc.items()}
    vol = dt.fetch_volumes()
    if ref_vol is None and vol.sum() > 0:
        ref_vol = vol
    verr = float(np.abs(vol - ref_vol).max()) if ref_vol is not None else -1
    print(dict(zip(names, combo)), "segs", total, acc, "Gseg/s %.2f" % (total / acc["total"] / 1e6), "vol_sum %.12f maxdiff %.2e" % (vol.sum(), verr), flush=True)
