"""Development: timings for options that are read at rt_tracks_create (name=v1,v2,... ; fresh DeviceTracks per value)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
mesh, na, d = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
opts = sys.argv[4:]
names = [o.split("=")[0] for o in opts]
vals = [[int(v) for v in o.split("=")[1].split(",")] for o in opts]
model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for rnd in range(2):
    for combo in itertools.product(*vals):
        for n, v in zip(names, combo):
            dm.set_option(n, v)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        acc = None
        for rep in range(7):
            total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
            tm = dt.timing()
            if rep >= 2:
                acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
        print(dict(zip(names, combo)), "segs", total, {k: round(v / 5 * 1e3, 1) for k, v in acc.items()}, flush=True)
        dt.close()
