"""Development: compact_mode 3 (lean staging) must reproduce compact_mode 2 bit for bit; timings of both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

for mesh, na, d, split in [("pincell.json", 8, 2e-2, -1), ("pincell.json", 32, 5e-3, -1), ("pincell.json", 128, 1e-3, -1), ("bwr_like.msh", 64, 2e-3, -1),
                           ("pincell.json", 32, 5e-3, 0), ("pincell.json", 32, 5e-3, 20)]:
    model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
    tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
    aq = tg.azimuthal_quadrature
    res = {}
    for mode in (2, 3):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("split", split); dm.set_option("compact_mode", mode)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        for _ in range(4):
            tot = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tm = dt.timing()
        off, st = dt.fetch_offsets(); seg = dt.fetch_segments(); vol = dt.fetch_volumes()
        res[mode] = (off, st, seg, vol, tm, tot)
        dt.close()
    a, b = res[2], res[3]
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and all(np.array_equal(a[2][k], b[2][k]) for k in a[2]) and np.array_equal(a[3], b[3])
    print(mesh, na, d, "split", split, "segs", a[5], "identical:", ok,
          "| mode2 march %.3f compact %.3f total %.3f | mode3 march %.3f compact %.3f total %.3f" % (a[4]["march"], a[4]["compact"], a[4]["total"], b[4]["march"], b[4]["compact"], b[4]["total"]), flush=True)
    if not ok:
        for name in a[2]:
            x, y = a[2][name], b[2][name]
            bad = np.nonzero(x != y)[0]
            print("   ", name, "mismatches", len(bad), bad[:5], x[bad[:3]], y[bad[:3]])
