"""Development check of the cheap steps (option "topo"): every output of a call with topo=1 against topo=0 — offsets, status
and the six record arrays bit for bit, volumes to 1e-12 — plus device timings of both.
usage (GPU box): python tools/topo_check.py [mesh nφ δ [name=value ...]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
extra = [a.split("=") for a in sys.argv[4:]]
if mesh.startswith("fuzz:"):  # fuzz:<seed> — the mesh of tools/fuzz_cpu.py's case <seed>
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_cpu
    kind, model, _, _, _ = fuzz_cpu.case(int(mesh[5:]))
    print("mesh", kind, model.num_cells, "cells")
else:
    model = rt.GmshDiscreteModel(rt.data_path(mesh)) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d)
rt.trace(tg)
aq = tg.azimuthal_quadrature
res = {}
for topo in (0, 1):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("topo", topo)
    dm.set_option("timing", 1)
    for n, v in extra:
        dm.set_option(n, int(v))
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    acc = None
    for rep in range(7):
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tm = dt.timing()
        if rep >= 2:
            acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
    acc = {k: round(v / 5, 4) for k, v in acc.items()}
    off, st = dt.fetch_offsets()
    seg = dt.fetch_segments()
    vol = dt.fetch_volumes()
    res[topo] = (off, st, seg, vol)
    info = dm.info()
    print("topo", topo, "segs", total, acc, "stats", dt.stats(), "failed", int(np.count_nonzero(st)),
          "records walk/cheap %d/%d" % (info["records_walk"], info["records_cheap"]), flush=True)
a, b = res[0], res[1]
ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
for k in ("px", "py", "qx", "qy", "ell", "element"):
    same = np.array_equal(a[2][k], b[2][k])
    ok = ok and same
    if not same:
        bad = np.flatnonzero(a[2][k] != b[2][k])
        print("MISMATCH", k, len(bad), bad[:8])
verr = float(np.abs(a[3] - b[3]).max() / max(np.abs(a[3]).max(), 1e-300))
print("volumes rel maxdiff %.2e" % verr)
print("EQUAL" if ok and verr < 1e-12 else "MISMATCH")
sys.exit(0 if ok and verr < 1e-12 else 1)
