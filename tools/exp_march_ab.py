"""Development: one line of rt_segmentize timings (ms per step of plain calls, per-kernel HIP-event times) + a hash of all
results for the library named by RT_SEGMENTIZE_LIB — same-box A/B of march kernel variants.
usage: python tools/exp_march_ab.py [mesh nφ δ [name=value ...]]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
na = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
model = rt.GmshDiscreteModel(rt.data_path(mesh))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg); aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); dm.set_option(k, int(v))
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
for _ in range(4): seg()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(40): total = seg()
    best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
dm.set_option("timing", 1)
tm = []
for _ in range(9):
    seg(); tm.append(dt.timing())
med = lambda k: sorted(t[k] for t in tm)[len(tm) // 2]
h = hashlib.sha256()
off, st = dt.fetch_offsets(); vol = dt.fetch_volumes()
recs = dt.fetch_segments() if not os.environ.get("AB_NOHASH") else {}  # (AB_NOHASH=1: batches of gigabytes — offsets and status only)
for a in (off, st, *[recs[k] for k in ("px", "py", "qx", "qy", "ell", "element") if k in recs]): h.update(np.ascontiguousarray(a).tobytes())
print(os.path.basename(os.environ.get("RT_SEGMENTIZE_LIB", "in-tree")), f"| {mesh} {na} {d}: {total} segments, {best:.4f} ms/step, march {med('march'):.4f} scan {med('scan'):.4f} "
      f"compact {med('compact'):.4f} | records sha {h.hexdigest()[:12]} volumes sum {float(vol.sum()):.15e} cheap {dt.stats()['cheap_records']}", flush=True)
