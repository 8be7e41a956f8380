"""debug: which arrays of a two-phase call differ from the oracle, and at which rows (development)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
orc.build()
na, d = int(sys.argv[1]), float(sys.argv[2])
model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
tg = rt.TrackGenerator(model, na, d); rt.trace(tg); aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dm.set_option("split", 0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); dm.set_option(k, int(v))
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, n_threads=0)
off, st = dt.fetch_offsets(); recs = dt.fetch_segments(); vol = dt.fetch_volumes()
print("total", total, "offsets equal", np.array_equal(off, ref["offsets"]), "status equal", np.array_equal(st, ref["status"]), dt.stats())
row = np.arange(total) - np.repeat(off[:-1], np.diff(off))
for k in ("px", "py", "qx", "qy", "ell", "element"):
    bad = np.nonzero(recs[k] != ref[k])[0]
    print(k, "mismatches", len(bad), "rows mod 32 histogram", np.bincount(row[bad] % 32, minlength=32).tolist() if len(bad) else "")
    if len(bad): print("   first", bad[:5], recs[k][bad[:5]], ref[k][bad[:5]], "rows", row[bad[:5]])
refv = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
print("volumes max rel err", np.abs(vol - refv).max() / refv.max(), vol.sum(), refv.sum())
# which tracks / units are affected
n = tg.n_total_tracks
nw = (n + 63) // 64
wmax = np.array([tg.ell[w * 64:(w + 1) * 64].max() for w in range(nw)])
worder = np.argsort(-wmax, kind="stable")
slot_of = np.empty(n, np.int64)
k = 0
for pos, w in enumerate(worder):
    m = min(64, n - w * 64)
    slot_of[w * 64:w * 64 + m] = pos * 64 + np.arange(m)
bad = np.nonzero(recs["element"] != ref["element"])[0]
tid = np.searchsorted(off, bad, side="right") - 1
ut = np.unique(tid)
print("tracks with bad records:", len(ut), "uids", ut[:40])
print("their counts", np.diff(off)[ut][:40])
sl = slot_of[ut]
print("march slots", sl[:40], "units", np.unique(sl // 16)[:40], "n_units", 4 * nw)
cnts = np.diff(off)
for un in np.unique(sl // 16)[:12]:
    w = worder[un // 4]; q = un % 4
    uu = w * 64 + 16 * q + np.arange(16); uu = uu[uu < n]
    print("unit", un, "wave", w, "q", q, "counts", cnts[uu].tolist(), "bad tracks", np.intersect1d(uu, ut).tolist(), "off0", off[uu[0]])
