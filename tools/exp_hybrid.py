"""Development: the hybrid plan (pieces for the longest waves only) with cheap steps on the whole-track kernel.
usage: python tools/exp_hybrid.py [mesh nφ δ]   -> ms per step and per kernel, per hybrid_pct; records compared with the default plan"""
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

def run(opts):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items(): dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    for _ in range(4): seg()
    t0 = time.perf_counter()
    for _ in range(100): total = seg()
    ms = (time.perf_counter() - t0) / 100 * 1e3
    dm.set_option("timing", 1)
    seg(); seg()
    tm = dt.timing()
    h = hashlib.sha256()
    off, st = dt.fetch_offsets(); recs = dt.fetch_segments(); vol = dt.fetch_volumes()
    for a in (off, st, *[recs[k] for k in ("px", "py", "qx", "qy", "ell", "element")]): h.update(np.ascontiguousarray(a).tobytes())
    stats = dt.stats()
    return ms, tm, h.hexdigest(), vol, total, stats

base = run({})
print("default      ms/step %.4f  march %.4f scan %.4f compact %.4f  total %d" % (base[0], base[1]["march"], base[1]["scan"], base[1]["compact"], base[4]), flush=True)
for pct in (92, 85, 78, 70, 60):
    r = run({"hybrid": 1, "hybrid_pct": pct})
    print("hybrid %2d%%   ms/step %.4f  march %.4f scan %.4f compact %.4f  records equal %s  volumes maxrel %.2e  split-mode %s cheap %s" % (
        pct, r[0], r[1]["march"], r[1]["scan"], r[1]["compact"], r[2] == base[2], float(np.max(np.abs(r[3] - base[3]) / np.abs(base[3]))), r[5].get("split"), r[5].get("cheap_records")), flush=True)
