"""Experiment: march time vs number of resident waves (subsets of C3's tracks in 64-track blocks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

pin = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(pin, 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dm.set_option("split", 0)
n = tg.n_total_tracks
blk = np.arange(n) // 64
for name, sel in [("all", np.ones(n, bool)), ("1/2 blocks", blk % 2 == 0), ("1/4 blocks", blk % 4 == 0), ("1/8 blocks", blk % 8 == 0),
                  ("first half", np.arange(n) < n // 2)]:
    idx = np.nonzero(sel)[0]
    a = {k: np.ascontiguousarray(getattr(tg, k)[idx]) for k in ("px", "py", "phi", "cos_phi", "sin_phi", "A", "B", "C", "ell", "azim_idx")}
    dt = _capi.DeviceTracks(dm, a["px"], a["py"], a["phi"], a["cos_phi"], a["sin_phi"], a["A"], a["B"], a["C"], a["ell"], a["azim_idx"])
    best = None
    for _ in range(6):
        tot = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tm = dt.timing()
        best = tm if best is None or tm["march"] < best["march"] else best
    print(f"{name:12s} tracks={len(idx):7d} waves={(len(idx)+63)//64:5d} segs={tot:9d} march={best['march']*1e3:7.1f} us compact={best['compact']*1e3:7.1f} us", flush=True)
    dt.close()
