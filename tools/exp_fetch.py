"""Development: how long does fetching the results to the host take (rt_fetch_*)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(model, 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
t0 = time.perf_counter()
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
t1 = time.perf_counter()
for _ in range(3):
    a = time.perf_counter(); tot = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2); b = time.perf_counter()
print("tracks_create (H2D of 9.9 MB + plan): %.2f ms; segmentize wall: %.3f ms" % ((t1 - t0) * 1e3, (b - a) * 1e3))
for rep in range(3):
    a = time.perf_counter(); off, st = dt.fetch_offsets(); b = time.perf_counter(); seg = dt.fetch_segments(); c = time.perf_counter()
    print("fetch_offsets %.2f ms, fetch_segments (%.0f MB) %.2f ms = %.1f GB/s" % ((b - a) * 1e3, tot * 44 / 1e6, (c - b) * 1e3, tot * 44 / (c - b) / 1e9)); c2 = time.perf_counter(); pv = dt.fetch_segments_pinned(); c3 = time.perf_counter(); print("   pinned fetch %.2f ms = %.1f GB/s" % ((c3 - c2) * 1e3, tot * 44 / (c3 - c2) / 1e9))
