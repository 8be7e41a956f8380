"""Two track sets segmentized from two host threads on two streams (development; the probe behind DESIGN.md §9 "record-writing in the
march's shadow"): the march of one call overlaps the record kernel of the other — what a march-fused record phase could overlap at
best.  Prints ms per step alone and overlapped; under `rocprofv3 --kernel-trace --stats` the kernels' durations WHILE overlapped.
usage: python tools/two_in_flight.py [mesh nazim delta] [steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
nazim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
delta = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
mode = sys.argv[5] if len(sys.argv) > 5 else "both"
path = rt.data_path(mesh)
model = rt.GmshDiscreteModel(path) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
tg = rt.TrackGenerator(model, nazim, delta); rt.trace(tg)
aq = tg.azimuthal_quadrature
hs = []
for _ in range(2):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    hs.append((dm, dt))
seg = lambda dt: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
for dm, dt in hs:
    for _ in range(3): total = seg(dt)
if mode in ("both", "alone"):
    t0 = time.perf_counter()
    for _ in range(steps): seg(hs[0][1])
    alone = (time.perf_counter() - t0) / steps * 1e3
    print(f"{mesh} {nazim} {delta}: {total} segments; one handle alone {alone:.4f} ms per step", flush=True)
if mode in ("both", "overlap"):
    def run(dt):
        for _ in range(steps): seg(dt)
    th = [threading.Thread(target=run, args=(dt,)) for _, dt in hs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    both = (time.perf_counter() - t0) / (2 * steps) * 1e3
    print(f"{mesh} {nazim} {delta}: two handles on two streams {both:.4f} ms per step", flush=True)
