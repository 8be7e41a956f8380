"""Probe (development; results are void): what would the march cost if every track were cut into P pieces?  The track set is replaced
by P x as many tracks — piece k of a track starts at p + (k/P)·ℓ·(cos ϕ, sin ϕ) — and the library is a -DRT_STOP_AFTER=N build in
which every lane stops after N records (N ≈ the longest track's record count / P): 2,039·P waves with chains of N, every piece's
first record by the generic step — the march of "pieces with cheap steps" without its bookkeeping.  Prints the march's HIP-event time.
usage: RT_SEGMENTIZE_LIB=build_ab/lib_stopN.so python tools/probe_pieces.py P [mesh nazim delta]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mesh = sys.argv[2] if len(sys.argv) > 2 else "pincell.msh"
nazim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
delta = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
path = rt.data_path(mesh)
model = rt.GmshDiscreteModel(path) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
tg = rt.TrackGenerator(model, nazim, delta); rt.trace(tg)
aq = tg.azimuthal_quadrature
rep = lambda a: np.repeat(np.ascontiguousarray(a), P)
k = np.tile(np.arange(P), len(tg.ell)).astype(np.float64)
ell = rep(tg.ell)
px = rep(tg.px) + (k / P * ell) * rep(tg.cos_phi)
py = rep(tg.py) + (k / P * ell) * rep(tg.sin_phi)
dm = _capi.DeviceMesh(tg.mesh, 0)
dm.set_option("split", 0)
dt = _capi.DeviceTracks(dm, px, py, rep(tg.phi), rep(tg.cos_phi), rep(tg.sin_phi), rep(tg.A), rep(tg.B), rep(tg.C), ell / P, rep(tg.azim_idx))
seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
for _ in range(4): total = seg()
dm.set_option("timing", 1)
tm = []
for _ in range(9):
    seg(); tm.append(dt.timing())
med = lambda key: sorted(t[key] for t in tm)[len(tm) // 2]
print(f"{os.path.basename(os.environ.get('RT_SEGMENTIZE_LIB', 'in-tree'))} | {mesh} {nazim} {delta} P={P}: {len(ell)} lanes = {(len(ell) + 63) // 64} waves, {total} records (void), "
      f"march {med('march'):.4f} ms, scan {med('scan'):.4f}, record kernel {med('compact'):.4f}", flush=True)
