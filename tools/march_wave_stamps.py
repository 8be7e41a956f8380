"""Development: per-wave stamps of k_march from a -DRT_TIMING build of the library (RT_SEGMENTIZE_LIB=build_ab/librt_timing.so):
start and duration of every march wave (s_memtime, 100 MHz), its iterations and the CU / SIMD it ran on (HW_REG_HW_ID, XCC_ID).

  RT_SEGMENTIZE_LIB=... python tools/march_wave_stamps.py [--nazim 128] [--delta 1e-3] [name=value ...]"""
import argparse, os, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--nazim", type=int, default=128)
ap.add_argument("--delta", type=float, default=1e-3)
ap.add_argument("opts", nargs="*")
a = ap.parse_args()
dump = os.path.join(tempfile.gettempdir(), "rt_timing_dump.bin")
os.environ["RT_TIMING_DUMP"] = dump
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
refl = rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)
tg = rt.TrackGenerator(model, a.nazim, a.delta, bcs=refl)
rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for kv in a.opts:
    k, v = kv.split("=")
    dm.set_option(k, int(v))
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
for _ in range(4):
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
h = np.fromfile(dump, dtype=np.uint64).reshape(-1, 4)
dur = h[:, 0].astype(np.int64)
its = h[:, 1].astype(np.int64)
hw = (h[:, 2] >> np.uint64(16)) & np.uint64(0xffffffff)
xcc = (h[:, 2] >> np.uint64(48)) & np.uint64(15)
tstart = (h[:, 3] >> np.uint64(16)).astype(np.int64)
ok = dur > 0
off, _ = dt.fetch_offsets()
cnt = np.diff(off)
perm = np.asarray(dt.march_order()) if hasattr(dt, "march_order") else None
nw = len(dur)
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = (hw >> np.uint64(8)) & np.uint64(15)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
xcc = xcc.astype(np.int64)
place = ((((xcc * 8 + se.astype(np.int64)) * 2 + sh.astype(np.int64)) * 16 + cu.astype(np.int64)) * 4 + simd.astype(np.int64))
start = np.zeros(nw, dtype=np.int64)
# (every XCD has its own counter, and the XCC_ID field does not separate them on every box: cluster the raw stamps instead —
#  counters of different XCDs are millions of cycles apart, a kernel lasts a few hundred thousand)
order_t = np.argsort(tstart)
cl = np.zeros(nw, dtype=np.int64)
c = 0
for a_, b_ in zip(order_t[:-1], order_t[1:]):
    if tstart[b_] - tstart[a_] > 2_000_000: c += 1
    cl[b_] = c
for x in range(c + 1):
    sel = ok & (cl == x)
    if sel.any():
        start[sel] = tstart[sel] - tstart[sel].min()
print(f"{c + 1} counter clusters (XCDs)")
end = start + dur
print(f"{ok.sum()} waves (march order: longest first); cycles; kernel span (per-XCD clocks, start of the XCD's first wave = 0) {end[ok].max()}")
print("start stamps: percentiles 0/10/25/50/75/90/99/100:", np.percentile(start[ok], [0, 10, 25, 50, 75, 90, 99, 100]).astype(int))
print("end stamps:   percentiles 0/50/90/99/100:", np.percentile(end[ok], [0, 50, 90, 99, 100]).astype(int))
cuid = place // 4
print("wave  start    end    dur   place(cu,simd)  same-SIMD partners (wave,start,end) | waves on the CU")
for w in list(range(0, 8)) + list(range(nw // 4, nw // 4 + 3)) + list(range(nw // 2, nw // 2 + 3)) + list(range(nw - 3, nw)):
    mates = [v for v in np.nonzero(place == place[w])[0] if v != w]
    ms = " ".join(f"({v},{start[v]},{end[v]})" for v in mates[:4])
    print(f"{w:5d} {start[w]:6d} {end[w]:7d} {dur[w]:7d}  ({int(cuid[w])},{int(place[w]) % 4})  {ms} | {int(np.sum(cuid == cuid[w]))}")
for lo, hi in ((0, nw // 8), (nw // 8, nw // 4), (nw // 4, nw // 2), (nw // 2, nw)):
    sel = np.arange(lo, hi)
    print(f"waves [{lo},{hi}): mean duration {dur[sel].mean():9.0f}, mean start {start[sel].mean():8.0f}, mean end {end[sel].mean():9.0f}")
