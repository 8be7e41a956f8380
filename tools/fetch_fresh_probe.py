"""development: rt_fetch_segments into fresh arrays, several times in a row (round 5, the one-shot fetch): does the time depend on
whether earlier results were freed, on the library's huge-page hint (RT_FETCH_NO_HUGEPAGE_HINT=1: off) and on the allocator's own (NUMPY_MADVISE_HUGEPAGE=0)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
tg = rt.TrackGenerator(rt.GmshDiscreteModel(rt.data_path("pincell.msh")), 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
held, out = [], []
for i in range(6):
    t0 = time.perf_counter(); r = dt.fetch_segments(); out.append((time.perf_counter() - t0) * 1e3)
    if keep: held.append(r)
    del r
thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip() + " defrag " + open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip()
print(f"huge-page hint {'off' if os.environ.get('RT_FETCH_NO_HUGEPAGE_HINT') else 'on'} numpy-hugepage {os.environ.get('NUMPY_MADVISE_HUGEPAGE', 'default')} "
      f"{'results kept' if keep else 'results freed'}: " + " ".join(f"{x:.1f}" for x in out) + f" ms  ({total * 44 / 1e6:.0f} MB; THP {thp})", flush=True)
