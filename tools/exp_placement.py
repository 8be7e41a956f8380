"""Development: where do the march waves run?  (RT_TIMING=1 build: per-wave HW_ID / XCC_ID in the timing dump.)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(model, 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
path = "/tmp/rt_wavedump.bin"; os.environ["RT_TIMING_DUMP"] = path
for _ in range(3):
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
cyc = a[:, 0].astype(np.int64); its = a[:, 1].astype(np.int64)
hw = (a[:, 2] >> np.uint64(16)) & np.uint64(0xffffffff); xcc = (a[:, 2] >> np.uint64(48)) & np.uint64(15)
t0 = (a[:, 3] >> np.uint64(16)).astype(np.int64)
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
wave_slot = hw & np.uint64(15); simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(15)
sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 12) | (sh.astype(np.int64) << 8) | (cu.astype(np.int64) << 4) | simd.astype(np.int64)
groups = collections.defaultdict(list)
for w, k in enumerate(key):
    groups[int(k)].append(w)
sizes = collections.Counter(len(v) for v in groups.values())
print("waves", len(a), "distinct SIMDs used", len(groups), "waves per SIMD histogram", dict(sizes))
print("first 12 waves: (wave, xcc, se, sh, cu, simd, slot)", [(w, int(xcc[w]), int(se[w]), int(sh[w]), int(cu[w]), int(simd[w]), int(wave_slot[w])) for w in range(12)])
# pairing: for each SIMD, the waves that shared it
pairs = sorted(groups.values(), key=lambda v: -sum(its[w] for w in v))[:10]
for v in pairs:
    print("  SIMD waves", v, "its", [int(its[w]) for w in v], "cycles", [int(cyc[w]) for w in v], "start offsets", [int(t0[w] - t0.min()) for w in v])
tot = np.array([sum(its[w] for w in v) for v in groups.values()])
print("sum of its per SIMD: mean %.1f max %d min %d" % (tot.mean(), tot.max(), tot.min()))
end = np.array([max((t0[w] - t0.min()) + cyc[w] for w in v) for v in groups.values()])
print("SIMD end time: mean %.0f max %.0f" % (end.mean(), end.max()))
