import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
import fuzz_cpu
orc.build()
kind, model, n_azim, delta, k = fuzz_cpu.case(7444)
n_azim = 256 if n_azim >= 1024 else n_azim
tg = rt.TrackGenerator(model, n_azim, delta); rt.trace(tg)
om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, k=k, iter_cap=4000000, n_threads=0)
print(kind, n_azim, delta, k, "status histogram (oracle):", np.bincount(ref["status"]))
aq = tg.azimuthal_quadrature
for opts in (dict(split=0), dict()):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for kk, v in opts.items(): dm.set_option(kk, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets(); seg = dt.fetch_segments()
    print(opts, "total", total, ref["total"], "status hist", np.bincount(st), "split", dt.stats()["split"])
    bad = np.nonzero(st != ref["status"])[0]
    print("  status differs on tracks", bad[:10], "gpu", st[bad[:10]], "ref", ref["status"][bad[:10]])
    cnt = np.diff(off); rc = np.diff(ref["offsets"])
    b2 = np.nonzero(cnt != rc)[0]
    print("  counts differ on", b2[:10], cnt[b2[:10]], rc[b2[:10]])
    vol = dt.fetch_volumes()
    rv = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    print("  volumes close:", np.allclose(vol, rv, rtol=1e-10, atol=1e-300), np.abs(vol-rv).max())
    if total == ref["total"]:
        for f in ("element","px","py","qx","qy","ell"):
            print("   ", f, np.array_equal(seg[f], ref[f]))
