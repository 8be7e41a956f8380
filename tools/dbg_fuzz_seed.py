"""development: one fuzz case (tools/fuzz_cpu.case(seed)) through both record kernels against the checker — which arrays differ, where"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
import fuzz_cpu
orc.build()
seed = int(sys.argv[1]); topo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind, model, n_azim, delta, k = fuzz_cpu.case(seed)
if n_azim >= 1024: n_azim = 256
tg = rt.TrackGenerator(model, n_azim, delta, tiny_step=1e-8); rt.trace(tg)
if kind == "steep":
    import meshgen; meshgen.steep_tracks(rt, tg, seed)
om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, k=k, iter_cap=4000000, n_threads=0)
aq = tg.azimuthal_quadrature
vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
print(kind, "tracks", tg.n_total_tracks, "segments", int(ref["offsets"][-1]), "failing", int(np.count_nonzero(ref["status"])))
for mk in (1, 0):
    dm = _capi.DeviceMesh(tg.mesh, 0)
    for kk, v in dict(walk=1, split=0, topo=topo, mat_kernel=mk).items(): dm.set_option(kk, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets(); seg = dt.fetch_segments(); v = dt.fetch_volumes()
    print("mat_kernel", mk, "total ok", total == int(ref["offsets"][-1]), "offsets", np.array_equal(off, ref["offsets"]), "status", np.array_equal(st, ref["status"]),
          {f: int((seg[f] != ref[f]).sum()) for f in ("px", "py", "qx", "qy", "ell", "element")}, "volumes max rel", float(np.max(np.abs(v - vol) / np.maximum(np.abs(vol), 1e-300))))
    bs = np.nonzero(st != ref["status"])[0]
    cnt = np.diff(off)
    for u in bs[:6]:
        a, b = off[u], off[u + 1]
        ell = ref["ell"][a:b]
        print("   uid", u, "cnt", cnt[u], "status", st[u], "ref", ref["status"][u], "L", tg.ell[u], "sum", ell.sum(), "L-sum", tg.ell[u] - ell.sum(),
              "chain ok", bool(np.array_equal(ref["px"][a + 1:b], ref["qx"][a:b - 1]) and np.array_equal(ref["py"][a + 1:b], ref["qy"][a:b - 1])))
    if mk == 0:  # (the records that keep their own p: the gaps of k_materialise_lin's Σℓ chain)
        for u in bs[:2]:
            a, b = off[u], off[u + 1]
            rows = np.nonzero((ref["px"][a+1:b] != ref["qx"][a:b-1]) | (ref["py"][a+1:b] != ref["qy"][a:b-1]))[0] + 1
            for r in rows[:6]:
                i = a + r
                print("         row", r, "prev q", repr(ref["qx"][i-1]), repr(ref["qy"][i-1]), "p", repr(ref["px"][i]), repr(ref["py"][i]), "q", repr(ref["qx"][i]), repr(ref["qy"][i]), "ell", ref["ell"][i], "el", ref["element"][i-1], ref["element"][i])
    bv = np.nonzero(np.abs(v - vol) > 1e-10 * np.abs(vol))[0]
    if len(bv): print("   volumes off at cells", bv[:8], v[bv[:4]], vol[bv[:4]])
    dt.close(); dm.close()
