"""development: deviation of the fused volumes (cheap records tallied from the vertices' distances) from the oracle's, per mesh class"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
import test_gpu_walk_regime as W
orc.build()
for kind, seed in W.CASES:
    model, delta_scale = W._make(rt, kind, seed)
    tg = rt.TrackGenerator(model, (8, 16, 32, 4, 64)[seed % 5], 0.008 * delta_scale)
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, iter_cap=4000000, n_threads=0)
    refv = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    dm = _capi.DeviceMesh(tg.mesh, 0); dm.set_option("split", 0); dm.set_option("topo", 2)
    if os.environ.get("VOLDEV_EPS"): dm.set_option("test_tally_tau", int(os.environ["VOLDEV_EPS"]))
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    v = dt.fetch_volumes(); st = dt.stats()
    nz = refv > 0
    rel = np.abs(v[nz] - refv[nz]) / refv[nz]
    xy = tg.mesh.node_coordinates if hasattr(tg.mesh, "node_coordinates") else None
    print(f"{kind:12s} {seed}: cells {len(refv)} records {st['records']} cheap {st['cheap_records']} from lengths {st['records_tallied_from_lengths']} max rel dev {rel.max():.2e} median {np.median(rel):.1e} "
          f"#>1e-10 {int((rel > 1e-10).sum())} #>1e-12 {int((rel > 1e-12).sum())} min cell vol {refv[nz].min():.2e}", flush=True)
    dt.close(); dm.close()
