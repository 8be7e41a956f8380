"""Experiment: one big batch (config 5) as S uid-contiguous shards on the SAME GPU through rt_multi_* (device_ids = [0]*S):
every shard has its own handle, stream and host thread, so one shard's compaction overlaps another's march.
usage (GPU box): python tools/exp_multi_same_device.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

shards = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]
model = rt.GmshDiscreteModel(rt.data_path("bwr_like.msh"))
tg = rt.TrackGenerator(model, 128, 5e-4)
rt.trace(tg)
aq = tg.azimuthal_quadrature
for S in shards:
    md = _capi.MultiDevice(tg.mesh, [0] * S, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    for _ in range(2):
        total = md.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        md.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    ms = (time.perf_counter() - t0) / K * 1e3
    print("S = %d shards on one GPU: %d segments, %.2f ms per step = %.1f G segments/s" % (S, total, ms, total / ms / 1e6), flush=True)
    md.close()
