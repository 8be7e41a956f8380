"""Experiment: all walk records of the mesh in LDS (option lds_records=1) against the same workgroup shape
fetching them from L2 (=2) and the default kernel (=0), on a mesh small enough to fit (a jittered 14x14 lattice).
usage (GPU box): python tools/exp_lds_records.py [mode ...]   (RT_OPTIONS is not needed; modes default to 0 2 1)"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
import meshgen

modes = [int(a) for a in sys.argv[1:]] or [0, 2, 1]
model = meshgen.lattice_model(rt, 7, 14, 14, jitter=0.25)
tg = rt.TrackGenerator(model, 128, 6e-4)
rt.trace(tg)
aq = tg.azimuthal_quadrature
print("mesh: %d cells (%d KB of walk records), %d tracks" % (model.num_cells, model.num_cells * 240 // 1024, tg.n_total_tracks))
ref = None
for rnd in range(2):
    for mode in modes:
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("lds_records", mode)
        dm.set_option("split", 0)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        acc = None
        for rep in range(8):
            total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
            tm = dt.timing()
            if rep >= 3:
                acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
        seg = dt.fetch_segments()
        sig = (total, zlib.crc32(seg["element"].tobytes()), zlib.crc32(seg["qx"].tobytes()), zlib.crc32(seg["ell"].tobytes()))
        ref = ref or sig
        print({0: "default (4-wave workgroups, L2)", 1: "8-wave workgroups, records in LDS", 2: "8-wave workgroups, records from L2"}[mode],
              "| segments", total, "| march %.1f us, step %.1f us" % (acc["march"] / 5 * 1e3, acc["total"] / 5 * 1e3),
              "| waves/workgroup", dt.stats()["march_waves"], "| identical records:", sig == ref, flush=True)
        dt.close()
        dm.close()
