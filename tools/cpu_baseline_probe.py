"""cpu_baseline of bench.py by thread count on one box (development): the box's CPU share (cgroup cpu.max) against all hardware threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import raytracing_jl_amd as rt
from oracle import oracle as orc
orc.build()
tg = bench.make_tg(rt, bench.WORKLOADS["c3"])
om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", "omp threads", orc.num_threads(), "cpu_count", os.cpu_count())
for nt in (0, 16, 24, 32, 64, 16, 0):
    t0 = time.perf_counter()
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, tiny_step=tg.tiny_step, n_threads=nt)
    dt = time.perf_counter() - t0
    print("n_threads %3d: %.3f s, %.1f M segments/s" % (nt, dt, r["total"] / dt / 1e6), flush=True)
    time.sleep(0.5)
