"""Development: two independent batches in flight (two handles, two streams, two host threads): does the
latency-bound march of one overlap the bandwidth-bound compaction of the other?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(model, 128, 1e-3); rt.trace(tg)
aq = tg.azimuthal_quadrature
K = 40
def make():
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    for _ in range(3):
        tot = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    return dm, dt, tot
hs = [make() for _ in range(3)]
tot = hs[0][2]
def run(dt, n):
    for _ in range(n):
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
for nthreads in (1, 2, 3, 2, 1):
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(hs[i][1], K // nthreads)) for i in range(nthreads)]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    n = (K // nthreads) * nthreads
    print(f"{nthreads} batch(es) in flight: {el / n * 1e3:.4f} ms per step, {tot * n / el / 1e9:.2f} G segments/s", flush=True)
