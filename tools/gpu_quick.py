"""Quick GPU check used during development: parity + timings at C1..C3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc

pin = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
bwr = rt.GmshDiscreteModel(rt.data_path("bwr_like.msh"))
for model, na, d in [(pin, 8, 2e-2), (pin, 32, 5e-3), (pin, 128, 1e-3), (bwr, 16, 0.02), (bwr, 64, 2e-3)]:
    tg = rt.TrackGenerator(model, na, d); rt.trace(tg)
    rt.segmentize(tg, walk=False, check=False)
    for _ in range(2):
        tg.device_tracks.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, tg.azimuthal_quadrature.delta_s, tg.azimuthal_quadrature.n_azim_2)
    print("generic-only:", tg.device_tracks.timing())
    t0 = time.time(); rt.segmentize(tg, check=False); t1 = time.time()
    tm = tg.device_tracks.timing()
    for _ in range(3):
        tg.device_tracks.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, tg.azimuthal_quadrature.delta_s, tg.azimuthal_quadrature.n_azim_2)
    tm = tg.device_tracks.timing()
    n = len(tg.segments)
    print(f"nphi={na} delta={d}: tracks={tg.n_total_tracks} segs={n} wall={t1-t0:.3f}s dev={tm} -> {n/tm['total']/1e3:.1f} Mseg/s", flush=True)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    t0 = time.time()
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, n_threads=0)
    t1 = time.time()
    s = tg.segments
    ok_off = np.array_equal(s.offsets, ref["offsets"])
    ok_el = ok_off and np.array_equal(s.element, ref["element"])
    bit = ok_off and all(np.array_equal(getattr(s, k), ref[k]) for k in ("px", "py", "qx", "qy", "ell"))
    print(f"   oracle {t1-t0:.2f}s ({orc.num_threads()} thr)  offsets_equal={ok_off} elements_equal={ok_el} coords_bitwise={bit}", flush=True)
    if ok_off and not bit:
        for k in ("px", "py", "qx", "qy", "ell"):
            e = np.abs(getattr(s, k) - ref[k]); print("    ", k, "max abs err", e.max(), "n diff", (e > 0).sum())
