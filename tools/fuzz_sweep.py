"""Many seeded meshes / quadratures / boundary conditions / group counts through rt_sweep on the GPU (both inputs, two consecutive
sweeps each, the second one on the (ℓ, cell) rows the first leaves) against the sequential numpy sweep over the CHECKER's records
(tests/sweep_ref.py) — tolerance 1e-12 of each array's largest value (device expm1 and tally order differ in the last ulps).
Meshes as in tools/fuzz_cpu.py (the classes of tests/meshgen.py); cases above 400 k segments are skipped (the numpy sweep is slow).
usage (GPU box): python tools/fuzz_sweep.py [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
import fuzz_cpu
import sweep_ref

orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = rt.BoundaryConditions
BCS = {"reflective": B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective),
       "periodic": B(top=rt.Periodic, bottom=rt.Periodic, left=rt.Periodic, right=rt.Periodic),
       "vacuum": B(top=rt.Vacuum, bottom=rt.Vacuum, left=rt.Vacuum, right=rt.Vacuum),
       "mixed": B(top=rt.Vacuum, bottom=rt.Reflective, left=rt.Periodic, right=rt.Periodic)}
bad = done = skipped = 0
worst = 0.0
t0 = time.time()
for seed in range(first, first + count):
    kind, model, n_azim, delta, k = fuzz_cpu.case(seed)
    rng = np.random.default_rng(seed * 7919 + 3)
    bc = list(BCS)[seed % 4]
    G = int(rng.integers(1, 9))
    try:
        tg = rt.TrackGenerator(model, n_azim, delta, bcs=BCS[bc])
        rt.trace(tg)
    except Exception as e:  # a quadrature the generator refuses
        skipped += 1
        continue
    if tg.n_total_tracks > 40000:
        skipped += 1
        continue
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, k=k, iter_cap=4000000, n_threads=0)
    if ref["total"] > 400000 or ref["total"] == 0:
        skipped += 1
        continue
    nc = tg.mesh.num_cells
    sigma_t = rng.uniform(0.0 if seed % 5 == 0 else 0.05, 40.0 if seed % 7 == 0 else 3.0, (nc, G))
    if seed % 5 == 0:
        sigma_t[rng.random((nc, G)) < 0.1] = 0.0  # void cells
    source = rng.uniform(0.0, 2.0, (nc, G))
    aq = tg.azimuthal_quadrature
    weight = aq.delta_s[tg.azim_idx - 1] * aq.omega_a[tg.azim_idx - 1]
    psi_in = rng.uniform(0.0, 1.5, (2, tg.n_total_tracks, G))
    links = (tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd)
    phi1, out1 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, psi_in)
    nxt1 = sweep_ref.link(out1, *links)
    phi2, out2 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, nxt1)
    errs = []
    for compact, inp, opts in ((0, "staged", {"split": 0}), (1, "compact", {})):  # (wide k cannot fuse fill_volumes and compacts anyway: whole tracks all the same)
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("compact", compact)
        for name, v in opts.items():
            dm.set_option(name, v)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dt.sweep_set_links(tg)
        r = dt.sweep(G, sigma_t, source, weight, psi_in, input=inp)
        r2 = dt.sweep(G)
        for got, want in ((r["phi"], phi1), (r["psi_out"], out1), (r["psi_next"], nxt1), (r2["phi"], phi2), (r2["psi_out"], out2)):
            errs.append(float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-300))
        dt.close(); dm.close()
    e = max(errs)
    worst = max(worst, e)
    ok = e <= 1e-12
    bad += not ok
    done += 1
    print(f"seed {seed} {kind:11s} {bc:10s} G {G} nφ {n_azim:4d} tracks {tg.n_total_tracks:6d} segs {ref['total']:7d} max rel err {e:.1e}"
          f"{'' if ok else '  MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
print(f"done: {done} cases ({skipped} skipped), {bad} mismatches, worst relative error {worst:.1e}")
sys.exit(1 if bad else 0)
