"""Development: one seed of tools/fuzz_sweep.py with the error of every array and leg, the track statuses and where the largest
deviation sits.   usage (GPU box): python tools/exp_fuzz_sweep_one.py seed"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
import fuzz_cpu, sweep_ref
orc.build()
seed = int(sys.argv[1])
kind, model, n_azim, delta, k = fuzz_cpu.case(seed)
rng = np.random.default_rng(seed * 7919 + 3)
B = rt.BoundaryConditions
BCS = {"reflective": B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective),
       "periodic": B(top=rt.Periodic, bottom=rt.Periodic, left=rt.Periodic, right=rt.Periodic),
       "vacuum": B(top=rt.Vacuum, bottom=rt.Vacuum, left=rt.Vacuum, right=rt.Vacuum),
       "mixed": B(top=rt.Vacuum, bottom=rt.Reflective, left=rt.Periodic, right=rt.Periodic)}
bc = list(BCS)[seed % 4]
G = int(rng.integers(1, 9))
tg = rt.TrackGenerator(model, n_azim, delta, bcs=BCS[bc]); rt.trace(tg)
om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, k=k, iter_cap=4000000, n_threads=0)
print(kind, bc, "G", G, "k", k, "tracks", tg.n_total_tracks, "segs", ref["total"], "statuses", np.unique(ref["status"], return_counts=True))
cnt = np.diff(ref["offsets"]); print("tracks without records:", int((cnt == 0).sum()), "max records", int(cnt.max()))
nc = tg.mesh.num_cells
sigma_t = rng.uniform(0.0 if seed % 5 == 0 else 0.05, 40.0 if seed % 7 == 0 else 3.0, (nc, G))
if seed % 5 == 0: sigma_t[rng.random((nc, G)) < 0.1] = 0.0
source = rng.uniform(0.0, 2.0, (nc, G))
aq = tg.azimuthal_quadrature
weight = aq.delta_s[tg.azim_idx - 1] * aq.omega_a[tg.azim_idx - 1]
psi_in = rng.uniform(0.0, 1.5, (2, tg.n_total_tracks, G))
links = (tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd)
phi1, out1 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, psi_in)
nxt1 = sweep_ref.link(out1, *links)
phi2, out2 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, nxt1)
for compact, inp, opts in ((0, "staged", {"split": 0}), (1, "compact", {}), (0, "staged", {"split": 0, "sweep_ell": 0})):
    dm = _capi.DeviceMesh(tg.mesh, 0); dm.set_option("compact", compact)
    for name, v in opts.items(): dm.set_option(name, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st = dt.fetch_offsets()
    print(inp, opts, "total", total, "offsets equal", np.array_equal(off, ref["offsets"]), "status equal", np.array_equal(st, ref["status"]), dt.stats()["split"])
    dt.sweep_set_links(tg)
    r = dt.sweep(G, sigma_t, source, weight, psi_in, input=inp)
    r2 = dt.sweep(G)
    for nm, got, want in (("phi", r["phi"], phi1), ("psi_out", r["psi_out"], out1), ("psi_next", r["psi_next"], nxt1), ("phi2", r2["phi"], phi2), ("psi_out2", r2["psi_out"], out2)):
        d = np.abs(got - want); e = float(d.max()) / max(float(np.abs(want).max()), 1e-300)
        w = np.unravel_index(int(d.argmax()), d.shape)
        extra = ""
        if nm.startswith("psi") and e > 1e-12:
            u = w[1]; extra = f" track {u} dir {w[0]} records {cnt[u]} status {ref['status'][u]} got {got[w]:.6g} want {want[w]:.6g}; tracks off: {int((d.max(axis=(0, 2)) > 1e-9).sum())}"
        print(f"   {nm:9s} rel err {e:.2e} at {w}{extra}")
    dt.close(); dm.close()
