"""Fuzz the walk step's certificates on the CPU: the device march's per-lane logic compiled for the host
(tests/host_march.hip) with the walk step on and off, against the CPU checker, bit for bit, over seeded meshes
of every shape class (tests/meshgen.py).  No GPU needed.
usage: python tools/fuzz_cpu.py [first_seed] [count] [processes]   -> one line per mesh + a summary; exit 1 on mismatch"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


KINDS = ["random", "cluster", "lattice", "sliver", "lattice_far", "sliver_fine", "near_vertex", "aligned", "lattice_mid", "steep"]


def case(seed):
    """(kind, model, nφ, δ, k) of a seed.  The last four kinds aim at the thresholds of the walk / cheap step certificates
    (tests/meshgen.py): nodes on or within 1e-8 … 3e-3 spacings of track lines, lattice rows within 1e-7 … 3e-3 rad of a track
    direction, lattices 10 – 60 units from the origin (not yet fragile, but the rounding terms k2 / κ0 are large), and
    hand-made tracks within 1e-5 … 1e-8 of ϕ = 0, π/2, π (the tracks are replaced in `make_tg`)."""
    import raytracing_jl_amd as rt
    import meshgen
    rng = np.random.default_rng(seed * 104729 + 7)
    kind = KINDS[seed % len(KINDS)]
    kw = dict(w=float(rng.choice([1.0, 0.3, 2.5, 7.0])), h=float(rng.choice([1.0, 0.4, 1.7])),
              x0=float(rng.choice([0.0, -3.25, 11.0])), y0=float(rng.choice([0.0, 2.5, -0.75])))
    if kind in ("near_vertex", "aligned"):
        n_azim = int(rng.choice([4, 8, 16, 32, 64]))
        delta = float(rng.choice([0.002, 0.004, 0.01])) * min(kw["w"], kw["h"])
        k = int(rng.choice([5, 5, 5, 2, 3, 8, 12, 1]))
        if kind == "near_vertex":
            model = meshgen.near_vertex_model(rt, seed, int(rng.integers(80, 1500)), n_azim, delta, nb=int(rng.choice([6, 12, 30])), **kw)
        else:
            model = meshgen.aligned_model(rt, seed, int(rng.integers(6, 30)), n_azim, delta, **kw)
        return kind, model, n_azim, delta, k
    if kind == "lattice_mid":
        n = int(rng.integers(6, 40))
        kw.update(x0=float(rng.choice([10.0, 30.0, -60.0])), y0=float(rng.choice([5.0, -40.0, 25.0])))
        model = meshgen.lattice_model(rt, seed, n, n, jitter=float(rng.choice([0.1, 0.3])), **kw)
    elif kind == "steep":
        n = int(rng.integers(6, 30))
        model = meshgen.lattice_model(rt, seed, n, n, jitter=float(rng.choice([0.0, 0.1, 0.3])), **kw) if seed % 20 < 10 else \
            meshgen.random_model(rt, seed, int(rng.integers(50, 1200)), **kw)
    elif kind == "random":
        model = meshgen.random_model(rt, seed, int(rng.integers(50, 2500)), nb=int(rng.choice([6, 12, 30])), **kw)
    elif kind == "cluster":
        model = meshgen.random_model(rt, seed, int(rng.integers(150, 2500)), nb=int(rng.choice([6, 12, 30])), cluster=True, **kw)
    elif kind == "lattice":
        n = int(rng.integers(6, 40))
        model = meshgen.lattice_model(rt, seed, n, n, jitter=float(rng.choice([0.1, 0.25, 0.4])), **kw)
    elif kind == "lattice_far":
        n = int(rng.integers(6, 40))
        kw.update(x0=float(rng.choice([100.0, -1000.0])), y0=float(rng.choice([50.0, 2000.0])))
        model = meshgen.lattice_model(rt, seed, n, n, jitter=0.3, **kw)
    elif kind == "sliver":
        n = int(rng.integers(6, 30))
        model = meshgen.sliver_model(rt, seed, n, n, gap=float(rng.choice([1e-2, 1e-3, 1e-4])), **kw)
    else:
        n = int(rng.integers(6, 30))
        model = meshgen.sliver_model(rt, seed, n, n, gap=float(rng.choice([1e-5, 1e-6, 1e-7])), **kw)
    n_azim = int(rng.choice([4, 8, 16, 32, 64, 256, 1024]))
    delta = float(rng.choice([0.002, 0.004, 0.01])) * min(kw["w"], kw["h"]) * (4.0 if n_azim >= 256 else 1.0)
    k = int(rng.choice([5, 5, 5, 2, 3, 8, 12, 1]))
    return kind, model, n_azim, delta, k


def tiny_of(seed):
    """tiny_step of the case (TrackGenerator's keyword, src/trackgenerator.jl:80): the reference's default mostly, sometimes
    much larger or smaller — the walk step's certificates must hold (or refuse) whatever the re-seeding distance is."""
    return float(np.random.default_rng(seed * 31 + 5).choice([1e-8, 1e-8, 1e-8, 1e-6, 1e-5, 1e-10, 1e-7]))


def run(seed):
    import raytracing_jl_amd as rt
    from oracle import oracle as orc
    import hostmarch as hm
    kind, model, n_azim, delta, k = case(seed)
    if os.environ.get("FUZZ_SHUFFLE"):
        # cells with their three nodes in random order (rotations and reflections: both edge orientations occur on either
        # side of an edge) — Gridap's oriented grids list them ascending, but the C ABI takes any order
        rs = np.random.default_rng(seed + 77)
        cells = np.asarray(model.cell_node_ids).copy()
        for c in range(len(cells)):
            cells[c] = cells[c][rs.permutation(3)]
        model = rt.DiscreteModel(model.node_coordinates, cells)
    tg = rt.TrackGenerator(model, n_azim, delta, tiny_step=tiny_of(seed) if os.environ.get("FUZZ_TINY") else 1e-8)
    rt.trace(tg)
    if kind == "steep":
        import meshgen
        meshgen.steep_tracks(rt, tg, seed)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=False)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, k=k, iter_cap=4000000, n_threads=1)
    res = {}
    bad = []
    for walk in (True, False, "topo"):
        r = hm.run(tg, k=k, walk=walk, n_threads=1)
        ok = r["total"] == ref["total"] and np.array_equal(r["status"], ref["status"]) and np.array_equal(r["offsets"], ref["offsets"]) and \
            np.array_equal(r["element"], ref["element"]) and all(np.array_equal(r[q], ref[q]) for q in ("px", "py", "qx", "qy", "ell"))
        if not ok:
            bad.append(walk)
        res[walk] = r
    # the Σℓ chain of k_materialise_lin (rt_device.hpp chain_*: the functions the kernel calls) on the records of the cheap-step
    # march just made — with the flags of the records that keep their own p —, against the left-to-right check, at the default
    # rtol and at tolerances tuned to this problem's own tracks
    cw, cm, cd, _cm0 = hm.chain_check(tg)
    if cw:
        bad.append("chain")
    s, info = res[True]["stats"], res[True]["info"]
    line = ("seed %d %-11s tiny %.0e cells %5d nφ %4d k %2d tracks %6d segs %8d failing %5d | records walkable %5d/%5d eps≤%.1e fragile %d degenerate %d | "
            "walk emits %8d skips %6d generic emits %7d refused %6d | cheap steps: emits %8d refused %5d restarts %d | Σℓ chain: %d decisions, %d left to the exact sum, %d wrong%s" %
            (seed, kind, tg.tiny_step, model.num_cells, n_azim, k, tg.n_total_tracks, ref["total"], int(np.count_nonzero(ref["status"])),
             int(info["records_walk"]), int(info["records"]), info["eps_max"], int(info["cells_fragile"]), int(info["cells_degenerate"]),
             s["walk_emits"], s["walk_skips"], s["generic_emits"], s["refused"], res["topo"]["stats"]["cheap_emits"],
             res["topo"]["stats"]["cheap_refused"], res["topo"]["stats"]["cheap_restarts"], cd, cm, cw, ("  MISMATCH walk=%s" % bad) if bad else ""))
    return seed, bool(bad), line, s["walk_emits"], ref["total"], res["topo"]["stats"]["cheap_emits"], kind


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    from oracle import oracle as orc
    import hostmarch as hm
    orc.build()
    import shutil, tempfile
    private = os.path.join(tempfile.mkdtemp(prefix="fuzz_cpu_"), "libhostmarch.so")
    shutil.copy(hm.build(), private)  # the workers load a private copy: rebuilding the tree does not disturb a long run
    os.environ["HOSTMARCH_LIB"] = private
    import multiprocessing as mp
    t0 = time.time()
    n_bad = 0; walk_total = 0; seg_total = 0; cheap_total = 0
    per = {}
    with mp.Pool(procs) as pool:
        for seed, bad, line, we, tot, ce, kind in pool.imap_unordered(run, range(first, first + count)):
            print(line, flush=True)
            n_bad += bad; walk_total += we; seg_total += tot; cheap_total += ce
            a = per.setdefault(kind, [0, 0, 0, 0]); a[0] += 1; a[1] += tot; a[2] += we; a[3] += ce
    for kind, a in sorted(per.items()):
        print("class %-11s: %4d meshes, %10d segments, %5.1f %% by the walk step, %5.1f %% by cheap steps" %
              (kind, a[0], a[1], 100.0 * a[2] / max(a[1], 1), 100.0 * a[3] / max(a[1], 1)))
    print("done: %d meshes, %d mismatches, %d segments, %.1f %% of them by the walk step (%.1f %% by cheap steps), %.0f s" %
          (count, n_bad, seg_total, 100.0 * walk_total / max(seg_total, 1), 100.0 * cheap_total / max(seg_total, 1), time.time() - t0))
    sys.exit(1 if n_bad else 0)
