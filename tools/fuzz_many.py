"""Development: many random Delaunay meshes / quadratures through the HIP path against the oracle, bit for bit.
usage (GPU box): python tools/fuzz_many.py [first_seed] [count]"""
import sys, os, time, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_jl_amd as rt
from oracle import oracle as orc
spec = importlib.util.spec_from_file_location("m", os.path.join(ROOT, "tests", "test_gpu_random_meshes.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed * 7919)
    n_int = int(rng.integers(50, 3000))
    kw = dict(w=float(rng.choice([1.0, 0.3, 2.5, 7.0])), h=float(rng.choice([1.0, 0.4, 1.7])),
              x0=float(rng.choice([0.0, -3.25, 11.0])), y0=float(rng.choice([0.0, 2.5, -0.75])),
              nb=int(rng.choice([6, 12, 30])), cluster=bool(rng.integers(0, 2)))
    n_azim = int(rng.choice([4, 8, 16, 32]))
    delta = float(rng.choice([0.002, 0.004, 0.01])) * min(kw["w"], kw["h"])
    model = m._random_model(rt, seed, n_int, **kw)
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    ref = m._oracle(orc, tg)
    opts = [dict(), dict(split=int(rng.choice([16, 24, 40])))][: 1 + int(rng.integers(0, 2))]
    for o in opts:
        total, off, st, seg, vol = m._run(rt, tg, o)
        ok = total == ref["total"] and np.array_equal(st, ref["status"]) and np.array_equal(off, ref["offsets"]) and \
            np.array_equal(seg["element"], ref["element"]) and all(np.array_equal(seg[k], ref[k]) for k in ("px", "py", "qx", "qy", "ell")) and \
            np.allclose(vol, ref["volumes"], rtol=1e-10, atol=1e-300)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, o, kw, n_azim, delta, flush=True)
    print("seed %d: %d cells, %d tracks, %d segments, %d failing tracks, opts %s  [%.0f s]" %
          (seed, model.num_cells, tg.n_total_tracks, int(ref["total"]), int(np.count_nonzero(ref["status"])), opts, time.time() - t0), flush=True)
print("done:", count, "meshes,", bad, "mismatches")
sys.exit(1 if bad else 0)
