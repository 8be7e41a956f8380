"""Development: run one random-mesh case / option set of tests/test_gpu_random_meshes.py in isolation."""
import sys, os, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_jl_amd as rt
spec = importlib.util.spec_from_file_location("m", os.path.join(ROOT, "tests", "test_gpu_random_meshes.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
CASES = {1: (200, {}), 2: (1500, {}), 3: (600, dict(cluster=True)), 4: (400, dict(w=3.0, h=0.7, x0=-1.5, y0=10.0)), 5: (3000, dict(nb=40))}
seed = int(sys.argv[1]); opts = dict(kv.split("=") for kv in sys.argv[2:]); opts = {k: int(v) for k, v in opts.items()}
n, kw = CASES[seed]
model = m._random_model(rt, seed, n, **kw)
w, h = kw.get("w", 1.0), kw.get("h", 1.0)
tg = rt.TrackGenerator(model, 16, 0.004 * min(w, h)); rt.trace(tg)
print("seed", seed, opts, "cells", model.num_cells, "tracks", tg.n_total_tracks, flush=True)
total, off, st, seg, vol = m._run(rt, tg, opts)
print("   total", total, "failed", int(np.count_nonzero(st)), "status", np.unique(st), flush=True)
