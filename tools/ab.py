"""One parametrised driver for same-box measurements of the C-ABI library (development; not part of the product path).

  python tools/ab.py [--what segmentize|sweep|calls] [--mesh pincell.msh] [--nazim 128] [--delta 1e-3] [--groups 7]
                     [--libs a.so b.so ...] [--reps 2] [--nohash] [--calls 100] [name=v1,v2 ...]

* every `name=v1,v2` is an rt_set_option sweep (cartesian product over all of them); options read at rt_tracks_create
  ("split", "sort_mode", "hybrid") work because every combination gets fresh handles;
* `--libs`: library builds to interleave (each run is a child process with RT_SEGMENTIZE_LIB set), `--reps` times — the A/B
  of kernel variants on ONE box; without it the in-tree library runs in this process;
* `--what segmentize`: ms per step of plain calls (no HIP events), then the per-kernel HIP-event times of nine more calls
  (option "timing"), a hash of all results (offsets, status, records) and the regime (records by cheap steps);
  `--what sweep`: rt_sweep over (ℓ, cell) rows and over the CSR records, best / median of 15, result hashes;
  `--what calls`: plain calls only (for rocprofv3 traces and counter passes: tools/prof.sh).
Replaces the one-off exp_*.py / gpu_modes.py / ab.sh scripts of rounds 1-3 (their results are in DESIGN.md and profiles/)."""
import argparse, hashlib, itertools, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="segmentize", choices=("segmentize", "sweep", "calls"))
    ap.add_argument("--mesh", default="pincell.msh")
    ap.add_argument("--nazim", type=int, default=128)
    ap.add_argument("--delta", type=float, default=1e-3)
    ap.add_argument("--groups", type=int, default=7)
    ap.add_argument("--libs", nargs="*", default=[])
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--calls", type=int, default=100)
    ap.add_argument("--nohash", action="store_true")
    ap.add_argument("opts", nargs="*")
    return ap.parse_args()


def combos(opts):
    names, vals = [], []
    for kv in opts:
        k, v = kv.split("=")
        names.append(k); vals.append([int(x) for x in v.split(",")])
    return [dict(zip(names, c)) for c in itertools.product(*vals)] if names else [{}]


def run_here(a):
    import numpy as np
    import raytracing_jl_amd as rt
    from raytracing_jl_amd import _capi

    path = rt.data_path(a.mesh)
    model = rt.GmshDiscreteModel(path) if a.mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
    refl = rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)
    tg = rt.TrackGenerator(model, a.nazim, a.delta, bcs=refl)
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    lib = os.path.basename(os.environ.get("RT_SEGMENTIZE_LIB", "in-tree"))
    for opt in combos(a.opts):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        for k, v in opt.items():
            dm.set_option(k, v)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        tag = f"{lib} | {a.mesh} {a.nazim} {a.delta} {opt if opt else ''}"
        if a.what == "calls":
            for _ in range(4): seg()
            t0 = time.perf_counter()
            for _ in range(a.calls): total = seg()
            print(tag, f": {total} segments, {(time.perf_counter() - t0) / a.calls * 1e3:.4f} ms per step", flush=True)
        elif a.what == "segmentize":
            for _ in range(4): seg()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(40): total = seg()
                best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
            dm.set_option("timing", 1)
            tm = []
            for _ in range(9):
                seg(); tm.append(dt.timing())
            med = lambda k: sorted(t[k] for t in tm)[len(tm) // 2]
            h = hashlib.sha256()
            off, st = dt.fetch_offsets(); vol = dt.fetch_volumes()
            recs = {} if a.nohash else dt.fetch_segments()  # (--nohash: batches of gigabytes — offsets and status only)
            for x in (off, st, *[recs[k] for k in ("px", "py", "qx", "qy", "ell", "element") if k in recs]): h.update(np.ascontiguousarray(x).tobytes())
            s = dt.stats()
            print(tag, f": {total} segments, {best:.4f} ms/step, march {med('march'):.4f} scan {med('scan'):.4f} compact {med('compact'):.4f} volumes {med('volumes'):.4f} | "
                  f"records sha {h.hexdigest()[:12]} volumes sum {float(vol.sum()):.15e} cheap {s['cheap_records']} generic {s['generic_records']} from lengths {s['records_tallied_from_lengths']} queued {s.get('lean_queued', 0)} "
                  f"held {s['device_bytes'] / 1e9:.3f} GB", flush=True)
        else:
            G, nc = a.groups, tg.mesh.num_cells
            sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G); src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
            out = []
            seg(); dt.sweep_set_links(tg)
            for inp in ("staged", "compact"):
                r = dt.sweep(G, sig, src, None, np.ones((2, tg.n_total_tracks, G)), input=inp)
                ms = sorted(dt.sweep(G, input=inp, fetch=False)["ms"] for _ in range(15))
                out.append(f"{inp} {ms[0]:.4f} (median {ms[7]:.4f}, first {r['ms']:.4f}, {r['passes']} passes) psi_out sha "
                           f"{hashlib.sha256(np.ascontiguousarray(r['psi_out']).tobytes()).hexdigest()[:10]} phi sum {float(r['phi'].sum()):.15e}")
            print(tag, f"G={G} |", " | ".join(out), flush=True)
        dt.close(); dm.close()


def main():
    a = parse()
    if not a.libs:
        return run_here(a)
    argv = [sys.executable, os.path.abspath(__file__)] + [x for x in sys.argv[1:]]
    # drop --libs ... and --reps from the child's arguments
    child, skip = [], False
    it = iter(sys.argv[1:])
    for x in it:
        if x == "--libs":
            skip = True; continue
        if x == "--reps":
            next(it); skip = False; continue
        if skip and not x.startswith("--") and x.endswith(".so"):
            continue
        skip = False
        child.append(x)
    for _ in range(a.reps):
        for lib in a.libs:
            env = dict(os.environ, RT_SEGMENTIZE_LIB=os.path.abspath(lib))
            subprocess.run([sys.executable, os.path.abspath(__file__)] + child, env=env, check=False)


if __name__ == "__main__":
    main()
