"""Many seeded meshes / quadratures / k through the HIP path (C ABI) against the checker, bit for bit, with the regime
of every run printed (rt_mesh_info / rt_last_stats).  Same cases as tools/fuzz_cpu.py (every mesh class of
tests/meshgen.py incl. the threshold-aimed ones, nφ up to 1024, k in {1, 2, 3, 5, 8, 12}); walk step on with exact steps only, off,
on with cheap steps as the library gates them, on with cheap steps FORCED (option "topo" = 2: no 90 % gate, no hand-back of
often-refused waves — every record that carries a cheap certificate is decided by it), the library's defaults (pieces for
small batches), and cheap steps forced with the records in COMPLETION order (option "record_order" 2: the per-track table against
the checker, then the CSR layout on demand).  Prints the share of cheap steps and the refusals by certificate term per mesh class.
FUZZ_SMALL_POOLS=1: every handle starts with pools and result arrays that are too small — every call's first attempt is void or short.
usage (GPU box): [FUZZ_TINY=1] [FUZZ_SHUFFLE=1] [FUZZ_SMALL_POOLS=1] python tools/fuzz_many.py [first_seed] [count]"""
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

orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
FIELDS = ("px", "py", "qx", "qy", "ell")


def lr_sums(off, ell):
    """Σℓ per track, added LEFT TO RIGHT (what src/track.jl:171 compares and k_finish forms): record r of every track in one vector
    addition, r = 0, 1, ... — bit for bit the sequential sum."""
    cnt = np.diff(off)
    acc = np.zeros(len(cnt))
    for r in range(int(cnt.max()) if len(cnt) else 0):
        m = cnt > r
        acc[m] = acc[m] + ell[off[:-1][m] + r]
    return acc


def tuned_rtols(L, E):
    """Tolerances next to this problem's own tracks: for up to four tracks whose |ℓ − Σℓ| / max is real (> 1e-11), that ratio
    x (1 ± 1e-7) — the track just fails / just passes the reference's check."""
    big = np.maximum(np.abs(L), np.abs(E))
    ratio = np.abs(L - E) / np.where(big > 0, big, 1.0)
    real = np.sort(ratio[(ratio > 1e-11) & (ratio < 1e-3)])
    pick = real[np.linspace(0, len(real) - 1, min(4, len(real))).astype(int)] if len(real) else []
    return [float(v) * f for v in pick for f in (1 - 1e-7, 1 + 1e-7)]


bad = 0
n_tuned = 0
n_completion = 0
t0 = time.time()
seg_total = walk_total = cheap_total = 0
per = {}
for seed in range(first, first + count):
    kind, model, n_azim, delta, k = fuzz_cpu.case(seed)
    if n_azim >= 1024:  # keep a GPU run short: the CPU fuzzer covers the finest quadratures
        n_azim = 256
    if os.environ.get("FUZZ_SHUFFLE"):  # cells with their three nodes in random order (as tools/fuzz_cpu.py)
        rs = np.random.default_rng(seed + 77)
        cells = np.asarray(model.cell_node_ids).copy()
        for c in range(len(cells)):
            cells[c] = cells[c][rs.permutation(3)]
        model = rt.DiscreteModel(model.node_coordinates, cells)
    tg = rt.TrackGenerator(model, n_azim, delta, tiny_step=fuzz_cpu.tiny_of(seed) if os.environ.get("FUZZ_TINY") else 1e-8)
    rt.trace(tg)
    if kind == "steep":
        import meshgen
        meshgen.steep_tracks(rt, tg, seed)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step,
                        k=k, iter_cap=4000000, n_threads=0)
    aq = tg.azimuthal_quadrature
    vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    regime = ""
    cheap_n = forced_n = 0
    for opts in (dict(walk=1, split=0, topo=0), dict(walk=0, split=0), dict(walk=1, split=0, topo=1), dict(walk=1, split=0, topo=2), dict(walk=1),
                 dict(walk=1, split=0, topo=2, record_order=2)):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        for kk, v in opts.items():
            dm.set_option(kk, v)
        if os.environ.get("FUZZ_SMALL_POOLS"):
            # every handle starts with a staging pool, a side list and result arrays that are too small (by a seeded amount): the first
            # attempt of every call is void or short, the re-run paths of every mode run on every mesh — with the record kernel beside
            # the march (sixth mode) the pools run out WHILE it serves units
            rs = np.random.default_rng(seed * 7 + len(opts))
            dm.set_option("pool_chunks_hint", int(rs.integers(2, 400)))
            dm.set_option("side_entries_hint", int(rs.integers(1, 3000)))
            dm.set_option("test_out_records", int(rs.integers(100, 200000)))
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        total = dt.segmentize(tg.tiny_step, k, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        ok_tab = True
        if total == ref["total"] and dt.record_order() == 1:
            # records in completion order: every track's records through the per-track table, bit for bit; the spans tile [0, total)
            n_completion += 1
            beg, cnt, st_t = dt.fetch_table()
            rec = dt.fetch_records()
            roff = np.asarray(ref["offsets"], np.int64)
            ok_tab = np.array_equal(cnt, np.diff(roff)) and np.array_equal(st_t, ref["status"])
            if ok_tab and total:
                b, e = np.sort(beg[cnt > 0]), np.sort((beg + cnt)[cnt > 0])
                idx = np.repeat(beg - roff[:-1], cnt) + np.arange(total)
                ok_tab = bool(b[0] == 0 and e[-1] == total and np.array_equal(b[1:], e[:-1])) and np.array_equal(rec["element"][idx], ref["element"]) and \
                    all(np.array_equal(rec[f][idx], ref[f]) for f in FIELDS)
        off, st = dt.fetch_offsets()
        seg = dt.fetch_segments()  # (a handle in completion order: the CSR layout on demand)
        ok = ok_tab and total == ref["total"] and np.array_equal(st, ref["status"]) and np.array_equal(off, ref["offsets"]) and \
            np.array_equal(seg["element"], ref["element"]) and all(np.array_equal(seg[f], ref[f]) for f in FIELDS) and \
            np.allclose(dt.fetch_volumes(), vol, rtol=1e-10, atol=1e-300)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, kind, opts, n_azim, delta, k, flush=True)
        if opts == dict(walk=1, split=0, topo=1):
            cheap_n = dt.stats()["cheap_records"]
            cheap_total += cheap_n
        if opts == dict(walk=1, split=0, topo=2) and ok:
            # the Σℓ check of the record kernel (its chain + k_finish's exact sums) at tolerances tuned to this problem's tracks:
            # the status must be the reference's check on the left-to-right sums at THAT rtol (records do not depend on rtol)
            L = np.asarray(tg.ell, np.float64)
            E = lr_sums(ref["offsets"], ref["ell"])
            big = np.maximum(np.abs(L), np.abs(E))
            for rtol_t in tuned_rtols(L, E):
                dt.segmentize(tg.tiny_step, k, rtol_t, aq.delta_s, aq.n_azim_2)
                _, st_t = dt.fetch_offsets()
                want = np.where((ref["status"] == 0) | (ref["status"] == 2), np.where((L == E) | (np.abs(L - E) <= rtol_t * big), 0, 2), ref["status"])
                n_tuned += 1
                if not np.array_equal(st_t, want):
                    bad += 1
                    print("MISMATCH (status at a tuned rtol) seed", seed, kind, rtol_t, "tracks", np.nonzero(st_t != want)[0][:8], flush=True)
        if opts == dict(walk=1, split=0, topo=2):
            st2 = dt.stats()
            forced_n = st2["cheap_records"]
            a = per.setdefault(kind, dict(meshes=0, segs=0, cheap=0, forced=0, refusals=dict.fromkeys(_capi.DeviceTracks.REFUSAL_TERMS, 0)))
            a["meshes"] += 1; a["segs"] += total; a["cheap"] += cheap_n; a["forced"] += forced_n
            for kk2, v in st2["cheap_refusals"].items():
                a["refusals"][kk2] += v
        if opts == dict(walk=1, split=0, topo=0):
            info, stats = dm.info(), dt.stats()
            regime = "walk %s, %d/%d records walkable, eps<=%.1e, %d of %d records by the walk step" % (
                "on" if info["walk_enabled"] else "off", info["records_walk"], info["records"], info["eps_max"], stats["walk_records"], total)
            seg_total += total
            walk_total += stats["walk_records"]
        dt.close()
        dm.close()
    print("seed %d %-11s cells %5d nφ %4d k %2d tracks %6d segs %8d failing %5d | %s, %d by cheap steps, %d when forced  [%.0f s]" %
          (seed, kind, model.num_cells, n_azim, k, tg.n_total_tracks, int(ref["total"]), int(np.count_nonzero(ref["status"])), regime,
           cheap_n, forced_n, time.time() - t0), flush=True)
for kind, a in sorted(per.items()):
    print("class %-11s: %4d meshes, %10d segments, %5.1f %% by cheap steps as gated, %5.1f %% forced; refusals when forced: %s" %
          (kind, a["meshes"], a["segs"], 100.0 * a["cheap"] / max(a["segs"], 1), 100.0 * a["forced"] / max(a["segs"], 1),
           ", ".join("%s %d" % kv for kv in a["refusals"].items() if kv[1])))
print("status at tuned tolerances: %d calls; records in completion order (per-track table): %d calls" % (n_tuned, n_completion))
print("done: %d meshes x 6 modes, %d mismatches, %d segments, %.1f %% of them by the walk step, %.1f %% by cheap steps as gated, %.1f %% forced" %
      (count, bad, seg_total, 100.0 * walk_total / max(seg_total, 1), 100.0 * cheap_total / max(seg_total, 1),
       100.0 * sum(a["forced"] for a in per.values()) / max(seg_total, 1)))
sys.exit(1 if bad else 0)
