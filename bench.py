#!/usr/bin/env python3
"""Benchmark of the HIP segmentize! path — BASELINE.json's metric on BASELINE.json's configurations.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full ``segmentize!`` (``rt_segmentize`` through the C ABI: device march of every track, CSR
offsets, compact segment records, fused ``fill_volumes``) over a batch of tracks that is already resident in HBM.

N = 1 — the headline configuration (BASELINE.json configs[2], the one the 50 M segments/s target is quoted on):
``demo/pincell.msh`` at nφ=128, δ=1e-3, 130,456 tracks, 9.32 M segments.  The line also carries, under
``config5_single_gpu``, the scaling workload below run on this one GPU, so that an N=1 point of the same
workload exists beside every N>1 line.

N > 1 (launched by ``torch.distributed.run``, one rank per GPU) — BASELINE.json configs[4]: the BWR-like assembly
mesh at nφ=128, δ=5e-4 (1,043,212 tracks, 114 M segments) as a FIXED global problem, i.e. strong scaling:
``tracks_by_uid`` is cut into N contiguous uid ranges balanced by Σℓ, every rank marches its own range, and the only
data-path exchange of the reference's algorithm (``fill_volumes``' cross-track sum) is one RCCL all-reduce of
``n_cells`` doubles inside the timed step.  The RCCL all-gather-v that reassembles the global segment list on
every rank (direct sends and receives into the final buffers, ``distributed.SegmentGather``) is measured after the
timed region and always reported beside it (``allgather.ms``, ``segments_per_s_including_allgather``); rank 0 then
runs the whole problem alone (``single_gpu_same_workload``) so that the speed-up is measured inside one run.

The JSON line also carries ``roofline`` (``definition_version`` 3: the DOMINANT kernel by HIP-event time — the longer of march and
record kernel: SURVEY §8(d)'s algorithmic bytes per launch ÷ its duration against the 8 TB/s HBM peak; ``record_kernel``: the kernel
that writes the 44-B records with its own 48 B per segment, whatever its share; ``pipeline_frac``: 45 B per segment ÷ the whole
step; ``march``: the march's own limits — latency / issue — with real bytes by counters; ``traffic`` only when the committed PMC
summary was taken from the very library that is running), ``latency`` (p50/p95 of single steps), ``e2e`` (the costs of the
boundary around the step: mesh preparation, track upload, record download) and, at N=1, ``cpu_baseline`` (the
oracle — a C port of the reference's algorithm — timed on the host cores of the same box).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
# Algorithmic bytes per segment of each kernel of a step.  The march's figure is
# SURVEY.md §8(d)'s: 44 B of record written + ≈1.3 B of amortised track input.  Round 4's two-phase march (whole-track batches
# with cheap steps) stages ONE 4-B word per record; k_materialise reads it, computes p, q, ℓ and writes the 44-B record to
# its CSR position: 48 B/segment ("compact" below is that phase: k_materialise + k_finish).  Batches that march with exact
# steps stage (q, ±cell) rows, 20 B, and k_compact3 moves 64 B/segment.
BYTES_PER_SEGMENT = {"march": 45.0, "compact": 64.0, "scan": 0.0}
BYTES_PER_SEGMENT_TWO_PHASE = {"march": 45.0, "compact": 48.0, "scan": 0.0}
STEP_BYTES_PER_SEGMENT = 45.0  # the whole step, by the same definition (what one segmentize! must at least move)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06", "pmc_summary.json")

WORKLOADS = {
    "c3": dict(mesh="pincell.msh", n_azim=128, delta=1e-3, name="BASELINE configs[2]: demo/pincell.msh, nφ=128, δ=1e-3"),
    "c5": dict(mesh="bwr_like.msh", n_azim=128, delta=5e-4,
               name="BASELINE configs[4]: BWR assembly (substitute mesh bwr_like.msh, tools/make_bwr_mesh.py), nφ=128, δ=5e-4"),
}


def lib_sha256():
    from raytracing_jl_amd import _capi

    return hashlib.sha256(open(_capi.LIB_PATH, "rb").read()).hexdigest()


PMC_KERNELS = {}  # raw counters per kernel of the committed PMC passes (filled by pmc_traffic when they belong to this library)


def pmc_traffic():
    """HBM bytes per launch per kernel from the committed rocprofv3 PMC passes — only if they were taken from the
    library that is running now (sha256 of the .so recorded by tools/pmc_summary.py).  FETCH_SIZE and WRITE_SIZE are
    in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads and is doubled
    (MI355X_MICROARCH.md §HBM)."""
    try:
        d = json.load(open(PMC_SUMMARY))
        if d.get("lib_sha256") != lib_sha256():
            return {}, "profiles/r06/pmc_summary.json was taken from another build of the library"
        global PMC_KERNELS
        PMC_KERNELS = d["kernels"]
        return ({k: (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for k, v in d["kernels"].items()
                 if "FETCH_SIZE" in v and "WRITE_SIZE" in v}, "profiles/r06/pmc_summary.json (same library, sha256 match)")
    except Exception as e:
        return {}, "no PMC summary: %r" % (e,)


def kernel_names(stats):
    """Names of the kernels behind the three timed phases.  split = 2 (hybrid): the whole-track kernel named here runs
    beside the split kernel that marches the longest waves in pieces; the phase time covers both."""
    w = stats["march_waves"]
    sp = stats["split"] == 1
    cheap = stats.get("cheap_records", 0) > 0  # the TOPO instantiation (cheap steps) ran
    # (the template's last argument: the lean plan's phase — 0 = the march in one kernel, what the default options run)
    return {"march": "rt::k_march<2, %d, %s, %s, %s, 0>" % (w, "true" if sp else "false", "true" if stats["wide_k"] else "false",
                                                              "true" if cheap else "false"),
            # (the record kernel by what the call launched — rt_last_stats[23]: a two-phase call falls back to k_materialise for
            #  arrays of 2^29 records or with option "mat_kernel" 1)
            "compact": (stats.get("record_kernel") or ("rt::k_materialise_lin<false>" if cheap else "rt::k_compact3")) +
                       ("<%s>" % ("true" if sp else "false") if (stats.get("record_kernel") or ("" if cheap else "rt::k_compact3")) == "rt::k_compact3" else ""),
            "scan": "rt::k_scan_fused" if cheap else "rt::k_scan_write"}


def bytes_per_segment(stats):
    return BYTES_PER_SEGMENT_TWO_PHASE if stats.get("cheap_records", 0) > 0 else BYTES_PER_SEGMENT


def make_tg(rt, wl):
    mesh_file = rt.data_path(wl["mesh"])
    model = rt.GmshDiscreteModel(mesh_file) if mesh_file.endswith(".msh") else rt.DiscreteModelFromFile(mesh_file)
    tg = rt.TrackGenerator(model, wl["n_azim"], wl["delta"])
    rt.trace(tg)
    return tg


def run_steps(rt, dt, tg, aq, n):
    t0 = time.perf_counter()
    for _ in range(n):
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    return total, time.perf_counter() - t0


def single_gpu_run(rt, _capi, wl, device, steps, warmup, stream_ptr=None, tg=None, e2e=False):
    """The given workload, unsharded, on one GPU: ms per step (host clock around K synchronous calls)."""
    tg = tg if tg is not None else make_tg(rt, wl)
    aq = tg.azimuthal_quadrature
    t0 = time.perf_counter()
    dm = _capi.DeviceMesh(tg.mesh, device)
    mesh_ms = (time.perf_counter() - t0) * 1e3
    if stream_ptr:
        dm.set_stream(stream_ptr)
    t0 = time.perf_counter()
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    h2d_ms = (time.perf_counter() - t0) * 1e3
    h2d_again_ms = None
    if e2e:  # a second handle on the same arrays: the library keeps its page-locked staging block from the second upload of a process on
        for _ in range(2):
            t0 = time.perf_counter()
            dt2 = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
            h2d_again_ms = (time.perf_counter() - t0) * 1e3
            dt2.close()
    run_steps(rt, dt, tg, aq, warmup)
    total, el = run_steps(rt, dt, tg, aq, steps)
    dm.set_option("timing", 1)
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    tm = dt.timing()
    out = {"workload": wl["name"], "tracks": int(tg.n_total_tracks), "segments": int(total), "steps": steps,
           "ms_per_step": el / steps * 1e3, "value": total * steps / el, "unit": "segments/s",
           "kernel_ms_last_step": {k: tm[k] for k in ("march", "scan", "compact", "total")},
           "device_GB_held_by_the_handle": dt.stats()["device_bytes"] / 1e9}
    if e2e:
        try:
            out["e2e"] = boundary_costs(dt, total)
            try:
                out["e2e"]["one_shot_sequence"] = one_shot_sequence(rt, _capi, dm, tg, aq)
            except Exception as e:  # pragma: no cover
                out["e2e"]["one_shot_sequence"] = {"error": repr(e)}
            out["e2e"].update({"mesh_create_ms": mesh_ms, "mesh_prep_host_ms": dm.info()["prep_ms"], "tracks_h2d_ms": h2d_ms,
                               "tracks_h2d_again_ms": h2d_again_ms, "segmentize_ms": out["ms_per_step"]})
            one_shot(out["e2e"])
        except Exception as e:  # pragma: no cover
            out["e2e"] = {"error": repr(e)}
    dt.close()
    dm.close()
    return out


def two_in_flight(tg, aq, dmesh, dt, steps, segments_per_step):
    """Extra, not `value`: the same K steps issued from two host threads on two handles / streams.  The march of
    one batch is a latency chain that leaves most of the chip idle and the compaction of another is HBM-bound,
    so independent batches overlap well (a host that segmentizes several track sets, e.g. one per geometry).
    `value` and `roofline` above stay the one-synchronous-call-at-a-time numbers."""
    import threading

    import raytracing_jl_amd as rt
    from raytracing_jl_amd import _capi

    try:
        dmesh2 = _capi.DeviceMesh(tg.mesh, dmesh.device)  # own stream
        dt2 = _capi.DeviceTracks(dmesh2, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        for _ in range(2):
            dt2.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)

        def run(h, n):
            for _ in range(n):
                h.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)

        n_each = max(1, steps // 2)
        th = [threading.Thread(target=run, args=(h, n_each)) for h in (dt, dt2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        dt2.close()
        dmesh2.close()
        return {"steps": 2 * n_each, "ms_per_step": el / (2 * n_each) * 1e3,
                "value": segments_per_step * 2 * n_each / el, "unit": "segments/s",
                "note": "two independent batches overlapped on two streams; not the headline value"}
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}


def completion_order_calls(rt, tg, aq, dmesh, steps, segments_per_step):
    """Extra, not `value`: the same K steps on a handle of its own with the option "record_order" 2 — a march workgroup that ends
    takes its tracks' span of the result arrays from an atomic cursor and the record kernel runs BESIDE the rest of the march
    (every track's records contiguous, the tracks in completion order, a per-track table: rt_device_table).  Built and measured in
    round 6: no gain inside one call (profiles/r06/exp_completion_order.log) — the line carries the number so that the driver's
    box says so too."""
    try:
        from raytracing_jl_amd import _capi
        dm2 = _capi.DeviceMesh(tg.mesh, dmesh.device)
        dm2.set_option("record_order", 2)
        dt2 = _capi.DeviceTracks(dm2, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        seg = lambda: dt2.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        for _ in range(3):
            seg()
        t0 = time.perf_counter()
        for _ in range(steps):
            seg()
        el = time.perf_counter() - t0
        st = dt2.stats()
        out = {"steps": steps, "ms_per_step": el / steps * 1e3, "value": segments_per_step * steps / el, "unit": "segments/s",
               "records_in_completion_order": bool(st.get("completion_order")), "record_order_after_call": dt2.record_order(),
               "note": "rt_set_option record_order=2: records written beside the march, tracks in completion order + per-track table; not the headline value"}
        dt2.close()
        dm2.close()
        return out
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}


def stream_ordered_calls(rt, tg, aq, dmesh, dt, steps, segments_per_step):
    """Extra, not `value`: the same K steps with the option "async" — rt_segmentize returns once the host knows total, status
    summary and offsets (after the scan) while the compaction still runs, and the next call queues behind it: the ≈25 µs of host
    turnaround between two synchronous calls disappear.  All K steps' kernels complete inside the timed region (rt_wait)."""
    try:
        import torch
        seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        dmesh.set_option("async", 1)
        for _ in range(3):
            seg()
        dt.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            seg()
        dt.wait()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        dmesh.set_option("async", 0)
        seg()
        return {"steps": steps, "ms_per_step": el / steps * 1e3, "value": segments_per_step * steps / el, "unit": "segments/s",
                "note": "rt_set_option async=1: calls return after march + scan, the compaction overlaps the host's turnaround; not the headline value"}
    except Exception as e:  # pragma: no cover
        dmesh.set_option("async", 0)
        return {"error": repr(e)}


def one_shot_sequence(rt, _capi, dm, tg, aq, lo=0, hi=None):
    """One call as the reference makes it (segmentize! runs once per TrackGenerator, src/trackgenerator.jl:357-369), as ONE timed
    sequence on a mesh handle that exists: rt_result_alloc (the library's host block: mapped, huge pages asked for, faulted in by its
    threads in the background from here on), rt_tracks_create (track arrays in), rt_segmentize (this handle's FIRST call: it also
    allocates its device pools), rt_result_fetch (offsets, status and the six record arrays out, copied behind the faulting front)."""
    hi = len(tg.ell) if hi is None else hi
    sl = slice(lo, hi)
    t0 = time.perf_counter()
    blk = dm.result_alloc(hi - lo, float(np.sum(tg.ell[sl])))
    t1 = time.perf_counter()
    dt = _capi.DeviceTracks(dm, tg.px[sl], tg.py[sl], tg.phi[sl], tg.cos_phi[sl], tg.sin_phi[sl], tg.A[sl], tg.B[sl], tg.C[sl], tg.ell[sl], tg.azim_idx[sl])
    t2 = time.perf_counter()
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    t3 = time.perf_counter()
    off, st, seg = dt.fetch_result(blk)
    t4 = time.perf_counter()
    ok = int(off[-1]) == total and len(seg["ell"]) == total
    dt.close()
    return {"result_alloc_ms": (t1 - t0) * 1e3, "tracks_h2d_ms": (t2 - t1) * 1e3, "segmentize_first_call_ms": (t3 - t2) * 1e3,
            "fetch_result_ms": (t4 - t3) * 1e3, "wall_ms": (t4 - t0) * 1e3, "records": int(total), "consistent": bool(ok)}


def one_shot(e):
    """e2e.one_shot_ms / fetch_fresh_ms (round 6): the wall clock of the real sequence above and its fetch; the sums of separately
    timed parts with the caller's own fresh arrays (rounds 4-5) stay beside them as *_caller_arrays_ms."""
    e["fetch_caller_arrays_ms"] = e["fetch_fresh_ms"]
    e["one_shot_caller_arrays_ms"] = e["tracks_h2d_ms"] + e["segmentize_ms"] + e["fetch_fresh_ms"]
    seq = e.get("one_shot_sequence")
    if seq and "wall_ms" in seq:
        e["one_shot_ms"] = seq["wall_ms"]
        e["fetch_fresh_ms"] = seq["fetch_result_ms"]
    else:
        e["one_shot_ms"] = e["one_shot_caller_arrays_ms"]
    if e.get("tracks_h2d_again_ms") is not None:
        e["one_shot_again_ms"] = e["tracks_h2d_again_ms"] + e["segmentize_ms"] + e["fetch_pinned_all_ms"]
    e["one_shot_note"] = ("one_shot_ms = wall clock of ONE timed sequence on an existing mesh handle: rt_result_alloc, rt_tracks_create, rt_segmentize "
                          "(the handle's first call), rt_result_fetch (`one_shot_sequence` has the parts; fetch_fresh_ms = its fetch: fresh host memory "
                          "that the library mapped inside the sequence and faults in beside the upload and the kernels); *_caller_arrays_ms = rounds 4-5's "
                          "numbers: rt_fetch_offsets + rt_fetch_segments into numpy arrays the caller has just allocated, and the sum of separately timed parts; "
                          "one_shot_again_ms = a later track set and the handle's page-locked buffers, already pinned (rt_fetch_pinned)")


def boundary_costs(dt, total):
    """What the boundary adds around one step when the caller wants host arrays (never part of `value`): all eight result
    arrays through the handle's page-locked buffers in one call (rt_fetch_pinned), and the older pair of calls beside it."""
    a = time.perf_counter()
    fresh = (dt.fetch_offsets(), dt.fetch_segments())  # into fresh pageable arrays, pipelined through the library's page-locked block
    fresh_ms = (time.perf_counter() - a) * 1e3
    del fresh
    a = time.perf_counter()
    dt.fetch_pinned()  # first use pins the handle's buffers (one-time: ≈0.08 ms per MB)
    pin_first_ms = (time.perf_counter() - a) * 1e3
    a = time.perf_counter()
    dt.fetch_pinned()
    all_ms = (time.perf_counter() - a) * 1e3
    a = time.perf_counter()
    dt.fetch_segments_pinned()
    seg_ms = (time.perf_counter() - a) * 1e3
    a = time.perf_counter()
    dt.fetch_offsets()
    off_ms = (time.perf_counter() - a) * 1e3
    return {"fetch_fresh_ms": fresh_ms, "fetch_pinned_first_ms": pin_first_ms, "fetch_pinned_all_ms": all_ms, "fetch_offsets_status_ms": max(all_ms - seg_ms, 0.0), "fetch_records_pinned_ms": seg_ms,
            "fetch_offsets_pageable_ms": off_ms, "fetch_GBs": 44.0 * total / (all_ms * 1e-3) / 1e9 if all_ms > 0 else 0.0,
            "note": "one call as the shim sees it: rt_mesh_create (once per mesh), rt_tracks_create (H2D of the track arrays), rt_segmentize, "
                    "rt_fetch_pinned (offsets, status and the 44-B records over PCIe into page-locked buffers, one synchronisation); "
                    "fetch_offsets_status_ms = what offsets + status add to the records' copy; fetch_offsets_pageable_ms = the older "
                    "rt_fetch_offsets into fresh pageable arrays"}


def sweep_bench(rt, tg, aq, dmesh, dt, total, steps, G=7):
    """a consumer that stays on the GPU and walks the cyclic tracks — one MOC transport sweep (rt_sweep) over the
    device-resident records, from the compact CSR records and from the march's staging rows directly.  With the latter a
    device-resident caller sets option "compact" = 0 and its step is march + offsets scan + sweep: no compaction.  Outside `value`."""
    nc = dmesh.n_cells
    sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
    src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
    dt.sweep_set_links(tg)
    out = {"groups": G, "note": "one transport sweep = every track forward and backward, per segment and group: τ = Σt·ℓ, "
                                "Δ = (ψ − q/Σt)(−expm1(−τ)), ψ −= Δ, φ[cell] += w·Δ; boundary fluxes handed on through "
                                "next_track_fwd/bwd + dir_next_track_* (Vacuum: 0); f64"}
    seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    # "compact": the CSR records (ℓ, cell) as a caller with the reference's layout holds them; "staged": a call that wrote no records
    # ("compact" = 0: k_materialise leaves (ℓ, cell) rows in the march's slot order instead); "auto": the default input after an
    # ordinary call — the rows are built from the staged words once per segmentation (`first_sweep_ms` includes that)
    # (round 6: the compact input is swept as rows too — "compact": as the library serves it by default, the staging's rows while the
    #  handle has them; "compact_rows_from_records": rows transposed once per segmentation from the CSR records themselves, what a
    #  handle without whole-track staging gets; "compact_in_place": the records where they lie, round 5's path)
    for key, compact, name, rows in (("compact", 1, "compact", 1), ("compact_rows_from_records", 1, "compact", 2), ("compact_in_place", 1, "compact", 0),
                                     ("staged", 0, "staged", 1), ("auto", 1, "auto", 1)):
        dmesh.set_option("compact", compact)
        dmesh.set_option("sweep_rows", rows)
        seg()
        r = dt.sweep(G, sig, src, None, None, input=name, fetch=False)
        ms = min(dt.sweep(G, input=name, fetch=False)["ms"] for _ in range(5))
        # per segment, direction and pass over a slab of groups: (ℓ, cell), 12 B — CSR records or rows
        row_bytes = 12.0
        nbytes = total * 2.0 * r["passes"] * row_bytes
        t0 = time.perf_counter()
        for _ in range(steps):
            seg()
            dt.sweep(G, input=name, fetch=False)
        step_ms = (time.perf_counter() - t0) / steps * 1e3
        # consecutive sweeps under the option "async": queued back to back (the solver's inner loop), wall clock per sweep
        dmesh.set_option("async", 1)
        dt.sweep(G, input=name, fetch=False)
        dt.wait()
        t0 = time.perf_counter()
        for _ in range(10):
            dt.sweep(G, input=name, fetch=False)
        dt.wait()
        b2b_ms = (time.perf_counter() - t0) / 10 * 1e3
        dmesh.set_option("async", 0)
        out[key] = {"input": r["input"], "rows": r.get("rows"), "sweep_ms": ms, "sweeps_back_to_back_ms": b2b_ms, "first_sweep_ms": r["ms"], "passes": r["passes"], "groups_per_pass": r["groups_per_pass"], "bytes_per_segment": 2.0 * r["passes"] * row_bytes,
                     "achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "segment_group_updates_per_s": total * 2.0 * G / (ms * 1e-3),
                     "ms_per_step_segmentize_plus_sweep": step_ms}
    dmesh.set_option("sweep_rows", 1)
    dmesh.set_option("compact", 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        seg()
    out["ms_per_step_without_compaction"] = (time.perf_counter() - t0) / steps * 1e3
    dmesh.set_option("compact", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        seg()
    out["ms_per_step_with_compaction"] = (time.perf_counter() - t0) / steps * 1e3
    return out


def cpu_share():
    """CPUs this process may use: the cgroup's quota (cpu.max) where there is one, else the hardware threads."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(round(float(q) / float(p)))))
    except Exception:
        pass
    return n


def cpu_baseline(tg):
    """Oracle (C port of the reference's algorithm, libm trig per advance_step as the
    reference does) on the host cores.  Checker code used as a reported baseline only.
    The reference's segmentize! is one thread (src/trackgenerator.jl:362-364: a plain loop over tracks_by_uid): `single_core` is the
    like-for-like number; `value` is the port parallelised over tracks with OpenMP, at the thread count that does best among the
    box's CPU share (a GPU box gives a job a cgroup quota of 16 CPUs of its 256 hardware threads: 256 threads burn a period's quota
    in an eighth of it and are throttled for the rest), four times the share, and all hardware threads — `legs` has all three.  Timed:
    the march into the port's own per-track storage (every record materialised), not the serial copy into numpy arrays behind it."""
    from oracle import oracle as orc

    orc.build()
    hw = int(orc.num_threads())
    share = cpu_share()
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    legs = []
    for nt in sorted({min(hw, share), min(hw, 4 * share), hw}):
        t0 = time.perf_counter()
        r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, tiny_step=tg.tiny_step, n_threads=nt, fetch=False)
        dt = time.perf_counter() - t0
        legs.append({"threads": nt, "value": r["total"] / dt, "seconds": dt})
        time.sleep(0.25)  # (a fresh quota period for the next leg)
    best = max(legs, key=lambda l: l["value"])
    # single-core leg: every 4th track (bounded)
    sel = np.arange(0, tg.n_total_tracks, 4)
    om1 = orc.OracleMesh.from_mesh(tg.mesh, omp=False)
    t0 = time.perf_counter()
    r1 = om1.segmentize(tg.px[sel], tg.py[sel], tg.phi[sel], tg.A[sel], tg.B[sel], tg.C[sel], tg.ell[sel],
                        tiny_step=tg.tiny_step, n_threads=1, fetch=False)
    dt_1 = time.perf_counter() - t0
    return {
        "value": best["value"], "unit": "segments/s", "cores": int(best["threads"]), "kind": "port",
        "sample": "full workload once per leg (%d tracks, %d segments), OpenMP over tracks, best of %d thread counts: %d threads, %.2f s wall"
                  % (tg.n_total_tracks, r["total"], len(legs), best["threads"], best["seconds"]),
        "cpu_share": share, "hardware_threads": hw, "legs": legs,
        "single_core": {"value": r1["total"] / dt_1, "unit": "segments/s", "cores": 1,
                        "sample": "every 4th track (%d tracks, %d segments), %.2f s" % (len(sel), r1["total"], dt_1)},
    }


def _launch_ranks(n, real_stdout):
    """`python bench.py --gpus N` without a launcher around it: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    bench.py <same arguments>` as a child process on 127.0.0.1 and a free port, pass its stdout (rank 0's JSON line) through, its
    stderr too, and return its exit code."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on these hosts)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, env=env, text=True)
    for line in proc.stdout:
        real_stdout.write(line)
        real_stdout.flush()
    return proc.wait()


def main():
    # RCCL prints a version banner on stdout when a communicator is created: keep the real stdout for the
    # one JSON line and send everything else that writes to fd 1 to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rc = 0
    try:
        rc = _main(real_stdout)
    finally:
        real_stdout.flush()
    return rc or 0


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3", "c5", "custom"],
                    help="auto: c3 (headline) on one GPU, c5 (strong scaling) on several; custom: --mesh/--n-azim/--delta")
    ap.add_argument("--n-azim", type=int, default=128)
    ap.add_argument("--delta", type=float, default=1e-3)
    ap.add_argument("--mesh", default="pincell.msh")
    ap.add_argument("--latency-steps", type=int, default=200, help="extra single steps for the p50/p95 latency (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the extra two-batches-in-flight measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip e2e, latency, config5_single_gpu / single_gpu_same_workload")
    ap.add_argument("--sharded-sweep", action="store_true",
                    help="N > 1: also time the sharded transport sweep (rt_sweep per rank + point-to-point exchange of the fluxes that leave a "
                         "shard); off by default: its RCCL exchange has only run over gloo and a one-rank group so far, and the headline line "
                         "must not depend on it")
    ap.add_argument("--extras-timeout", type=int, default=300,
                    help="N > 1: seconds after which rank 0 prints the headline line without the extras (all-gather-v, same problem on one "
                         "GPU) if they have not returned, and every rank leaves; 0: wait for ever")
    ap.add_argument("--force-dist", action="store_true", help="development: run the multi-GPU code path with a one-rank RCCL group")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the launcher — it starts the N ranks with torch.distributed.run as a
        # CHILD process (no exec; it has not imported torch, loaded the library or touched a GPU), relays the child's one JSON line
        # and leaves with the child's exit code.
        return _launch_ranks(args.gpus, real_stdout)

    import torch
    import torch.distributed as dist

    import raytracing_jl_amd as rt
    from raytracing_jl_amd import _capi
    from raytracing_jl_amd import distributed as rtd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and (world > 1 or args.gpus > 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal of the N > 1 code path on a box with ONE GPU (development; RT_BENCH_REHEARSAL=1): every rank uses GPU 0 and
    # the collectives run over gloo on host copies — RCCL refuses two ranks on one device.  Timings of such a run mean nothing.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the small collective payloads live
    if rehearsal:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=dev)
    elif args.force_dist:  # development: the multi-GPU code path with a one-rank RCCL group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist_on = world > 1 or args.force_dist

    # ---- workload (deterministic, no RNG): host-side trace! of the GLOBAL problem, then upload the rank's uid range
    wkey = args.workload if args.workload != "auto" else ("c3" if world == 1 else "c5")
    wl = WORKLOADS[wkey] if wkey in WORKLOADS else dict(mesh=args.mesh, n_azim=args.n_azim, delta=args.delta,
                                                        name="%s, nφ=%d, δ=%g" % (args.mesh, args.n_azim, args.delta))
    tg = make_tg(rt, wl)
    aq = tg.azimuthal_quadrature
    t0 = time.perf_counter()
    dmesh = _capi.DeviceMesh(tg.mesh, local_rank)
    mesh_create_ms = (time.perf_counter() - t0) * 1e3
    # one explicit stream for everything: the library's kernels, torch's own ops and the stream-level waits
    # of the pipelined all-reduce (torch's default stream has handle 0, which the library reads as "use
    # your own stream" — the waits would then order nothing)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dmesh.set_stream(stream.cuda_stream)
    t0 = time.perf_counter()
    dt, (lo, hi) = rtd.segmentize_shard(tg, rank, world, device=local_rank, dmesh=dmesh)
    tracks_h2d_ms = (time.perf_counter() - t0) * 1e3

    # fill_volumes is the one reduction across tracks: every rank's partial `volumes` are summed by an RCCL
    # all-reduce per step, in place in the library's buffer.  It is pipelined by one step: the library
    # alternates between two volumes buffers, and the all-reduce of step i's buffer is issued from the enqueue
    # hook of step i+1 (when step i+1's kernels are already queued) on a side stream — its launch and its
    # latency sit beside the next march instead of between two steps.  The last one is waited for inside the
    # timed region (`drain`).
    if dist_on:
        if rehearsal:
            def _host_all_reduce(vol):  # gloo: through a host copy, synchronously
                h = vol.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                vol.copy_(h)
                return None
            pipe = rtd.PipelinedVolumesAllReduce(device=None, all_reduce=_host_all_reduce)
        else:
            pipe = rtd.PipelinedVolumesAllReduce(device=dev)
        views = {}
        dmesh.set_enqueue_hook(pipe.hook)

    def step():
        if dist_on:
            k = pipe.before_call()  # the library alternates between two volumes buffers: this call writes buffer k
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)  # hook: all-reduce of the previous step
        if dist_on:
            if k not in views:  # first two calls: wrap the buffer once
                ptr = dt.device_pointers()["volumes"]
                views[k] = torch.as_tensor(rtd.DevArray(ptr, dmesh.n_cells, "<f8", dt), device=dev)
            pipe.after_call(k, views[k])
        return total

    def drain():
        """Issue and finish the all-reduce still owed for the last step (inside the timed region)."""
        if dist_on:
            pipe.drain()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        local_total = step()
    drain()
    sync()
    # ---- the timed region: K steps, no HIP events between the kernels (each costs ≈4 µs of stream time)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        local_total = step()
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    # ---- the same K steps again with the library's HIP events on (option "timing"): per-kernel durations, measured live on
    #      the stream the kernels are launched on — what `roofline` and `kernel_ms` report; never part of `value`
    kern = {"march": 0.0, "compact": 0.0, "scan": 0.0, "volumes": 0.0, "plan": 0.0, "total": 0.0}
    dmesh.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = dt.timing()  # HIP events recorded on the launch stream inside rt_segmentize
        for k in kern:
            kern[k] += tm[k]
    drain()
    sync()
    elapsed_with_events = time.perf_counter() - t0
    dmesh.set_option("timing", 0)
    n_failed, _, _ = dt.failed()  # tracks on which the reference itself would have thrown (never fatal here:
    #                               a rank that exits alone would deadlock the others)
    stats = dt.stats()
    info = dmesh.info()
    info_cus = torch.cuda.get_device_properties(dev).multi_processor_count

    # ---- N > 1: the same K steps once more WITHOUT the volumes all-reduce — what the pipelined all-reduce still costs a step
    #      (its launch and whatever of its latency the next march does not cover) is the difference; load balance per rank
    per_rank = None
    allreduce_exposed_ms = None
    if dist_on:
        dmesh.set_enqueue_hook(None)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        sync()
        elapsed_no_allreduce = time.perf_counter() - t0
        dmesh.set_enqueue_hook(pipe.hook)
        mine = torch.tensor([float(local_total), float(hi - lo), elapsed / args.steps * 1e3, elapsed_no_allreduce / args.steps * 1e3,
                             kern["march"] / args.steps], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        allr = torch.stack(allr).cpu().numpy()
        per_rank = {"segments": [int(v) for v in allr[:, 0]], "tracks": [int(v) for v in allr[:, 1]],
                    "ms_per_step": [float(v) for v in allr[:, 2]], "march_ms": [float(v) for v in allr[:, 4]]}
        allreduce_exposed_ms = float(max(allr[:, 2]) - max(allr[:, 3]))
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    tot = torch.tensor([float(local_total), float(n_failed)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    t_max = float(t_max.item())
    global_segments = float(tot[0].item())
    failed_tracks = int(tot[1].item())

    # ---- N > 1: nothing after the timed region may cost the run its line.  The extras below use collectives that have never run
    #      between several GPUs here (point-to-point all-gather-v); should one of them not return, rank 0 prints the headline
    #      line without them after `--extras-timeout` seconds and every rank leaves.
    import threading
    line_lock = threading.Lock()
    line_printed = [False]
    watchdog = None
    if world > 1 and args.extras_timeout > 0:
        def _give_up():
            with line_lock:
                if line_printed[0]:
                    return
                line_printed[0] = True
                if rank == 0:
                    ms_step = t_max / args.steps * 1e3
                    per_k = {key: kern[key] / args.steps for key in ("march", "compact")}
                    dom = "march" if per_k["march"] >= per_k["compact"] else "compact"  # (the dominant kernel by HIP-event time)
                    bps_fb = dict(bytes_per_segment(stats), march=STEP_BYTES_PER_SEGMENT)
                    ach = bps_fb[dom] * local_total / (per_k[dom] * 1e-3) / 1e9 if per_k[dom] > 0 else 0.0
                    fb = {"metric": "segments/sec (whole node)", "value": global_segments * args.steps / t_max, "unit": "segments/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic (deterministic tracks from trace! on the mesh; no RNG)",
                          "config": {"workload": "%s; FIXED global problem split over %d GPUs (strong scaling)" % (wl["name"], world),
                                     "tracks_global": int(tg.n_total_tracks), "segments_global": int(global_segments),
                                     "failed_tracks": failed_tracks},
                          "roofline": {"definition_version": 3, "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_segment": bps_fb[dom],
                                       "segments_per_launch": int(local_total), "kernel_ms_avg": per_k[dom]},
                          "kernel_ms": {k: v / args.steps for k, v in kern.items()}, "per_rank": per_rank,
                          "allreduce_ms_exposed": allreduce_exposed_ms,
                          "extras": "not reported: an extra after the timed region did not return within %d s" % args.extras_timeout}
                    real_stdout.write(json.dumps(fb, ensure_ascii=False) + "\n")
                    real_stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(args.extras_timeout, _give_up)
        watchdog.daemon = True
        watchdog.start()

    # ---- N > 1: reassemble the global segment list on every rank (RCCL all-gather-v), always reported
    allgather = None
    if world > 1 or args.force_dist:
        p = dt.device_pointers()
        off = torch.as_tensor(rtd.DevArray(p["offsets"], dt.n + 1, "<i8", dt), device=dev)
        local = {"counts": off[1:] - off[:-1]}
        for name in ("px", "py", "qx", "qy", "ell"):
            local[name] = torch.as_tensor(rtd.DevArray(p[name], local_total, "<f8", dt), device=dev)
        local["element"] = torch.as_tensor(rtd.DevArray(p["element"], local_total, "<i4", dt), device=dev)
        if rehearsal:
            local = {k: v.cpu() for k, v in local.items()}
        try:
            gather = rtd.SegmentGather()
            gather(local)  # warm-up (allocates the final buffers, opens the peer connections)
            sync()
            reps = 3
            g0 = time.perf_counter()
            for _ in range(reps):
                g = gather(local)
            sync()
            g_ms = (time.perf_counter() - g0) / reps * 1e3
            gt = torch.tensor([g_ms], dtype=torch.float64, device=cdev)
            if world > 1:
                dist.all_reduce(gt, op=dist.ReduceOp.MAX)
            assert int(g["offsets"][-1].item()) == int(global_segments)
            g_ms = float(gt.item())
            allgather = {"ms": g_ms, "bytes_received_per_rank": 44.0 * (global_segments - local_total),
                         "GBs_received_per_rank": 44.0 * (global_segments - local_total) / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0,
                         "segments_per_s_including_allgather": global_segments / (t_max / args.steps + g_ms / 1e3),
                         "method": "batched RCCL send/recv pairs straight into the final buffers (all-gather-v, no padding)"}
            del g, gather
        except Exception as e:  # the headline line must survive a failure of the extra
            allgather = {"error": repr(e)}

    # ---- N > 1: the consumer that never gathers — a transport sweep over every rank's own tracks (rt_sweep on the shard's staging
    #      rows), the fluxes that leave a shard exchanged by RCCL send/recv pairs, the tallies all-reduced (distributed.ShardedSweep)
    sharded_sweep = None
    if dist_on and args.sharded_sweep and not args.no_extras:
        if rehearsal:
            sharded_sweep = {"skipped": "rehearsal (the exchange needs device tensors on an RCCL group; tests/test_gpu_sharded_sweep.py covers it over gloo)"}
        else:
            try:
                G = 7
                nc = dmesh.n_cells
                sig = np.linspace(0.2, 1.6, nc * G).reshape(nc, G)
                src = np.linspace(0.0, 1.0, nc * G).reshape(nc, G)
                ss = rtd.ShardedSweep(tg, dt, rank, world, device=dev)
                ss.sweep(G, sig, src, None, np.ones((2, hi - lo, G)), input="auto")  # warm-up: uploads, peer connections
                sync()
                reps, sw_ms = 3, 0.0
                t0 = time.perf_counter()
                for _ in range(reps):
                    sw_ms += ss.sweep(G)["ms"]
                sync()
                wall_ms = (time.perf_counter() - t0) / reps * 1e3
                tt = torch.tensor([wall_ms, sw_ms / reps], dtype=torch.float64, device=cdev)
                if world > 1:
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                n_cross = sum(len(v[0]) for v in ss.plan.send.values())
                sharded_sweep = {"groups": G, "ms_per_sweep": float(tt[0].item()), "rt_sweep_kernel_ms": float(tt[1].item()),
                                 "fluxes_sent_by_rank0": int(n_cross), "bytes_sent_by_rank0": int(n_cross) * G * 8,
                                 "segment_group_updates_per_s": global_segments * 2.0 * G / (float(tt[0].item()) * 1e-3),
                                 "note": "every rank sweeps its own uid range (forward and backward over the staging rows), sends the fluxes whose "
                                         "linked track lives on another rank (point-to-point), all-reduces the tallies; slowest rank; outside `value`"}
            except Exception as e:  # the headline line must survive a failure of the extra
                sharded_sweep = {"error": repr(e)}

    # ---- extras measured outside the timed region (rank 0)
    latency = e2e = same_workload = config5 = None
    if rank == 0 and not args.no_extras and not dist_on:
        # single-step latency distribution
        lat = []
        for _ in range(max(args.latency_steps, 0)):
            a = time.perf_counter()
            dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
            lat.append((time.perf_counter() - a) * 1e3)
        if lat:
            q = np.percentile(lat, [50, 95, 99])
            latency = {"steps": len(lat), "ms_p50": float(q[0]), "ms_p95": float(q[1]), "ms_p99": float(q[2]), "ms_min": float(min(lat)),
                       "ms_max": float(max(lat)), "note": "host clock around single synchronous rt_segmentize calls, after the timed region"}
        # the boundary's real costs around one step (host buffers in, host buffers out)
        a = time.perf_counter()
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        seg_ms = (time.perf_counter() - a) * 1e3
        e2e = boundary_costs(dt, local_total)
        try:
            e2e["one_shot_sequence"] = one_shot_sequence(rt, _capi, dmesh, tg, aq, lo, hi)
        except Exception as e:  # pragma: no cover
            e2e["one_shot_sequence"] = {"error": repr(e)}
        e2e.update({"mesh_create_ms": mesh_create_ms, "mesh_prep_host_ms": info["prep_ms"], "tracks_h2d_ms": tracks_h2d_ms, "segmentize_ms": seg_ms})
        try:
            for _ in range(2):
                a = time.perf_counter()
                dt2 = _capi.DeviceTracks(dmesh, tg.px[lo:hi], tg.py[lo:hi], tg.phi[lo:hi], tg.cos_phi[lo:hi], tg.sin_phi[lo:hi], tg.A[lo:hi], tg.B[lo:hi],
                                         tg.C[lo:hi], tg.ell[lo:hi], tg.azim_idx[lo:hi])
                e2e["tracks_h2d_again_ms"] = (time.perf_counter() - a) * 1e3
                dt2.close()
        except Exception as e:  # pragma: no cover
            e2e["tracks_h2d_again_ms"] = None
        one_shot(e2e)
    downstream_sweep = None
    if rank == 0 and not args.no_extras and not dist_on:
        try:
            downstream_sweep = sweep_bench(rt, tg, aq, dmesh, dt, local_total, max(3, args.steps // 2))
        except Exception as e:  # pragma: no cover
            downstream_sweep = {"error": repr(e)}
    downstream = None
    if rank == 0 and not args.no_extras and not dist_on:
        # a consumer that stays on the GPU: Segment.τ for 7 energy groups from the device-resident records
        try:
            G = 7
            sig = np.linspace(0.2, 1.6, dmesh.n_cells * G).reshape(dmesh.n_cells, G)
            dt.fill_tau(sig, fetch=False)
            ms_tau = min(dt.fill_tau(sig, fetch=False)[2] for _ in range(5))
            nbytes = local_total * (12.0 + 8.0 * G)  # ℓ + element read, G values written per segment
            downstream = {"kernel": "rt::k_fill_tau", "groups": G, "ms": ms_tau, "bytes_per_segment": 12.0 + 8.0 * G,
                          "achieved_GBs": nbytes / (ms_tau * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (ms_tau * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "note": "τ[s,g] = Σt[element[s],g]·ℓ[s] over the records where they lie (rt_fill_tau); outside `value`"}
        except Exception as e:  # pragma: no cover
            downstream = {"error": repr(e)}
    if not args.no_extras and world > 1:
        # the same global problem on rank 0's GPU alone, in the same run: the N=1 point of this strong-scaling line
        sync()
        if rank == 0:
            same_workload = single_gpu_run(rt, _capi, wl, local_rank, max(3, args.steps // 4), 2, stream.cuda_stream, tg=tg)
        sync()
    if rank == 0 and not args.no_extras and world == 1 and not dist_on and wkey == "c3":
        try:
            config5 = single_gpu_run(rt, _capi, WORKLOADS["c5"], local_rank, 5, 2, stream.cuda_stream, e2e=True)
        except Exception as e:  # pragma: no cover
            config5 = {"error": repr(e)}

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        names = kernel_names(stats)
        traffic, traffic_src = pmc_traffic()
        per_kernel = []
        bps_by_phase = bytes_per_segment(stats)
        for key in ("march", "compact", "scan"):
            ms = kern[key] / args.steps
            bps = bps_by_phase[key]
            per_kernel.append({"kernel": names[key], "ms_avg": ms, "bytes_per_segment": bps,
                               "achieved_GBs": bps * local_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                               "traffic": traffic.get(names[key]) if wkey == "c3" and world == 1 else None})
        # The roofline's kernel is the one that moves SURVEY §8(d)'s bytes — the record-writing kernel (k_materialise_lin: 4 B read +
        # 44 B written per segment; k_compact3 for calls that march with exact steps) — priced with its own bytes and its own HIP-event
        # time.  The march writes 4 B per segment and is bound by its lanes' dependent chains: it is reported as such (`march`: real
        # bytes and VALU issue by counters when the committed PMC passes belong to this library), never priced with bytes it does not move.
        rec_k = next(k for k in per_kernel if k["kernel"] == names["compact"])
        march_k = next(k for k in per_kernel if k["kernel"] == names["march"])
        # definition_version 3 (round 6): `roofline` is the DOMINANT kernel by HIP-event time again (rounds 1-4; round 5 named the
        # record kernel whatever its share) — SURVEY §8(d)'s 45 B per segment x the segments of a launch ÷ that kernel's duration when
        # it is the march (which stages 4 B per segment: `march` says what it is bound by), the record kernel's own 48 B when it is
        # the record kernel; the record kernel always has its own object, `roofline.record_kernel`.
        dom_is_march = march_k["ms_avg"] >= rec_k["ms_avg"]
        dom_k = march_k if dom_is_march else rec_k
        dom_bps = STEP_BYTES_PER_SEGMENT if dom_is_march else rec_k["bytes_per_segment"]
        achieved = dom_bps * local_total / (dom_k["ms_avg"] * 1e-3) / 1e9 if dom_k["ms_avg"] > 0 else 0.0
        step_GBs = STEP_BYTES_PER_SEGMENT * global_segments / (ms_per_step * 1e-3) / 1e9
        tr_all = [k["traffic"] for k in per_kernel]
        traffic_ratio = (sum(tr_all) / (STEP_BYTES_PER_SEGMENT * local_total)) if all(t is not None for t in tr_all) else None
        march_pmc = PMC_KERNELS.get(names["march"], {}) if (wkey == "c3" and world == 1) else {}
        march_report = {"kernel": march_k["kernel"], "bound": "latency/issue", "ms_avg": march_k["ms_avg"],
                        "bytes_by_counters": march_k["traffic"],
                        "achieved_GBs_by_counters": (march_k["traffic"] / (march_k["ms_avg"] * 1e-3) / 1e9) if march_k["traffic"] and march_k["ms_avg"] > 0 else None,
                        "frac_of_hbm_peak_by_counters": (march_k["traffic"] / (march_k["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if march_k["traffic"] and march_k["ms_avg"] > 0 else None,
                        # SQ_INSTS_VALU wave-instructions x 4 issue cycles over (SIMDs x the kernel's cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                        "valu_issue_frac": (march_pmc["SQ_INSTS_VALU"] * 4.0 / (4 * info_cus * march_pmc["GRBM_GUI_ACTIVE"] / 8.0))
                                           if "SQ_INSTS_VALU" in march_pmc and march_pmc.get("GRBM_GUI_ACTIVE") else None,
                        "wait_frac": (march_pmc["SQ_WAIT_ANY"] / march_pmc["SQ_WAVE_CYCLES"]) if march_pmc.get("SQ_WAVE_CYCLES") else None,
                        "note": "one lane = one track: the kernel lasts as long as its longest lane's chain of dependent 32-B fetches; "
                                "it stages 4 B per segment"}
        out = {
            "metric": "segments/sec (whole node)",
            "value": global_segments * args.steps / t_max,
            "unit": "segments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (deterministic tracks from trace! on the mesh; no RNG)",
            "config": {
                "workload": "%s; all tracks of tracks_by_uid; segmentize! = march + CSR offsets + compact segment records + "
                            "fill_volumes%s" % (wl["name"], "; FIXED global problem split over %d GPUs (strong scaling)" % world if world > 1 else ""),
                "tracks_global": int(tg.n_total_tracks), "segments_global": int(global_segments),
                "tracks_rank0": int(hi - lo), "segments_rank0": int(local_total),
                "sharding": "contiguous uid ranges balanced by Σℓ; all-reduce(sum) of volumes inside the step" if world > 1 else "none",
                "tiny_step": tg.tiny_step, "k": 5, "rtol": rt.RTOL_DEFAULT, "failed_tracks": failed_tracks,
                "march_plan": {0: "whole tracks", 1: "every track in pieces"}[stats["split"]],
                "regime": {"walk_enabled": info["walk_enabled"], "records_walkable": info["records_walk"], "records": info["records"],
                           "walk_records_rank0": stats["walk_records"], "generic_records_rank0": stats["generic_records"],
                           "cheap_records_rank0": stats["cheap_records"], "two_phase": stats["cheap_records"] > 0,
                           "cheap_refusals_rank0": stats["cheap_refusals"], "tracks_restarted_rank0": stats["tracks_restarted"],
                           # tracks whose Σℓ check (src/track.jl:171) lies within summation-order noise of its threshold: Julia's
                           # pairwise / @simd sum could decide them the other way — what cannot be pinned without the real package
                           "tracks_near_rtol_rank0": stats["tracks_near_rtol"],
                           # cheap records whose fill_volumes term the second kernel added from the record's own length
                           "records_tallied_from_lengths_rank0": stats["records_tallied_from_lengths"]},
                "device_GB_held_by_rank0_handle": stats["device_bytes"] / 1e9,
                "library_sha256": lib_sha256(),
            },
            "roofline": {
                "definition_version": 3,
                "bound": "hbm", "kernel": dom_k["kernel"],
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": dom_k["traffic"], "traffic_source": traffic_src,
                "bytes_per_segment": dom_bps, "segments_per_launch": int(local_total),
                "kernel_ms_avg": dom_k["ms_avg"],
                "record_kernel": {"kernel": rec_k["kernel"], "bound": "hbm", "achieved": rec_k["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": rec_k["achieved_GBs"] / HBM_PEAK_GBS, "traffic": rec_k["traffic"],
                                  "bytes_per_segment": rec_k["bytes_per_segment"], "kernel_ms_avg": rec_k["ms_avg"],
                                  "note": "the kernel that moves SURVEY §8(d)'s bytes: its own 4 B read + 44 B written per segment ÷ its HIP-event time"},
                "pipeline_frac": step_GBs / (HBM_PEAK_GBS * world),
                "traffic_over_algorithmic": traffic_ratio,
                "march": march_report,
                "pipeline": {"achieved": step_GBs, "frac": step_GBs / (HBM_PEAK_GBS * world), "unit": "GB/s",
                             "bytes_per_segment": STEP_BYTES_PER_SEGMENT, "ms_per_step": ms_per_step,
                             "traffic_over_algorithmic": traffic_ratio,
                             "note": "whole step: algorithmic bytes (45 B/segment) ÷ ms_per_step against the peak of the GPUs in use"},
                "note": "kernel = the longer of march and record kernel by HIP-event time (definition_version 3); the march is priced with "
                        "SURVEY §8(d)'s 45 B/segment (it stages 4 B/segment and is bound by its per-track dependent chain: see `march`), the record "
                        "kernel with its own 48 B (`record_kernel`); pipeline_frac = 45 B/segment ÷ ms_per_step",
            },
            "kernels": per_kernel,
            "kernel_ms": {k: v / args.steps for k, v in kern.items()},
            "kernel_ms_note": "from a second pass of the same K steps with the library's HIP events on (option \"timing\"); "
                              "that pass took %.4f ms per step" % (elapsed_with_events / args.steps * 1e3),
            # what a synchronous call adds around its kernels: completion seen by the host, the return through the caller, the next
            # call's launches (the kernels of one call run back to back)
            "host_gap_us": (ms_per_step - (kern["march"] + kern["scan"] + kern["compact"] + kern["volumes"]) / args.steps) * 1e3 if world == 1 else None,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank  # load balance of the Σℓ-balanced uid ranges
            for r, v in enumerate(per_rank["segments"]):
                out["config"]["segments_rank%d" % r] = v
            out["allreduce_ms_exposed"] = allreduce_exposed_ms
            out["allreduce_note"] = ("slowest rank's ms per step with the pipelined volumes all-reduce minus the same K steps without "
                                     "any all-reduce (max over ranks each): what the reduction adds to a step")
        if rehearsal:
            out["rehearsal"] = "RT_BENCH_REHEARSAL=1: all ranks on GPU 0, collectives over gloo on host copies — a functional run, its timings mean nothing"
        if latency is not None:
            out["latency"] = latency
        if e2e is not None:
            out["e2e"] = e2e
        if allgather is not None:
            out["allgather"] = allgather
        if sharded_sweep is not None:
            out["sharded_sweep"] = sharded_sweep
        if same_workload is not None:
            out["single_gpu_same_workload"] = same_workload
            out["speedup_vs_single_gpu"] = same_workload["ms_per_step"] / ms_per_step
        if dist_on:
            # what a sub-linear curve is made of, in the line itself: a rank's shard is a smaller batch, and the march of a batch that
            # no longer fills the chip lasts as long as its longest track's chain whatever the number of waves
            waves = (int(hi - lo) + 63) // 64
            simds = 4 * info_cus
            out["shard_regime"] = {
                "march_waves_rank0": waves, "simds": simds, "waves_per_simd_rank0": waves / simds,
                "resident_waves_per_simd": 2,  # k_march<..., TOPO>: 221 VGPRs
                "march_rounds_rank0": max(1.0, waves / (2.0 * simds)),
                "ideal_ms_per_step": (same_workload["ms_per_step"] / world) if same_workload is not None else None,
                "note": "strong scaling of a FIXED problem: with N ranks a shard has 1/N of the waves; below ~2 waves per SIMD every wave is "
                        "resident from the start and the march is bound by the longest track's dependent chain (it does not shorten with N), "
                        "while k_materialise / the scans scale with the shard — the step tends to (chain) + (records/N)/(HBM rate); "
                        "ideal_ms_per_step = single_gpu_same_workload / N",
            }
        if config5 is not None:
            out["config5_single_gpu"] = config5
        if downstream is not None:
            out["downstream_tau"] = downstream
        if downstream_sweep is not None:
            out["downstream_sweep"] = downstream_sweep
        if world == 1 and not dist_on and not args.no_concurrent:
            out["two_batches_in_flight"] = two_in_flight(tg, aq, dmesh, dt, args.steps, local_total)
            out["stream_ordered_calls"] = stream_ordered_calls(rt, tg, aq, dmesh, dt, args.steps, local_total)
            out["completion_order_calls"] = completion_order_calls(rt, tg, aq, dmesh, args.steps, local_total)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tg)
        with line_lock:
            if not line_printed[0]:
                line_printed[0] = True
                real_stdout.write(json.dumps(out, ensure_ascii=False) + "\n")
                real_stdout.flush()
    if watchdog is not None:
        watchdog.cancel()
    if dist_on:
        dmesh.set_enqueue_hook(None)
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
