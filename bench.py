#!/usr/bin/env python3
"""Benchmark of the HIP segmentize! path — BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full ``segmentize!`` (``rt_segmentize``: device march of every track, CSR
offsets, compact segment records, fused ``fill_volumes``) over a batch of tracks that is
already resident in HBM.  N=1: the headline configuration, ``demo/pincell.msh`` at nφ=128,
δ=1e-3 (130,456 tracks, 9.32 M segments).  N>1 (launched by ``torch.distributed.run``, one
rank per GPU): weak scaling — the global problem is the same mesh at nφ=128, δ=1e-3/N (≈N×
the tracks), ``tracks_by_uid`` is split into N contiguous uid ranges balanced by Σℓ, every
rank marches its own range, and the only data-path exchange of the reference's algorithm
(``fill_volumes``' cross-track sum) is one RCCL all-reduce of ``n_cells`` doubles inside the
timed step.  Reassembling the global segment list on every rank (an RCCL all-gather of the
segment arrays) is measured after the timed region and reported under ``"allgather"``.

The JSON line also carries ``roofline`` (dominant kernel, algorithmic bytes ÷ HIP-event
duration against the 8 TB/s HBM peak) and, at N=1, ``cpu_baseline`` (the oracle — a C port
of the reference's algorithm — timed on the host cores of the same box).
"""
from __future__ import annotations

import argparse
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
# SURVEY.md §8(d)'s: 44 B of record written + ≈1.3 B of amortised track input (the march itself stages
# only 20 B of it: q and the cell).  The compaction reads those 20 B from the staging pool, rebuilds p
# and ℓ, and writes the 44-B record to its CSR position.
KERNELS = {
    "march": ("rt::k_march<2, 4, false>", 45.0),   # single-pass staged march, fill_volumes fused (LDS-private)
    "compact": ("rt::k_compact3<false>", 64.0),
    "scan": ("rt::k_scan_write", 0.0),  # three small kernels: CSR offsets of the counts; volumes ./= n_azim_2 rides along
}
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")


def hbm_traffic_per_launch(kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01): FETCH_SIZE and
    WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads and is
    doubled (MI355X_MICROARCH.md §HBM).  None when no profile of this build is committed."""
    try:
        d = json.load(open(PMC_SUMMARY))
        k = d["kernels"][kernel_name]
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


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


def cpu_baseline(tg, max_seconds=30.0):
    """Oracle (C port of the reference's algorithm, libm trig per advance_step as the
    reference does) on the host cores.  Checker code used as a reported baseline only."""
    from oracle import oracle as orc

    orc.build()
    cores = orc.num_threads()
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    # all-core leg: the full workload once (≈10–30 core-seconds on this box class)
    t0 = time.perf_counter()
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, tiny_step=tg.tiny_step, n_threads=0)
    dt_all = time.perf_counter() - t0
    # single-core leg: every 4th track (bounded)
    sel = np.arange(0, tg.n_total_tracks, 4)
    om1 = orc.OracleMesh.from_mesh(tg.mesh, omp=False)
    t0 = time.perf_counter()
    r1 = om1.segmentize(tg.px[sel], tg.py[sel], tg.phi[sel], tg.A[sel], tg.B[sel], tg.C[sel], tg.ell[sel],
                        tiny_step=tg.tiny_step, n_threads=1)
    dt_1 = time.perf_counter() - t0
    return {
        "value": r["total"] / dt_all, "unit": "segments/s", "cores": int(cores), "kind": "port",
        "sample": "full workload once (%d tracks, %d segments), OpenMP over tracks, %.2f s wall"
                  % (tg.n_total_tracks, r["total"], dt_all),
        "single_core": {"value": r1["total"] / dt_1, "unit": "segments/s", "cores": 1,
                        "sample": "every 4th track (%d tracks, %d segments), %.2f s" % (len(sel), r1["total"], dt_1)},
    }


def main():
    # RCCL prints a version banner on stdout when a communicator is created: keep the real stdout for the
    # one JSON line and send everything else that writes to fd 1 to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    try:
        _main(real_stdout)
    finally:
        real_stdout.flush()


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-azim", type=int, default=128)
    ap.add_argument("--delta", type=float, default=1e-3)
    ap.add_argument("--mesh", default="pincell.msh")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the extra two-batches-in-flight measurement")
    ap.add_argument("--allgather", action="store_true",
                    help="N>1: also time the RCCL all-gather that reassembles the global segment list on every rank (after the timed region)")
    ap.add_argument("--no-allgather", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--force-dist", action="store_true", help="development: run the multi-GPU code path with a one-rank RCCL group")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracing_jl_amd as rt
    from raytracing_jl_amd import _capi
    from raytracing_jl_amd import distributed as rtd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    elif args.force_dist:  # development: the multi-GPU code path with a one-rank RCCL group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    # ---- workload (deterministic, no RNG): host-side trace! then upload the rank's uid range
    mesh_file = rt.data_path(args.mesh)
    model = rt.GmshDiscreteModel(mesh_file) if mesh_file.endswith(".msh") else rt.DiscreteModelFromFile(mesh_file)
    delta = args.delta / world
    tg = rt.TrackGenerator(model, args.n_azim, delta)
    rt.trace(tg)
    aq = tg.azimuthal_quadrature
    dmesh = _capi.DeviceMesh(tg.mesh, local_rank)
    # one explicit stream for everything: the library's kernels, torch's own ops and the stream-level waits
    # of the pipelined all-reduce (torch's default stream has handle 0, which the library reads as "use
    # your own stream" — the waits would then order nothing)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dmesh.set_stream(stream.cuda_stream)
    dt, (lo, hi) = rtd.segmentize_shard(tg, rank, world, device=local_rank, dmesh=dmesh)

    # fill_volumes is the one reduction across tracks: every rank's partial `volumes` are summed by an RCCL
    # all-reduce per step, in place in the library's buffer.  It is pipelined by one step: the library
    # alternates between two volumes buffers, and the all-reduce of step i's buffer is issued from the enqueue
    # hook of step i+1 (when step i+1's kernels are already queued) on a side stream — its launch and its
    # latency sit beside the next march instead of between two steps.  The last one is waited for inside the
    # timed region (`drain`).
    dist_on = world > 1 or args.force_dist
    if dist_on:
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
    kern = {"march": 0.0, "compact": 0.0, "scan": 0.0, "volumes": 0.0, "plan": 0.0, "total": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        local_total = step()
        tm = dt.timing()  # HIP events recorded on the launch stream inside rt_segmentize
        for k in kern:
            kern[k] += tm[k]
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    n_failed, _, _ = dt.failed()  # tracks on which the reference itself would have thrown (never fatal here:
    #                               a rank that exits alone would deadlock the others)

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(local_total), float(n_failed)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    t_max = float(t_max.item())
    global_segments = float(tot[0].item())
    failed_tracks = int(tot[1].item())

    # ---- optional: reassemble the global segment list on every rank (RCCL all-gather)
    allgather = None
    if world > 1 and args.allgather:
        p = dt.device_pointers()
        off = torch.as_tensor(rtd.DevArray(p["offsets"], dt.n + 1, "<i8", dt), device=dev)
        local = {"counts": off[1:] - off[:-1]}
        for name in ("px", "py", "qx", "qy", "ell"):
            local[name] = torch.as_tensor(rtd.DevArray(p[name], local_total, "<f8", dt), device=dev)
        local["element"] = torch.as_tensor(rtd.DevArray(p["element"], local_total, "<i4", dt), device=dev)
        rtd.allgather_segments(local)  # warm-up
        sync()
        reps = 3
        g0 = time.perf_counter()
        for _ in range(reps):
            g = rtd.allgather_segments(local)
        sync()
        g_ms = (time.perf_counter() - g0) / reps * 1e3
        gt = torch.tensor([g_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        assert int(g["offsets"][-1].item()) == int(global_segments)
        allgather = {"ms": float(gt.item()), "bytes_per_rank_received": 44.0 * global_segments,
                     "segments_per_s_including_allgather": global_segments / (t_max / args.steps + float(gt.item()) / 1e3)}
        del g

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        per_kernel = []
        for key, (kname, bps) in KERNELS.items():
            ms = kern[key] / args.steps
            per_kernel.append({"kernel": kname, "ms_avg": ms, "bytes_per_segment": bps,
                               "achieved_GBs": bps * local_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                               "traffic": hbm_traffic_per_launch(kname)})
        dom_k = max(per_kernel, key=lambda k: k["ms_avg"])
        achieved = dom_k["achieved_GBs"]
        out = {
            "metric": "segments/sec (whole node)",
            "value": global_segments * args.steps / t_max,
            "unit": "segments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (deterministic tracks from trace! on the reference's demo/pincell.msh; no RNG)",
            "config": {
                "workload": "%s, nφ=%d, δ=%g (global), all tracks of tracks_by_uid; segmentize! = march + CSR offsets "
                            "+ compact segment records + fill_volumes" % (args.mesh, args.n_azim, delta),
                "tracks_global": int(tg.n_total_tracks), "segments_global": int(global_segments),
                "tracks_rank0": int(hi - lo), "segments_rank0": int(local_total),
                "sharding": "contiguous uid ranges balanced by Σℓ; all-reduce(sum) of volumes" if world > 1 else "none",
                "tiny_step": tg.tiny_step, "k": 5, "rtol": rt.RTOL_DEFAULT, "failed_tracks": failed_tracks,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom_k["kernel"],
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": dom_k["traffic"],
                "bytes_per_segment": dom_k["bytes_per_segment"], "segments_per_launch": int(local_total),
                "kernel_ms_avg": dom_k["ms_avg"],
                "note": "dominant kernel by HIP-event duration; algorithmic bytes = bytes_per_segment x segments per launch "
                        "(DESIGN.md §4).  The march is FP64 traversal bound by its per-track dependent chain, not by HBM",
            },
            "kernels": per_kernel,
            "kernel_ms": {k: v / args.steps for k, v in kern.items()},
        }
        if allgather is not None:
            out["allgather"] = allgather
        if world == 1 and not dist_on and not args.no_concurrent:
            out["two_batches_in_flight"] = two_in_flight(tg, aq, dmesh, dt, args.steps, local_total)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tg)
        real_stdout.write(json.dumps(out, ensure_ascii=False) + "\n")
        real_stdout.flush()
    if dist_on:
        dmesh.set_enqueue_hook(None)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
