"""N>1 path on CPU: world_size-2 `gloo` run of the sharding + reassembly logic
(raytracing.jl_amd/distributed.py).  The oracle stands in for the device march (tests may
call it); on the GPU box the same functions run over RCCL with the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        import raytracing_jl_amd as rt
        from oracle import oracle as orc
        from raytracing_jl_amd import distributed as rtd

        dist.init_process_group("gloo", rank=rank, world_size=world)
        model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
        tg = rt.TrackGenerator(model, 8, 0.04)
        rt.trace(tg)
        aq = tg.azimuthal_quadrature
        lo, hi = rtd.shard_ranges(tg.ell, world)[rank]
        a = rtd.shard_arrays(tg, lo, hi)
        om = orc.OracleMesh.from_mesh(tg.mesh)
        r = om.segmentize(a["px"], a["py"], a["phi"], a["A"], a["B"], a["C"], a["ell"],
                          cos_phi=a["cos_phi"], sin_phi=a["sin_phi"])
        vol = torch.from_numpy(om.fill_volumes(r["offsets"], a["azim_idx"], aq.delta_s, aq.n_azim_2))
        rtd.allreduce_volumes(vol)
        local = {"counts": torch.from_numpy(np.diff(r["offsets"]))}
        for k in ("px", "py", "qx", "qy", "ell", "element"):
            local[k] = torch.from_numpy(r[k])
        g = rtd.allgather_segments(local)
        # single-process answer
        full = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi)
        vfull = om.fill_volumes(full["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
        ok = np.array_equal(g["offsets"].numpy(), full["offsets"])
        for k in ("px", "py", "qx", "qy", "ell", "element"):
            ok = ok and np.array_equal(g[k].numpy(), full[k])
        ok = ok and np.allclose(vol.numpy(), vfull, rtol=1e-12, atol=0)
        dist.destroy_process_group()
        q.put((rank, bool(ok), int(hi - lo)))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_shard_and_gather(world):
    """Sharding by Σℓ, all-reduce of volumes and the all-gather-v of the segment arrays (direct sends and receives into
    the final buffers, distributed.SegmentGather) against the unsharded answer."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) > 0


def _pipe_worker(rank, world, port, q):
    """The bench's pipelined all-reduce of volumes, with CPU tensors over gloo: a fake `segmentize` that
    alternates between two buffers like the library and fires the enqueue hook in the middle of a call."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        from raytracing_jl_amd import distributed as rtd

        dist.init_process_group("gloo", rank=rank, world_size=world)
        n_cells, steps = 257, 7
        bufs = [torch.zeros(n_cells, dtype=torch.float64), torch.zeros(n_cells, dtype=torch.float64)]
        pipe = rtd.PipelinedVolumesAllReduce(device="cpu")
        base = torch.arange(n_cells, dtype=torch.float64)
        reduced = {}

        def local_volumes(step):  # what rank `rank` contributes at `step`
            return (rank + 1) * base + 1000.0 * step

        owner = {}  # buffer index -> step whose all-reduce result it will hold once waited for
        for i in range(steps):
            k = pipe.before_call()
            if k in owner:  # before_call waited for the all-reduce of step i-2: its result is in bufs[k] now
                reduced[owner.pop(k)] = bufs[k].clone()
            bufs[k].zero_()          # k_prologue
            pipe.hook()              # the library's enqueue hook: all-reduce of step i-1 starts here
            bufs[k] += local_volumes(i)  # the march + scale of step i
            pipe.after_call(k, bufs[k])
            owner[k] = i
        pipe.drain()
        for k, i in owner.items():
            reduced[i] = bufs[k].clone()
        ok = len(reduced) == steps
        for i in range(steps):
            want = sum((r + 1) * base + 1000.0 * i for r in range(world))
            ok = ok and torch.equal(reduced[i], want)
        dist.destroy_process_group()
        q.put((rank, bool(ok), steps))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(300)
def test_two_rank_gloo_pipelined_volumes_allreduce():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _sweep_worker(rank, world, port, q, bc):
    """The sharded transport sweep (distributed.ShardedSweep) over gloo: every rank runs the sequential sweep of tests/sweep_ref.py
    over ITS uid range of the checker's records (standing in for rt_sweep), hands on the fluxes that stay in the shard itself (as
    the library's k_sweep_link does, with next uid 0 for tracks of other ranks), and ShardedSweep.exchange moves the rest and sums
    the tallies.  Two sweeps must equal two sweeps of the unsharded problem."""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        import raytracing_jl_amd as rt
        import sweep_ref
        from oracle import oracle as orc
        from raytracing_jl_amd import distributed as rtd

        dist.init_process_group("gloo", rank=rank, world_size=world)
        B = rt.BoundaryConditions
        bcs = {"reflective": B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective),
               "periodic": B(top=rt.Periodic, bottom=rt.Periodic, left=rt.Periodic, right=rt.Periodic),
               "mixed": B(top=rt.Vacuum, bottom=rt.Reflective, left=rt.Periodic, right=rt.Periodic)}[bc]
        model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
        tg = rt.TrackGenerator(model, 8, 0.04, bcs=bcs)
        rt.trace(tg)
        om = orc.OracleMesh.from_mesh(tg.mesh)
        full = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi)
        n, nc, G = tg.n_total_tracks, tg.mesh.num_cells, 3
        rng = np.random.default_rng(4)
        sig, src, w = rng.uniform(0.1, 2.0, (nc, G)), rng.uniform(0.0, 1.0, (nc, G)), rng.uniform(0.5, 1.5, n)
        psi0 = rng.uniform(0.5, 1.5, (2, n, G))
        links = (tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd)
        # the unsharded answer: two sweeps
        phi1, out1 = sweep_ref.sweep(full["offsets"], full["ell"], full["element"], sig, src, w, psi0)
        nxt1 = sweep_ref.link(out1, *links)
        phi2, out2 = sweep_ref.sweep(full["offsets"], full["ell"], full["element"], sig, src, w, nxt1)
        nxt2 = sweep_ref.link(out2, *links)
        # this rank's shard
        ranges = rtd.shard_ranges(tg.ell, world)
        lo, hi = ranges[rank]
        nl = hi - lo
        off = full["offsets"][lo:hi + 1] - full["offsets"][lo]
        s0, s1 = full["offsets"][lo], full["offsets"][hi]
        state = {"phi": torch.zeros((nc, G), dtype=torch.float64), "out": torch.zeros((2 * nl, G), dtype=torch.float64),
                 "in": torch.from_numpy(psi0[:, lo:hi].reshape(2 * nl, G).copy())}
        ss = rtd.ShardedSweep(tg, None, rank, world, ranges=ranges, tensors=lambda: (state["phi"], state["out"], state["in"]))
        ll = ss.plan.local_links

        def local_sweep():  # what rt_sweep does on the shard: traversals, then the hand-over inside the shard
            psi_in = state["in"].numpy().reshape(2, nl, G)
            phi, out = sweep_ref.sweep(off, full["ell"][s0:s1], full["element"][s0:s1], sig, src, w[lo:hi], psi_in)
            nxt = np.zeros_like(out)
            for u in range(nl):
                for d, (nx, dr, b) in enumerate(((ll["next_fwd"], ll["dir_fwd"], ll["bc_fwd"]), (ll["next_bwd"], ll["dir_bwd"], ll["bc_bwd"]))):
                    if nx[u] > 0:
                        nxt[int(dr[u]), int(nx[u]) - 1] = 0.0 if int(b[u]) == 0 else out[d, u]
            state["phi"].copy_(torch.from_numpy(phi)); state["out"].copy_(torch.from_numpy(out.reshape(2 * nl, G)))
            state["in"].copy_(torch.from_numpy(nxt.reshape(2 * nl, G)))

        ok = True
        for want_phi, want_out, want_next in ((phi1, out1, nxt1), (phi2, out2, nxt2)):
            local_sweep()
            phi, out, nin = ss.exchange(G)
            ok = ok and np.allclose(phi.numpy(), want_phi, rtol=1e-12, atol=1e-14)
            ok = ok and np.array_equal(out.numpy().reshape(2, nl, G), want_out[:, lo:hi])
            ok = ok and np.array_equal(nin.numpy().reshape(2, nl, G), want_next[:, lo:hi])
        n_cross = sum(len(v[0]) for v in ss.plan.send.values())
        dist.destroy_process_group()
        q.put((rank, bool(ok), int(n_cross)))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,bc", [(2, "reflective"), (3, "mixed"), (2, "periodic")])
def test_gloo_sharded_sweep(world, bc):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, q, bc)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sum(c for _, _, c in res) > 0  # fluxes did cross ranks
