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
