"""The sharded transport sweep on the real kernels: two ranks (two processes on the one GPU of the test box, gloo on host copies of
the boundary fluxes — RCCL refuses two ranks on one device) each segmentize their uid range, run rt_sweep over the staging rows of
their shard with the links restricted to it (next uid 0 for tracks of the other rank) and exchange the fluxes that leave the shard
(distributed.ShardedSweep); two sweeps must equal two sweeps of the unsharded problem on one handle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


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
        from raytracing_jl_amd import _capi
        from raytracing_jl_amd import distributed as rtd

        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        B = rt.BoundaryConditions
        model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
        tg = rt.TrackGenerator(model, 32, 5e-3, bcs=B(top=rt.Reflective, bottom=rt.Vacuum, left=rt.Reflective, right=rt.Reflective))
        rt.trace(tg)
        aq = tg.azimuthal_quadrature
        n, nc, G = tg.n_total_tracks, tg.mesh.num_cells, 5
        rng = np.random.default_rng(9)
        sig, src, w = rng.uniform(0.1, 2.0, (nc, G)), rng.uniform(0.0, 1.0, (nc, G)), rng.uniform(0.5, 1.5, n)
        psi0 = rng.uniform(0.5, 1.5, (2, n, G))
        dm = _capi.DeviceMesh(tg.mesh, 0)
        dm.set_option("compact", 0)
        dt, (lo, hi) = rtd.segmentize_shard(tg, rank, world, device=0, dmesh=dm)
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        nl = hi - lo
        host = {}

        def tensors():  # host copies for gloo (a real multi-GPU run hands RCCL the device views, ShardedSweep's default)
            p = dt.sweep_pointers()
            mk = lambda ptr, rows: torch.as_tensor(rtd.DevArray(ptr, rows * G, "<f8", dt), device=dev).view(rows, G)
            host["dev_in"] = mk(p["psi_in"], 2 * nl)
            host["phi"], host["out"], host["in"] = mk(p["phi"], nc).cpu(), mk(p["psi_out"], 2 * nl).cpu(), host["dev_in"].cpu()
            return host["phi"], host["out"], host["in"]

        ss = rtd.ShardedSweep(tg, dt, rank, world, tensors=tensors)
        res = []
        for it in range(2):
            dt.sweep(G, sig if it == 0 else None, src if it == 0 else None, w[lo:hi] if it == 0 else None,
                     psi0[:, lo:hi] if it == 0 else None, input="staged", fetch=False)
            phi, out, nin = ss.exchange(G)
            host["dev_in"].copy_(nin)  # the completed boundary flux goes back to the library's buffer for the next sweep
            torch.cuda.synchronize()
            res.append((phi.numpy().copy(), out.numpy().reshape(2, nl, G).copy(), nin.numpy().reshape(2, nl, G).copy()))
        ok, err = True, 0.0
        if True:  # the unsharded problem on one handle (every rank checks its own slice)
            dm1 = _capi.DeviceMesh(tg.mesh, 0)
            dm1.set_option("compact", 0)
            d1 = _capi.DeviceTracks(dm1, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
            d1.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
            d1.sweep_set_links(tg)
            for it in range(2):
                r = d1.sweep(G, sig if it == 0 else None, src if it == 0 else None, w if it == 0 else None, psi0 if it == 0 else None, input="staged")
                phi, out, nin = res[it]
                ok = ok and np.array_equal(out, r["psi_out"][:, lo:hi]) and np.array_equal(nin, r["psi_next"][:, lo:hi])
                e = float(np.abs(phi - r["phi"]).max() / np.abs(r["phi"]).max())
                err = max(err, e)
                ok = ok and e <= 1e-12
        n_cross = sum(len(v[0]) for v in ss.plan.send.values())
        dist.destroy_process_group()
        q.put((rank, bool(ok), (int(n_cross), err)))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(600)
def test_two_rank_sharded_sweep_equals_unsharded():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(info[0] > 0 for _, _, info in res)  # fluxes did cross ranks
    print("sharded sweep, 2 ranks:", res)
