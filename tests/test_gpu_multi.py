"""rt_multi_*: several devices behind one call of the C ABI, rehearsed on the one GPU of the test box with
device_ids = [0, 0, ...] (every shard its own mesh replica, stream and host thread) — the sharded result must be the
unsharded one bit for bit (offsets, status, records; volumes to 1e-12: shards are summed in a different order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("px", "py", "qx", "qy", "ell", "element")


def _multi(tg, ids):
    from raytracing_jl_amd import _capi

    return _capi.MultiDevice(tg.mesh, ids, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)


@pytest.mark.parametrize("n_azim,delta,ids", [(8, 2e-2, [0, 0]), (32, 5e-3, [0, 0, 0, 0]), (16, 0.02, [0] * 7), (4, 0.5, [0] * 6)])
def test_sharded_equals_unsharded(rt, traced, oracle_run, n_azim, delta, ids):
    tg = traced(n_azim, delta)
    aq = tg.azimuthal_quadrature
    ref = oracle_run(tg)
    rt.segmentize(tg)
    one = {k: getattr(tg.segments, k).copy() for k in FIELDS + ("offsets",)}
    md = _multi(tg, ids)
    total = md.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    assert total == len(tg.segments) == ref["total"]
    ub, sb = md.shards()
    assert ub[0] == 0 and ub[-1] == tg.n_total_tracks and np.all(np.diff(ub) >= 0) and sb[-1] == total
    from raytracing_jl_amd import distributed as rtd
    assert [(int(a), int(b)) for a, b in zip(ub[:-1], ub[1:])] == rtd.shard_ranges(tg.ell, len(ids))  # same partition as the Python host's
    off, st = md.fetch_offsets()
    seg = md.fetch_segments()
    assert np.array_equal(off, one["offsets"]) and np.array_equal(off, ref["offsets"]) and st.max() == 0
    for k in FIELDS:
        assert np.array_equal(seg[k], one[k]), k
        assert np.array_equal(seg[k], ref[k]), k
    assert np.allclose(md.fetch_volumes(), tg.volumes, rtol=1e-12, atol=0)
    assert md.failed() == (0, 0, 0)
    md.close()


def test_allgather_on_every_device_and_failures(rt, traced, oracle_run):
    """Peer-copy reassembly on every shard's device, read back through torch; a failing track reports its global uid."""
    import torch

    from raytracing_jl_amd import distributed as rtd

    tg = traced(16, 0.02)
    aq = tg.azimuthal_quadrature
    ref = oracle_run(tg)
    md = _multi(tg, [0, 0, 0])
    md.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    ms, ptrs = md.allgather()
    assert ms >= 0 and len(ptrs) == 3
    dev = torch.device("cuda", 0)
    for i in range(3):
        for a, k in enumerate(FIELDS):
            t = torch.as_tensor(rtd.DevArray(ptrs[i][a], md.total, "<i4" if k == "element" else "<f8", md), device=dev)
            assert np.array_equal(t.cpu().numpy(), ref[k]), (i, k)
    # every (destination, source) pair ran on its own stream: a rate per pair, none for an empty shard, and a second
    # all-gather (buffers and streams reused) gives the same arrays
    rates = md.link_rates()
    _, sb = md.shards()
    assert rates.shape == (3, 3) and np.all((rates > 0) == (np.diff(sb) > 0)[None, :])
    ms2, ptrs2 = md.allgather()
    assert ptrs2 == ptrs
    t = torch.as_tensor(rtd.DevArray(ptrs2[2][4], md.total, "<f8", md), device=dev)
    assert np.array_equal(t.cpu().numpy(), ref["ell"])
    md.close()
    # the Σℓ check of one track in the last shard fails: global uid, reference status
    ell = tg.ell.copy()
    u = tg.n_total_tracks - 3
    ell[u] *= 1.0 + 1e-6
    from raytracing_jl_amd import _capi
    md = _capi.MultiDevice(tg.mesh, [0, 0, 0, 0], tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, ell, tg.azim_idx)
    md.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    assert md.failed() == (1, u + 1, 2)
    _, st = md.fetch_offsets()
    assert st[u] == 2 and np.count_nonzero(st) == 1
    md.close()


def test_bad_device_is_refused(rt, traced):
    from raytracing_jl_amd import _capi

    tg = traced(8, 2e-2)
    with pytest.raises(_capi.RtError, match="out of range"):
        _multi(tg, [0, 99])
