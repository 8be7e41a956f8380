"""Full-size and sharded runs of the HIP segmentize! path — GPU only.

At BASELINE.json's largest configuration (config 5: BWR-like mesh, nφ=128, δ=5e-4, ≈1.04 M tracks,
≈1.1·10⁸ segments) the results are checked twice.  Record for record against the oracle on a sample of uids
(``test_bwr_config5_records_equal_the_oracle_on_a_uid_sample``: tracks are independent, ``src/trackgenerator.jl:362-364``, so
the oracle's march of every 16th track is an exact comparison for those tracks of the FULL GPU run — the regime with 16 k
march waves, eight rounds deep, and the compaction in output order; ``RT_C5_STRIDE=1`` compares all 114 M records where the
host has the cores and ≈12 GB of memory for it).  And, for all tracks, through properties that do not depend on size,
evaluated on the device:

* the reference's own invariants (``test/runtests.jl:30-43``): first segment starts at the track's
  ``p``, Σℓ of a track's segments ≈ the track's ℓ (this is also the reference's run-time check,
  ``src/track.jl:171-175``);
* contiguity: segment i+1 starts within ``tiny_step`` (+ rounding) of where segment i ended, on the
  track's line (``src/track.jl:165``), or further only where the reference skips a sliver;
* Σ volumes ≈ the domain area (``fill_volumes``, ``src/trackgenerator.jl:371-386``);
* a checksum of checksums: marching the 8 uid shards of ``distributed.shard_ranges`` one after
  the other must reproduce the unsharded result bit for bit (segments) and to rounding (volumes) —
  the N>1 data path of bench.py without the collective.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_views(dt):
    import torch

    from raytracing_jl_amd import distributed as rtd

    p = dt.device_pointers()
    dev = torch.device("cuda", 0)
    n = dt.total
    v = {"offsets": torch.as_tensor(rtd.DevArray(p["offsets"], dt.n + 1, "<i8", dt), device=dev),
         "status": torch.as_tensor(rtd.DevArray(p["status"], dt.n, "<i4", dt), device=dev),
         "element": torch.as_tensor(rtd.DevArray(p["element"], n, "<i4", dt), device=dev),
         "volumes": torch.as_tensor(rtd.DevArray(p["volumes"], dt.dmesh.n_cells, "<f8", dt), device=dev)}
    for name in ("px", "py", "qx", "qy", "ell"):
        v[name] = torch.as_tensor(rtd.DevArray(p[name], n, "<f8", dt), device=dev)
    return v


def _checksums(v):
    """Order-sensitive integer checksums of a segment list (bit patterns, not float sums)."""
    import torch

    n = v["ell"].numel()
    w = (torch.arange(n, device=v["ell"].device, dtype=torch.int64) % 1000003) + 1
    out = {"n": n, "element": int((v["element"].to(torch.int64) * w).sum().item())}
    for name in ("px", "py", "qx", "qy", "ell"):
        bits = v[name].view(torch.int64)
        out[name] = int(((bits >> 11) * w).sum().item())  # wraps mod 2^64: fine for a checksum
    return out


def _shard_run(rt, tg, world, dmesh):
    """March the `world` uid shards one after the other on this GPU; concatenated results."""
    import torch

    from raytracing_jl_amd import distributed as rtd

    aq = tg.azimuthal_quadrature
    parts, vol = [], None
    for r in range(world):
        dt, (lo, hi) = rtd.segmentize_shard(tg, r, world, dmesh=dmesh)
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        v = _device_views(dt)
        parts.append({k: v[k].clone() for k in ("px", "py", "qx", "qy", "ell", "element", "status")}
                     | {"counts": (v["offsets"][1:] - v["offsets"][:-1]).clone()})
        vol = v["volumes"].clone() if vol is None else vol + v["volumes"]
        dt.close()
    cat = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
    cat["volumes"] = vol
    return cat


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_march_equals_unsharded(rt, traced, oracle_run, world):
    """Config 2 sharded 2 and 8 ways: concatenation in rank order is the oracle's uid order."""
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    cat = _shard_run(rt, tg, world, _capi.DeviceMesh(tg.mesh, 0))
    assert np.array_equal(cat["counts"].cpu().numpy(), np.diff(ref["offsets"]))
    assert np.array_equal(cat["element"].cpu().numpy(), ref["element"])
    assert np.array_equal(cat["status"].cpu().numpy(), ref["status"])
    for name in ("px", "py", "qx", "qy", "ell"):
        assert np.array_equal(cat[name].cpu().numpy(), ref[name]), name
    assert np.allclose(cat["volumes"].cpu().numpy(), ref["volumes"], rtol=1e-10, atol=0)


_C5 = {}


def _config5(rt):
    """BASELINE configs[4], traced once per session."""
    if "tg" not in _C5:
        model = rt.GmshDiscreteModel(rt.data_path("bwr_like.msh"))
        tg = rt.TrackGenerator(model, 128, 5e-4)
        rt.trace(tg)
        assert tg.n_total_tracks == 1_043_212
        _C5["tg"] = tg
    return _C5["tg"]


def test_bwr_config5_records_equal_the_oracle_on_a_uid_sample(rt, orc):
    """The FULL config-5 march on one GPU (16 k march waves, many rounds, compaction in output order), compared record for
    record — offsets, status, element ids and the five coordinate arrays, bit for bit — with the oracle's march
    (oracle/rt_oracle.c, following src/track.jl:106-178) of every 16th uid: ≈65 k tracks, ≈7 M segments."""
    import os

    import torch

    from raytracing_jl_amd import _capi

    tg = _config5(rt)
    aq = tg.azimuthal_quadrature
    stride = int(os.environ.get("RT_C5_STRIDE", "16"))
    sel = np.arange(0, tg.n_total_tracks, stride)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px[sel], tg.py[sel], tg.phi[sel], tg.A[sel], tg.B[sel], tg.C[sel], tg.ell[sel],
                        cos_phi=tg.cos_phi[sel], sin_phi=tg.sin_phi[sel], tiny_step=tg.tiny_step, n_threads=0)
    dmesh = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dmesh, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    assert total > 1.0e8
    st = dt.stats()
    v = _device_views(dt)
    dev = v["ell"].device
    off = v["offsets"]
    tsel = torch.as_tensor(sel, device=dev)
    cnt = (off[1:] - off[:-1])[tsel]
    assert np.array_equal(cnt.cpu().numpy(), np.diff(ref["offsets"])), "segment counts of the sampled uids"
    assert np.array_equal(v["status"][tsel].cpu().numpy(), ref["status"]), "per-track status of the sampled uids"
    # positions of the sampled tracks' records inside the full CSR arrays
    starts = off[:-1][tsel]
    ref_off = torch.as_tensor(ref["offsets"][:-1], device=dev)
    idx = torch.repeat_interleave(starts - ref_off, cnt) + torch.arange(int(cnt.sum().item()), device=dev)
    assert idx.numel() == len(ref["element"])
    assert np.array_equal(v["element"][idx].cpu().numpy(), ref["element"]), "element ids"
    for name in ("px", "py", "qx", "qy", "ell"):
        got = v[name][idx].cpu().numpy()
        assert np.array_equal(got.view(np.int64), ref[name].view(np.int64)), name  # bit for bit
    print(f"config 5: {total} segments on the GPU ({st['cheap_records']} by cheap steps); {len(sel)} sampled tracks "
          f"(every {stride}th uid), {idx.numel()} records equal the oracle's bit for bit")
    dt.close()
    torch.cuda.empty_cache()


def test_bwr_config5_full_size_properties(rt):
    """BASELINE config 5 on one GPU (the 8-GPU run marches 1/8 of it per rank)."""
    import torch

    from raytracing_jl_amd import _capi

    tg = _config5(rt)
    aq = tg.azimuthal_quadrature
    dmesh = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dmesh, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell,
                            tg.azim_idx)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    v = _device_views(dt)
    dev = v["ell"].device
    off = v["offsets"]
    counts = off[1:] - off[:-1]
    assert int(off[-1].item()) == total and total > 1.0e8
    assert int(counts.min().item()) >= 1
    assert int(v["element"].min().item()) >= 1 and int(v["element"].max().item()) <= dmesh.n_cells

    # Σℓ per track vs the track's ℓ — the reference's own acceptance test (src/track.jl:171)
    tid = torch.repeat_interleave(torch.arange(dt.n, device=dev), counts)
    ell_sum = torch.zeros(dt.n, dtype=torch.float64, device=dev).index_add_(0, tid, v["ell"])
    del tid
    tell = torch.as_tensor(tg.ell, device=dev)
    ok = v["status"] == 0
    n_bad = int((~ok).sum().item())
    print("config 5:", total, "segments,", n_bad, "tracks fail the reference Σℓ check")
    assert n_bad <= 100, n_bad  # a handful of tracks fail the reference's own Σℓ check on this mesh
    assert int(((v["status"] != 0) & (v["status"] != 2)).sum().item()) == 0  # only length mismatches
    rel = (ell_sum - tell).abs() / tell
    assert float(rel[ok].max().item()) <= rt.RTOL_DEFAULT * (1 + 1e-9)  # exactly the reference's criterion
    assert float(rel.max().item()) < 1e-3  # the failing ones miss a sliver, not a cell
    # ... in BOTH directions: isapprox(ℓ, Σℓ; rtol) is |ℓ − Σℓ| <= rtol·max(|ℓ|, |Σℓ|) (src/track.jl:171) — every track marked OK
    # satisfies it, and every track marked LENGTH_MISMATCH really violates it (1e-9: this sum's order is not the kernel's)
    big = torch.maximum(tell.abs(), ell_sum.abs())
    ratio = (ell_sum - tell).abs() / big
    assert float(ratio[ok].max().item()) <= rt.RTOL_DEFAULT * (1 + 1e-9)
    if n_bad:
        assert float(ratio[~ok].min().item()) > rt.RTOL_DEFAULT * (1 - 1e-9), "a track marked LENGTH_MISMATCH passes the reference's check"
    print("config 5: the Σℓ status holds in both directions; smallest ratio of a failing track / rtol =",
          float(ratio[~ok].min().item()) / rt.RTOL_DEFAULT if n_bad else None)

    # first / last segment ends at the track's p / q (test/runtests.jl:30-35)
    first = off[:-1]
    last = off[1:] - 1
    for got, want in ((v["px"][first], tg.px), (v["py"][first], tg.py), (v["qx"][last], tg.qx), (v["qy"][last], tg.qy)):
        err = (got - torch.as_tensor(want, device=dev)).abs()
        assert float(err[ok].max().item()) < 1.5e-8 * 6.4 * 1.5  # `≈` of the reference test: rtol √eps on the norm

    # contiguity inside a track: p[i+1] = q[i] + tiny_step·(cos ϕ, sin ϕ) up to the skipped slivers
    same = torch.ones(total, dtype=torch.bool, device=dev)
    same[off[1:-1]] = False  # segment i (i ≥ 1) continues segment i-1 unless it opens a track
    same[0] = False
    gx = v["px"][1:] - v["qx"][:-1]
    gy = v["py"][1:] - v["qy"][:-1]
    gap = torch.sqrt(gx * gx + gy * gy)[same[1:]]
    assert float(gap.min().item()) >= 0.0
    assert float((gap > 1.0e-6).double().mean().item()) < 1e-5
    assert float(gap.median().item()) < 2.0e-8  # ≈ tiny_step

    # Σ volumes ≈ area of the 6.4 × 6.4 domain
    assert abs(float(v["volumes"].sum().item()) - 6.4 * 6.4) < 1e-6

    whole = _checksums(v)
    whole_vol = v["volumes"].clone()
    whole_status = v["status"].clone()
    del v
    dt.close()
    torch.cuda.empty_cache()

    cat = _shard_run(rt, tg, 8, dmesh)
    assert torch.equal(cat["status"], whole_status)
    assert _checksums(cat) == whole
    assert torch.allclose(cat["volumes"], whole_vol, rtol=1e-10, atol=0)
