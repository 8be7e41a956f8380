"""rt_sweep — the device-side consumer that walks the cyclic tracks — against a sequential numpy sweep
over the ORACLE's records (tests/sweep_ref.py).  Both inputs are covered: the compact CSR records and the march's staging
rows read directly (option "compact" = 0: march + scan + sweep, no compaction).  Tolerance 1e-12 relative to the largest
value of each array: the device's expm1 and numpy's differ in the last ulp, and the tallies add in a different order."""
import numpy as np
import pytest

import sweep_ref

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _close(a, b, what):
    scale = max(float(np.abs(b).max()), 1e-300)
    err = float(np.abs(a - b).max()) / scale
    assert err <= RTOL, (what, err)
    return err


def _problem(rt, tg, G, seed):
    rng = np.random.default_rng(seed)
    nc = tg.mesh.num_cells
    sigma_t = rng.uniform(0.05, 3.0, (nc, G))
    source = rng.uniform(0.0, 2.0, (nc, G))
    aq = tg.azimuthal_quadrature
    weight = aq.delta_s[tg.azim_idx - 1] * aq.omega_a[tg.azim_idx - 1]
    psi_in = rng.uniform(0.0, 1.5, (2, tg.n_total_tracks, G))
    return sigma_t, source, weight, psi_in


def _links(tg):
    return (tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd)


def _bcs(rt, kind):
    B = rt.BoundaryConditions
    if kind == "reflective":
        return B(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)
    if kind == "periodic":
        return B(top=rt.Periodic, bottom=rt.Periodic, left=rt.Periodic, right=rt.Periodic)
    if kind == "vacuum":
        return B(top=rt.Vacuum, bottom=rt.Vacuum, left=rt.Vacuum, right=rt.Vacuum)
    return B(top=rt.Vacuum, bottom=rt.Reflective, left=rt.Periodic, right=rt.Periodic)  # mixed


def _device(rt, tg, compact, **opts):
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("compact", compact)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    dt.sweep_set_links(tg)
    return dm, dt


CASES = [("pincell.json", 32, 5e-3, "reflective", 7), ("pincell.json", 32, 5e-3, "vacuum", 2), ("pincell.json", 32, 5e-3, "periodic", 1),
         ("bwr_like.msh", 16, 0.02, "reflective", 7), ("bwr_like.msh", 16, 0.02, "mixed", 3)]


@pytest.mark.parametrize("mesh,n_azim,delta,bc,G", CASES)
def test_sweep_matches_sequential_sweep_over_oracle_records(rt, orc, mesh, n_azim, delta, bc, G):
    path = rt.data_path(mesh)
    model = rt.GmshDiscreteModel(path) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
    tg = rt.TrackGenerator(model, n_azim, delta, bcs=_bcs(rt, bc))
    rt.trace(tg)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, n_threads=0)
    sigma_t, source, weight, psi_in = _problem(rt, tg, G, 7)
    phi1, out1 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, psi_in)
    nxt1 = sweep_ref.link(out1, *_links(tg))
    phi2, out2 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, nxt1)  # second iteration
    if bc == "vacuum":
        assert not nxt1.any()
    # (a batch this small is marched in pieces by default: the staging rows of a compacting call are whole tracks only with
    #  "split" = 0; "compact" = 0 marches whole tracks by itself)
    # (round 6: the compact records are swept as (ℓ, cell) rows — made once from the records themselves when the call left no
    #  whole-track staging, "rows from compact", or the staging's; "sweep_rows" 0 = the records where they lie, 2 = always from the records)
    for compact, inp, opts, rows in ((1, "compact", {}, "from compact"), (0, "staged", {}, "staging"), (1, "staged", {"split": 0}, "staging"),
                                     (0, "compact", {}, "staging"), (1, "compact", {"sweep_rows": 0}, None),
                                     (1, "compact", {"sweep_rows": 2, "split": 0}, "from compact"), (0, "compact", {"sweep_rows": 0}, None)):
        dm, dt = _device(rt, tg, compact, **opts)
        r = dt.sweep(G, sigma_t, source, weight, psi_in, input=inp)
        assert r["input"] == inp and r["rows"] == rows, (r["input"], r["rows"], inp, rows, opts)
        e = [_close(r["phi"], phi1, "phi"), _close(r["psi_out"], out1, "psi_out"), _close(r["psi_next"], nxt1, "psi_next")]
        r2 = dt.sweep(G)  # everything from the device: cross sections, weights, the boundary flux handed on
        e += [_close(r2["phi"], phi2, "phi, 2nd sweep"), _close(r2["psi_out"], out2, "psi_out, 2nd sweep")]
        if compact == 0:
            # the records were never compacted for the sweep; asking for them now produces them, equal to the oracle's
            s = dt.fetch_segments()
            assert np.array_equal(s["element"], ref["element"]) and all(np.array_equal(s[k], ref[k]) for k in ("px", "py", "qx", "qy", "ell"))
        print(f"{mesh} nφ={n_azim} {bc} G={G} compact={compact} input={inp}: {r['passes']} passes of {r['groups_per_pass']} groups, "
              f"{r['ms']:.3f} ms, max rel err {max(e):.1e}")
        dt.close(); dm.close()


@pytest.mark.parametrize("scale,regime", [(0.02, "thin"), (4.0, "mixed"), (1000.0, "thick")])
def test_sweep_attenuation_regimes(rt, orc, scale, regime):
    """rt_sweep takes the series-only attenuation factor on rows whose every lane and group has τ < 1/8 and the general form
    elsewhere (rt_device.hpp, one_minus_exp_neg_thin / one_minus_exp_neg): cross sections scaled so that every row is thin, rows
    of both kinds occur, next to no record is thin (some optical lengths beyond 41.5, where the factor is 1) — each against the sequential sweep over the oracle's records, and the default against
    option "sweep_debug" 4 (the general form everywhere)."""
    model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
    tg = rt.TrackGenerator(model, 32, 5e-3, bcs=_bcs(rt, "reflective"))
    rt.trace(tg)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, n_threads=0)
    G = 5
    sigma_t, source, weight, psi_in = _problem(rt, tg, G, 11)
    sigma_t = sigma_t * scale
    tau = sigma_t[ref["element"] - 1] * ref["ell"][:, None]
    thin_share = float((tau.max(axis=1) < 0.125).mean())
    assert {"thin": thin_share == 1.0, "mixed": 0.02 < thin_share < 0.98, "thick": thin_share < 0.02}[regime], thin_share
    phi1, out1 = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, psi_in)
    res = {}
    for dbg in (0, 4):
        dm, dt = _device(rt, tg, 0, sweep_debug=dbg)
        r = dt.sweep(G, sigma_t, source, weight, psi_in, input="staged")
        _close(r["phi"], phi1, "phi"); _close(r["psi_out"], out1, "psi_out")
        res[dbg] = r
        dt.close(); dm.close()
    _close(res[0]["phi"], res[4]["phi"], "phi, thin form against the general one")
    _close(res[0]["psi_out"], res[4]["psi_out"], "psi_out, thin form against the general one")


def test_sweep_at_config4_size(rt, orc):
    """BASELINE configs[3] (BWR-like mesh, nφ=64, δ=2e-3: 130,472 tracks, 14.3 M segments): the full batch — 2,039 march waves, the
    BWR-like mesh's two groups of tallies per pass in LDS — swept twice on the device, against the sequential sweep over ALL of
    the oracle's records (no sampling: the tallies of a cell need every track that crosses it)."""
    model = rt.GmshDiscreteModel(rt.data_path("bwr_like.msh"))
    tg = rt.TrackGenerator(model, 64, 2e-3, bcs=_bcs(rt, "reflective"))
    rt.trace(tg)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi,
                        tiny_step=tg.tiny_step, n_threads=0)
    G = 3
    sigma_t, source, weight, psi_in = _problem(rt, tg, G, 21)
    phi1, out1 = sweep_ref.sweep_fast(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, psi_in)
    nxt1 = sweep_ref.link(out1, *_links(tg))
    phi2, out2 = sweep_ref.sweep_fast(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, weight, nxt1)
    for compact, inp in ((1, "auto"), (0, "staged"), (1, "compact")):
        dm, dt = _device(rt, tg, compact)
        assert dt.total == ref["total"]
        r = dt.sweep(G, sigma_t, source, weight, psi_in, input=inp)
        e = [_close(r["phi"], phi1, "phi"), _close(r["psi_out"], out1, "psi_out"), _close(r["psi_next"], nxt1, "psi_next")]
        r2 = dt.sweep(G)
        e += [_close(r2["phi"], phi2, "phi, 2nd sweep"), _close(r2["psi_out"], out2, "psi_out, 2nd sweep")]
        print(f"config 4, G={G}, compact={compact}, input {inp} -> {r['input']}: {r['passes']} passes of {r['groups_per_pass']} groups, {r['ms']:.3f} ms, "
              f"max rel err {max(e):.1e}")
        dt.close(); dm.close()


def test_sweep_default_weight_and_group_slabs(rt, orc, traced, oracle_run):
    """Default weights are fill_volumes' δs[azim_idx]; 1 to 4 groups per pass (and global atomics) give the same tallies."""
    tg = traced(16, 1e-2)
    ref = oracle_run(tg)
    G = 5
    sigma_t, source, _, psi_in = _problem(rt, tg, G, 11)
    aq = tg.azimuthal_quadrature
    phi, out = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, aq.delta_s[tg.azim_idx - 1], psi_in)
    for gp in (0, 1, 2, 3, 4):
        for waves in (0, 4):
            dm, dt = _device(rt, tg, 0, sweep_gp=gp, sweep_waves=waves)
            r = dt.sweep(G, sigma_t, source, None, psi_in)
            assert r["input"] == "staged" and (gp == 0 or r["groups_per_pass"] == gp)
            _close(r["phi"], phi, f"phi gp={gp}"); _close(r["psi_out"], out, f"psi_out gp={gp}")
            dt.close(); dm.close()


def test_sweep_argument_errors(rt, traced):
    from raytracing_jl_amd import _capi

    tg = traced(8, 2e-2)
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    sig = np.ones((tg.mesh.num_cells, 2))
    with pytest.raises(_capi.RtError, match="rt_segmentize has not run"):
        dt.sweep(2, sig)
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)  # a small batch: marched in pieces by default
    with pytest.raises(_capi.RtError, match="rt_sweep_set_links"):
        dt.sweep(2, sig)
    dt.sweep_set_links(tg)
    with pytest.raises(_capi.RtError, match="cross sections"):
        dt.sweep(2)
    assert dt.stats()["split"] == 1
    with pytest.raises(_capi.RtError, match="staging rows"):
        dt.sweep(2, sig, input="staged")
    assert dt.sweep(2, sig)["input"] == "compact"
    bad = dict(next_fwd=tg.next_fwd_uid.copy(), next_bwd=tg.next_bwd_uid, dir_fwd=tg.dir_next_fwd, dir_bwd=tg.dir_next_bwd,
               bc_fwd=tg.bc_fwd, bc_bwd=tg.bc_bwd)
    bad["next_fwd"][3] = tg.n_total_tracks + 1
    with pytest.raises(_capi.RtError, match="bad link"):
        dt.sweep_set_links(bad)
    dt.close(); dm.close()


@pytest.mark.parametrize("mesh,n_azim,delta", [("pincell.msh", 128, 1e-3), ("bwr_like.msh", 64, 2e-3)])
def test_sweep_properties_at_full_size(rt, mesh, n_azim, delta):
    """BASELINE configs[2] and [3] (9.3 M and 14.3 M segments), where the sequential checker would take minutes: properties that do
    not depend on the size.  (a) A flat source in equilibrium with the incoming flux (ψ = q/Σt everywhere) is a fixed point: nothing
    is attenuated, nothing is tallied, every traversal ends with what it started with.  (b) A pure absorber is in balance: what the
    tallies hold is what the traversals lost, Σ φ = Σ w·(ψ_in − ψ_out), and with Σt uniform ψ_out = ψ_in·exp(−Σt·ℓ_track) — the
    records of a track add up to its length (src/track.jl:171).  (c) Both inputs — compact records and staging rows — give the same
    fluxes bit for bit (same ℓ, same order) and the same tallies to rounding."""
    from raytracing_jl_amd import _capi

    model = rt.GmshDiscreteModel(rt.data_path(mesh))
    tg = rt.TrackGenerator(model, n_azim, delta, bcs=_bcs(rt, "reflective"))
    rt.trace(tg)
    n, nc, G = tg.n_total_tracks, tg.mesh.num_cells, 7
    rng = np.random.default_rng(2)
    sigma_t = rng.uniform(0.05, 3.0, (nc, G))
    level = rng.uniform(0.5, 2.0, G)
    w = rng.uniform(0.5, 1.5, n)
    dm, dt = _device(rt, tg, 0)
    # (a)
    r = dt.sweep(G, sigma_t, sigma_t * level, w, np.broadcast_to(level, (2, n, G)).copy(), input="staged")
    assert np.abs(r["psi_out"] - level).max() <= 1e-13 * level.max() and np.abs(r["phi"]).max() <= 1e-10
    # (b)
    psi_in = rng.uniform(0.5, 1.5, (2, n, G))
    st = np.broadcast_to(np.linspace(0.3, 1.1, G), (nc, G)).copy()
    r = dt.sweep(G, st, np.zeros((nc, G)), w, psi_in, input="staged")
    lost = (w[None, :, None] * (psi_in - r["psi_out"])).sum(axis=(0, 1))
    assert np.allclose(r["phi"].sum(axis=0), lost, rtol=1e-11)
    off, status = dt.fetch_offsets()
    good = status == 0  # (a track on which the reference's own Σℓ check fails misses a sliver: its Σℓ is not its length)
    expect = psi_in[:, good, :] * np.exp(-st[0][None, None, :] * tg.ell[good][None, :, None])
    assert np.allclose(r["psi_out"][:, good, :], expect, rtol=1e-6)  # Σℓ ≈ ℓ to rtol = √eps, times Σt·ℓ ≤ 10
    # (c)
    rc = dt.sweep(G, st, np.zeros((nc, G)), w, psi_in, input="compact")
    assert np.array_equal(rc["psi_out"], r["psi_out"]) and np.array_equal(rc["psi_next"], r["psi_next"])
    assert np.allclose(rc["phi"], r["phi"], rtol=1e-11, atol=1e-300)
    print(f"{mesh} nφ={n_azim} δ={delta}: {dt.total} segments, staged {r['ms']:.3f} ms ({r['passes']} passes of {r['groups_per_pass']}), compact {rc['ms']:.3f} ms")
    dt.close(); dm.close()


def test_ell_rows_of_the_first_staged_pass_change_nothing(rt):
    """Staged input: the first pass after a segmentize leaves ℓ per row, later passes and sweeps read (ℓ, cell) rows (option
    "sweep_ell", default 1).  Same bits as deriving ℓ from the exit points in every pass — also after a new rt_segmentize."""
    model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    tg = rt.TrackGenerator(model, 32, 5e-3, bcs=_bcs(rt, "mixed"))
    rt.trace(tg)
    G = 7
    sigma_t, source, weight, psi_in = _problem(rt, tg, G, 11)
    aq = tg.azimuthal_quadrature
    res = []
    for ell in (0, 1):
        dm, dt = _device(rt, tg, 0, split=0, sweep_ell=ell)
        a = dt.sweep(G, sigma_t, source, weight, psi_in, input="staged")
        b = dt.sweep(G, input="staged")                       # second sweep: every pass on (ℓ, cell) rows
        dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)  # the rows are rewritten: ℓ must be derived again
        c = dt.sweep(G, None, None, None, psi_in, input="staged")
        res.append((a, b, c))
        dt.close(); dm.close()
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x["psi_out"], y["psi_out"])
        _close(x["phi"], y["phi"], "phi")  # (the tallies add in another order from run to run)
    assert np.array_equal(res[1][0]["psi_out"], res[1][2]["psi_out"])  # same fluxes in: the first sweep again


def test_source_updated_on_the_device(rt):
    """rt_sweep_xs_pointer: a solver writes q/Σt into the sweep's device copy of the cross sections (torch here) and sweeps again
    with sigma_t = source = None — the same result as handing the new source over from the host."""
    import torch
    from raytracing_jl_amd import distributed as rtd

    model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    tg = rt.TrackGenerator(model, 16, 1e-2, bcs=_bcs(rt, "reflective"))
    rt.trace(tg)
    G = 3
    sigma_t, source, weight, psi_in = _problem(rt, tg, G, 5)
    source2 = source[::-1].copy() * 0.5
    dm, dt = _device(rt, tg, 0, split=0)
    dt.sweep(G, sigma_t, source, weight, psi_in)
    want = dt.sweep(G, sigma_t, source2, None, psi_in)      # new source from the host
    dt.sweep(G, sigma_t, source, None, psi_in)               # back to the first one
    nc = tg.mesh.num_cells
    xs = torch.as_tensor(rtd.DevArray(dt.sweep_xs_pointer(), nc * G * 2, "<f8", dt), device="cuda:0").view(nc * G, 2)
    assert np.array_equal(xs[:, 0].cpu().numpy().reshape(nc, G), sigma_t)
    xs[:, 1] = torch.as_tensor((source2 / sigma_t).reshape(-1), device="cuda:0")
    torch.cuda.synchronize()
    got = dt.sweep(G, None, None, None, psi_in)              # cross sections "of the previous call": the updated device copy
    assert np.array_equal(got["psi_out"], want["psi_out"])
    _close(got["phi"], want["phi"], "phi")
    dt.close(); dm.close()
