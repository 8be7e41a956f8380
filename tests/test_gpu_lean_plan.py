"""The lean plan of the two-phase march (GPU only; option "lean", default off — DESIGN.md §4 "Round 6").

`_segmentize_track!` (src/track.jl:106-178) carries xp, prev_element and i between iterations; the lean plan hands exactly that
state — as the cheap step holds it — through memory between three kernels: k_first (start band, first record, `topo_enter`),
k_cheap (ONLY the cheap loop, four waves per SIMD; a refused lane queues its slot and leaves) and k_serve (the one-kernel march
resuming queued lanes; behind k_cheap with "lean" 1, beside it on a second stream with "lean" 2).  Whatever kernel decides a
record, records / status / offsets must be the checker's bit for bit, and the one-kernel plan's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(rt, tg, opts):
    from raytracing_jl_amd import _capi

    dm = _capi.DeviceMesh(tg.mesh, 0)
    for k, v in opts.items():
        dm.set_option(k, v)
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    aq = tg.azimuthal_quadrature
    out = []
    for _ in range(2):  # (the second call starts from the state the first one left: queue, control blocks)
        total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
        off, st = dt.fetch_offsets()
        out.append((total, off, st, dt.fetch_segments(), dt.fetch_volumes(), dt.stats()))
    dt.close()
    dm.close()
    return out


def _same(a, ref, vol, what):
    total, off, st, seg, v, _stats = a
    assert total == ref["total"], what
    assert np.array_equal(st, ref["status"]) and np.array_equal(off, ref["offsets"]), what
    assert np.array_equal(seg["element"], ref["element"]), what
    for k in ("px", "py", "qx", "qy", "ell"):
        assert np.array_equal(seg[k], ref[k]), (k, what)
    assert np.allclose(v, vol, rtol=1e-10, atol=1e-300), what


@pytest.mark.parametrize("lean", [1, 2])
def test_lean_plan_on_the_pincell(rt, traced, oracle_run, lean):
    """C2 marched whole (a batch this small is cut into pieces by default) in three kernels."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    for opts in (dict(split=0, lean=lean), dict(split=0, lean=lean, topo=2), dict(split=0, lean=lean, serve_blocks=3)):
        for k, r in enumerate(_run(rt, tg, opts)):
            _same(r, ref, ref["volumes"], (opts, k))
            assert r[5]["lean"] == lean and r[5]["cheap_records"] > 0, r[5]
            assert r[5]["lean_queued"] > 0  # some lane always leaves: refusals, tracks that end without a certified last step


@pytest.mark.parametrize("lean", [1, 2])
def test_lean_plan_on_fuzz_meshes(rt, orc, lean):
    """Meshes on which many records carry no cheap certificate (k_serve then finishes a large share of the tracks), cheap steps as
    gated and forced; a staging pool and a side list that are too small on the first attempt (the re-run must start clean)."""
    from meshgen import random_model

    for seed, n_int, kw in ((11, 500, {}), (12, 900, dict(cluster=True)), (13, 2500, dict(nb=40))):
        model = random_model(rt, seed, n_int, **kw)
        tg = rt.TrackGenerator(model, 16, 0.004)
        rt.trace(tg)
        om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
        ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step,
                            iter_cap=4000000, n_threads=0)
        aq = tg.azimuthal_quadrature
        vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
        for opts in (dict(split=0, lean=lean, topo=2), dict(split=0, lean=lean, topo=2, pool_chunks_hint=8, side_entries_hint=4)):
            for k, r in enumerate(_run(rt, tg, opts)):
                _same(r, ref, vol, (seed, opts, k))
                assert r[5]["lean"] == lean, r[5]


def test_lean_plan_at_the_headline_configuration(rt, traced, oracle_run):
    """C3: 130,456 tracks, 9.3 M records — one residency round of k_cheap; the records equal the one-kernel plan's and the checker's."""
    tg = traced(128, 1e-3)
    ref = oracle_run(tg)
    for lean in (1, 2):
        r = _run(rt, tg, dict(lean=lean))[1]
        _same(r, ref, ref["volumes"], lean)
        print(f"lean {lean}: {r[5]['lean_queued']} lanes finished by k_serve of {tg.n_total_tracks}")
