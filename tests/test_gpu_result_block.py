"""The host block the library owns for a call's results (GPU only; rt_result_alloc / rt_result_fetch / rt_result_free, round 6).

`segmentize!` leaves `track.segments` on the host (src/trackgenerator.jl:357-369; README.md:127-135 reads them from there); the copy
into fresh host memory is most of one call.  The block is mapped and faulted in by the library's own threads beside the upload and the
kernels; what it holds must be exactly what rt_fetch_offsets / rt_fetch_segments return, and the arrays must outlive the handles."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracks(_capi, dm, tg):
    return _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)


@pytest.mark.parametrize("hint", [0, 7, 10 ** 7])
def test_result_block_holds_what_the_fetches_return(rt, traced, oracle_run, hint):
    from raytracing_jl_amd import _capi

    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    aq = tg.azimuthal_quadrature
    dm = _capi.DeviceMesh(tg.mesh, 0)
    blk = dm.result_alloc(tg.n_total_tracks, float(tg.ell.sum()), hint)  # (7: far too small — the fetch replaces the block)
    dt = _tracks(_capi, dm, tg)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st, seg = dt.fetch_result(blk)
    off2, st2 = dt.fetch_offsets()
    seg2 = dt.fetch_segments()
    assert total == ref["total"] and np.array_equal(off, off2) and np.array_equal(st, st2)
    for k in ("px", "py", "qx", "qy", "ell", "element"):
        assert np.array_equal(seg[k], seg2[k]) and np.array_equal(seg[k], ref[k]), k
    # a second segmentation into the same block (the arrays are rewritten in place), then everything else goes away
    dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off3, st3, seg3 = dt.fetch_result(blk)
    assert np.array_equal(off3, off2) and np.array_equal(seg3["ell"], ref["ell"])
    dt.close(); dm.close()
    del blk, off, st, seg, off3, st3
    gc.collect()
    assert np.array_equal(seg3["qx"], ref["qx"]) and np.array_equal(seg3["element"], ref["element"])  # the arrays keep the block alive
    seg3["qx"][0] = 1.0  # ... and are ordinary writable memory
    del seg3
    gc.collect()


def test_reference_shaped_call_goes_through_the_block(rt, traced, oracle_run):
    """rt.segmentize(tg) — the reference's call — fetches through a result block by default; "fresh" is rounds 4-5's path."""
    tg = traced(32, 5e-3)
    ref = oracle_run(tg)
    for fetch in (True, "fresh", "pinned"):
        rt.segmentize(tg, fetch=fetch)
        s = tg.segments
        assert np.array_equal(s.offsets, ref["offsets"]) and np.array_equal(s.element, ref["element"]), fetch
        for k in ("px", "py", "qx", "qy", "ell"):
            assert np.array_equal(getattr(s, k), ref[k]), (k, fetch)


def test_result_block_at_the_headline_configuration(rt, traced):
    from raytracing_jl_amd import _capi

    tg = traced(128, 1e-3)
    aq = tg.azimuthal_quadrature
    dm = _capi.DeviceMesh(tg.mesh, 0)
    import time
    t0 = time.perf_counter()
    blk = dm.result_alloc(tg.n_total_tracks, float(tg.ell.sum()))
    dt = _tracks(_capi, dm, tg)
    total = dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    off, st, seg = dt.fetch_result(blk)
    ms = (time.perf_counter() - t0) * 1e3
    seg2 = dt.fetch_segments()
    assert int(off[-1]) == total and all(np.array_equal(seg[k], seg2[k]) for k in seg)
    print(f"C3 in one sequence (alloc, upload, first call, fetch): {ms:.1f} ms for {total} records")
    dt.close(); dm.close()
