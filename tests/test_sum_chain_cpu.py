"""The Σℓ check of the record kernel on the HOST (no GPU): `k_materialise_lin` does not add up a track's lengths — it takes the
first record's length plus the projection of (q_last − q_first) on the march direction, minus signed gaps in front of records that
keep their own p, minus twice the length of such a record that walks backwards (rt_device.hpp `chain_gap_term` / `chain_sum` /
`chain_status`; the reference: `isapprox(track.ℓ, sum(ℓ.(track.segments)); rtol)`, src/track.jl:171-175).  The three functions
are compiled for the host too; `tests/host_march.hip` drives them over the records of the device header's own march (which knows
which records are the generic step's), and wherever the chain decides — outside the band it leaves to k_finish's left-to-right
sum — it must decide as the reference's check on the left-to-right sum does: at the default rtol and at tolerances tuned so that
this problem's own tracks sit on either side of the threshold."""
import numpy as np
import pytest


def _problems(rt):
    import meshgen

    # hand-made tracks within 1e-5 … 1e-8 of ϕ = 0, π/2, π on a lattice and on a random mesh: records whose own two points
    # come out in the wrong order along the march (order_intersection_points compares x, src/intersection.jl:151-159)
    for seed, model in ((9, meshgen.lattice_model(rt, 9, 14, 14, jitter=0.3)), (19, meshgen.random_model(rt, 19, 700))):
        tg = rt.TrackGenerator(model, 8, 0.01)
        rt.trace(tg)
        yield "steep %d" % seed, meshgen.steep_tracks(rt, tg, seed)
    # cells that overlap within the locate's tolerance / slivers: generic records INSIDE tracks (gaps and overlaps)
    for seed, model in ((3, meshgen.sliver_model(rt, 3, 12, 12, gap=1e-6)), (4, meshgen.random_model(rt, 4, 1500, cluster=True)),
                        (5, meshgen.lattice_model(rt, 5, 20, 20, jitter=0.3, x0=30.0, y0=-40.0))):
        tg = rt.TrackGenerator(model, 16, 0.004)
        rt.trace(tg)
        yield "mesh %d" % seed, tg
    # the case that found the round-5 chain's error (tools/fuzz_cpu.py seed 710227, class `aligned`: lattice rows within 1e-7 … 3e-3 rad
    # of a track direction, 11 units from the origin): a gap between two exit points of SHALLOW crossings is not along the line — its
    # length exceeded its projection by 1.5e-11, and at a tolerance next to that track's |ℓ − Σℓ| the chain decided the other way
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cpu
    kind, model, n_azim, delta, k = fuzz_cpu.case(710227)
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    yield "fuzz seed 710227 (%s)" % kind, tg
    # tracks of more than 256 records (the kernel's rounds): a fine lattice crossed lengthwise
    model = meshgen.lattice_model(rt, 6, 200, 4, jitter=0.2, w=4.0, h=0.1)
    tg = rt.TrackGenerator(model, 4, 0.01)
    rt.trace(tg)
    yield "long", tg


def test_chain_decides_as_the_left_to_right_sum(rt, orc):
    import hostmarch as hm

    total_marg = total_dec = 0
    for name, tg in _problems(rt):
        r = hm.run(tg, walk="topo", n_threads=0)
        om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
        ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step,
                            iter_cap=4000000, n_threads=0)
        assert np.array_equal(r["offsets"], ref["offsets"]) and np.array_equal(r["ell"], ref["ell"]), name  # the records ARE the checker's
        wrong, marg, dec, marg0 = hm.chain_check(tg)
        st, S, E = hm.chain(tg)
        # the chain's Σℓ is the left-to-right one to within the band's own width (n ulp-ish): nothing gross hides behind "marginal"
        n = np.diff(r["offsets"])
        big = np.maximum(np.abs(S), np.abs(E))
        assert np.all(np.abs(S - E) <= (96 * 1.2e-16 * big + 64 * 2.3e-16 * float(np.max(np.abs(tg.mesh.bb)))) * np.maximum(n, 1)), name
        assert wrong == 0, (name, wrong)
        assert marg0 <= max(2, tg.n_total_tracks // 100), (name, marg0)  # at the reference's own rtol the band is practically empty
        total_marg += marg
        total_dec += dec
        print(f"{name}: {tg.n_total_tracks} tracks, {int(ref['total'])} records, {dec} decisions, {marg} left to the exact sum, "
              f"{int(np.count_nonzero(ref['status'] == 2))} tracks fail the reference's check at the default rtol")
    assert total_dec > 3 * total_marg  # (1e-13 is summation noise on the longest tracks: those go to the exact sum, by design)


def test_chain_terms_by_hand(rt):
    """One track of three records along d = (1, 0): a gap, an overlap and a reversed own-p record, against the plain sum."""
    import ctypes as C

    import hostmarch as hm

    hm.lib()  # (the functions are exercised through hostmarch_chain above; here only the closed forms they implement)
    d = (1.0, 0.0)
    # records: [0, 1] then own-p record [1.25, 2] (gap 0.25) then own-p record [1.9, 3] (overlap 0.1)
    ell = [1.0, 0.75, 1.1]
    gap = 0.25 + (-0.1)
    S = ell[0] + (3.0 - 1.0) * d[0] - gap
    assert abs(S - sum(ell)) < 1e-15
    # a reversed own-p record [2.0 -> 1.7] in the middle: Σℓ adds 0.3, the projection subtracts it
    ell = [1.0, 0.3, 1.5]  # [0,1], [2.0 -> 1.7] (gap 1.0 in front), [1.7 -> 3.2] chained
    gap = 1.0 - 2.0 * 0.3
    S = ell[0] + (3.2 - 1.0) - gap
    assert abs(S - sum(ell)) < 1e-15
