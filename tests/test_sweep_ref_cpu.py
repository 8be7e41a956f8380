"""The sequential sweep that checks rt_sweep (tests/sweep_ref.py), on the CPU: properties a transport sweep must have,
evaluated over the oracle's records — so the checker itself is pinned before the GPU tests lean on it."""
import numpy as np

import sweep_ref


def test_flat_source_equilibrium_and_pure_absorber(rt, traced, oracle_run):
    tg = traced(8, 2e-2)
    ref = oracle_run(tg)
    nc, n, G = tg.mesh.num_cells, tg.n_total_tracks, 3
    rng = np.random.default_rng(3)
    sigma_t = rng.uniform(0.1, 2.0, (nc, G))
    w = np.ones(n)
    # ψ = q/Σt everywhere is the fixed point: nothing is attenuated, nothing is tallied
    level = np.array([0.5, 1.0, 2.0])
    phi, out = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, sigma_t * level, w, np.broadcast_to(level, (2, n, G)))
    assert np.allclose(out, level, rtol=1e-13) and np.abs(phi).max() < 1e-12
    # no source, one group, uniform Σt: ψ_out = ψ_in·exp(−Σt·Σℓ) and the tally is what was absorbed
    st = np.full((nc, 1), 0.7)
    psi_in = rng.uniform(0.5, 1.5, (2, n, 1))
    phi, out = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], st, np.zeros((nc, 1)), w, psi_in)
    sum_ell = np.add.reduceat(ref["ell"], ref["offsets"][:-1])
    assert np.allclose(out[:, :, 0], psi_in[:, :, 0] * np.exp(-0.7 * sum_ell), rtol=1e-12)
    assert np.isclose(phi.sum(), (psi_in - out).sum(), rtol=1e-12)


def test_linking_is_a_permutation_for_reflective_and_zero_for_vacuum(rt, pincell):
    for kind, bcs in (("reflective", rt.BoundaryConditions(top=rt.Reflective, bottom=rt.Reflective, left=rt.Reflective, right=rt.Reflective)),
                      ("periodic", rt.BoundaryConditions(top=rt.Periodic, bottom=rt.Periodic, left=rt.Periodic, right=rt.Periodic)),
                      ("vacuum", rt.BoundaryConditions(top=rt.Vacuum, bottom=rt.Vacuum, left=rt.Vacuum, right=rt.Vacuum))):
        tg = rt.TrackGenerator(pincell, 8, 2e-2, bcs=bcs)
        rt.trace(tg)
        n = tg.n_total_tracks
        out = np.arange(1, 2 * n + 1, dtype=np.float64).reshape(2, n, 1)
        nxt = sweep_ref.link(out, tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd)
        if kind == "vacuum":
            assert not nxt.any()
        else:  # every entry receives exactly one outgoing flux (the tracks form closed cycles, src/track.jl:36-38)
            assert np.array_equal(np.sort(nxt.ravel()), out.ravel())


def test_attenuation_factor_is_accurate_to_an_ulp():
    """rt_sweep evaluates 1 − e^{−τ} with its own routine (rt_device.hpp, one_minus_exp_neg: fewer issue cycles than the device
    library's expm1); here it runs on the host against numpy's expm1 over the whole range of optical lengths — denormal to
    beyond the point where the result is 1 — and across the range-reduction boundaries."""
    import hostmarch as hm

    rng = np.random.default_rng(5)
    k = np.arange(1, 61)
    edges = np.concatenate([(k - 0.5) * np.log(2.0), np.nextafter((k - 0.5) * np.log(2.0), 0), np.nextafter((k - 0.5) * np.log(2.0), 100)])
    tau = np.concatenate([10.0 ** rng.uniform(-310, 2, 200000), rng.uniform(0, 2, 200000), rng.uniform(0, 60, 100000), edges,
                          [0.0, 5e-324, 41.4999, 41.5, 41.5001, 700.0, 1e300]])
    got = hm.one_minus_exp_neg(tau)
    ref = -np.expm1(-tau)
    assert np.all(got >= 0) and np.all(got <= 1)
    assert np.all(np.abs(got - ref) <= 4.5e-16 * ref)  # 2 ulp of headroom over numpy's own rounding
    assert got[tau == 0.0][0] == 0.0 and got[-1] == 1.0


def test_thin_attenuation_factor_is_accurate_to_an_ulp():
    """Where every lane's segment is optically thin (τ < 1/8 in every group of the pass) rt_sweep takes the series without range
    reduction (rt_device.hpp, one_minus_exp_neg_thin): same accuracy on [0, 1/8), and within 2 ulp of the general form there."""
    import hostmarch as hm

    rng = np.random.default_rng(6)
    tau = np.concatenate([10.0 ** rng.uniform(-310, -0.91, 200000), rng.uniform(0, 0.125, 300000),
                          [0.0, 5e-324, 1e-300, 0.0625, np.nextafter(0.125, 0)]])
    tau = tau[tau < 0.125]
    got = hm.one_minus_exp_neg(tau, thin=True)
    ref = -np.expm1(-tau)
    assert np.all(got >= 0) and np.all(got < 0.1176)
    assert np.all(np.abs(got - ref) <= 4.5e-16 * ref)
    assert np.all(np.abs(got - hm.one_minus_exp_neg(tau)) <= 4.5e-16 * ref)
    assert got[0] >= 0 and hm.one_minus_exp_neg([0.0], thin=True)[0] == 0.0


def test_fast_sweep_equals_the_plain_one(rt, traced, oracle_run):
    """sweep_fast (bincount tallies, the active tracks as a prefix of the tracks sorted by length) is what the C4-sized GPU test
    uses: the same numbers as `sweep` up to the order of the tallies' additions."""
    tg = traced(16, 1e-2)
    ref = oracle_run(tg)
    nc, n, G = tg.mesh.num_cells, tg.n_total_tracks, 3
    rng = np.random.default_rng(5)
    sigma_t = rng.uniform(0.05, 3.0, (nc, G)); source = rng.uniform(0.0, 2.0, (nc, G))
    w = rng.uniform(0.5, 1.5, n); psi_in = rng.uniform(0.0, 1.5, (2, n, G))
    phi, out = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, w, psi_in)
    phi2, out2 = sweep_ref.sweep_fast(ref["offsets"], ref["ell"], ref["element"], sigma_t, source, w, psi_in)
    assert np.array_equal(out, out2)
    assert np.abs(phi - phi2).max() <= 1e-13 * np.abs(phi).max()
