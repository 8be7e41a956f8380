"""Sequential restatement (numpy, vectorised over tracks) of the transport sweep `rt_sweep` runs on the device — the checker of
tests/test_gpu_sweep.py, evaluated over the ORACLE's segment records.  The consumption pattern is the reference's
(README.md:127-135: for track in tg.tracks_by_uid, for segment in track.segments; backward traversals read the segments
reversed, demo/makie.jl:103; fluxes pass through next_track_fwd / next_track_bwd with dir_next_track_*, src/track.jl:42-77)."""
import numpy as np

VACUUM = 0


def sweep(offsets, ell, element, sigma_t, source, weight, psi_in):
    """One sweep.  offsets [n+1], ell / element [total] (element 1-based), sigma_t / source [n_cells, G], weight [n],
    psi_in [2, n, G].  Returns (phi [n_cells, G], psi_out [2, n, G])."""
    offsets = np.asarray(offsets, np.int64)
    n = len(offsets) - 1
    G = sigma_t.shape[1]
    cnt = np.diff(offsets)
    qs = np.where(sigma_t > 0, source / np.where(sigma_t > 0, sigma_t, 1.0), 0.0)
    phi = np.zeros_like(sigma_t, dtype=np.float64)
    psi_out = np.zeros((2, n, G))
    for d in (0, 1):
        psi = np.array(psi_in[d], np.float64, copy=True)
        for t in range(int(cnt.max()) if n else 0):
            act = np.nonzero(cnt > t)[0]
            idx = offsets[act] + (t if d == 0 else cnt[act] - 1 - t)
            e = element[idx] - 1
            tau = sigma_t[e] * ell[idx][:, None]
            dd = (psi[act] - qs[e]) * (-np.expm1(-tau))
            psi[act] = psi[act] - dd
            np.add.at(phi, e, weight[act][:, None] * dd)
        psi_out[d] = psi
    return phi, psi_out


def sweep_fast(offsets, ell, element, sigma_t, source, weight, psi_in):
    """`sweep` for batches of 10^5 tracks: the same steps, the tallies with one bincount per group instead of np.add.at (which takes
    ~0.1 s per step there).  Pinned against `sweep` by tests/test_sweep_ref_cpu.py."""
    offsets = np.asarray(offsets, np.int64)
    n = len(offsets) - 1
    nc, G = sigma_t.shape
    cnt = np.diff(offsets)
    qs = np.where(sigma_t > 0, source / np.where(sigma_t > 0, sigma_t, 1.0), 0.0)
    phi = np.zeros_like(sigma_t, dtype=np.float64)
    psi_out = np.zeros((2, n, G))
    order = np.argsort(-cnt, kind="stable")  # tracks by length: the active set of step t is a prefix
    cs = cnt[order]
    for d in (0, 1):
        psi = np.array(psi_in[d], np.float64, copy=True)
        for t in range(int(cnt.max()) if n else 0):
            act = order[:int(np.searchsorted(-cs, -t, side="left"))]  # tracks with cnt > t
            idx = offsets[act] + (t if d == 0 else cnt[act] - 1 - t)
            e = element[idx] - 1
            dd = (psi[act] - qs[e]) * (-np.expm1(-sigma_t[e] * ell[idx][:, None]))
            psi[act] = psi[act] - dd
            wd = weight[act][:, None] * dd
            for g in range(G):
                phi[:, g] += np.bincount(e, weights=wd[:, g], minlength=nc)
        psi_out[d] = psi
    return phi, psi_out


def link(psi_out, next_fwd, next_bwd, dir_fwd, dir_bwd, bc_fwd, bc_bwd):
    """The boundary flux of the next sweep: the flux track u ends its forward (backward) traversal with becomes the incoming
    flux of next_track_fwd (next_track_bwd) in direction dir_next_track_fwd (.._bwd); 0 behind a Vacuum boundary."""
    n = psi_out.shape[1]
    nxt = np.zeros_like(psi_out)
    for u in range(n):  # uid ascending, forward before backward: the order of a sequential sweep
        for d, (nx, dr, bc) in enumerate(((next_fwd, dir_fwd, bc_fwd), (next_bwd, dir_bwd, bc_bwd))):
            nxt[int(dr[u]), int(nx[u]) - 1] = 0.0 if int(bc[u]) == VACUUM else psi_out[d, u]
    return nxt
