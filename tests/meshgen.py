"""Deterministic mesh generators for the parity tests and fuzzers (scipy Delaunay of seeded point clouds; cells
kept in the generator's order with ascending node ids per cell, like Gridap's oriented grids)."""
import numpy as np


def _model(rt, xy, drop_area2=1e-14):
    from scipy.spatial import Delaunay

    xy = np.asarray(xy, dtype=np.float64)
    tri = Delaunay(xy)
    cells = np.sort(tri.simplices.astype(np.int32) + 1, axis=1)
    # drop degenerate (zero-area) triangles Delaunay may emit on the collinear boundary points
    a, b, c = xy[cells[:, 0] - 1], xy[cells[:, 1] - 1], xy[cells[:, 2] - 1]
    area2 = np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1]))
    cells = cells[area2 > drop_area2]
    return rt.DiscreteModel(xy, cells)


def _border(w, h, nb, x0, y0):
    tx = np.arange(nb) * (w / nb)
    ty = np.arange(nb) * (h / nb)
    return [(x0 + v, y0) for v in tx] + [(x0 + w, y0 + v) for v in ty] + [(x0 + w - v, y0 + h) for v in tx] + \
           [(x0, y0 + h - v) for v in ty]


def random_model(rt, seed, n_interior, w=1.0, h=1.0, nb=12, x0=0.0, y0=0.0, cluster=False):
    """Uniformly random interior points (slivers, very small and very obtuse triangles occur); ``cluster``: a third
    of them in a tight Gaussian cluster — many tiny cells next to large ones."""
    rng = np.random.default_rng(seed)
    pts = _border(w, h, nb, x0, y0)
    inner = rng.uniform(0.02, 0.98, (n_interior, 2))
    if cluster:
        inner[: n_interior // 3] = 0.5 + 0.01 * rng.standard_normal((n_interior // 3, 2))
        inner = np.clip(inner, 0.02, 0.98)
    pts += [(x0 + w * a, y0 + h * b) for a, b in inner]
    return _model(rt, pts)


def lattice_model(rt, seed, nx, ny, jitter=0.25, w=1.0, h=1.0, x0=0.0, y0=0.0):
    """Jittered lattice: well-shaped cells (what a mesh generator produces); jitter is a fraction of the spacing."""
    rng = np.random.default_rng(seed)
    hx, hy = w / nx, h / ny
    pts = _border(w, h, nx, x0, y0) if nx == ny else _border(w, h, max(nx, ny), x0, y0)
    gx, gy = np.meshgrid(np.arange(1, nx) * hx, np.arange(1, ny) * hy, indexing="xy")
    inner = np.column_stack((gx.ravel(), gy.ravel()))
    inner += rng.uniform(-jitter, jitter, inner.shape) * (hx, hy)
    pts += [(x0 + a, y0 + b) for a, b in inner]
    return _model(rt, pts)


def sliver_model(rt, seed, nx, ny, gap=1e-4, w=1.0, h=1.0, x0=0.0, y0=0.0):
    """Jittered lattice in which every third row of points is pushed to within ``gap``·spacing of the row below:
    bands of needle-shaped cells between well-shaped ones."""
    rng = np.random.default_rng(seed)
    hx, hy = w / nx, h / ny
    pts = _border(w, h, max(nx, ny), x0, y0)
    for j in range(1, ny):
        for i in range(1, nx):
            yy = j * hy + rng.uniform(-0.2, 0.2) * hy
            if j % 3 == 2:
                yy = (j - 1) * hy + 0.25 * hy + gap * hy * (1.0 + rng.uniform(0, 1))
            pts.append((x0 + i * hx + rng.uniform(-0.2, 0.2) * hx, y0 + yy))
    return _model(rt, pts)


# ---- generators aimed at the thresholds of the walk / cheap step certificates ---------------------------------
def box_tracks(rt, n_azim, delta, w=1.0, h=1.0, x0=0.0, y0=0.0):
    """The tracks of a w x h box: trace! depends on the bounding box, nφ and δ only, never on the cells."""
    box = rt.DiscreteModel(np.array([(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)]), np.array([[1, 2, 3], [1, 3, 4]], np.int32))
    tg = rt.TrackGenerator(box, n_azim, delta)
    rt.trace(tg)
    return tg


def near_vertex_model(rt, seed, n_interior, n_azim, delta, w=1.0, h=1.0, x0=0.0, y0=0.0, nb=12):
    """Interior nodes placed ON track lines of the (nφ, δ) quadrature, pushed off them by 10^[-8, -2.5] of the local spacing
    (some exactly on the line): the distances m = min(|s0|, |s1|) and |s_v| of many crossings then sit at the vertex clearance
    d_vertex, at the isolation threshold E·D + g1 and at the chord threshold lc·lcf — from either side."""
    rng = np.random.default_rng(seed)
    tg = box_tracks(rt, n_azim, delta, w, h, x0, y0)
    L = np.sqrt(w * h / max(n_interior, 1))
    u = rng.integers(0, tg.n_total_tracks, n_interior)
    t = rng.uniform(0.03, 0.97, n_interior)
    px = tg.px[u] + t * (tg.qx[u] - tg.px[u])
    py = tg.py[u] + t * (tg.qy[u] - tg.py[u])
    off = L * 10.0 ** rng.uniform(-8.0, -2.5, n_interior) * rng.choice([-1.0, 1.0], n_interior)
    off[rng.uniform(size=n_interior) < 0.05] = 0.0
    px = px - off * tg.sin_phi[u]
    py = py + off * tg.cos_phi[u]
    keep = (px > x0 + 0.01 * w) & (px < x0 + 0.99 * w) & (py > y0 + 0.01 * h) & (py < y0 + 0.99 * h)
    pts = _border(w, h, nb, x0, y0) + list(zip(px[keep], py[keep]))
    return _model(rt, pts)


def aligned_model(rt, seed, n, n_azim, delta, w=1.0, h=1.0, x0=0.0, y0=0.0):
    """A lattice whose rows run along one of the quadrature's track directions, rotated off it by ±10^[-7, -2.5] rad: those
    tracks cross its row edges at vanishing angles — D = |s0| + |s1| at k2 (rounding of the entry / exit point), at dtf / c1
    (the bound on the reference's tiny steps, kub near 4096) — and run through long chains of shallow crossings."""
    rng = np.random.default_rng(seed)
    tg = box_tracks(rt, n_azim, delta, w, h, x0, y0)
    phis = tg.azimuthal_quadrature.phis
    phi = float(phis[rng.integers(0, len(phis))]) + float(rng.choice([-1.0, 1.0])) * 10.0 ** rng.uniform(-7.0, -2.5)
    ux, uy = np.cos(phi), np.sin(phi)
    hs = min(w, h) / n
    cx, cy = x0 + 0.5 * w, y0 + 0.5 * h
    m = int(np.ceil(np.hypot(w, h) / hs)) + 2
    ii, jj = np.meshgrid(np.arange(-m, m + 1), np.arange(-m, m + 1), indexing="xy")
    jit = rng.uniform(-0.02, 0.02, ii.shape + (2,)) * hs * (rng.uniform() < 0.5)
    X = cx + ii * hs * ux - jj * hs * uy + jit[..., 0] * ux
    Y = cy + ii * hs * uy + jj * hs * ux + jit[..., 0] * uy  # (jitter ALONG the rows only: the rows stay straight)
    keep = (X > x0 + 0.02 * w) & (X < x0 + 0.98 * w) & (Y > y0 + 0.02 * h) & (Y < y0 + 0.98 * h)
    pts = _border(w, h, max(4, n), x0, y0) + list(zip(X[keep], Y[keep]))
    return _model(rt, pts)


def steep_tracks(rt, tg, seed, n_per_angle=48):
    """Replace the tracks of a traced TrackGenerator by hand-made ones at angles within 1e-5 … 1e-8 of 0, π/2 and π — angles
    trace! only reaches on domains of extreme aspect — entering on the boundary and leaving on it, with the fields trace! fills
    (src/trackgenerator.jl:179-273: p, q, ϕ, ℓ, ABC = general_form(p, q)).  At ϕ ≈ π/2 the order guard of the walk / cheap
    step (order_intersection_points compares x coordinates, src/intersection.jl:151-159) is what decides."""
    import math

    rng = np.random.default_rng(seed)
    x0, y0, x1, y1 = tg.mesh.bb
    w, h = x1 - x0, y1 - y0
    phis = []
    for e in (1e-5, 1e-6, 1e-7, 1e-8):
        phis += [e, math.pi / 2 - e, math.pi / 2 + e, math.pi - e]
    P = {k: [] for k in ("px", "py", "qx", "qy", "phi", "cs", "sn")}
    for phi in phis:
        cs, sn = math.cos(phi), math.sin(phi)
        for _ in range(n_per_angle):
            if abs(cs) < 0.5:  # near vertical: from the bottom side to the top side
                px, py = x0 + rng.uniform(0.02, 0.98) * w, y0
                t = h / sn
            else:  # near horizontal: from the left (ϕ < π/2) or right side to the opposite one
                px, py = (x0 if cs > 0 else x1), y0 + rng.uniform(0.02, 0.98) * h
                t = w / abs(cs)
            qx, qy = px + t * cs, py + t * sn
            qx, qy = min(max(qx, x0), x1), min(max(qy, y0), y1)
            if abs(cs) < 0.5:
                qy = y1
            else:
                qx = x1 if cs > 0 else x0
            for k, v in zip(("px", "py", "qx", "qy", "phi", "cs", "sn"), (px, py, qx, qy, phi, cs, sn)):
                P[k].append(v)
    a = {k: np.array(v) for k, v in P.items()}
    from raytracing_jl_amd.trackgenerator import _general_form

    A, B, C = _general_form(a["px"], a["py"], a["qx"], a["qy"])
    ex, ey = a["px"] - a["qx"], a["py"] - a["qy"]
    n = len(a["px"])
    tg.px, tg.py, tg.qx, tg.qy = a["px"], a["py"], a["qx"], a["qy"]
    tg.phi, tg.cos_phi, tg.sin_phi = a["phi"], a["cs"], a["sn"]
    tg.ell = np.sqrt(ex * ex + ey * ey)
    tg.A, tg.B, tg.C = A, B, C
    tg.azim_idx = np.ones(n, np.int32)
    tg.track_idx = np.arange(1, n + 1, dtype=np.int32)
    tg.n_total_tracks = n
    return tg
