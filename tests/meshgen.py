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
