import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_TESTS = os.path.dirname(os.path.abspath(__file__))
if _TESTS not in sys.path:  # tests/meshgen.py, tests/hostmarch.py
    sys.path.insert(0, _TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import raytracing_jl_amd as rt_

    return rt_


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as orc_

    orc_.build()
    return orc_


@pytest.fixture(scope="session")
def pincell(rt):
    return rt.DiscreteModelFromFile(rt.data_path("pincell.json"))


_TG_CACHE = {}


@pytest.fixture(scope="session")
def traced(rt, pincell):
    """traced(n_azim, delta) -> TrackGenerator on the pincell mesh (cached per session)."""

    def make(n_azim, delta, model=None, **kw):
        key = (id(model), n_azim, delta, tuple(sorted(kw.items())))
        if key not in _TG_CACHE:
            tg = rt.TrackGenerator(model if model is not None else pincell, n_azim, delta, **kw)
            rt.trace(tg)
            _TG_CACHE[key] = tg
        return _TG_CACHE[key]

    return make


@pytest.fixture(scope="session")
def oracle_run(orc):
    """oracle_run(tg) -> dict of the oracle's segmentize on tg's tracks (cached)."""
    cache = {}

    def run(tg, **kw):
        key = (id(tg), tuple(sorted(kw.items())))
        if key not in cache:
            om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
            r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi,
                              sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, n_threads=0, **kw)
            aq = tg.azimuthal_quadrature
            r["volumes"] = om.fill_volumes(r["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
            cache[key] = r
        return cache[key]

    return run


def make_grid_model(rt, nx, ny, x0=0.0, y0=0.0, hx=1.0, hy=1.0, flip=False):
    """Structured right-triangle mesh (every square cut along one diagonal): many parallel and
    collinear edges; with 45-degree tracks the rays pass exactly through vertices."""
    import numpy as np

    xs, ys = np.meshgrid(np.arange(nx + 1) * hx + x0, np.arange(ny + 1) * hy + y0, indexing="xy")
    xy = np.column_stack((xs.ravel(), ys.ravel()))
    cells = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i + 1
            b, c, d = a + 1, a + nx + 1, a + nx + 2
            if flip and (i + j) % 2:
                cells += [sorted((a, b, d)), sorted((a, d, c))]
            else:
                cells += [sorted((a, b, c)), sorted((b, d, c))]
    return rt.DiscreteModel(xy, np.asarray(cells, dtype=np.int32))


@pytest.fixture(scope="session")
def grid_model(rt):
    return lambda *a, **k: make_grid_model(rt, *a, **k)
