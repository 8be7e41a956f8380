"""The library's -DRT_EXPERIMENTAL build (GPU only): the two-pass march — count, scan, march again writing at the CSR offsets;
fill_volumes with global atomics or as its own pass — is round 1's first correct path and an independent cross-check of the
staging / compaction / two-phase machinery.  The default build does not carry it (`rt_set_option` refuses its options); this
test builds the library with the flag into a scratch directory, loads it in a child process (RT_SEGMENTIZE_LIB) and compares
the two-pass results with the checker's, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
from oracle import oracle as orc
orc.build()
for mesh, n_azim, delta in (("pincell.json", 8, 2e-2), ("pincell.json", 32, 5e-3), ("bwr_like.msh", 16, 2e-2)):
    path = rt.data_path(mesh)
    model = rt.GmshDiscreteModel(path) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
    tg = rt.TrackGenerator(model, n_azim, delta); rt.trace(tg); aq = tg.azimuthal_quadrature
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, tiny_step=tg.tiny_step, n_threads=0)
    vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    for opts in (dict(single_pass=0), dict(single_pass=0, volumes_mode=1), dict(single_pass=0, walk=0), dict(single_pass=1, split=0)):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        for k, v in opts.items():
            dm.set_option(k, v)
        dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
        assert dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2) == ref["total"], opts
        off, st = dt.fetch_offsets(); s = dt.fetch_segments()
        assert np.array_equal(off, ref["offsets"]) and np.array_equal(st, ref["status"]) and np.array_equal(s["element"], ref["element"]), opts
        for f in ("px", "py", "qx", "qy", "ell"):
            assert np.array_equal(s[f], ref[f]), (f, opts)
        assert np.allclose(dt.fetch_volumes(), vol, rtol=1e-10, atol=0), opts
        dt.close(); dm.close()
print("experimental build ok")
"""


def test_two_pass_march_of_the_experimental_build(tmp_path):
    csrc = os.path.join(ROOT, "raytracing.jl_amd", "csrc")
    lib = str(tmp_path / "librt_segmentize_experimental.so")
    build = subprocess.run(["make", "-j4", "EXTRA=-DRT_EXPERIMENTAL", "LIB=" + lib, "BUILD=" + str(tmp_path / "obj")], cwd=csrc, capture_output=True,
                           text=True, timeout=1200)
    assert build.returncode == 0, build.stderr[-2000:]
    # the default library refuses the option; the experimental one takes it
    from raytracing_jl_amd import _capi
    import raytracing_jl_amd as rt

    model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
    dm = _capi.DeviceMesh(rt.TrackGenerator(model, 8, 2e-2).mesh, 0)
    with pytest.raises(RuntimeError, match="RT_EXPERIMENTAL"):
        dm.set_option("single_pass", 0)
    dm.close()
    child = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=dict(os.environ, RT_SEGMENTIZE_LIB=lib), capture_output=True,
                           text=True, timeout=900)
    assert child.returncode == 0 and "experimental build ok" in child.stdout, (child.stdout[-1500:], child.stderr[-3000:])
