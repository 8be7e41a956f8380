"""tests/c_abi_smoke.c: a plain-C caller of the C ABI that makes the Julia shim's call sequence (dlopen by path,
1-based CSR ptrs, pinned fetch, "%d" substitution into the status message).  Without a GPU the harness must build and
the library must refuse to compute; on the GPU box its result checksums must equal the checker's, bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "build", "c_abi_smoke")
LIB = os.path.join(ROOT, "raytracing.jl_amd", "csrc", "librt_segmentize.so")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-o", EXE, SRC, "-ldl"])
    return EXE


def _bits64(a):
    w = np.ascontiguousarray(a).view(np.uint64)
    return int((np.arange(1, len(w) + 1, dtype=np.uint64) * w).sum(dtype=np.uint64))


def _bits32(a):
    w = np.ascontiguousarray(a, np.int32).view(np.uint32).astype(np.uint64)
    return int((np.arange(1, len(w) + 1, dtype=np.uint64) * w).sum(dtype=np.uint64))


def test_harness_builds_and_library_refuses_without_gpu(rt):
    exe = _build()
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, LIB, rt.data_path("pincell.msh"), "8", "0.02"], capture_output=True, text=True)
    assert r.returncode == 3 and "no GPU" in r.stderr  # rt_device_count() == 0: nothing is computed on the host


@pytest.mark.gpu
@pytest.mark.parametrize("n_azim,delta,shards", [(8, 0.02, 0), (32, 5e-3, 0), (32, 5e-3, 3)])
def test_c_caller_matches_checker(rt, orc, n_azim, delta, shards):
    """shards > 0: the same through rt_multi_* (device_ids = {0, 0, 0}), what segmentize_amd_multi! of the shim calls."""
    exe = _build()
    r = subprocess.run([exe, LIB, rt.data_path("pincell.msh"), str(n_azim), str(delta), "0", str(shards)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    model = rt.GmshDiscreteModel(rt.data_path("pincell.msh"))
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    om = orc.OracleMesh.from_mesh(tg.mesh, omp=True)
    ref = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell, cos_phi=tg.cos_phi, sin_phi=tg.sin_phi, n_threads=0)
    aq = tg.azimuthal_quadrature
    vol = om.fill_volumes(ref["offsets"], tg.azim_idx, aq.delta_s, aq.n_azim_2)
    assert got["n_tracks"] == tg.n_total_tracks and got["tracks_px"] == _bits64(tg.px)  # same inputs
    assert got["total"] == ref["total"] == got["walked"] and got["n_failed"] == 0 and got["walk_enabled"] == (-1 if shards else 1)
    assert got["sum_offsets"] == _bits64(ref["offsets"]) and got["sum_status"] == 0
    for k in ("px", "py", "qx", "qy", "ell"):
        assert got[k] == _bits64(ref[k]), k
    assert got["element"] == _bits32(ref["element"])
    assert abs(got["volumes_sum"] - vol.sum()) < 1e-10
    if not shards:  # the records in completion order through the per-track table (a second handle, option "record_order" 2)
        assert got["table_ok"] == 1 and got["table_order"] == 1, got
    if not shards:  # the sweep the C caller ran on the device against a sequential sweep over the checker's records
        import sweep_ref

        G, nc, n = 2, tg.mesh.num_cells, tg.n_total_tracks
        i = np.arange(nc * G, dtype=np.float64)
        sig = (0.2 + 1.4 * i / (nc * G - 1)).reshape(nc, G)
        src = (i / (nc * G - 1)).reshape(nc, G)
        phi, out = sweep_ref.sweep(ref["offsets"], ref["ell"], ref["element"], sig, src, aq.delta_s[tg.azim_idx - 1], np.ones((2, n, G)))
        assert got["sweep_input"] in (1, 2)
        assert abs(got["sweep_phi_sum"] - phi.sum()) <= 1e-11 * np.abs(phi).sum()
        assert abs(got["sweep_psi_out_sum"] - out.sum()) <= 1e-12 * out.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [0, 4])
def test_c_caller_reports_the_reference_error_text(rt, shards):
    exe = _build()
    r = subprocess.run([exe, LIB, rt.data_path("pincell.msh"), "8", "0.02", "17", str(shards)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert got["n_failed"] == 1 and got["first_uid"] == 17 and got["first_status"] == 2
    assert got["message"].startswith("Track with `uid` 17 has a length that do not match")
