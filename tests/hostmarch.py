"""ctypes wrapper of tests/host_march.hip — TEST INFRASTRUCTURE (see that file's header): the device march's
per-lane logic and the mesh preprocessing compiled for the host, to fuzz the walk step's certificates against the
CPU checker without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "host_march.hip")
_OUT = os.path.join(_HERE, "build", "libhostmarch.so")
_CSRC = os.path.join(_HERE, "..", "raytracing.jl_amd", "csrc")
_lib = None
_dp, _ip, _lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def build() -> str:
    if os.environ.get("HOSTMARCH_LIB"):  # a private copy (tools/fuzz_cpu.py: long runs must not see rebuilds)
        return os.environ["HOSTMARCH_LIB"]
    deps = [_SRC, os.path.join(_CSRC, "rt_device.hpp"), os.path.join(_CSRC, "rt_mesh_prep.hpp")]
    if not os.path.exists(_OUT) or any(os.path.getmtime(d) > os.path.getmtime(_OUT) for d in deps):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-fPIC", "-shared", "-pthread", "-Wno-unused-function", "-o", _OUT, _SRC])
    return _OUT


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.hostmarch_run.restype = C.c_int64
        L.hostmarch_run.argtypes = [_dp, _dp, C.c_int32, _ip, C.c_int32, _ip, _ip, _dp, C.c_int64] + [_dp] * 9 + \
            [C.c_double, C.c_int32, C.c_double, C.c_int64, C.c_int32, C.c_int32, _dp]
        L.hostmarch_fetch.argtypes = [_lp, _ip] + [_dp] * 5 + [_ip, _lp]
        L.hostmarch_prep.restype = C.c_int32
        L.hostmarch_prep.argtypes = [_dp, _dp, C.c_int32, _ip, C.c_int32, _dp, _ip, _ip, _ip, _dp, C.c_char_p, C.c_int32]
        L.hostmarch_bf16.argtypes = [_dp, C.c_int64, C.POINTER(C.c_uint16), _dp]
        L.hostmarch_one_minus_exp_neg.argtypes = [_dp, C.c_int64, _dp]
        L.hostmarch_one_minus_exp_neg_thin.argtypes = [_dp, C.c_int64, _dp]
        L.hostmarch_chain.argtypes = [C.c_int64, _dp, _dp, _dp, C.c_double, C.c_double, _ip, _dp, _dp]
        L.hostmarch_topo.restype = C.c_int32
        L.hostmarch_topo.argtypes = [_dp, _dp, C.c_int32, _ip, C.c_int32, _dp, _ip] + [_dp] * 6
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, np.float64)


def _i(a):
    return np.ascontiguousarray(a, np.int32)


INFO = ("walk_ok", "records", "records_walk", "eps_min", "eps_max", "d_vertex", "cells_fragile", "cells_degenerate")


def run(tg, *, k=5, rtol=None, walk=True, iter_cap=4000000, n_threads=0):
    """March every track of ``tg`` on the host with the device header's logic.  Returns a dict shaped like the
    checker's (offsets, status, element, px, py, qx, qy, ell, total) + ``stats`` and ``info``."""
    mesh = tg.mesh
    x, y = _f(mesh.x), _f(mesh.y)
    cn = _i(np.asarray(mesh.cell_nodes).reshape(-1))
    ptr = _i(np.asarray(mesh.node_cells_ptrs) - int(mesh.node_cells_ptrs[0]))
    dat = _i(mesh.node_cells_data)
    bb = _f(mesh.bb)
    arrs = [_f(a) for a in (tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell)]
    n = len(arrs[0])
    info = np.zeros(8)
    p = lambda a, t: a.ctypes.data_as(t)
    total = lib().hostmarch_run(p(x, _dp), p(y, _dp), len(x), p(cn, _ip), len(cn) // 3, p(ptr, _ip), p(dat, _ip), p(bb, _dp),
                                n, *[p(a, _dp) for a in arrs], float(tg.tiny_step), int(k),
                                float(rtol if rtol is not None else 1.4901161193847656e-8), int(iter_cap),
                                (2 if walk == 'topo' else 1) if walk else 0, int(n_threads), p(info, _dp))
    if total < 0:
        raise RuntimeError("hostmarch_run failed")
    out = {k2: np.zeros(total, np.float64) for k2 in ("px", "py", "qx", "qy", "ell")}
    out["element"] = np.zeros(total, np.int32)
    out["offsets"] = np.zeros(n + 1, np.int64)
    out["status"] = np.zeros(max(n, 1), np.int32)
    stats = np.zeros(8, np.int64)
    lib().hostmarch_fetch(p(out["offsets"], _lp), p(out["status"], _ip), *[p(out[k2], _dp) for k2 in ("px", "py", "qx", "qy", "ell")],
                          p(out["element"], _ip), p(stats, _lp))
    out["status"] = out["status"][:n]
    out["total"] = int(total)
    out["stats"] = dict(walk_emits=int(stats[0]), walk_skips=int(stats[1]), generic_emits=int(stats[2]),
                        generic_iterations=int(stats[3]), refused=int(stats[4]),
                        cheap_emits=int(stats[5]), cheap_refused=int(stats[6]), cheap_restarts=int(stats[7]))
    out["info"] = {k2: float(info[i]) for i, k2 in enumerate(INFO)}
    return out


def prep(mesh):
    """The preprocessing's per-record certificate fields: extras[n_cells,3], epscode[n_cells,3] (-1: walk off),
    cls[n_cells], info dict, note."""
    x, y = _f(mesh.x), _f(mesh.y)
    cn = _i(np.asarray(mesh.cell_nodes).reshape(-1))
    bb = _f(mesh.bb)
    nc = len(cn) // 3
    extras, code, cls = np.zeros(3 * nc, np.int32), np.zeros(3 * nc, np.int32), np.zeros(nc, np.int32)
    info = np.zeros(8)
    note = C.create_string_buffer(256)
    p = lambda a, t: a.ctypes.data_as(t)
    lib().hostmarch_prep(p(x, _dp), p(y, _dp), len(x), p(cn, _ip), nc, p(bb, _dp), p(extras, _ip), p(code, _ip), p(cls, _ip),
                         p(info, _dp), note, 256)
    return dict(extras=extras.reshape(nc, 3), epscode=code.reshape(nc, 3), cls=cls,
                info={k2: float(info[i]) for i, k2 in enumerate(INFO)}, note=note.value.decode())


def bf16_up(v):
    """(pattern, value) of rtprep::bf16_up for every entry of ``v``: the smallest bfloat16 >= v (v > 0)."""
    v = _f(v)
    pat, val = np.zeros(len(v), np.uint16), np.zeros(len(v))
    lib().hostmarch_bf16(v.ctypes.data_as(_dp), len(v), pat.ctypes.data_as(C.POINTER(C.c_uint16)), val.ctypes.data_as(_dp))
    return pat, val


def topo_records(mesh):
    """The cheap-step records of ``mesh`` as the device decodes them: dict of [n_cells, 3] arrays extras, E, g1, k2, dtf, lc +
    the scalars tiny_max, rmax, end_err, l_min, d_vertex, on."""
    x, y = _f(mesh.x), _f(mesh.y)
    cn = _i(np.asarray(mesh.cell_nodes).reshape(-1))
    bb = _f(mesh.bb)
    nc = len(cn) // 3
    extras = np.zeros(3 * nc, np.int32)
    arr = [np.zeros(3 * nc) for _ in range(5)]
    sc = np.zeros(6)
    p = lambda a, t: a.ctypes.data_as(t)
    lib().hostmarch_topo(p(x, _dp), p(y, _dp), len(x), p(cn, _ip), nc, p(bb, _dp), p(extras, _ip), *[p(a, _dp) for a in arr], p(sc, _dp))
    out = dict(extras=extras.reshape(nc, 3), **{k: a.reshape(nc, 3) for k, a in zip(("E", "g1", "k2", "dtf", "lc"), arr)})
    out.update(tiny_max=sc[0], rmax=sc[1], end_err=sc[2], l_min=sc[3], d_vertex=sc[4], on=bool(sc[5]))
    return out


def one_minus_exp_neg(tau, thin=False):
    """rt::one_minus_exp_neg (rt_device.hpp): the sweep's 1 - exp(-tau), evaluated on the host; `thin`: its form for
    optically thin segments (tau < 1/8: the series without range reduction)."""
    tau = _f(tau)
    out = np.zeros(len(tau))
    f = lib().hostmarch_one_minus_exp_neg_thin if thin else lib().hostmarch_one_minus_exp_neg
    f(tau.ctypes.data_as(_dp), len(tau), out.ctypes.data_as(_dp))
    return out


def chain(tg, rtol=None):
    """The Σℓ check of ``k_materialise_lin`` (rt_device.hpp ``chain_*``) on the records of the LAST ``run(tg)``: per track the
    chain's status (0 OK, 1 LENGTH_MISMATCH, 2 inside the band that k_finish decides with its left-to-right sum), the chain's Σℓ
    and the left-to-right Σℓ of the records."""
    n = len(tg.ell)
    st = np.zeros(n, np.int32)
    S, E = np.zeros(n), np.zeros(n)
    cs, sn, ell = _f(tg.cos_phi), _f(tg.sin_phi), _f(tg.ell)
    p = lambda a, t: a.ctypes.data_as(t)
    lib().hostmarch_chain(n, p(cs, _dp), p(sn, _dp), p(ell, _dp), float(rtol if rtol is not None else 1.4901161193847656e-8),
                          float(np.max(np.abs(tg.mesh.bb))), p(st, _ip), p(S, _dp), p(E, _dp))
    return st, S, E


def chain_check(tg, rtols=None):
    """Does the chain decide as the reference's check (src/track.jl:171: isapprox(ℓ, Σℓ; rtol) on the LEFT-TO-RIGHT sum of the
    records' lengths) wherever it decides at all?  Run on the records of the last ``run(tg)``: at the default rtol, at three fixed
    ones, and at tolerances put next to this problem's own tracks — for up to eight tracks whose |ℓ − Σℓ| / max(ℓ, Σℓ) is real
    (above 1e-11: a skipped sliver, an overlap), that ratio times (1 ± 1e-7): the track just fails / just passes.  Returns
    (tracks decided wrongly, tracks left to the exact sum, decisions made, tracks left to the exact sum at the default rtol)."""
    L = _f(tg.ell)
    wrong = marginal = decided = 0
    st, S, E = chain(tg)
    marg_default = int(np.count_nonzero(st == 2))
    big = np.maximum(np.abs(L), np.abs(E))
    ratio = np.abs(L - E) / np.where(big > 0, big, 1.0)
    if rtols is None:
        real = np.sort(ratio[ratio > 1e-11])
        pick = real[np.linspace(0, len(real) - 1, min(8, len(real))).astype(int)] if len(real) else []
        rtols = [1.4901161193847656e-8, 1e-6, 1e-10, 1e-13] + [float(v) * f for v in pick for f in (1 - 1e-7, 1 + 1e-7)]
    for rtol in rtols:
        st, S, E = chain(tg, rtol)
        ref_fail = ~((L == E) | (np.abs(L - E) <= rtol * big))
        dec = st != 2
        wrong += int(np.count_nonzero(dec & ((st == 1) != ref_fail)))
        marginal += int(np.count_nonzero(~dec))
        decided += int(np.count_nonzero(dec))
    return wrong, marginal, decided, marg_default
