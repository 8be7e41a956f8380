"""ctypes front-end of the CPU oracle (``oracle/rt_oracle.c``).  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does (``tests/test_no_oracle_in_product.py``
greps for it).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "build")

STATUS = {0: "ok", 1: "locate_failed", 2: "length_mismatch", 3: "undef_intersection", 4: "iter_cap"}
RTOL_DEFAULT = math.sqrt(np.finfo(np.float64).eps)

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the oracle's shared libraries (gcc; single-thread and OpenMP builds)."""
    src = os.path.join(_HERE, "rt_oracle.c")
    libs = [os.path.join(_BUILD, n) for n in ("liboracle.so", "liboracle_omp.so")]
    fresh = all(os.path.exists(l) and os.path.getmtime(l) >= os.path.getmtime(src) for l in libs)
    if force or not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


_libs = {}


def _lib(omp: bool = False):
    key = "omp" if omp else "st"
    if key in _libs:
        return _libs[key]
    build()
    lib = C.CDLL(os.path.join(_BUILD, "liboracle_omp.so" if omp else "liboracle.so"))
    lib.orc_mesh_create.restype = C.c_void_p
    lib.orc_mesh_create.argtypes = [_f64p, _f64p, C.c_int32, _i32p, C.c_int32, _i32p, _i32p, _f64p]
    lib.orc_mesh_destroy.argtypes = [C.c_void_p]
    lib.orc_mesh_set_bruteforce.argtypes = [C.c_void_p, C.c_int]
    lib.orc_segmentize.restype = C.c_int64
    lib.orc_segmentize.argtypes = [C.c_void_p, C.c_int64, _f64p, _f64p, _f64p, C.c_void_p, C.c_void_p,
                                   _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_int32, C.c_double,
                                   C.c_int64, C.c_int32, _i64p, _i32p, _i64p]
    lib.orc_fetch.restype = C.c_int64
    lib.orc_fetch.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, _f64p, _i32p]
    lib.orc_fill_volumes.argtypes = [C.c_void_p, C.c_int64, _i64p, _i32p, _f64p, C.c_int32, _f64p]
    lib.orc_find_element.restype = C.c_int32
    lib.orc_find_element.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32]
    lib.orc_nn.restype = C.c_int32
    lib.orc_nn.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.orc_knn.restype = C.c_int32
    lib.orc_knn.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, _i32p]
    lib.orc_point_in_triangle.restype = C.c_int32
    lib.orc_point_in_triangle.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double]
    lib.orc_intersections.restype = C.c_int32
    lib.orc_intersections.argtypes = [C.c_void_p, C.c_int32, C.c_double, _f64p, _f64p]
    lib.orc_track_counts.restype = C.c_int64
    lib.orc_track_counts.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_double, _i64p, _i64p]
    lib.orc_trace.restype = C.c_int32
    lib.orc_trace.argtypes = [_f64p, C.c_int32, _i64p, _i64p, _i32p, _f64p, _f64p, _f64p, _i32p, _i32p,
                              _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p,
                              _i8p, _i8p, _i8p, _i8p, _i64p, _i64p]
    lib.orc_num_threads.restype = C.c_int32
    _libs[key] = lib
    return lib


class OracleMesh:
    """Mesh handle of the oracle.  Arguments are the flat arrays of ``Mesh`` (1-based ids)."""

    def __init__(self, x, y, cell_nodes, nc_ptrs, nc_data, bb, omp: bool = False):
        self._lib = _lib(omp)
        self.n_nodes = len(x)
        self.n_cells = len(cell_nodes)
        self._h = self._lib.orc_mesh_create(
            np.ascontiguousarray(x, np.float64), np.ascontiguousarray(y, np.float64), self.n_nodes,
            np.ascontiguousarray(cell_nodes, np.int32).reshape(-1), self.n_cells,
            np.ascontiguousarray(nc_ptrs, np.int32), np.ascontiguousarray(nc_data, np.int32),
            np.ascontiguousarray(bb, np.float64))

    @classmethod
    def from_mesh(cls, mesh, omp: bool = False) -> "OracleMesh":
        return cls(mesh.x, mesh.y, mesh.cell_nodes, mesh.node_cells_ptrs, mesh.node_cells_data, mesh.bb, omp)

    def close(self):
        if self._h:
            self._lib.orc_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bruteforce(self, on: bool):
        self._lib.orc_mesh_set_bruteforce(self._h, int(on))

    # single-point probes
    def find_element(self, x, y, k=2):
        return int(self._lib.orc_find_element(self._h, x, y, k))

    def nn(self, x, y):
        return int(self._lib.orc_nn(self._h, x, y))

    def knn(self, x, y, k, skip=0):
        out = np.zeros(max(int(k), 1), np.int32)
        n = self._lib.orc_knn(self._h, x, y, k, skip, out)
        return out[:n].copy()

    def point_in_triangle(self, cell, x, y):
        return bool(self._lib.orc_point_in_triangle(self._h, cell, x, y))

    def intersections(self, cell, phi, abc):
        pq = np.zeros(4)
        rc = self._lib.orc_intersections(self._h, cell, phi, np.ascontiguousarray(abc, np.float64), pq)
        return rc, pq

    def segmentize(self, px, py, phi, A, B, Cc, ell, cos_phi=None, sin_phi=None, tiny_step=1e-8, k=5,
                   rtol=RTOL_DEFAULT, iter_cap=0, n_threads=1, fetch=True):
        """Returns dict(offsets, status, n_iters, px, py, qx, qy, ell, element); fetch=False: without the six record arrays (they
        stay inside the C handle — bench.py's cpu_baseline times the march alone, not a serial copy into numpy arrays)."""
        n = len(px)
        a = lambda v: np.ascontiguousarray(v, np.float64)
        offsets = np.zeros(n + 1, np.int64)
        status = np.zeros(n, np.int32)
        n_iters = np.zeros(n, np.int64)
        cs = a(cos_phi) if cos_phi is not None else None
        sn = a(sin_phi) if sin_phi is not None else None
        total = self._lib.orc_segmentize(
            self._h, n, a(px), a(py), a(phi),
            cs.ctypes.data_as(C.c_void_p) if cs is not None else None,
            sn.ctypes.data_as(C.c_void_p) if sn is not None else None,
            a(A), a(B), a(Cc), a(ell), tiny_step, k, rtol, iter_cap, n_threads, offsets, status, n_iters)
        out = dict(offsets=offsets, status=status, n_iters=n_iters, total=int(total))
        self._last_total = int(total)
        if not fetch:
            return out
        for name in ("px", "py", "qx", "qy", "ell"):
            out[name] = np.zeros(total, np.float64)
        out["element"] = np.zeros(total, np.int32)
        self._lib.orc_fetch(self._h, out["px"], out["py"], out["qx"], out["qy"], out["ell"], out["element"])
        self._last_total = int(total)
        return out

    def fill_volumes(self, offsets, azim_idx, delta_s, n_azim_2):
        """fill_volumes over the segments of this handle's LAST segmentize (they live inside the C handle)."""
        if getattr(self, "_last_total", None) is None or int(offsets[-1]) != self._last_total:
            raise RuntimeError("fill_volumes: call segmentize on this OracleMesh first (offsets must be that run's)")
        vol = np.zeros(self.n_cells)
        self._lib.orc_fill_volumes(self._h, len(azim_idx), np.ascontiguousarray(offsets, np.int64),
                                   np.ascontiguousarray(azim_idx, np.int32),
                                   np.ascontiguousarray(delta_s, np.float64), n_azim_2, vol)
        return vol


def track_counts(Dx, Dy, n_azim, delta):
    n2 = n_azim // 2
    ntx = np.zeros(max(n2, 1), np.int64)
    nty = np.zeros(max(n2, 1), np.int64)
    total = _lib().orc_track_counts(Dx, Dy, n_azim, delta, ntx, nty)
    return int(total), ntx[:n2], nty[:n2]


def trace(bb, n_azim, delta, bcs=(0, 0, 0, 0)):
    """Oracle restatement of TrackGenerator ctor + trace!.  bcs = (top, bottom, right, left)."""
    bb = np.ascontiguousarray(bb, np.float64)
    total, ntx, nty = track_counts(bb[2] - bb[0], bb[3] - bb[1], n_azim, delta)
    if total < 0:
        raise ValueError("DomainError")
    n2 = n_azim // 2
    f = lambda: np.zeros(total, np.float64)
    out = dict(n_total_tracks=total, n_tracks_x=ntx, n_tracks_y=nty, n_tracks=ntx + nty,
               phis=np.zeros(n2), delta_s=np.zeros(n2), omega=np.zeros(n2),
               azim_idx=np.zeros(total, np.int32), track_idx=np.zeros(total, np.int32),
               px=f(), py=f(), qx=f(), qy=f(), phi=f(), ell=f(), A=f(), B=f(), C=f(),
               bc_fwd=np.zeros(total, np.int8), bc_bwd=np.zeros(total, np.int8),
               dir_fwd=np.zeros(total, np.int8), dir_bwd=np.zeros(total, np.int8),
               next_fwd=np.zeros(total, np.int64), next_bwd=np.zeros(total, np.int64))
    rc = _lib().orc_trace(bb, n_azim, ntx, nty, np.asarray(bcs, np.int32), out["phis"], out["delta_s"],
                          out["omega"], out["azim_idx"], out["track_idx"], out["px"], out["py"], out["qx"],
                          out["qy"], out["phi"], out["ell"], out["A"], out["B"], out["C"], out["bc_fwd"],
                          out["bc_bwd"], out["dir_fwd"], out["dir_bwd"], out["next_fwd"], out["next_bwd"])
    if rc == -2:
        raise ValueError("DomainError: could not found track exit point.")
    if rc == -3:
        raise RuntimeError("Boundaries do not match!")
    return out


def num_threads() -> int:
    return int(_lib(True).orc_num_threads())
