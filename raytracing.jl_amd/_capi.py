"""ctypes binding of the C ABI (``include/rt_segmentize.h``) — the same symbols a Julia
``ccall`` shim binds (``julia/RayTracingAMD.jl``).  No CPU fallback: if the shared library
is missing, or no GPU is visible, loading / compute raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("RT_SEGMENTIZE_LIB") or os.path.join(_CSRC, "librt_segmentize.so")

# every symbol include/rt_segmentize.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = (
    "rt_abi_version", "rt_last_error", "rt_status_message", "rt_device_count",
    "rt_mesh_create", "rt_mesh_destroy", "rt_mesh_info", "rt_last_stats", "rt_mesh_set_stream", "rt_mesh_get_stream", "rt_mesh_set_enqueue_hook",
    "rt_tracks_create", "rt_tracks_destroy", "rt_segmentize", "rt_failed_tracks", "rt_wait",
    "rt_fetch_offsets", "rt_fetch_segments", "rt_fetch_segments_pinned", "rt_fetch_pinned", "rt_fetch_volumes", "rt_device_pointers",
    "rt_record_order", "rt_device_table", "rt_fetch_table", "rt_fetch_records",
    "rt_result_alloc", "rt_result_fetch", "rt_result_free",
    "rt_last_timing", "rt_set_option", "rt_fill_tau", "rt_fetch_tau",
    "rt_sweep_set_links", "rt_sweep", "rt_sweep_fetch", "rt_sweep_info", "rt_sweep_rows_kind", "rt_sweep_xs_pointer", "rt_multi_link_rates",
    "rt_multi_create", "rt_multi_destroy", "rt_multi_set_option", "rt_multi_segmentize", "rt_multi_shards", "rt_multi_shard",
    "rt_multi_failed_tracks", "rt_multi_fetch_offsets", "rt_multi_fetch_segments", "rt_multi_fetch_volumes", "rt_multi_allgather",
    "rt_trace_counts", "rt_trace", "rt_msh_load", "rt_msh_sizes", "rt_msh_fetch", "rt_msh_free",
)

RT_TRACK_OK = 0
RT_TRACK_LOCATE_FAILED = 1
RT_TRACK_LENGTH_MISMATCH = 2
RT_TRACK_UNDEF_INTERSECTION = 3
RT_TRACK_ITER_CAP = 4


class RtError(RuntimeError):
    pass


def build(force: bool = False, extra: str = "") -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp", ".cpp", "Makefile"))]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "rt_segmentize.h"))
    fresh = os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    if force or not fresh:
        cmd = ["make", "-j4", "-C", _CSRC] + (["-B"] if force else []) + ([f"EXTRA={extra}"] if extra else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _share_hip_runtime_with_torch() -> None:
    """PyTorch's ROCm wheels bundle their own ``libamdhip64.so`` (same SONAME as ROCm's).  Two
    copies of the HIP runtime in one process cannot both see the GPU: whichever initialises second
    reports "No HIP GPUs are available".  If torch is installed, load its copy first — this
    library then binds to it by SONAME, and a later ``import torch`` finds it already loaded."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_lib = None
_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def lib():
    """Load ``librt_segmentize.so`` (raises ``RtError`` when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback for segmentize)")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.rt_abi_version.restype = C.c_int32
    L.rt_last_error.restype = C.c_char_p
    L.rt_status_message.restype = C.c_char_p
    L.rt_status_message.argtypes = [C.c_int32]
    L.rt_device_count.restype = C.c_int32
    L.rt_mesh_create.restype = _vp
    L.rt_mesh_create.argtypes = [C.c_int32, _dp, _dp, C.c_int32, _ip, C.c_int32, _ip, _ip, _dp]
    L.rt_mesh_destroy.argtypes = [_vp]
    try:
        L.rt_mesh_info.restype = C.c_int32
        L.rt_mesh_info.argtypes = [_vp, _dp, C.c_int32, C.c_char_p, C.c_int32]
        L.rt_last_stats.restype = C.c_int32
        L.rt_last_stats.argtypes = [_vp, _lp, C.c_int32]
    except AttributeError:
        if not os.environ.get("RT_SEGMENTIZE_LIB"):  # development A/B against an older build (tools/ab.sh) only
            raise
    L.rt_mesh_set_stream.restype = C.c_int32
    L.rt_mesh_set_stream.argtypes = [_vp, _vp]
    L.rt_mesh_get_stream.restype = _vp
    L.rt_mesh_get_stream.argtypes = [_vp]
    L.rt_mesh_set_enqueue_hook.restype = C.c_int32
    L.rt_mesh_set_enqueue_hook.argtypes = [_vp, _vp, _vp]
    L.rt_tracks_create.restype = _vp
    L.rt_tracks_create.argtypes = [_vp, C.c_int64] + [_dp] * 9 + [_ip]
    L.rt_tracks_destroy.argtypes = [_vp]
    L.rt_segmentize.restype = C.c_int64
    L.rt_segmentize.argtypes = [_vp, C.c_double, C.c_int32, C.c_double, _dp, C.c_int32]
    L.rt_wait.restype = C.c_int32
    L.rt_wait.argtypes = [_vp]
    L.rt_failed_tracks.restype = C.c_int32
    L.rt_failed_tracks.argtypes = [_vp, _lp, _lp, _ip]
    L.rt_fetch_offsets.restype = C.c_int32
    L.rt_fetch_offsets.argtypes = [_vp, _lp, _ip]
    L.rt_fetch_segments.restype = C.c_int32
    L.rt_fetch_segments.argtypes = [_vp, _dp, _dp, _dp, _dp, _dp, _ip]
    L.rt_fetch_segments_pinned.restype = C.c_int32
    L.rt_fetch_segments_pinned.argtypes = [_vp, C.POINTER(C.c_void_p)]
    L.rt_fetch_pinned.restype = C.c_int32
    L.rt_fetch_pinned.argtypes = [_vp, C.POINTER(C.c_void_p)]
    L.rt_fetch_volumes.restype = C.c_int32
    L.rt_fetch_volumes.argtypes = [_vp, _dp]
    L.rt_device_pointers.restype = C.c_int32
    L.rt_device_pointers.argtypes = [_vp, C.POINTER(_vp)]
    L.rt_record_order.restype = C.c_int32
    L.rt_record_order.argtypes = [_vp]
    L.rt_device_table.restype = C.c_int32
    L.rt_device_table.argtypes = [_vp, C.POINTER(_vp)]
    L.rt_fetch_table.restype = C.c_int32
    L.rt_fetch_table.argtypes = [_vp, _lp, _ip, _ip]
    L.rt_fetch_records.restype = C.c_int32
    L.rt_fetch_records.argtypes = [_vp, _dp, _dp, _dp, _dp, _dp, _ip]
    L.rt_last_timing.restype = C.c_int32
    L.rt_last_timing.argtypes = [_vp, _dp, C.c_int32]
    L.rt_set_option.restype = C.c_int32
    L.rt_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
    try:
        L.rt_fill_tau.restype = C.c_int32
        L.rt_fill_tau.argtypes = [_vp, _dp, C.c_int32, C.POINTER(_vp), _dp]
        L.rt_fetch_tau.restype = C.c_int32
        L.rt_fetch_tau.argtypes = [_vp, _dp]
        _bp8 = C.POINTER(C.c_int8)
        L.rt_sweep_set_links.restype = C.c_int32
        L.rt_sweep_set_links.argtypes = [_vp, _lp, _lp, _bp8, _bp8, _bp8, _bp8]
        L.rt_sweep.restype = C.c_int32
        L.rt_sweep.argtypes = [_vp, C.c_int32, _dp, _dp, _dp, _dp, C.c_int32, _dp]
        L.rt_sweep_fetch.restype = C.c_int32
        L.rt_sweep_fetch.argtypes = [_vp, _dp, _dp, _dp]
        L.rt_sweep_xs_pointer.restype = C.c_int32
        L.rt_sweep_xs_pointer.argtypes = [_vp, C.POINTER(_vp)]
        L.rt_result_alloc.restype = _vp
        L.rt_result_alloc.argtypes = [_vp, C.c_int64, C.c_double, C.c_int64]
        L.rt_result_fetch.restype = C.c_int32
        L.rt_result_fetch.argtypes = [_vp, _vp, C.POINTER(_vp), _lp]
        L.rt_result_free.argtypes = [_vp]
        L.rt_sweep_info.restype = C.c_int32
        L.rt_sweep_rows_kind.restype = C.c_int32
        L.rt_sweep_rows_kind.argtypes = [_vp]
        L.rt_sweep_info.argtypes = [_vp, C.POINTER(_vp), _ip]
        L.rt_multi_link_rates.restype = C.c_int32
        L.rt_multi_link_rates.argtypes = [_vp, _dp]
        L.rt_multi_create.restype = _vp
        L.rt_multi_create.argtypes = [_ip, C.c_int32, _dp, _dp, C.c_int32, _ip, C.c_int32, _ip, _ip, _dp, C.c_int64] + [_dp] * 9 + [_ip]
        L.rt_multi_destroy.argtypes = [_vp]
        L.rt_multi_set_option.restype = C.c_int32
        L.rt_multi_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
        L.rt_multi_segmentize.restype = C.c_int64
        L.rt_multi_segmentize.argtypes = [_vp, C.c_double, C.c_int32, C.c_double, _dp, C.c_int32]
        L.rt_multi_shards.restype = C.c_int32
        L.rt_multi_shards.argtypes = [_vp, _lp, _lp]
        L.rt_multi_shard.restype = _vp
        L.rt_multi_shard.argtypes = [_vp, C.c_int32]
        L.rt_multi_failed_tracks.restype = C.c_int32
        L.rt_multi_failed_tracks.argtypes = [_vp, _lp, _lp, _ip]
        L.rt_multi_fetch_offsets.restype = C.c_int32
        L.rt_multi_fetch_offsets.argtypes = [_vp, _lp, _ip]
        L.rt_multi_fetch_segments.restype = C.c_int32
        L.rt_multi_fetch_segments.argtypes = [_vp, _dp, _dp, _dp, _dp, _dp, _ip]
        L.rt_multi_fetch_volumes.restype = C.c_int32
        L.rt_multi_fetch_volumes.argtypes = [_vp, _dp]
        L.rt_multi_allgather.restype = C.c_int32
        L.rt_multi_allgather.argtypes = [_vp, C.POINTER(_vp), _dp]
    except AttributeError:
        if not os.environ.get("RT_SEGMENTIZE_LIB"):
            raise
    _bp = C.POINTER(C.c_int8)
    L.rt_trace_counts.restype = C.c_int64
    L.rt_trace_counts.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_double, _lp, _lp]
    L.rt_trace.restype = C.c_int32
    L.rt_trace.argtypes = [_dp, C.c_int32, _lp, _lp, _ip, _dp, _dp, _dp, _ip, _ip] + [_dp] * 11 + [_bp] * 4 + [_lp, _lp]
    L.rt_msh_load.restype = _vp
    L.rt_msh_load.argtypes = [C.c_char_p]
    L.rt_msh_sizes.restype = C.c_int32
    L.rt_msh_sizes.argtypes = [_vp, _ip, _ip, _ip]
    L.rt_msh_fetch.restype = C.c_int32
    L.rt_msh_fetch.argtypes = [_vp, _dp, _dp, _ip, _ip, _ip, _dp]
    L.rt_msh_free.argtypes = [_vp]
    if L.rt_abi_version() != 1:
        raise RtError("librt_segmentize.so: ABI version mismatch")
    _lib = L
    return L


def last_error() -> str:
    return lib().rt_last_error().decode("utf-8", "replace")


def _check(rc: int):
    if rc < 0:
        raise RtError(f"rt error {rc}: {last_error()}")
    return rc


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class DeviceMesh:
    """``rt_mesh`` handle: the flattened mesh resident in HBM."""

    def __init__(self, mesh, device: int = 0):
        L = lib()
        keep = []
        x, xp = _f64(mesh.x); y, yp = _f64(mesh.y)
        cn, cnp = _i32(np.asarray(mesh.cell_nodes).reshape(-1))
        ptr, ptrp = _i32(mesh.node_cells_ptrs); dat, datp = _i32(mesh.node_cells_data)
        bb, bbp = _f64(mesh.bb)
        keep += [x, y, cn, ptr, dat, bb]
        self.n_nodes, self.n_cells, self.device = len(x), len(cn) // 3, device
        self._h = L.rt_mesh_create(device, xp, yp, self.n_nodes, cnp, self.n_cells, ptrp, datp, bbp)
        if not self._h:
            raise RtError(f"rt_mesh_create failed: {last_error()}")

    INFO_NAMES = ("walk_enabled", "records", "records_walk", "eps_min", "eps_max", "d_vertex", "l_min", "cells_fragile",
                  "cells_degenerate", "edges_nonmanifold", "extras_max", "prep_ms", "kappa", "walk_available", "records_cheap",
                  "cheap_tiny_max")

    def info(self) -> dict:
        """``rt_mesh_info``: which regime the march of this mesh runs in (walk step on / off, how many
        (cell, entry edge) records carry valid certificates, the margins) + the preprocessing's remark."""
        v = (C.c_double * len(self.INFO_NAMES))()
        note = C.create_string_buffer(256)
        _check(lib().rt_mesh_info(self._h, v, len(self.INFO_NAMES), note, 256))
        d = {k: float(v[i]) for i, k in enumerate(self.INFO_NAMES)}
        for k in ("walk_enabled", "records", "records_walk", "cells_fragile", "cells_degenerate", "edges_nonmanifold",
                  "extras_max", "walk_available", "records_cheap"):
            d[k] = int(d[k])
        d["note"] = note.value.decode("utf-8", "replace")
        return d

    def set_stream(self, stream_ptr: int | None):
        _check(lib().rt_mesh_set_stream(self._h, stream_ptr))

    def get_stream(self) -> int:
        return lib().rt_mesh_get_stream(self._h) or 0

    def set_enqueue_hook(self, fn):
        """``fn()`` is called by every ``segmentize`` of this mesh's track sets once the call's kernels are
        enqueued and before it waits for them (``rt_mesh_set_enqueue_hook``); ``None`` removes it."""
        if fn is None:
            self._hook = None
            _check(lib().rt_mesh_set_enqueue_hook(self._h, None, None))
            return
        self._hook = C.CFUNCTYPE(None, C.c_void_p)(lambda _user: fn())  # keep the thunk alive
        _check(lib().rt_mesh_set_enqueue_hook(self._h, C.cast(self._hook, C.c_void_p), None))

    def result_alloc(self, n_tracks: int, sum_ell: float, n_records_hint: int = 0) -> "ResultBlock":
        """``rt_result_alloc``: a host block for the results of a track set of ``n_tracks`` tracks with Σℓ = ``sum_ell`` — returns at
        once, the library faults the block in in the background.  Call it before ``DeviceTracks`` / ``segmentize``."""
        h = lib().rt_result_alloc(self._h, int(n_tracks), float(sum_ell), int(n_records_hint))
        if not h:
            raise RtError(f"rt_result_alloc: {last_error()}")
        return ResultBlock(h)

    def set_option(self, name: str, value: int):
        _check(lib().rt_set_option(self._h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResultBlock:
    """``rt_result``: a host block owned by the library (2-MB aligned anonymous memory, huge pages asked for before its first touch,
    faulted in by the library's threads in the background) for everything a fetch returns — see ``include/rt_segmentize.h``."""

    def __init__(self, handle):
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTracks:
    """``rt_tracks`` handle: the per-track inputs resident in HBM + the results of segmentize."""

    def __init__(self, dmesh: DeviceMesh, px, py, phi, cos_phi, sin_phi, A, B, Cc, ell, azim_idx):
        L = lib()
        self.dmesh = dmesh  # keeps the mesh alive
        self.n = len(px)
        arrs = [_f64(a) for a in (px, py, phi, cos_phi, sin_phi, A, B, Cc, ell)]
        az, azp = _i32(azim_idx)
        for a, _ in arrs:
            if len(a) != self.n:
                raise ValueError("track arrays must have equal lengths")
        self._h = L.rt_tracks_create(dmesh._h, self.n, *[p for _, p in arrs], azp)
        if not self._h:
            raise RtError(f"rt_tracks_create failed: {last_error()}")
        self.total = None

    def segmentize(self, tiny_step: float, k: int, rtol: float, delta_s, n_azim_2: int) -> int:
        c = getattr(self, "_ds_cache", None)
        if c is None or c[0] is not delta_s:
            ds, dsp = _f64(delta_s)
            c = (delta_s, ds, dsp)
            # repeated calls with the same float64 array object reuse its pointer (no copy was made, so the
            # pointer sees the live data)
            self._ds_cache = c if ds is delta_s else None
        self.total = int(_check(lib().rt_segmentize(self._h, tiny_step, k, rtol, c[2], n_azim_2)))
        return self.total

    def failed(self):
        n, u, st = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _check(lib().rt_failed_tracks(self._h, C.byref(n), C.byref(u), C.byref(st)))
        return n.value, u.value, st.value

    def wait(self):
        """``rt_wait``: with the mesh option ``"async"`` a ``segmentize`` returns while its compaction is still on the stream;
        every accessor waits by itself, this is for consumers that use the device pointers on a stream of their own."""
        _check(lib().rt_wait(self._h))

    def fetch_offsets(self):
        off = np.zeros(self.n + 1, np.int64)
        st = np.zeros(max(self.n, 1), np.int32)
        _check(lib().rt_fetch_offsets(self._h, off.ctypes.data_as(_lp), st.ctypes.data_as(_ip)))
        return off, st[: self.n]

    def fetch_result(self, block):
        """``rt_result_fetch``: offsets, status and the six record arrays of the last ``segmentize`` in the host block ``block``
        (``DeviceMesh.result_alloc``, best called BEFORE this handle was created: the block is faulted in in the background).  Returns
        (offsets, status, records dict) as numpy arrays over the block's memory; they keep the block alive."""
        ptrs = (_vp * 8)()
        tot = C.c_int64(0)
        _check(lib().rt_result_fetch(self._h, block._h, ptrs, C.byref(tot)))
        n = int(tot.value)

        def view(addr, ctype, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            raw = (ctype * count).from_address(addr)
            raw._rt_block = block  # (the arrays' base: the block lives as long as any of them)
            return np.ctypeslib.as_array(raw)

        off = view(ptrs[0], C.c_int64, self.n + 1, np.int64)
        st = view(ptrs[1], C.c_int32, self.n, np.int32)
        seg = {k: view(ptrs[2 + i], C.c_double if i < 5 else C.c_int32, n, np.float64 if i < 5 else np.int32)
               for i, k in enumerate(("px", "py", "qx", "qy", "ell", "element"))}
        return off, st, seg

    def fetch_segments_pinned(self):
        """The records in page-locked host buffers owned by this handle (``rt_fetch_segments_pinned``): numpy
        views, valid until the next ``segmentize`` / ``fetch_segments_pinned`` / ``close`` of this handle —
        the copy runs at the PCIe rate instead of page-faulting into fresh arrays."""
        n = self.total
        ptrs = (C.c_void_p * 6)()
        _check(lib().rt_fetch_segments_pinned(self._h, ptrs))
        out = {}
        for i, k in enumerate(("px", "py", "qx", "qy", "ell", "element")):
            ctype = C.c_double if i < 5 else C.c_int32
            if n == 0:
                out[k] = np.zeros(0, np.float64 if i < 5 else np.int32)
                continue
            a = np.ctypeslib.as_array((ctype * n).from_address(ptrs[i]))
            a.flags.writeable = False
            out[k] = a
        self._pinned_views = out  # keep the handle alive as long as the views are reachable through it
        return out

    def fetch_pinned(self):
        """``rt_fetch_pinned``: (offsets, status, records dict) as read-only views of page-locked buffers owned by this handle —
        one call, one synchronisation (valid until the next ``segmentize`` / pinned fetch / ``close`` of this handle)."""
        n = self.total
        ptrs = (C.c_void_p * 8)()
        _check(lib().rt_fetch_pinned(self._h, ptrs))

        def view(addr, ctype, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            a = np.ctypeslib.as_array((ctype * count).from_address(addr))
            a.flags.writeable = False
            return a

        off = view(ptrs[0], C.c_int64, self.n + 1, np.int64)
        st = view(ptrs[1], C.c_int32, self.n, np.int32)
        out = {}
        for i, k in enumerate(("px", "py", "qx", "qy", "ell", "element")):
            out[k] = view(ptrs[2 + i], C.c_double if i < 5 else C.c_int32, n, np.float64 if i < 5 else np.int32)
        self._pinned_views = (off, st, out)
        return off, st, out

    def fetch_segments(self):
        n = self.total
        out = {k: np.empty(n, np.float64) for k in ("px", "py", "qx", "qy", "ell")}
        out["element"] = np.empty(n, np.int32)
        _check(lib().rt_fetch_segments(self._h, *[out[k].ctypes.data_as(_dp) for k in ("px", "py", "qx", "qy", "ell")],
                                       out["element"].ctypes.data_as(_ip)))
        return out

    def fetch_volumes(self):
        v = np.zeros(self.dmesh.n_cells, np.float64)
        _check(lib().rt_fetch_volumes(self._h, v.ctypes.data_as(_dp)))
        return v

    def device_pointers(self):
        """Raw device addresses: offsets, status, px, py, qx, qy, ell, element, volumes."""
        arr = (_vp * 9)()
        _check(lib().rt_device_pointers(self._h, arr))
        names = ("offsets", "status", "px", "py", "qx", "qy", "ell", "element", "volumes")
        return {k: (arr[i] or 0) for i, k in enumerate(names)}

    # ---- records in completion order (mesh option "record_order"): the per-track table
    def record_order(self) -> int:
        """``rt_record_order``: 1 if the handle's records lie in completion order (tracks in the order in which the march's
        workgroups ended; every track's records contiguous), 0 if in CSR order (uid order)."""
        rc = lib().rt_record_order(self._h)
        if rc < 0:
            _check(rc)
        return rc

    def device_table(self):
        """``rt_device_table``: device addresses of seg_begin, seg_count, status, px, py, qx, qy, ell, element, volumes — in
        whichever order the records lie; nothing is rewritten."""
        arr = (_vp * 10)()
        _check(lib().rt_device_table(self._h, arr))
        names = ("seg_begin", "seg_count", "status", "px", "py", "qx", "qy", "ell", "element", "volumes")
        return {k: (arr[i] or 0) for i, k in enumerate(names)}

    def fetch_table(self):
        """``rt_fetch_table``: (seg_begin, seg_count, status) — track u's records are [seg_begin[u], seg_begin[u] + seg_count[u])
        of the arrays ``fetch_records`` returns."""
        n = max(self.n, 1)
        beg, cnt, st = np.zeros(n, np.int64), np.zeros(n, np.int32), np.zeros(n, np.int32)
        _check(lib().rt_fetch_table(self._h, beg.ctypes.data_as(_lp), cnt.ctypes.data_as(_ip), st.ctypes.data_as(_ip)))
        return beg[: self.n], cnt[: self.n], st[: self.n]

    def fetch_records(self):
        """``rt_fetch_records``: the six record arrays as they lie on the device (see ``fetch_table``)."""
        n = self.total
        out = {k: np.empty(n, np.float64) for k in ("px", "py", "qx", "qy", "ell")}
        out["element"] = np.empty(n, np.int32)
        _check(lib().rt_fetch_records(self._h, *[out[k].ctypes.data_as(_dp) for k in ("px", "py", "qx", "qy", "ell")],
                                      out["element"].ctypes.data_as(_ip)))
        return out

    def fill_tau(self, sigma_t, fetch=True):
        """``rt_fill_tau``: τ[s, g] = Σt[element[s], g]·ℓ[s] on the device (``Segment.τ``, src/segment.jl:14,28).  ``sigma_t``:
        [n_cells, n_groups].  Returns (τ as a [total, n_groups] host array or None, device pointer, kernel ms)."""
        sg = np.ascontiguousarray(sigma_t, np.float64)
        if sg.ndim == 1:
            sg = sg.reshape(-1, 1)
        if sg.shape[0] != self.dmesh.n_cells:
            raise ValueError("sigma_t must have one row per cell")
        ptr, ms = _vp(), C.c_double(0.0)
        _check(lib().rt_fill_tau(self._h, sg.ctypes.data_as(_dp), sg.shape[1], C.byref(ptr), C.byref(ms)))
        tau = None
        if fetch:
            tau = np.empty((self.total, sg.shape[1]), np.float64)
            _check(lib().rt_fetch_tau(self._h, tau.ctypes.data_as(_dp)))
        return tau, (ptr.value or 0), ms.value

    def sweep_set_links(self, tg):
        """``rt_sweep_set_links`` with the link arrays ``trace`` left in the TrackGenerator (``tg`` must hold this handle's
        tracks, in uid order) — or pass a dict with next_fwd, next_bwd, dir_fwd, dir_bwd, bc_fwd, bc_bwd."""
        if isinstance(tg, dict):
            d = tg
        else:
            d = dict(next_fwd=tg.next_fwd_uid, next_bwd=tg.next_bwd_uid, dir_fwd=tg.dir_next_fwd, dir_bwd=tg.dir_next_bwd,
                     bc_fwd=tg.bc_fwd, bc_bwd=tg.bc_bwd)
        nf = np.ascontiguousarray(d["next_fwd"], np.int64); nb = np.ascontiguousarray(d["next_bwd"], np.int64)
        b8 = [np.ascontiguousarray(d[k], np.int8) for k in ("dir_fwd", "dir_bwd", "bc_fwd", "bc_bwd")]
        if any(len(a) != self.n for a in [nf, nb] + b8):
            raise ValueError("link arrays must have one entry per track")
        p8 = C.POINTER(C.c_int8)
        _check(lib().rt_sweep_set_links(self._h, nf.ctypes.data_as(_lp), nb.ctypes.data_as(_lp), *[a.ctypes.data_as(p8) for a in b8]))

    SWEEP_INPUT = {"auto": 0, "compact": 1, "staged": 2}

    def sweep(self, n_groups, sigma_t=None, source=None, track_weight=None, psi_in=None, input="auto", fetch=True):
        """``rt_sweep``: one transport sweep over the cyclic tracks on the device.  ``sigma_t`` / ``source``: [n_cells, G];
        ``track_weight``: [n_tracks]; ``psi_in``: [2, n_tracks, G] (None: what the previous sweep handed on).  Returns a dict
        with ``ms``, ``input`` ("compact" / "staged"), ``groups_per_pass``, ``passes`` and, with ``fetch``, ``phi`` [n_cells, G],
        ``psi_out`` and ``psi_next`` [2, n_tracks, G]."""
        G = int(n_groups)

        def arr(a, shape):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, np.float64)
            if a.size != int(np.prod(shape)):
                raise ValueError("expected an array of shape %r" % (shape,))
            return a, a.ctypes.data_as(_dp)

        st, stp = arr(sigma_t, (self.dmesh.n_cells, G)); q, qp = arr(source, (self.dmesh.n_cells, G))
        w, wp = arr(track_weight, (self.n,)); pi, pip = arr(psi_in, (2, self.n, G))
        ms = C.c_double(0.0)
        _check(lib().rt_sweep(self._h, G, stp, qp, wp, pip, self.SWEEP_INPUT[input], C.byref(ms)))
        info = (C.c_int32 * 4)()
        _check(lib().rt_sweep_info(self._h, None, info))
        out = dict(ms=ms.value, input={1: "compact", 2: "staged"}[int(info[0])], groups_per_pass=int(info[1]), passes=int(info[2]),
                   rows={0: None, 1: "staging", 2: "from compact"}.get(int(lib().rt_sweep_rows_kind(self._h))))
        if fetch:
            out["phi"] = np.empty((self.dmesh.n_cells, G)); out["psi_out"] = np.empty((2, self.n, G)); out["psi_next"] = np.empty((2, self.n, G))
            _check(lib().rt_sweep_fetch(self._h, *[out[k].ctypes.data_as(_dp) for k in ("phi", "psi_out", "psi_next")]))
        return out

    def sweep_pointers(self) -> dict:
        """``rt_sweep_info``: device addresses of the last sweep's ``phi`` [n_cells, G], ``psi_out`` and ``psi_in`` [2, n_tracks, G]
        (``psi_in`` holds what the sweep handed on: the boundary flux of the next one) + ``groups``."""
        ptrs = (_vp * 3)()
        info = (C.c_int32 * 4)()
        _check(lib().rt_sweep_info(self._h, ptrs, info))
        return dict(phi=ptrs[0] or 0, psi_out=ptrs[1] or 0, psi_in=ptrs[2] or 0, groups=int(info[3]))

    def sweep_xs_pointer(self) -> int:
        """``rt_sweep_xs_pointer``: device address of the cross sections as the sweep reads them, [n_cells * G][2] =
        {Σt, q/Σt} — a solver updates the sources there and sweeps again with ``sigma_t = source = None``."""
        p = _vp()
        _check(lib().rt_sweep_xs_pointer(self._h, C.byref(p)))
        return p.value or 0

    def stats(self) -> dict:
        """``rt_last_stats``: records of the last call and how many of them the literal step produced."""
        v = (C.c_int64 * 28)()
        _check(lib().rt_last_stats(self._h, v, 28))
        return dict(records=int(v[0]), generic_records=int(v[1]), walk_records=int(v[0]) - int(v[1]),
                    chunks_used=int(v[2]), chunks_allocated=int(v[3]), march_waves=int(v[4]), split=int(v[5]), wide_k=int(v[6]), device_bytes=int(v[7]),
                    cheap_records=int(v[8]), cheap_refusals={k: int(v[9 + i]) for i, k in enumerate(self.REFUSAL_TERMS)},
                    tracks_near_rtol=int(v[18]), tracks_restarted=int(v[19]), records_tallied_from_lengths=int(v[20]),
                    lean=int(v[21]), lean_queued=int(v[22]), completion_order=int(v[24]), side_entries_used=int(v[25]), side_entries_allocated=int(v[26]), attempts=int(v[27]),
                    record_kernel={0: None, 1: "rt::k_compact3", 2: "rt::k_materialise<true, false>", 3: "rt::k_materialise_lin<true>" if int(v[24]) else "rt::k_materialise_lin<false>",
                                   4: "rt::k_materialise<false, true>"}.get(int(v[23])))

    # the nine terms of the cheap step's certificate (rt_device.hpp, topo_certified), in rt_last_stats' order
    REFUSAL_TERMS = ("no_record", "scan_window", "vertex_clearance", "entry_not_crossed", "isolation_margin", "entry_rounding",
                     "exit_rounding", "tiny_step_bound", "chord_order")

    def timing(self):
        """``rt_last_timing`` — all zeros unless the mesh option ``"timing"`` is 1 (HIP events cost stream time)."""
        ms = getattr(self, "_ms_buf", None)
        if ms is None:
            ms = self._ms_buf = (C.c_double * 8)()
        _check(lib().rt_last_timing(self._h, ms, 8))
        # compact: staging -> CSR compaction (single-pass mode) or the second, writing march (two-pass mode)
        return dict(total=float(ms[0]), plan=float(ms[1]), march=float(ms[2]), scan=float(ms[3]), compact=float(ms[4]),
                    volumes=float(ms[5]))

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_tracks_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevice:
    """``rt_multi`` handle: ``tracks_by_uid`` sharded over several devices behind one call (``device_ids`` may
    name a device more than once)."""

    def __init__(self, mesh, device_ids, px, py, phi, cos_phi, sin_phi, A, B, Cc, ell, azim_idx):
        L = lib()
        ids, idp = _i32(device_ids)
        x, xp = _f64(mesh.x); y, yp = _f64(mesh.y)
        cn, cnp = _i32(np.asarray(mesh.cell_nodes).reshape(-1))
        ptr, ptrp = _i32(mesh.node_cells_ptrs); dat, datp = _i32(mesh.node_cells_data)
        bb, bbp = _f64(mesh.bb)
        arrs = [_f64(a) for a in (px, py, phi, cos_phi, sin_phi, A, B, Cc, ell)]
        az, azp = _i32(azim_idx)
        self.n_devices, self.n_tracks, self.n_cells = len(ids), len(arrs[0][0]), len(cn) // 3
        self._h = L.rt_multi_create(idp, len(ids), xp, yp, len(x), cnp, self.n_cells, ptrp, datp, bbp, self.n_tracks,
                                    *[p for _, p in arrs], azp)
        if not self._h:
            raise RtError(f"rt_multi_create failed: {last_error()}")
        self.total = None

    def set_option(self, name: str, value: int):
        _check(lib().rt_multi_set_option(self._h, name.encode(), int(value)))

    def segmentize(self, tiny_step: float, k: int, rtol: float, delta_s, n_azim_2: int) -> int:
        ds, dsp = _f64(delta_s)
        self.total = int(_check(lib().rt_multi_segmentize(self._h, tiny_step, k, rtol, dsp, n_azim_2)))
        return self.total

    def shards(self):
        ub, sb = np.zeros(self.n_devices + 1, np.int64), np.zeros(self.n_devices + 1, np.int64)
        _check(lib().rt_multi_shards(self._h, ub.ctypes.data_as(_lp), sb.ctypes.data_as(_lp) if self.total is not None else None))
        return ub, sb

    def failed(self):
        n, u, st = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _check(lib().rt_multi_failed_tracks(self._h, C.byref(n), C.byref(u), C.byref(st)))
        return n.value, u.value, st.value

    def fetch_offsets(self):
        off = np.zeros(self.n_tracks + 1, np.int64)
        st = np.zeros(max(self.n_tracks, 1), np.int32)
        _check(lib().rt_multi_fetch_offsets(self._h, off.ctypes.data_as(_lp), st.ctypes.data_as(_ip)))
        return off, st[: self.n_tracks]

    def fetch_segments(self):
        n = self.total
        out = {k: np.empty(n, np.float64) for k in ("px", "py", "qx", "qy", "ell")}
        out["element"] = np.empty(n, np.int32)
        _check(lib().rt_multi_fetch_segments(self._h, *[out[k].ctypes.data_as(_dp) for k in ("px", "py", "qx", "qy", "ell")],
                                             out["element"].ctypes.data_as(_ip)))
        return out

    def fetch_volumes(self):
        v = np.zeros(self.n_cells, np.float64)
        _check(lib().rt_multi_fetch_volumes(self._h, v.ctypes.data_as(_dp)))
        return v

    def link_rates(self):
        """``rt_multi_link_rates``: GB/s of the last all-gather per (destination, source) pair."""
        v = np.zeros((self.n_devices, self.n_devices))
        _check(lib().rt_multi_link_rates(self._h, v.ctypes.data_as(_dp)))
        return v

    def allgather(self):
        """Reassemble the global arrays on every shard's device (peer copies).  Returns (ms, ptrs[n_devices][6])."""
        ptrs = (_vp * (6 * self.n_devices))()
        ms = C.c_double(0.0)
        _check(lib().rt_multi_allgather(self._h, ptrs, C.byref(ms)))
        return ms.value, [[ptrs[6 * i + a] or 0 for a in range(6)] for i in range(self.n_devices)]

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def status_message(status: int, uid: int = 0) -> str:
    msg = lib().rt_status_message(status).decode("utf-8")
    return msg.replace("%d", str(uid)) if "%d" in msg else msg


def device_count() -> int:
    return int(lib().rt_device_count())


# ---- native host rows (CPU only; they work without a GPU) ---------------------------------

def native_trace_counts(width: float, height: float, n_azim: int, delta: float):
    n2 = max(n_azim // 2, 1)
    ntx = np.zeros(n2, np.int64)
    nty = np.zeros(n2, np.int64)
    total = lib().rt_trace_counts(width, height, n_azim, delta, ntx.ctypes.data_as(_lp), nty.ctypes.data_as(_lp))
    if total < 0:
        raise ValueError(last_error())
    return int(total), ntx[: n_azim // 2], nty[: n_azim // 2]


def native_trace(bb, n_azim: int, ntx, nty, bcs):
    """``rt_trace``: dict of per-angle and per-track arrays (uid order)."""
    n2 = n_azim // 2
    ntx = np.ascontiguousarray(ntx, np.int64)
    nty = np.ascontiguousarray(nty, np.int64)
    total = int((ntx + nty).sum())
    f = lambda n=total: np.zeros(n, np.float64)
    out = dict(phis=f(n2), delta_s=f(n2), omega=f(n2), azim_idx=np.zeros(total, np.int32),
               track_idx=np.zeros(total, np.int32), px=f(), py=f(), qx=f(), qy=f(), phi=f(), cos_phi=f(), sin_phi=f(),
               ell=f(), A=f(), B=f(), C=f(), bc_fwd=np.zeros(total, np.int8), bc_bwd=np.zeros(total, np.int8),
               dir_fwd=np.zeros(total, np.int8), dir_bwd=np.zeros(total, np.int8),
               next_fwd=np.zeros(total, np.int64), next_bwd=np.zeros(total, np.int64))
    bb_a, bbp = _f64(bb)
    bc_a, bcp = _i32(bcs)
    d = lambda k: out[k].ctypes.data_as(_dp)
    b = lambda k: out[k].ctypes.data_as(C.POINTER(C.c_int8))
    rc = lib().rt_trace(bbp, n_azim, ntx.ctypes.data_as(_lp), nty.ctypes.data_as(_lp), bcp, d("phis"), d("delta_s"),
                        d("omega"), out["azim_idx"].ctypes.data_as(_ip), out["track_idx"].ctypes.data_as(_ip),
                        d("px"), d("py"), d("qx"), d("qy"), d("phi"), d("cos_phi"), d("sin_phi"), d("ell"), d("A"),
                        d("B"), d("C"), b("bc_fwd"), b("bc_bwd"), b("dir_fwd"), b("dir_bwd"),
                        out["next_fwd"].ctypes.data_as(_lp), out["next_bwd"].ctypes.data_as(_lp))
    if rc != 0:
        msg = last_error()
        raise (ValueError if "DomainError" in msg else RuntimeError)(msg)
    return out


def native_load_msh(path: str):
    """``rt_msh_load``: (x, y, cell_nodes[n,3], nc_ptrs, nc_data, bb) straight from a gmsh 4.1 file."""
    L = lib()
    h = L.rt_msh_load(os.fsencode(path))
    if not h:
        raise RtError(last_error())
    try:
        nn, nc, nnz = C.c_int32(), C.c_int32(), C.c_int32()
        _check(L.rt_msh_sizes(h, C.byref(nn), C.byref(nc), C.byref(nnz)))
        x, y = np.zeros(nn.value), np.zeros(nn.value)
        cells = np.zeros(3 * nc.value, np.int32)
        ptrs, data = np.zeros(nn.value + 1, np.int32), np.zeros(nnz.value, np.int32)
        bb = np.zeros(4)
        _check(L.rt_msh_fetch(h, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), cells.ctypes.data_as(_ip),
                              ptrs.ctypes.data_as(_ip), data.ctypes.data_as(_ip), bb.ctypes.data_as(_dp)))
    finally:
        L.rt_msh_free(h)
    return x, y, cells.reshape(-1, 3), ptrs, data, bb
