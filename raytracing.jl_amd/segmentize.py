"""``segmentize!`` — host entry point of the HIP path (mirror of
``src/trackgenerator.jl:357-369``; Python spells it ``segmentize``).

Flattens the ``TrackGenerator`` into the C ABI's arrays, runs the device march and
``fill_volumes``, and refills ``track.segments`` (as an SoA ``SegmentStore`` with CSR offsets
per uid) and ``t.volumes`` in place.  Failures surface as ``RuntimeError`` with the
reference's messages (``src/trackgenerator.jl:360``, ``src/track.jl:141``, ``:172``).
The compute always goes through ``librt_segmentize.so``; there is no CPU path here.
"""
from __future__ import annotations

import math

import numpy as np

from . import _capi
from .trackgenerator import TrackGenerator

__all__ = ["segmentize", "SegmentStore", "RTOL_DEFAULT"]

RTOL_DEFAULT = math.sqrt(np.finfo(np.float64).eps)  # Base.rtoldefault(Float64)

NOT_TRACED_MSG = "Segmentation is intended after tracing. Please, call `trace!` first!"


class SegmentStore:
    """All segments of a ``TrackGenerator`` as SoA arrays; track ``u`` (0-based) owns
    ``[offsets[u], offsets[u+1])`` in march order.  ``tau(i)`` hands out the per-segment
    ``τ`` vector (``Segment.τ``, ``src/segment.jl:14,28``): every segment owns a distinct
    list, created on first use."""

    def __init__(self, offsets, px, py, qx, qy, ell, element):
        self.offsets = offsets
        self.px, self.py, self.qx, self.qy, self.ell, self.element = px, py, qx, qy, ell, element
        self._tau = {}

    def __len__(self):
        return len(self.ell)

    def tau(self, i: int) -> list:
        return self._tau.setdefault(int(i), [])


def segmentize(t: TrackGenerator, *, k: int = 5, rtol: float = RTOL_DEFAULT, device: int = 0,
               fetch=True, walk: bool = True, check: bool = True) -> TrackGenerator:
    """``segmentize!(t; k=5, rtol=√eps)``.  ``fetch=False`` leaves the results on the device
    (``t.device_tracks.device_pointers()``) for consumers that stay on the GPU; ``fetch="pinned"``
    returns views of page-locked buffers owned by ``t.device_tracks`` (PCIe rate, no page faults;
    valid until the next ``segmentize`` of ``t``) instead of fresh arrays; ``fetch=True`` (default) returns arrays over a host block
    that the library owns and faults in in the background while the tracks are uploaded and the kernels run
    (``rt_result_alloc``, round 6: what makes one call independent of the box's huge-page state; the arrays keep the block
    alive); ``fetch="fresh"``: numpy arrays allocated here, filled by ``rt_fetch_*`` (rounds 4-5).  ``walk=False``
    disables the certified walk step of the device march (every iteration then runs the
    literal locate + intersect step); results are identical either way.  ``check=False`` does
    not raise for failed tracks (the reference would have thrown at the first one) and leaves
    the per-track ``RT_TRACK_*`` codes in ``t.track_status`` instead."""
    if not t.traced:
        raise RuntimeError(NOT_TRACED_MSG)
    dm = getattr(t, "device_mesh", None)
    if dm is None or dm._h is None or dm.device != device:
        dm = _capi.DeviceMesh(t.mesh, device)
        t.device_mesh = dm
    dm.set_option("walk", 1 if walk else 0)
    old = getattr(t, "device_tracks", None)
    if old is not None:
        old.close()
    # (the destination first: its pages are faulted in beside the upload and the kernels)
    blk = dm.result_alloc(len(t.ell), float(np.sum(t.ell))) if fetch is True else None
    dt = _capi.DeviceTracks(dm, t.px, t.py, t.phi, t.cos_phi, t.sin_phi, t.A, t.B, t.C, t.ell, t.azim_idx)
    t.device_tracks = dt
    aq = t.azimuthal_quadrature
    dt.segmentize(t.tiny_step, int(k), float(rtol), aq.delta_s, aq.n_azim_2)
    n_failed, uid, st = dt.failed()
    if n_failed and check:
        raise RuntimeError(_capi.status_message(st, uid))
    if fetch:
        if fetch == "pinned":
            off, t.track_status, s = dt.fetch_pinned()
        elif blk is not None:
            off, t.track_status, s = dt.fetch_result(blk)
        else:
            off, t.track_status = dt.fetch_offsets()
            s = dt.fetch_segments()
        t.segments = SegmentStore(off, s["px"], s["py"], s["qx"], s["qy"], s["ell"], s["element"])
        t.volumes = dt.fetch_volumes()
    return t
