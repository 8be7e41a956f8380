"""``TrackGenerator`` / ``trace!`` host side (mirror of ``src/trackgenerator.jl:34-348``).

``trace!`` stays on the host in this build (BASELINE north_star); it is the producer of the
per-track inputs the HIP ``segmentize!`` path consumes.  Tracks are held as SoA numpy
arrays in uid order (azimuthal index major, track index minor — the order in which the
reference assigns ``uid``, ``src/trackgenerator.jl:179-273``) so that they can be handed to
the C ABI without repacking; ``Track`` / ``Segment`` objects are thin views over them.

Python cannot spell ``trace!`` / ``segmentize!``; the functions are ``trace`` and
``segmentize`` (``segmentize`` lives in ``segmentize.py``).
"""
from __future__ import annotations

import math
from typing import List

import numpy as np

from .boundary import BoundaryConditions, BoundaryType, Periodic, Reflective, Vacuum
from .mesh import DiscreteModel, Mesh
from .quadrature import AzimuthalQuadrature

__all__ = ["TrackGenerator", "trace", "Track", "Segment", "Forward", "Backward",
           "bc_fwd", "bc_bwd", "dir_next_track_fwd", "dir_next_track_bwd"]

# DirectionType (src/track.jl:11-14)
Forward = 0
Backward = 1

_RTOL = math.sqrt(np.finfo(np.float64).eps)  # Base.rtoldefault(Float64)


class TrackGenerator:
    """``TrackGenerator(model, n_azim, δ; bcs, tiny_step=1e-8, volume_correction=false)``
    (``src/trackgenerator.jl:80-125``)."""

    def __init__(self, model: DiscreteModel, n_azim: int, delta: float, *,
                 bcs: BoundaryConditions | None = None, tiny_step: float = 1e-8,
                 volume_correction: bool = False):
        self.mesh = Mesh(model)
        self.bcs = bcs if bcs is not None else BoundaryConditions()
        self.azimuthal_quadrature = aq = AzimuthalQuadrature(n_azim, delta)
        self.tiny_step = float(tiny_step)
        self.volume_correction = bool(volume_correction)  # stored, never acted on (as in the reference)

        dx, dy = self.mesh.width(), self.mesh.height()
        n2, n4 = aq.n_azim_2, aq.n_azim_4
        ntx = np.zeros(n2, dtype=np.int64)
        nty = np.zeros(n2, dtype=np.int64)
        for i in range(1, n4 + 1):  # right_dir, src/trackgenerator.jl:96-108
            phi = math.pi / n2 * (i - 1 / 2)
            ntx[i - 1] = math.floor(dx / delta * abs(math.sin(phi))) + 1
            nty[i - 1] = math.floor(dy / delta * abs(math.cos(phi))) + 1
            j = aq.suplementary_idx(i)
            ntx[j - 1] = ntx[i - 1]
            nty[j - 1] = nty[i - 1]
        self.n_tracks_x = ntx
        self.n_tracks_y = nty
        self.n_tracks = ntx + nty
        self.n_total_tracks = int(self.n_tracks.sum())
        self.volumes = np.full(self.mesh.num_cells, np.nan)

        # filled by trace(): SoA per-track arrays in uid order
        self.traced = False
        self.uid_offsets = np.concatenate(([0], np.cumsum(self.n_tracks)))  # uid of tracks[i][1] - 1
        self.azim_idx = self.track_idx = None
        self.px = self.py = self.qx = self.qy = None
        self.phi = self.cos_phi = self.sin_phi = self.ell = None
        self.A = self.B = self.C = None
        self.bc_fwd = self.bc_bwd = None
        self.dir_next_fwd = self.dir_next_bwd = None
        self.next_fwd_uid = self.next_bwd_uid = None
        # filled by segmentize(): SoA segments + CSR offsets per track (uid order)
        self.segments = None

    # -- reference-style containers ------------------------------------------------------
    @property
    def tracks_by_uid(self) -> "_TrackList":
        return _TrackList(self, np.arange(self.n_total_tracks))

    @property
    def tracks(self) -> List["_TrackList"]:
        return [_TrackList(self, np.arange(self.uid_offsets[i], self.uid_offsets[i + 1]))
                for i in range(self.azimuthal_quadrature.n_azim_2)]

    def __repr__(self) -> str:  # show(io, ::TrackGenerator), src/trackgenerator.jl:52-63
        aq = self.azimuthal_quadrature
        return ("  Number of azimuthal angles in (0, π): %d\n  Azimuthal angles in (0, π): %s\n"
                "  Effective azimuthal spacings: %s\n  Total tracks: %d\n  Correct volumes: %s" % (
                    aq.n_azim_2, np.round(np.degrees(aq.phis), 2), np.round(aq.delta_s, 3),
                    self.n_total_tracks, str(self.volume_correction).lower()))


def _general_form(px, py, qx, qy):
    """``general_form`` (``src/intersection.jl:11-18``), vectorised, same operation order."""
    A = py - qy
    B = qx - px
    C = px * qy - qx * py
    nrm = np.sqrt(A * A + B * B + C * C)
    return A / nrm, B / nrm, C / nrm


def _point_in_segment(p1, p2, x):
    """``point_in_segment`` (``src/segment.jl:39-44``), vectorised over ``x``."""
    lpx = np.sqrt((p1[0] - x[0]) ** 2 + (p1[1] - x[1]) ** 2)
    lqx = np.sqrt((p2[0] - x[0]) ** 2 + (p2[1] - x[1]) ** 2)
    lpq = math.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2)
    s = lpx + lqx
    return (s == lpq) | (np.abs(s - lpq) <= _RTOL * np.maximum(np.abs(s), abs(lpq)))


def _boundary_condition(x, sides, bcs):
    """``boundary_condition`` (``src/boundary.jl:48-63``), vectorised; -1 where no side matches."""
    out = np.full(x[0].shape, -1, dtype=np.int64)
    for name in ("left", "right", "bottom", "top"):  # reversed: earlier names in the reference win
        hit = _point_in_segment(sides[name][0], sides[name][1], x)
        out[hit] = int(getattr(bcs, name))
    return out


def trace(t: TrackGenerator, backend: str = "native") -> TrackGenerator:
    """``trace!(t)`` (``src/trackgenerator.jl:134-280``) + ``next_tracks`` (``:282-348``).

    ``backend="native"`` runs ``rt_trace`` of the C-ABI library (host C++, works without a GPU);
    ``backend="numpy"`` is the vectorised Python restatement below.  Both give bit-identical
    arrays (``tests/test_native_host.py``)."""
    if backend == "native":
        return _trace_native(t)
    if backend != "numpy":
        raise ValueError("backend must be 'native' or 'numpy'")
    mesh, bcs, aq = t.mesh, t.bcs, t.azimuthal_quadrature
    n2, n4 = aq.n_azim_2, aq.n_azim_4
    ntx, nty, nt = t.n_tracks_x, t.n_tracks_y, t.n_tracks
    Dx, Dy = mesh.width(), mesh.height()
    dxs = np.zeros(n2)
    dys = np.zeros(n2)
    for i in range(1, n4 + 1):  # src/trackgenerator.jl:150-166
        phi = math.atan((Dy * ntx[i - 1]) / (Dx * nty[i - 1]))
        aq.phis[i - 1] = phi
        dxs[i - 1] = Dx / ntx[i - 1]
        dys[i - 1] = Dy / nty[i - 1]
        aq.delta_s[i - 1] = dxs[i - 1] * math.sin(phi)
        j = aq.suplementary_idx(i)
        aq.phis[j - 1] = math.pi - phi
        dxs[j - 1] = dxs[i - 1]
        dys[j - 1] = dys[i - 1]
        aq.delta_s[j - 1] = aq.delta_s[i - 1]
    aq.init_weights()

    # mesh sides (src/trackgenerator.jl:171-177)
    p1 = (mesh.bb_min[0], mesh.bb_min[1])
    p2 = (mesh.bb_min[0], mesh.bb_max[1])
    p3 = (mesh.bb_max[0], mesh.bb_max[1])
    p4 = (mesh.bb_max[0], mesh.bb_min[1])
    sides = dict(top=(p2, p3), bottom=(p4, p1), right=(p3, p4), left=(p1, p2))

    N = t.n_total_tracks
    azim = np.repeat(np.arange(1, n2 + 1), nt)  # 1-based azimuthal index per uid
    jj = np.arange(N) - t.uid_offsets[azim - 1] + 1  # 1-based track index within its angle
    nx = ntx[azim - 1]
    ny = nty[azim - 1]
    right = azim <= n4  # points_right
    phi = aq.phis[azim - 1]
    # libm per angle (not numpy's SIMD kernels): same cos/sin/tan the C host code gets
    tan_a = np.array([math.tan(v) for v in aq.phis])
    cos_a = np.array([math.cos(v) for v in aq.phis])
    sin_a = np.array([math.sin(v) for v in aq.phis])
    dx = dxs[azim - 1]
    dy = dys[azim - 1]
    in_x = jj <= nx  # origins_in_x

    # origins (src/trackgenerator.jl:188-200)
    px = np.where(in_x, np.where(right, dx * ((nx - jj) + 1 / 2), dx * (jj - 1 / 2)),
                  np.where(right, 0.0, Dx))
    py = np.where(in_x, 0.0, dy * ((jj - nx) - 1 / 2))
    # exits (src/trackgenerator.jl:203-221)
    m = tan_a[azim - 1]
    qx = px - (py - Dy) / m
    qy = np.full(N, Dy)
    bad = ~((0 <= qx) & (qx <= Dx))
    qx2 = np.where(right, Dx, 0.0)
    qy2 = np.where(right, py + m * (Dx - px), py - m * px)
    qx = np.where(bad, qx2, qx)
    qy = np.where(bad, qy2, qy)
    if np.any(bad & ~((0 <= qy) & (qy <= Dy))):
        raise ValueError("DomainError: could not found track exit point.")
    # recalibrate (src/trackgenerator.jl:224-228)
    px = px + mesh.bb_min[0]
    py = py + mesh.bb_min[1]
    qx = qx + mesh.bb_min[0]
    qy = qy + mesh.bb_min[1]
    ex = px - qx
    ey = py - qy
    ell = np.sqrt(ex * ex + ey * ey)
    A, B, C = _general_form(px, py, qx, qy)

    # boundary conditions, both ways (src/trackgenerator.jl:231-245)
    bcf = _boundary_condition((qx, qy), sides, bcs)
    bcb = _boundary_condition((px, py), sides, bcs)
    if np.any(bcf < 0) or np.any(bcb < 0):
        raise RuntimeError("Point do not lie in the boundary.")
    bcf1 = np.where(jj <= ny, np.where(right, int(bcs.right), int(bcs.left)), int(bcs.top))
    bcb1 = np.where(jj <= nx, int(bcs.bottom), np.where(right, int(bcs.left), int(bcs.right)))
    if np.any(bcf != bcf1) or np.any(bcb != bcb1):
        raise RuntimeError("Boundaries do not match!")

    # direction of the linked tracks (src/trackgenerator.jl:247-265)
    dirf = np.where(jj <= ny, Forward, np.where(bcf == int(Periodic), Forward, Backward))
    dirb = np.where(jj <= nx, np.where(bcb == int(Periodic), Backward, Forward), Backward)

    # next_tracks (src/trackgenerator.jl:294-348); uids are 1-based
    sup = n2 - azim + 1
    off_i = t.uid_offsets[azim - 1]
    off_k = t.uid_offsets[sup - 1]
    n_i = nt[azim - 1]
    per_f = bcf == int(Periodic)
    nf = np.where(jj <= ny,
                  np.where(per_f, off_i + jj + nx, off_k + jj + nx),
                  np.where(per_f, off_i + jj - ny, off_k + n_i + ny - jj + 1))
    per_b = bcb == int(Periodic)
    nb = np.where(jj <= nx,
                  np.where(per_b, off_i + jj + ny, off_k + nx - jj + 1),
                  np.where(per_b, off_i + jj - nx, off_k + jj - nx))

    t.azim_idx = azim.astype(np.int32)
    t.track_idx = jj.astype(np.int32)
    t.px, t.py, t.qx, t.qy = (np.ascontiguousarray(a, dtype=np.float64) for a in (px, py, qx, qy))
    t.phi = np.ascontiguousarray(phi, dtype=np.float64)
    t.cos_phi = np.ascontiguousarray(cos_a[azim - 1])
    t.sin_phi = np.ascontiguousarray(sin_a[azim - 1])
    t.ell = np.ascontiguousarray(ell)
    t.A, t.B, t.C = (np.ascontiguousarray(a) for a in (A, B, C))
    t.bc_fwd = bcf.astype(np.int8)
    t.bc_bwd = bcb.astype(np.int8)
    t.dir_next_fwd = dirf.astype(np.int8)
    t.dir_next_bwd = dirb.astype(np.int8)
    t.next_fwd_uid = nf.astype(np.int64)
    t.next_bwd_uid = nb.astype(np.int64)
    t.segments = None
    t.traced = True
    return t


def _trace_native(t: TrackGenerator) -> TrackGenerator:
    from . import _capi

    bcs, aq = t.bcs, t.azimuthal_quadrature
    o = _capi.native_trace(t.mesh.bb, aq.n_azim, t.n_tracks_x, t.n_tracks_y,
                           (int(bcs.top), int(bcs.bottom), int(bcs.right), int(bcs.left)))
    aq.phis[:] = o["phis"]
    aq.delta_s[:] = o["delta_s"]
    aq.omega_a[:] = o["omega"]
    t.azim_idx, t.track_idx = o["azim_idx"], o["track_idx"]
    t.px, t.py, t.qx, t.qy = o["px"], o["py"], o["qx"], o["qy"]
    t.phi, t.cos_phi, t.sin_phi, t.ell = o["phi"], o["cos_phi"], o["sin_phi"], o["ell"]
    t.A, t.B, t.C = o["A"], o["B"], o["C"]
    t.bc_fwd, t.bc_bwd = o["bc_fwd"], o["bc_bwd"]
    t.dir_next_fwd, t.dir_next_bwd = o["dir_fwd"], o["dir_bwd"]
    t.next_fwd_uid, t.next_bwd_uid = o["next_fwd"], o["next_bwd"]
    t.segments = None
    t.traced = True
    return t


# ---- views ---------------------------------------------------------------------------------

class Segment:
    """View of one segment record (``Segment{T}``, ``src/segment.jl:23-29``): ``p``, ``q``,
    ``ℓ`` (ASCII alias ``ell``), ``τ`` (``tau``; per-segment list owned by the segment store)
    and ``element`` (1-based ``Int32`` cell id)."""

    __slots__ = ("_s", "_i")

    def __init__(self, store, i: int):
        self._s, self._i = store, int(i)

    @property
    def p(self):
        return (float(self._s.px[self._i]), float(self._s.py[self._i]))

    @property
    def q(self):
        return (float(self._s.qx[self._i]), float(self._s.qy[self._i]))

    @property
    def ell(self) -> float:
        return float(self._s.ell[self._i])

    l = ell  # `segment.ℓ` in Python source normalises (NFKC) to `segment.l`

    @property
    def element(self) -> int:
        return int(self._s.element[self._i])

    @property
    def tau(self) -> list:
        return self._s.tau(self._i)

    τ = tau


class _SegmentList:
    def __init__(self, store, lo: int, hi: int):
        self._s, self._lo, self._hi = store, int(lo), int(hi)

    def __len__(self):
        return self._hi - self._lo

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [Segment(self._s, self._lo + k) for k in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return Segment(self._s, self._lo + i)

    def __iter__(self):
        return (Segment(self._s, k) for k in range(self._lo, self._hi))


class Track:
    """View of one track (``Track``, ``src/track.jl:42-77``)."""

    __slots__ = ("_t", "_u")

    def __init__(self, tg: TrackGenerator, u0: int):
        self._t, self._u = tg, int(u0)  # u0 = uid - 1

    def _need(self):
        if not self._t.traced:
            raise RuntimeError("UndefRefError: access to undefined reference (call `trace` first)")

    @property
    def uid(self) -> int:
        return self._u + 1

    @property
    def azim_idx(self) -> int:
        self._need()
        return int(self._t.azim_idx[self._u])

    @property
    def track_idx(self) -> int:
        self._need()
        return int(self._t.track_idx[self._u])

    @property
    def p(self):
        self._need()
        return (float(self._t.px[self._u]), float(self._t.py[self._u]))

    @property
    def q(self):
        self._need()
        return (float(self._t.qx[self._u]), float(self._t.qy[self._u]))

    @property
    def phi(self) -> float:
        self._need()
        return float(self._t.phi[self._u])

    φ = phi  # `track.ϕ` normalises to `track.φ`

    @property
    def ell(self) -> float:
        self._need()
        return float(self._t.ell[self._u])

    l = ell

    @property
    def ABC(self):
        self._need()
        return (float(self._t.A[self._u]), float(self._t.B[self._u]), float(self._t.C[self._u]))

    @property
    def segments(self) -> _SegmentList:
        s = self._t.segments
        if s is None:
            return _SegmentList(None, 0, 0)
        return _SegmentList(s, s.offsets[self._u], s.offsets[self._u + 1])

    @property
    def next_track_fwd(self) -> "Track":
        self._need()
        return Track(self._t, int(self._t.next_fwd_uid[self._u]) - 1)

    @property
    def next_track_bwd(self) -> "Track":
        self._need()
        return Track(self._t, int(self._t.next_bwd_uid[self._u]) - 1)

    def __repr__(self) -> str:  # show(io, ::Track), src/track.jl:87-100
        return ("  Azimuthal angle: %.2f\n  Entry point: %s\n  Exit point: %s\n  Length: %r\n"
                "  # of segments: %d\n  Boundary fwd: %s\n  Boundary bwd: %s" % (
                    math.degrees(self.phi), self.p, self.q, self.ell, len(self.segments),
                    bc_fwd(self).name, bc_bwd(self).name))


class _TrackList:
    def __init__(self, tg: TrackGenerator, uids0: np.ndarray):
        self._t, self._u = tg, uids0

    def __len__(self):
        return len(self._u)

    def __getitem__(self, i):
        """0-based Python indexing: ``tracks_by_uid[uid - 1]``."""
        if isinstance(i, slice):
            return [Track(self._t, u) for u in self._u[i]]
        return Track(self._t, self._u[i])

    def __iter__(self):
        return (Track(self._t, u) for u in self._u)


def bc_fwd(track: Track) -> BoundaryType:
    return BoundaryType(int(track._t.bc_fwd[track._u]))


def bc_bwd(track: Track) -> BoundaryType:
    return BoundaryType(int(track._t.bc_bwd[track._u]))


def dir_next_track_fwd(track: Track) -> int:
    return int(track._t.dir_next_fwd[track._u])


def dir_next_track_bwd(track: Track) -> int:
    return int(track._t.dir_next_bwd[track._u])
