"""MI355X-native ``segmentize!`` for RayTracing.jl-style cyclic ray tracing.

Host-side mirror of the reference's public surface (``src/RayTracing.jl:32-35``):
``BoundaryConditions, Vacuum, Reflective, Periodic, TrackGenerator, trace!, segmentize!``
(Python spells the last two ``trace`` and ``segmentize``).  ``segmentize`` runs the
hand-written HIP kernels in ``csrc/`` through the C ABI declared in
``include/rt_segmentize.h``; there is no CPU fallback — without the built library or
without a GPU it raises.
"""
from .boundary import BoundaryConditions, BoundaryType, Periodic, Reflective, Vacuum
from .mesh import DiscreteModel, DiscreteModelFromFile, GmshDiscreteModel, Mesh, data_path
from .quadrature import AzimuthalQuadrature
from .trackgenerator import (Backward, Forward, Segment, Track, TrackGenerator, bc_bwd, bc_fwd,
                             dir_next_track_bwd, dir_next_track_fwd, trace)
from .segmentize import RTOL_DEFAULT, SegmentStore, segmentize

__all__ = [
    "BoundaryConditions", "BoundaryType", "Vacuum", "Reflective", "Periodic",
    "DiscreteModel", "DiscreteModelFromFile", "GmshDiscreteModel", "Mesh", "data_path",
    "AzimuthalQuadrature", "TrackGenerator", "trace", "segmentize", "SegmentStore", "RTOL_DEFAULT",
    "Track", "Segment",
    "Forward", "Backward", "bc_fwd", "bc_bwd", "dir_next_track_fwd", "dir_next_track_bwd",
]
