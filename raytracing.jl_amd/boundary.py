"""Boundary-condition enums (mirror of ``src/boundary.jl:12-46``).

Boundary conditions only influence ``trace!`` (track linking); ``segmentize!`` never reads
them.  They are kept so that ``TrackGenerator(model, nφ, δ; bcs=...)`` reads the same.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass

__all__ = ["BoundaryType", "Vacuum", "Reflective", "Periodic", "BoundaryConditions"]


class BoundaryType(enum.IntEnum):
    Vacuum = 0
    Reflective = 1
    Periodic = 2


Vacuum = BoundaryType.Vacuum
Reflective = BoundaryType.Reflective
Periodic = BoundaryType.Periodic


@dataclass(frozen=True)
class BoundaryConditions:
    """``BoundaryConditions(; top=Vacuum, bottom=Vacuum, right=Vacuum, left=Vacuum)``
    (``src/boundary.jl:38-46``)."""

    top: BoundaryType = Vacuum
    bottom: BoundaryType = Vacuum
    right: BoundaryType = Vacuum
    left: BoundaryType = Vacuum

    def __post_init__(self):
        for name in ("top", "bottom", "right", "left"):
            object.__setattr__(self, name, BoundaryType(getattr(self, name)))
