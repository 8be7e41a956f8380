"""Azimuthal quadrature (mirror of ``src/azimuthal_quad.jl:8-63``).

Only ``δs`` and ``nazim2`` are read by the segmentize! path (``fill_volumes``,
``src/trackgenerator.jl:373-386``); the rest is what ``trace!`` fills.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = ["AzimuthalQuadrature"]


class AzimuthalQuadrature:
    def __init__(self, n_azim: int, delta: float):
        # argument validation: src/azimuthal_quad.jl:21-25
        if not n_azim > 0:
            raise ValueError("DomainError: number of azimuthal angles must be positive.")
        if n_azim % 4 != 0:
            raise ValueError("DomainError: number of azimuthal angles must be a multiple of 4.")
        if not delta > 0:
            raise ValueError("DomainError: azimuthal spacing must be positive.")
        self.n_azim = int(n_azim)
        self.n_azim_2 = self.n_azim // 2
        self.n_azim_4 = self.n_azim // 4
        self.delta = float(delta)
        self.delta_s = np.full(self.n_azim_2, np.nan)  # δs
        self.phis = np.full(self.n_azim_2, np.nan)  # ϕs
        self.omega_a = np.full(self.n_azim_2, np.nan)  # ωₐ

    # reference accessor names (src/azimuthal_quad.jl:15-17)
    def nazim(self) -> int:
        return self.n_azim

    def nazim2(self) -> int:
        return self.n_azim_2

    def nazim4(self) -> int:
        return self.n_azim_4

    def suplementary_idx(self, i: int) -> int:
        """1-based supplementary index ``N2 - i + 1`` (``src/azimuthal_quad.jl:63``)."""
        return self.n_azim_2 - i + 1

    def points_right(self, i: int) -> bool:
        return i <= self.n_azim_4

    def init_weights(self) -> None:
        """``init_weights!`` (``src/azimuthal_quad.jl:35-53``)."""
        n4 = self.n_azim_4
        ph = self.phis
        for i in range(1, n4 + 1):
            # branch order as in the reference: `isone(i)` is tested before `i == n_azim_4`
            if i == 1:
                w = ph[i] - ph[i - 1]
            elif i == n4:
                w = math.pi - ph[i - 1] - ph[i - 2]
            else:
                w = ph[i] - ph[i - 2]
            w /= 4 * math.pi
            self.omega_a[i - 1] = w
            self.omega_a[self.suplementary_idx(i) - 1] = w
