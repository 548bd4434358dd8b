"""Quadrature providers with the reference's interface names (Python mirror used by tests/bench).

GaussLegendreQuadrature(n, a, b)   -- Quadratures/GaussLegendre.hpp:10-24 (GSL glfixed semantics: ascending nodes
                                      x_i = (a+b)/2 + (b-a)/2 t_i, weights (b-a)/2 w_i); Newton on P_n, no GSL.
SphericalDesign(N)                 -- Quadratures/SphericalDesign.cpp:6-50: N in {6,12,32,48,70,94,120,156,192},
                                      equal weights 4*pi/N, nodes from the shipped data tables (data/sph_design/,
                                      configurable directory instead of the reference's hard-coded absolute path).
"""
import os

import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "sph_design")
_DEGREE_OF = {6: 3, 12: 5, 32: 7, 48: 9, 70: 11, 94: 13, 120: 15, 156: 17, 192: 19}


class GaussLegendreQuadrature:
    def __init__(self, n_points, a, b):
        if n_points < 1:
            raise ValueError("n_points must be positive")
        ld = np.longdouble
        n = int(n_points)
        i = np.arange((n + 1) // 2, dtype=ld)
        x = np.cos(ld(np.pi) * (i + ld(0.75)) / (ld(n) + ld(0.5)))      # largest roots first
        for _ in range(100):
            p0, p1 = np.ones_like(x), x.copy()
            for k in range(2, n + 1):
                p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
            if n == 1:
                p0, p1 = np.ones_like(x), x.copy()
            dp = n * (x * p1 - p0) / (x * x - 1)
            dx = p1 / dp
            x = x - dx
            if np.max(np.abs(dx)) < 1e-18:
                break
        p0, p1 = np.ones_like(x), x.copy()
        for k in range(2, n + 1):
            p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
        dp = n * (x * p1 - p0) / (x * x - 1)
        w = 2 / ((1 - x * x) * dp * dp)
        half, mid = (ld(b) - ld(a)) / 2, (ld(a) + ld(b)) / 2
        nodes = np.empty(n, dtype=ld)
        weights = np.empty(n, dtype=ld)
        for j in range(len(x)):                                       # x[j] is the j-th largest root
            nodes[n - 1 - j] = mid + half * x[j]
            nodes[j] = mid - half * x[j]
            weights[n - 1 - j] = half * w[j]
            weights[j] = half * w[j]
        if n % 2 == 1:
            nodes[n // 2] = mid
        self._nodes = nodes.astype(np.float64)
        self._weights = weights.astype(np.float64)

    def getWeights(self):
        return self._weights

    def getNodes(self):
        return self._nodes

    def getNumberOfPoints(self):
        return len(self._weights)


class SphericalDesign:
    def __init__(self, N, data_dir=None):
        if N <= 0:
            raise ValueError("Number of points N must be a positive integer")
        if N not in _DEGREE_OF:
            raise ValueError("Invalid value of N")
        d = data_dir or os.environ.get("BFSM_DESIGN_DIR", _DATA_DIR)
        path = os.path.join(d, f"sym_design_t{_DEGREE_OF[N]:03d}_n{N:03d}.dat")
        header = 1
        if not os.path.exists(path):
            # the reference's own table of the same design (N rows of x y z, no header): lets BFSM_DESIGN_DIR point
            # at the directory a user of the reference already has
            alt = os.path.join(d, f"ss{_DEGREE_OF[N]:03d}.{N:03d}.txt")
            if not os.path.exists(alt):
                raise RuntimeError("Could not open file " + path + " (nor " + alt + ")")
            path, header = alt, 0
        rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]
        pts = np.array([[float(v) for v in r] for r in rows[header:]], dtype=np.float64)
        if pts.shape != (N, 3):
            raise RuntimeError(f"{path}: expected {N} points, found {pts.shape}")
        self._x, self._y, self._z = (np.ascontiguousarray(pts[:, k]) for k in range(3))
        self._w = np.full(N, (4 * np.pi) / N)

    def getWeights(self):
        return self._w

    def getx(self):
        return self._x

    def gety(self):
        return self._y

    def getz(self):
        return self._z

    def getNumberOfPoints(self):
        return len(self._w)
