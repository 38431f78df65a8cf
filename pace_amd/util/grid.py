"""GridData / DampingCoefficients containers (util/pace/util/grid/helper.py:21-45,306-530): read-only
metric terms as device Quantities, plus the packed pointer table the C ABI takes."""
import ctypes as C
import math

import numpy as np

from .. import _lib
from . import constants as c

_SCALARS = ("ptop", "da_min", "da_min_c")
_K_FIELDS = ("ak", "bk", "p", "p_ref", "dp_ref")
_VEC_J = ("edge_w", "edge_e")
_VEC_I = ("edge_s", "edge_n")
_HOST_ONLY = ("lon", "lat", "lon_agrid", "lat_agrid")


def _gcd(p1a, p1b, p2a, p2b):
    tb = math.sin((p1b - p2b) / 2.0) ** 2.0
    ta = math.sin((p1a - p2a) / 2.0) ** 2.0
    return math.asin(math.sqrt(tb + math.cos(p1b) * math.cos(p2b) * ta)) * 2.0


def _gcd_np(p1a, p1b, p2a, p2b):
    tb = np.sin((p1b - p2b) / 2.0) ** 2.0
    ta = np.sin((p1a - p2a) / 2.0) ** 2.0
    return np.arcsin(np.sqrt(tb + np.cos(p1b) * np.cos(p2b) * ta)) * 2.0


def a2b_corner_weights(lon, lat, lon_agrid, lat_agrid, n, n_halo=3):
    """x1 / (x2 - x1) of a2b_ord4.extrap_corner (a2b_ord4.py:43-56) for the 4 corner points x 3 diagonals,
    in the order the reference's corner stencils use them (a2b_ord4.py:59-273,570-583)."""
    is_, ie = n_halo, n + n_halo - 1
    diag = {0: ((0, 0), (1, 1)), 1: ((-1, 0), (-2, 1)), 2: ((0, -1), (1, -2)), 3: ((-1, -1), (-2, -2))}
    sets = ((0, 1, 2), (1, 3, 0), (3, 2, 1), (2, 3, 0))
    pts = ((is_, is_), (ie + 1, is_), (ie + 1, ie + 1), (is_, ie + 1))
    out = np.zeros((4, 3))
    for w, ((i, j), ds) in enumerate(zip(pts, sets)):
        for t, d in enumerate(ds):
            o1, o2 = diag[d]
            x1 = _gcd_np(lon_agrid[i + o1[0], j + o1[1]], lat_agrid[i + o1[0], j + o1[1]], lon[i, j], lat[i, j])
            x2 = _gcd_np(lon_agrid[i + o2[0], j + o2[1]], lat_agrid[i + o2[0], j + o2[1]], lon[i, j], lat[i, j])
            out[w, t] = x1 / (x2 - x1)
    return out


class GridData:
    """Attribute access to metric Quantities by the reference's names (grid_data.area, .dxa, ...)."""

    def __init__(self, quantity_factory, metrics: dict):
        self._qf = quantity_factory
        self._names = []
        n = quantity_factory.sizer.nx
        for name, val in metrics.items():
            if name in _SCALARS:
                setattr(self, name, float(val))
            elif name in _K_FIELDS:
                setattr(self, name, np.asarray(val, dtype=float))
            elif name in _HOST_ONLY:
                setattr(self, name, np.asarray(val, dtype=float))
            elif name in _VEC_J + _VEC_I:
                a = np.asarray(val, dtype=float)
                a = a[0, :] if a.ndim == 2 else a
                import torch

                setattr(self, name, torch.as_tensor(np.ascontiguousarray(a), dtype=quantity_factory.real, device=quantity_factory.device))
            else:
                a = np.asarray(val, dtype=float)
                if a.ndim != 2:
                    continue
                q = quantity_factory.zeros([c.X_DIM, c.Y_DIM], units="")
                q.set(a)
                q._grid_data = self  # lets operators that are handed one metric (updatedzc's `area`) find the table
                setattr(self, name, q)
                self._names.append(name)
        if all(hasattr(self, k) for k in _HOST_ONLY):
            self.a2b_corner_w = a2b_corner_weights(self.lon, self.lat, self.lon_agrid, self.lat_agrid, n)
        else:
            self.a2b_corner_w = np.zeros((4, 3))
        self._struct = None

    @classmethod
    def new_from_metric_terms(cls, metric_terms) -> "GridData":
        """grid/helper.py:306-350: the container of a rank's metric terms, from pace_amd.util.gridgen.MetricTerms."""
        keep = {k: v for k, v in metric_terms.terms.items() if k not in ("ee1", "ee2", "es1", "ew2")}
        gd = cls(metric_terms.quantity_factory, keep)
        gd._unit_vectors = {k: metric_terms.terms[k] for k in ("ee1", "ee2", "es1", "ew2")}
        gd._tile = metric_terms._tile
        return gd

    def c_struct(self) -> _lib.Metrics:
        if self._struct is None:
            m = _lib.Metrics()
            for name in _lib.METRIC_2D:
                setattr(m, name, getattr(self, name).ptr)
            for name in _lib.METRIC_1D:
                setattr(m, name, getattr(self, name).data_ptr())
            for w in range(4):
                for t in range(3):
                    m.a2b_corner_w[w][t] = float(self.a2b_corner_w[w, t])
            m.da_min = float(self.da_min)
            m.da_min_c = float(self.da_min_c)
            self._struct = m
        return self._struct


class DampingCoefficients:
    """helper.py:21-45: view of the same storage under the reference's second container name."""

    @classmethod
    def new_from_metric_terms(cls, metric_terms, grid_data: "GridData" = None) -> "DampingCoefficients":
        """helper.py:33-42.  Pass the GridData made from the same metric terms to share its device storage."""
        return cls(grid_data if grid_data is not None else GridData.new_from_metric_terms(metric_terms))

    def __init__(self, grid_data: GridData):
        self._grid_data = grid_data
        for n in ("del6_u", "del6_v", "divg_u", "divg_v", "da_min", "da_min_c"):
            setattr(self, n, getattr(grid_data, n))


def geom_struct(quantity_factory) -> _lib.Geom:
    s = quantity_factory.sizer
    g = _lib.Geom(s.nx, s.nz, quantity_factory.row_stride, 0, quantity_factory.level_stride)
    g._real = quantity_factory.real  # (python-side only: what check_layout expects of the fields)
    return g


from .gridgen import MetricTerms  # noqa: E402,F401  (pace.util.grid exports the three names together)
