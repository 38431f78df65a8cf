"""ORACLE (test infrastructure, not product code) -- numpy helpers shared by the CPU
restatement of the FV3 acoustic step.

Conventions (reference: dsl/pace/dsl/stencil.py:542-759, util/pace/util/initialization/sizer.py:132-155)
* every 3-D field is an ndarray indexed [i, j, k] of shape (N+7, N+7, nz+1), origin (3, 3, 0);
  2-D metric fields are (N+7, N+7); K-fields are (nz+1,).
* one tile per rank (layout 1x1): the tile edges are at is_=3, ie=N+2 (and the same in j), so the
  reference's ``i_start/i_end/local_is/local_ie`` externals all collapse to these global indices.
* ``sh(a, di, dj)`` is the value of ``a[di, dj, 0]`` in gtscript notation evaluated on the whole
  storage; cells whose source lies outside the storage are NaN.  Restated stencils therefore
  compute on the whole array and *commit* only the window the reference stencil's
  origin/domain covers (``put``), which reproduces GT4Py's temporaries-with-extents semantics
  without hand-derived extents.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np


class Grid:
    """Index bounds + metric terms for one tile (all attributes are plain numpy arrays/scalars)."""

    def __init__(self, n, nk, metrics):
        self.n = n
        self.nk = nk  # number of layers held (levels in K-arrays may be nk+1)
        self.is_ = 3
        self.ie = n + 2
        self.js = 3
        self.je = n + 2
        self.ni = n + 7
        self.nj = n + 7
        for k, v in metrics.items():
            setattr(self, k, np.asarray(v) if not np.isscalar(v) else v)
        self.I = np.arange(self.ni).reshape(-1, 1, 1)
        self.J = np.arange(self.nj).reshape(1, -1, 1)

    def m2(self, name):
        """2-D metric as a (ni, nj, 1) array that broadcasts against 3-D fields."""
        return getattr(self, name)[:, :, None]

    # region masks in global indices, inclusive bounds, None = unbounded
    def reg(self, i0=None, i1=None, j0=None, j1=None):
        m = np.ones((self.ni, self.nj, 1), dtype=bool)
        if i0 is not None:
            m &= self.I >= i0
        if i1 is not None:
            m &= self.I <= i1
        if j0 is not None:
            m &= self.J >= j0
        if j1 is not None:
            m &= self.J <= j1
        return m


def sh(a, di=0, dj=0, dk=0):
    """a[di, dj, dk] on the whole storage (NaN outside).  Works for (ni,nj,nk) and (ni,nj,1)."""
    if di == 0 and dj == 0 and dk == 0:
        return a
    out = np.full(a.shape, np.nan)
    ni, nj, nk = a.shape
    i0, i1 = max(0, di), min(ni, ni + di)
    j0, j1 = max(0, dj), min(nj, nj + dj)
    if nk == 1:
        k0, k1, dk = 0, 1, 0
    else:
        k0, k1 = max(0, dk), min(nk, nk + dk)
    if i0 < i1 and j0 < j1 and k0 < k1:
        out[i0 - di : i1 - di, j0 - dj : j1 - dj, k0 - dk : k1 - dk] = a[i0:i1, j0:j1, k0:k1]
    return out


def put(dst, src, origin, domain, mask=None, k0=0, k1=None):
    """Commit src into dst on the stencil window (global origin (i,j), domain (ni,nj)), levels k0:k1."""
    i0, j0 = origin[0], origin[1]
    i1, j1 = i0 + domain[0], j0 + domain[1]
    if k1 is None:
        k1 = dst.shape[2]
    src = np.broadcast_to(src, dst.shape) if np.ndim(src) else src
    if mask is None:
        dst[i0:i1, j0:j1, k0:k1] = src[i0:i1, j0:j1, k0:k1] if np.ndim(src) else src
    else:
        m = np.broadcast_to(mask, dst.shape)[i0:i1, j0:j1, k0:k1]
        d = dst[i0:i1, j0:j1, k0:k1]
        if np.ndim(src):
            d[m] = src[i0:i1, j0:j1, k0:k1][m]
        else:
            d[m] = src


def kcol(a, nk_total):
    """K-field (len >= nk) as a (1,1,nk_total) broadcastable array (NaN padded)."""
    out = np.full((1, 1, nk_total), np.nan)
    a = np.asarray(a, dtype=float)
    n = min(nk_total, a.shape[0])
    out[0, 0, :n] = a[:n]
    return out
