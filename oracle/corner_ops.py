"""ORACLE (test infrastructure) -- cubed-sphere corner halo fills, restated from the index
tables in the reference as closed-form index maps.

Follows stencils/pace/stencils/corners.py:
  copy_corners_x_stencil_defn :307-365, copy_corners_y_stencil_defn :367-425,
  fill_corners_bgrid_x_defn :591-650, fill_corners_bgrid_y_defn :653-712,
  fill_corners_dgrid_defn :987-1151.
In every map ``a`` counts cells away from the tile edge in i and ``b`` in j.  All sources lie in
the edge halos (never in a corner block), so the in-place form equals the reference's
read-old/write-new PARALLEL semantics.
"""
import numpy as np


def _corner_pairs_agrid(g, direction):
    """(dest_i, dest_j, src_i, src_j) for the 4x(3x3) A-grid corner cells."""
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    out = []
    for a in range(3):
        for b in range(3):
            if direction == "x":
                out.append((is_ - 1 - a, js - 1 - b, is_ - 1 - b, js + a))  # SW
                out.append((ie + 1 + a, js - 1 - b, ie + 1 + b, js + a))  # SE
                out.append((is_ - 1 - a, je + 1 + b, is_ - 1 - b, je - a))  # NW
                out.append((ie + 1 + a, je + 1 + b, ie + 1 + b, je - a))  # NE
            else:
                out.append((is_ - 1 - a, js - 1 - b, is_ + b, js - 1 - a))  # SW
                out.append((is_ - 1 - a, je + 1 + b, is_ + b, je + 1 + a))  # NW
                out.append((ie + 1 + a, js - 1 - b, ie - b, js - 1 - a))  # SE
                out.append((ie + 1 + a, je + 1 + b, ie - b, je + 1 + a))  # NE
    return out


def copy_corners(q, g, direction, ks=slice(None)):
    """corners.py:307-425 (CopyCorners, corners.py:17-59), in place on q[:, :, ks]."""
    pairs = _corner_pairs_agrid(g, direction)
    vals = [q[si, sj, ks].copy() for (_, _, si, sj) in pairs]
    for (di, dj, _, _), v in zip(pairs, vals):
        q[di, dj, ks] = v


def corner_source_agrid(g, direction):
    """Index maps (src_i, src_j) of shape (ni, nj): identity except in the corner blocks."""
    si, sj = np.meshgrid(np.arange(g.ni), np.arange(g.nj), indexing="ij")
    si, sj = si.copy(), sj.copy()
    for di, dj, i2, j2 in _corner_pairs_agrid(g, direction):
        si[di, dj], sj[di, dj] = i2, j2
    return si, sj


def _corner_pairs_bgrid(g, direction):
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    out = []
    for a in range(1, 4):
        for b in range(1, 4):
            if direction == "x":
                out.append((is_ - a, js - b, is_ - b, js + a))  # SW
                out.append((ie + 1 + a, js - b, ie + 1 + b, js + a))  # SE
                out.append((is_ - a, je + 1 + b, is_ - b, je + 1 - a))  # NW
                out.append((ie + 1 + a, je + 1 + b, ie + 1 + b, je + 1 - a))  # NE
            else:
                out.append((is_ - a, js - b, is_ + b, js - a))  # SW
                out.append((is_ - a, je + 1 + b, is_ + b, je + 1 + a))  # NW
                out.append((ie + 1 + a, js - b, ie + 1 - b, js - a))  # SE
                out.append((ie + 1 + a, je + 1 + b, ie + 1 - b, je + 1 + a))  # NE
    return out


def fill_corners_bgrid(q, g, direction, ks=slice(None)):
    """corners.py:591-712 (FillCornersBGrid :545-588), in place."""
    pairs = _corner_pairs_bgrid(g, direction)
    vals = [q[si, sj, ks].copy() for (_, _, si, sj) in pairs]
    for (di, dj, _, _), v in zip(pairs, vals):
        q[di, dj, ks] = v


def fill_corners_dgrid(x, y, g, mysign, ks=slice(None)):
    """corners.py:987-1151: vector (x on y-faces ... as the caller passes them), in place."""
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    xs, ys = [], []
    for a in range(1, 4):
        for b in range(1, 4):
            # SW
            xs.append((is_ - a, js - b, mysign, is_ - b, js + a - 1))
            ys.append((is_ - a, js - b, mysign, is_ + b - 1, js - a))
            # NE
            xs.append((ie + a, je + 1 + b, mysign, ie + 1 + b, je + 1 - a))
            ys.append((ie + 1 + a, je + b, mysign, ie + 1 - b, je + 1 + a))
            # NW
            xs.append((is_ - a, je + 1 + b, 1.0, is_ - b, je + 1 - a))
            ys.append((is_ - a, je + b, 1.0, is_ + b - 1, je + 1 + a))
            # SE
            xs.append((ie + a, js - b, 1.0, ie + 1 + b, js + a - 1))
            ys.append((ie + 1 + a, js - b, 1.0, ie + 1 - b, js - a))
    xv = [s * y[si, sj, ks] for (_, _, s, si, sj) in xs]
    yv = [s * x[si, sj, ks] for (_, _, s, si, sj) in ys]
    for (di, dj, _, _, _), v in zip(xs, xv):
        x[di, dj, ks] = v
    for (di, dj, _, _, _), v in zip(ys, yv):
        y[di, dj, ks] = v
