"""ORACLE (test infrastructure) -- cubed-sphere halo exchange for a (1, 1) layout (one tile per rank),
restated in numpy for all six tiles at once.

Follows util/pace/util: partitioner.py:425-523 (edge neighbours + rotations), _boundary_utils.py:58-95 (send = interior
strip, recv = halo strip, compute extent along the edge), rotate.py:4-50, halo_data_transformer.py:387-461 (rotate by
-n_clockwise_rotations before sending, plain copy on receipt), halo_updater.py:385-536 (shared-edge synchronisation).
Tile corners are never exchanged (partitioner.py:525-590).  Pinned by the reference's own halo tests, which run
in-container, and by tools/crosscheck_oracle.py halo.
"""
import numpy as np

W, E, N, S = 0, 1, 2, 3


def neighbour(tile, edge):
    """(to_tile, n_clockwise_rotations)"""
    if tile % 2 == 0:
        return {W: ((tile - 2) % 6, 1), E: ((tile + 1) % 6, 0), N: ((tile + 2) % 6, 3), S: ((tile - 1) % 6, 0)}[edge]
    return {W: ((tile - 1) % 6, 0), E: ((tile + 2) % 6, 1), N: ((tile + 1) % 6, 0), S: ((tile - 2) % 6, 3)}[edge]


def facing_edge(tile, to_tile):
    for e in (W, E, N, S):
        if neighbour(tile, e)[0] == to_tile:
            return e
    raise ValueError("tiles are not neighbours")


def _slices(n, edge, n_pts, interior, xi, yi, n_halo=3):
    """Boundary slice of a field with x extent n+xi, y extent n+yi (xi/yi = 1 on interface dims)."""
    o = n_halo

    def along(ext):
        return slice(o, o + ext)

    def across(ext, overlap, at_start):
        if at_start:
            edge_i = o
            return slice(edge_i + overlap, edge_i + overlap + n_pts) if interior else slice(edge_i - n_pts, edge_i)
        edge_i = o + ext
        return slice(edge_i - overlap - n_pts, edge_i - overlap) if interior else slice(edge_i, edge_i + n_pts)

    ex, ey = n + xi, n + yi
    if edge == W:
        return across(ex, xi, True), along(ey)
    if edge == E:
        return across(ex, xi, False), along(ey)
    if edge == S:
        return along(ex), across(ey, yi, True)
    return along(ex), across(ey, yi, False)


def _rot(a, nrot):
    """rotate_scalar_data for arrays whose axes 0, 1 are x, y."""
    nrot %= 4
    if nrot == 1:
        return np.rot90(a, axes=(1, 0))
    if nrot == 3:
        return np.rot90(a, axes=(0, 1))
    if nrot == 2:
        return a[::-1, ::-1]
    return a


def halo_update(fields, n, n_pts=3, xi=0, yi=0, nk=None):
    """Scalar halo update of one field given on all 6 tiles (list of arrays), in place.
    nk: number of levels exchanged (the reference exchanges the field's own vertical extent only)."""
    kz = slice(0, nk)
    msgs = {}
    for t in range(6):
        for e in (W, E, N, S):
            to, r = neighbour(t, e)
            sx, sy = _slices(n, e, n_pts, True, xi, yi)
            msgs[(t, to)] = _rot(fields[t][sx, sy, kz].copy(), -r)
    for t in range(6):
        for e in (W, E, N, S):
            frm, _ = neighbour(t, e)
            rx, ry = _slices(n, e, n_pts, False, xi, yi)
            dst = fields[t][rx, ry, kz]
            dst[...] = msgs[(frm, t)].reshape(dst.shape)


def vector_halo_update(xs, ys, n, n_pts=3, grid="d", nk=None):
    """Vector halo update.  grid 'd': x on (X, Y_INTERFACE), y on (X_INTERFACE, Y)  [u, v];
    grid 'c': x on (X_INTERFACE, Y), y on (X, Y_INTERFACE)  [uc, vc]."""
    (xxi, xyi), (yxi, yyi) = ((0, 1), (1, 0)) if grid == "d" else ((1, 0), (0, 1))
    kz = (slice(0, nk),)
    msgs = {}
    for t in range(6):
        for e in (W, E, N, S):
            to, r = neighbour(t, e)
            sx = _slices(n, e, n_pts, True, xxi, xyi)
            sy = _slices(n, e, n_pts, True, yxi, yyi)
            xd = _rot(xs[t][sx + kz].copy(), -r)
            yd = _rot(ys[t][sy + kz].copy(), -r)
            k = (-r) % 4
            if k == 1:
                xd, yd = yd, -xd
            elif k == 2:
                xd, yd = -xd, -yd
            elif k == 3:
                xd, yd = -yd, xd
            msgs[(t, to)] = (xd, yd)
    for t in range(6):
        for e in (W, E, N, S):
            frm, _ = neighbour(t, e)
            rx = _slices(n, e, n_pts, False, xxi, xyi)
            ry = _slices(n, e, n_pts, False, yxi, yyi)
            xd, yd = msgs[(frm, t)]
            dx, dy = xs[t][rx + kz], ys[t][ry + kz]
            dx[...] = xd.reshape(dx.shape)
            dy[...] = yd.reshape(dy.shape)


def synchronize_vector_interfaces(xs, ys, n, n_halo=3, nk=None):
    """halo_updater.py:385-536 for D-grid style staggering (x on (X, Y_INTERFACE), y on (X_INTERFACE, Y)):
    south row of x and west column of y overwrite the shared edge on the neighbouring tile."""
    o = n_halo
    kz = slice(0, nk)
    msgs = {}
    for t in range(6):
        to, r = neighbour(t, S)
        d = xs[t][o : o + n, o, kz].copy()
        if (-r) % 4 == 1:
            d = d[::-1]
        if r in (3, 2):
            d = -d
        msgs[(t, to)] = d
        to, r = neighbour(t, W)
        d = ys[t][o, o : o + n, kz].copy()
        if (-r) % 4 == 3:
            d = d[::-1]
        if r in (1, 2):
            d = -d
        msgs[(t, to)] = d
    for t in range(6):
        frm, _ = neighbour(t, N)
        xs[t][o : o + n, o + n, kz] = msgs[(frm, t)]
        frm, _ = neighbour(t, E)
        ys[t][o + n, o : o + n, kz] = msgs[(frm, t)]
