"""Corner positions of the six tiles of the equidistant gnomonic cubed sphere, with halos, for a (1, 1) layout.

Reference behaviour reproduced (generation.py:1476-1619 `_init_dgrid`, gnomonic.py:26-170 `local_gnomonic_ed`,
mirror.py:9-214 `mirror_grid`): face 0 is the cube face x = -1/sqrt(3) seen from lon 3/4 pi ... 5/4 pi; along a face edge the
corner points are EQUIDISTANT IN ANGLE (lat_j = -alpha + j * 2 alpha / N, alpha = asin(1/sqrt 3)); the interior is the tensor
product of the two edge distributions on the cube face, projected to the sphere; longitudes are then turned by -pi, the other
five faces are rotations of face 0 (in the reference's left-handed (lon, lat) -> (x, y, -z) convention), all longitudes are
shifted by -pi/18 ("away from Japan"), wrapped to [0, 2 pi) and values below 1e-10 set to zero.  Halo corners come from the
neighbouring tiles (scalar exchange of the corner-point arrays) and the 3 x 3 corner halos from `fill_corners_2d` (B grid,
x direction).

Everything here is vectorised numpy on (N + 7, N + 7) arrays, corner point (i, j) = storage index (i, j), halo 3.
"""
import numpy as np

PI = np.pi
W, E, N_, S = 0, 1, 2, 3


def _neighbour(tile, edge):
    """(neighbour tile, clockwise rotations of its axes relative to ours) -- partitioner.py:425-523."""
    if tile % 2 == 0:
        return {W: ((tile - 2) % 6, 1), E: ((tile + 1) % 6, 0), N_: ((tile + 2) % 6, 3), S: ((tile - 1) % 6, 0)}[edge]
    return {W: ((tile - 1) % 6, 0), E: ((tile + 2) % 6, 1), N_: ((tile + 1) % 6, 0), S: ((tile - 2) % 6, 3)}[edge]


def _strip(n, edge, n_pts, interior, xi, yi, o=3):
    """Slices of the send (interior) / receive (halo) strip of a field with n + xi by n + yi compute points
    (_boundary_utils.py:58-95: on an interface dimension the shared edge point itself is not sent)."""
    def across(ext, overlap, at_start):
        if at_start:
            return slice(o + overlap, o + overlap + n_pts) if interior else slice(o - n_pts, o)
        e = o + ext
        return slice(e - overlap - n_pts, e - overlap) if interior else slice(e, e + n_pts)

    ex, ey = n + xi, n + yi
    if edge == W:
        return across(ex, xi, True), slice(o, o + ey)
    if edge == E:
        return across(ex, xi, False), slice(o, o + ey)
    if edge == S:
        return slice(o, o + ex), across(ey, yi, True)
    return slice(o, o + ex), across(ey, yi, False)


def _turn(a, nrot):
    nrot %= 4
    if nrot == 1:
        return np.rot90(a, axes=(1, 0))
    if nrot == 3:
        return np.rot90(a, axes=(0, 1))
    if nrot == 2:
        return a[::-1, ::-1]
    return a


def exchange_scalar(fields, n, xi=0, yi=0, n_pts=3):
    """Halo update of a scalar given on the six tiles (list of 2-D / 3-D arrays whose first two axes are x, y), in place."""
    msgs = {}
    for t in range(6):
        for e in (W, E, N_, S):
            to, r = _neighbour(t, e)
            sx, sy = _strip(n, e, n_pts, True, xi, yi)
            msgs[(t, to)] = _turn(fields[t][sx, sy].copy(), -r)
    for t in range(6):
        for e in (W, E, N_, S):
            frm, _ = _neighbour(t, e)
            rx, ry = _strip(n, e, n_pts, False, xi, yi)
            dst = fields[t][rx, ry]
            dst[...] = msgs[(frm, t)].reshape(dst.shape)


def exchange_vector_unsigned(xs, ys, n, grid, n_pts=3):
    """Halo update of a pair of POSITIVE quantities that live on the two families of cell faces / edges (grid spacings):
    across a rotated edge the x- and y-members swap; the sign flips a true vector would get are dropped, as the reference
    does after its vector_halo_update (generation.py:1663-1676).  grid 'd': x on (X, Y_INTERFACE), y on (X_INTERFACE, Y);
    'c': x on (X_INTERFACE, Y), y on (X, Y_INTERFACE); 'a': both on cell centres."""
    (xxi, xyi), (yxi, yyi) = {"d": ((0, 1), (1, 0)), "c": ((1, 0), (0, 1)), "a": ((0, 0), (0, 0))}[grid]
    msgs = {}
    for t in range(6):
        for e in (W, E, N_, S):
            to, r = _neighbour(t, e)
            xd = _turn(xs[t][_strip(n, e, n_pts, True, xxi, xyi)].copy(), -r)
            yd = _turn(ys[t][_strip(n, e, n_pts, True, yxi, yyi)].copy(), -r)
            if (-r) % 4 in (1, 3):
                xd, yd = yd, xd
            msgs[(t, to)] = (xd, yd)
    for t in range(6):
        for e in (W, E, N_, S):
            frm, _ = _neighbour(t, e)
            xd, yd = msgs[(frm, t)]
            dx, dy = xs[t][_strip(n, e, n_pts, False, xxi, xyi)], ys[t][_strip(n, e, n_pts, False, yxi, yyi)]
            dx[...] = np.abs(xd).reshape(dx.shape)
            dy[...] = np.abs(yd).reshape(dy.shape)


def exchange_vector(xs, ys, n, grid="d", n_pts=3):
    """Halo update of a true vector field given by its two staggered components (rotate.py:37-50: across an edge whose
    neighbour is turned, the components swap and one changes sign).  grid 'd': x on (X, Y_INTERFACE), y on (X_INTERFACE, Y)."""
    (xxi, xyi), (yxi, yyi) = {"d": ((0, 1), (1, 0)), "c": ((1, 0), (0, 1)), "a": ((0, 0), (0, 0))}[grid]
    msgs = {}
    for t in range(6):
        for e in (W, E, N_, S):
            to, r = _neighbour(t, e)
            xd = _turn(xs[t][_strip(n, e, n_pts, True, xxi, xyi)].copy(), -r)
            yd = _turn(ys[t][_strip(n, e, n_pts, True, yxi, yyi)].copy(), -r)
            k = (-r) % 4
            if k == 1:
                xd, yd = yd, -xd
            elif k == 2:
                xd, yd = -xd, -yd
            elif k == 3:
                xd, yd = -yd, xd
            msgs[(t, to)] = (xd, yd)
    for t in range(6):
        for e in (W, E, N_, S):
            frm, _ = _neighbour(t, e)
            xd, yd = msgs[(frm, t)]
            dx, dy = xs[t][_strip(n, e, n_pts, False, xxi, xyi)], ys[t][_strip(n, e, n_pts, False, yxi, yyi)]
            dx[...] = xd.reshape(dx.shape)
            dy[...] = yd.reshape(dy.shape)


def _to_xyz_lh(lon, lat):
    """(lon, lat) -> Cartesian in the reference's convention for the face rotations (mirror.py:251-260: z = -sin lat)."""
    return np.stack([np.cos(lon) * np.cos(lat), np.sin(lon) * np.cos(lat), -np.sin(lat)])


def _from_xyz_lh(p):
    x, y, z = p
    r = np.sqrt(x * x + y * y + z * z)
    lon = np.where(np.abs(x) + np.abs(y) < 1.0e-10, 0.0, np.arctan2(y, x))
    lat = np.arccos(z / r) - PI / 2.0
    return lon, lat


def _rot(axis, deg, p):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    x, y, z = p
    if axis == 1:
        return np.stack([x, c * y + s * z, -s * y + c * z])
    if axis == 2:
        return np.stack([c * x - s * z, y, s * x + c * z])
    return np.stack([c * x + s * y, -s * x + c * y, z])


# the rotations that carry face 0 to faces 1 .. 5 (mirror.py:74-207), applied left to right
_FACE_ROTATIONS = {1: ((3, -90.0),), 2: ((3, -90.0), (1, 90.0)), 3: ((3, -180.0), (1, 90.0)), 4: ((3, 90.0), (2, 90.0)),
                   5: ((2, 90.0), (3, 0.0))}


def face0(n):
    """(lon, lat) of the (N + 1) x (N + 1) corner points of face 0 (before the -pi/18 shift)."""
    alpha = np.arcsin(3.0 ** -0.5)
    ang = -alpha + np.arange(n + 1) * (2.0 * alpha / n)
    # an edge point at angle a lies at height tan(a) * sqrt(2) / sqrt(3) on the cube face x = -1 / sqrt(3)
    t = np.tan(ang) * (2.0 ** 0.5) * (3.0 ** -0.5)
    y = -t[:, None] * np.ones((1, n + 1))
    z = np.ones((n + 1, 1)) * t[None, :]
    x = np.full((n + 1, n + 1), -(3.0 ** -0.5))
    r = np.sqrt(x * x + y * y + z * z)
    x, y, z = x / r, y / r, z / r
    lon = np.where(np.abs(x) + np.abs(y) < 1.0e-10, 0.0, np.arctan2(y, x))
    lon = np.where(lon < 0.0, lon + 2.0 * PI, lon)
    lat = np.arcsin(z)
    lon = lon - PI
    # the reference averages the four mirror images of face 0 (mirror.py:40-72); the images of this construction agree to
    # rounding, the average only symmetrises the last bits: do the same
    alon = 0.25 * (np.abs(lon) + np.abs(lon[::-1, :]) + np.abs(lon[:, ::-1]) + np.abs(lon[::-1, ::-1]))
    alat = 0.25 * (np.abs(lat) + np.abs(lat[::-1, :]) + np.abs(lat[:, ::-1]) + np.abs(lat[::-1, ::-1]))
    lon, lat = np.copysign(alon, lon), np.copysign(alat, lat)
    if n % 2 == 0:
        lon[n // 2, :] = 0.0  # dateline / Greenwich consistency (mirror.py:67-71)
    return lon, lat


def corner_positions(n, halo=3):
    """lon, lat: lists of six (N + 7, N + 7) arrays (corner point (i, j) of the tile at storage index (i, j); the last row /
    column of the storage beyond the N + 1 + 2 halo corner points does not exist: arrays are (N + 1 + 2 halo) square)."""
    lon0, lat0 = face0(n)
    size = n + 1 + 2 * halo
    lons, lats = [], []
    mid = n // 2
    for tile in range(6):
        lon, lat = lon0.copy(), lat0.copy()
        if tile > 0:
            p = _to_xyz_lh(lon, lat)
            for axis, deg in _FACE_ROTATIONS[tile]:
                lon_, lat_ = _from_xyz_lh(_rot(axis, deg, p))
                p = _to_xyz_lh(lon_, lat_)  # the reference converts back and forth between the two rotations
            lon, lat = lon_, lat_
            if n % 2 == 0:  # pole and dateline consistency (mirror.py:104-207)
                if tile == 2:
                    lon[mid, mid], lat[mid, mid] = 0.0, PI / 2.0
                    lon[: mid + 1, mid] = 0.0
                    lon[mid + 1:, mid] = PI
                elif tile == 3:
                    lon[:, mid] = PI
                elif tile == 5:
                    lon[mid, mid], lat[mid, mid] = 0.0, -PI / 2.0
                    lon[mid, mid + 1:] = 0.0
                    lon[mid, :mid] = PI
        full_lon, full_lat = np.zeros((size, size)), np.zeros((size, size))
        full_lon[halo:halo + n + 1, halo:halo + n + 1] = lon
        full_lat[halo:halo + n + 1, halo:halo + n + 1] = lat
        c = (slice(halo, halo + n + 1), slice(halo, halo + n + 1))
        full_lon[c] -= PI / 18.0
        v = full_lon[c]
        v[v < 0.0] += 2.0 * PI
        full_lon[np.abs(full_lon) < 1.0e-10] = 0.0
        full_lat[np.abs(full_lat) < 1.0e-10] = 0.0
        lons.append(full_lon)
        lats.append(full_lat)
    exchange_scalar(lons, n, xi=1, yi=1, n_pts=halo)
    exchange_scalar(lats, n, xi=1, yi=1, n_pts=halo)
    for t in range(6):
        fill_corners_b_x(lons[t], n, halo)
        fill_corners_b_x(lats[t], n, halo)
    return lons, lats


def fill_corners_b_x(q, n, halo=3):
    """fill_corners_2d(..., gridtype='B', direction='x') (gnomonic/geometry helpers of the reference: the 3 x 3 corner halos
    of a corner-point field take the values of the adjacent edge halo, turned)."""
    o = halo
    e = o + n  # last compute corner index
    for i in range(1, 1 + halo):
        for j in range(1, 1 + halo):
            q[o - i, o - j] = q[o - j, o + i]          # south-west
            q[e + i, o - j] = q[e + j, o + i]          # south-east
            q[e + i, e + j] = q[e + j, e - i]          # north-east
            q[o - i, e + j] = q[o - j, e - i]          # north-west
