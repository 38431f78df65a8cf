"""Grid metrics of the six cubed-sphere tiles from their corner positions (reference: util/pace/util/grid/generation.py
`MetricTerms`, geometry.py, gnomonic.py -- re-derived here as vectorised numpy over all six tiles; the reference's choices at
tile edges and corners (doubled half-cells, triangle areas, one-sided angles) are kept because the dynamical core is
discretised against them).

`generate(n, nz)` returns a list of six dicts with the names of pace_amd.util.grid.GridData.  Arrays are (N + 7, N + 7)
storages with halo 3: cell (i, j) / corner (i, j) of the compute domain at index (i + 3, j + 3).
"""
import numpy as np

from .. import constants as c
from .cornerfill import fill_pair, fill_scalar
from .positions import corner_positions, exchange_scalar, exchange_vector_unsigned

PI = np.pi
R = c.RADIUS
O = 3  # halo


# ---------------------------------------------------------------------------------------------------------------- geometry
def to_xyz(lon, lat):
    p = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1)
    return p / np.sqrt((p * p).sum(axis=-1, keepdims=True))


def to_lonlat(p):
    p = p / np.sqrt((p * p).sum(axis=-1, keepdims=True))
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    lon = np.where(np.abs(x) + np.abs(y) >= 1.0e-10, np.arctan2(y, x), 0.0)
    lon = np.where(lon < 0.0, lon + 2.0 * PI, lon)
    return lon, np.arcsin(z)


def unit(p):
    return p / np.sqrt((p * p).sum(axis=-1, keepdims=True))


def arc(lon1, lat1, lon2, lat2):
    """great-circle distance on the unit sphere (haversine form, gnomonic.py:329-343)"""
    return 2.0 * np.arcsin(np.sqrt(np.sin((lat1 - lat2) / 2.0) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin((lon1 - lon2) / 2.0) ** 2))


def arc_xyz(p, q):
    return arc(*to_lonlat(p), *to_lonlat(q))


def vertex_angle(pc, p2, p3):
    """angle at pc between the great circles pc-p2 and pc-p3 (gnomonic.py:627-654)"""
    a, b = np.cross(pc, p2), np.cross(pc, p3)
    with np.errstate(invalid="ignore", divide="ignore"):
        ang = np.arccos((a * b).sum(axis=-1) / np.sqrt((a * a).sum(axis=-1) * (b * b).sum(axis=-1)))
    return np.where(np.isnan(ang), 0.0, ang)


def quad_area(p1, p2, p3, p4):
    """spherical excess of the quadrilateral p1 p2 p3 p4 (corners in order), times R^2 (gnomonic.py:560-578)"""
    tot = vertex_angle(p2, p3, p1)
    for a, b, cc in ((p3, p2, p4), (p4, p3, p1), (p1, p4, p2)):
        tot = tot + vertex_angle(a, b, cc)
    return (tot - 2.0 * PI) * R ** 2


def tri_area(p1, p2, p3):
    tot = vertex_angle(p1, p2, p3)
    for a, b, cc in ((p2, p3, p1), (p3, p1, p2)):
        tot = tot + vertex_angle(a, b, cc)
    return (tot - PI) * R ** 2


def cell_areas(lon, lat):
    p = to_xyz(lon, lat)
    return quad_area(p[:-1, :-1], p[:-1, 1:], p[1:, 1:], p[1:, :-1])


# ------------------------------------------------------------------------------------------------------------- the metrics
def generate(n, nz=79):
    size = n + 7
    lons, lats = corner_positions(n)            # (N + 7)^2 corner arrays incl. halo
    T = range(6)
    G = [dict(lon=lons[t], lat=lats[t]) for t in T]
    e = O + n  # storage index of the last compute corner

    # ---- cell centres (generation.py:1621-1642): mean of the four corners in Cartesian space, exchanged, corner-filled
    for t in T:
        p = to_xyz(lons[t], lats[t])
        ctr = unit(p[1:, 1:] + p[:-1, :-1] + p[1:, :-1] + p[:-1, 1:])
        la, ta = np.zeros((size, size)), np.zeros((size, size))
        la[:-1, :-1], ta[:-1, :-1] = to_lonlat(ctr)
        G[t]["lon_agrid"], G[t]["lat_agrid"] = la, ta
    exchange_scalar([G[t]["lon_agrid"] for t in T], n)
    exchange_scalar([G[t]["lat_agrid"] for t in T], n)
    for t in T:
        fill_scalar(G[t]["lon_agrid"], n, "A", "x")
        fill_scalar(G[t]["lat_agrid"], n, "A", "y")

    # ---- dx, dy: corner-to-corner arcs on the compute domain, exchanged as an unsigned pair (generation.py:1644-1676)
    for t in T:
        lo, la = lons[t], lats[t]
        dx, dy = np.zeros((size, size)), np.zeros((size, size))
        cw = slice(O, e + 1)
        dx[O:e, cw] = R * arc(lo[O:e, cw], la[O:e, cw], lo[O + 1:e + 1, cw], la[O + 1:e + 1, cw])
        dy[cw, O:e] = R * arc(lo[cw, O:e], la[cw, O:e], lo[cw, O + 1:e + 1], la[cw, O + 1:e + 1])
        G[t]["dx"], G[t]["dy"] = dx, dy
    exchange_vector_unsigned([G[t]["dx"] for t in T], [G[t]["dy"] for t in T], n, "d")
    for t in T:
        fill_pair(G[t]["dx"], G[t]["dy"], n, "d")

    # ---- dxa, dya: arcs between face midpoints, on the whole storage, corner-filled, then exchanged (:1678-1712)
    for t in T:
        p = to_xyz(lons[t], lats[t])
        ym = to_lonlat(unit(p[:, :-1] + p[:, 1:]))   # midpoints of the x-faces' ... (i, j+1/2)
        xm = to_lonlat(unit(p[:-1, :] + p[1:, :]))   # (i+1/2, j)
        dxa_t = R * arc(ym[0][:-1, :], ym[1][:-1, :], ym[0][1:, :], ym[1][1:, :])
        dya_t = R * arc(xm[0][:, :-1], xm[1][:, :-1], xm[0][:, 1:], xm[1][:, 1:])
        dxa, dya = np.zeros((size, size)), np.zeros((size, size))
        dxa[:-1, :-1], dya[:-1, :-1] = dxa_t, dya_t
        fill_pair(dxa, dya, n, "a")
        G[t]["dxa"], G[t]["dya"] = dxa, dya
    exchange_vector_unsigned([G[t]["dxa"] for t in T], [G[t]["dya"] for t in T], n, "a")

    # ---- dxc, dyc: arcs between cell centres; at a tile edge twice the arc from the edge midpoint to the first centre
    #      (:1714-1772, gnomonic.py:505-557)
    for t in T:
        la, ta = G[t]["lon_agrid"][:-1, :-1], G[t]["lat_agrid"][:-1, :-1]
        dxc, dyc = np.zeros((size, size)), np.zeros((size, size))
        dxc_t = R * arc(la[:-1, :], ta[:-1, :], la[1:, :], ta[1:, :])
        dyc_t = R * arc(la[:, :-1], ta[:, :-1], la[:, 1:], ta[:, 1:])
        dxc[1:-1, :-1], dxc[0, :-1], dxc[-1, :-1] = dxc_t, dxc_t[0, :], dxc_t[-1, :]
        dyc[:-1, 1:-1], dyc[:-1, 0], dyc[:-1, -1] = dyc_t, dyc_t[:, 0], dyc_t[:, -1]
        pd = to_xyz(lons[t], lats[t])
        pa = to_xyz(G[t]["lon_agrid"], G[t]["lat_agrid"])
        cs = slice(O, e)
        west = 0.5 * (pd[O, O + 1:e + 1] + pd[O, O:e])
        dxc[O, cs] = 2.0 * R * arc_xyz(west, pa[O, cs])
        east = 0.5 * (pd[e, O + 1:e + 1] + pd[e, O:e])
        dxc[e, cs] = 2.0 * R * arc_xyz(east, pa[e - 1, cs])
        south = 0.5 * (pd[O + 1:e + 1, O] + pd[O:e, O])
        dyc[cs, O] = 2.0 * R * arc_xyz(south, pa[cs, O])
        north = 0.5 * (pd[O + 1:e + 1, e] + pd[O:e, e])
        dyc[cs, e] = 2.0 * R * arc_xyz(north, pa[cs, e - 1])
        G[t]["dxc"], G[t]["dyc"] = dxc, dyc
    exchange_vector_unsigned([G[t]["dxc"] for t in T], [G[t]["dyc"] for t in T], n, "c")
    for t in T:
        fill_pair(G[t]["dxc"], G[t]["dyc"], n, "c")

    # ---- area: spherical excess of the compute cells, exchanged (:1774-1785); area_c: cells of the dual grid, halved /
    #      third-ed shapes at tile edges / corners (:1787-1826, gnomonic.py:386-503)
    for t in T:
        area = np.full((size, size), -1.0e8)
        area[O:e, O:e] = cell_areas(lons[t][O:e + 1, O:e + 1], lats[t][O:e + 1, O:e + 1])
        G[t]["area"] = area
    exchange_scalar([G[t]["area"] for t in T], n)
    for t in T:
        la, ta = G[t]["lon_agrid"], G[t]["lat_agrid"]
        ac = np.zeros((size, size))
        ac[O:e + 1, O:e + 1] = cell_areas(la[O - 1:e + 1, O - 1:e + 1], ta[O - 1:e + 1, O - 1:e + 1])
        pa = to_xyz(la, ta)
        pd = to_xyz(lons[t], lats[t])
        # corners: the triangle of the three cell centres that are on the tile (the fourth "centre" is a corner-fill copy)
        ac[O, O] = tri_area(pa[O - 1, O], pa[O, O], pa[O, O - 1])
        ac[e, O] = tri_area(pa[e, O], pa[e - 1, O], pa[e - 1, O - 1])
        ac[e, e] = tri_area(pa[e, e - 1], pa[e - 1, e - 1], pa[e - 1, e])
        ac[O, e] = tri_area(pa[O - 1, e - 1], pa[O, e - 1], pa[O, e])
        # edges: twice the half cell on this side of the edge (between the edge's face midpoints and the first row of centres);
        # the reference's helper receives arrays trimmed by two points, which makes it cover corner points 0 .. N incl. the
        # tile corners (they keep this doubled half-cell value, the triangle above is overwritten)
        cw = slice(O, e + 1)
        ym = 0.5 * (pd[O, O - 1:e + 1] + pd[O, O:e + 2])      # midpoints along the west edge, j - 1/2 for j = 0 .. N + 1
        ac[O, cw] = 2.0 * quad_area(ym[:-1], pa[O, O - 1:e], pa[O, O:e + 1], ym[1:])
        ym = 0.5 * (pd[e, O - 1:e + 1] + pd[e, O:e + 2])
        ac[e, cw] = 2.0 * quad_area(ym[:-1], pa[e - 1, O - 1:e], pa[e - 1, O:e + 1], ym[1:])
        xm = 0.5 * (pd[O - 1:e + 1, e] + pd[O:e + 2, e])
        ac[cw, e] = 2.0 * quad_area(xm[:-1], pa[O - 1:e, e - 1], pa[O:e + 1, e - 1], xm[1:])
        xm = 0.5 * (pd[O - 1:e + 1, O] + pd[O:e + 2, O])
        ac[cw, O] = 2.0 * quad_area(xm[:-1], pa[O - 1:e, O], pa[O:e + 1, O], xm[1:])
        G[t]["area_c"] = ac
    exchange_scalar([G[t]["area_c"] for t in T], n, xi=1, yi=1)
    for t in T:
        fill_scalar(G[t]["area_c"], n, "B", "x")
    for t in T:
        for k in ("dx", "dy", "dxa", "dya", "dxc", "dyc", "area", "area_c"):
            with np.errstate(divide="ignore"):
                G[t]["r" + k] = 1.0 / G[t][k]
    for t in T:
        _angles(G[t], n)
    # the divergence / del-6 factors are exchanged like a pair of C-grid spacings (generation.py:2198-2206)
    exchange_vector_unsigned([G[t]["divg_v"] for t in T], [G[t]["divg_u"] for t in T], n, "c")
    exchange_vector_unsigned([G[t]["del6_v"] for t in T], [G[t]["del6_u"] for t in T], n, "c")
    da_min = min(float(G[t]["area"][O:e, O:e].min()) for t in T)
    da_min_c = min(float(G[t]["area_c"][O:e, O:e].min()) for t in T)
    from .eta import hybrid_coefficients

    ak, bk, ptop = hybrid_coefficients(nz)
    p_ref = ak + bk * 1.0e5
    for t in T:
        G[t].update(da_min=da_min, da_min_c=da_min_c, ak=ak, bk=bk, ptop=ptop, p_ref=1.0e5, dp_ref=ak[1:] - ak[:-1] + (bk[1:] - bk[:-1]) * 1.0e5)
        G[t]["p"] = _reference_layer_pressure(ak, bk)
        G[t]["fC"] = 2.0 * c.OMEGA * np.sin(G[t]["lat"])
        G[t]["fC_agrid"] = 2.0 * c.OMEGA * np.sin(G[t]["lat_agrid"])
    return G


def _reference_layer_pressure(ak, bk):
    """GridData.p (grid/helper.py): layer-mean reference pressure, (p_int[k+1] - p_int[k]) / log(p_int[k+1] / p_int[k])."""
    p_int = ak + bk * 1.0e5
    return (p_int[1:] - p_int[:-1]) / np.log(p_int[1:] / p_int[:-1])


def _cosang(pc, p2, p3):
    a, b = np.cross(pc, p2), np.cross(pc, p3)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (a * b).sum(axis=-1) / np.sqrt((a * a).sum(axis=-1) * (b * b).sum(axis=-1))


def _mirrors(n):
    """index maps that turn each corner of the tile into the south-west one: (flip x, flip y)"""
    return ((False, False), (True, False), (False, True), (True, True))


def _angles(g, n):
    """cos / sin of the grid angle at the nine supergrid points of every cell, what is derived from them on faces, corners and
    centres, the divergence / del-6 factors, the A -> B edge weights and the local-to-lat-lon matrix
    (generation.py:1838-2358, geometry.py:13-720).  All of it is local to a tile: it works on the whole storage incl. halo."""
    BIG, TINY = 1.0e8, 1.0e-8
    size = n + 7
    e = O + n
    pd = to_xyz(g["lon"], g["lat"])                                   # corners, (size, size, 3)
    pa = to_xyz(g["lon_agrid"][:-1, :-1], g["lat_agrid"][:-1, :-1])  # centres, (size - 1, size - 1, 3)
    c00, c10, c01, c11 = pd[:-1, :-1], pd[1:, :-1], pd[:-1, 1:], pd[1:, 1:]

    # unit vectors at cell centres along x and y (get_center_vector); the 3 x 3 corner halos carry no meaning
    ctr = unit(c00 + c10 + c01 + c11)
    ec1 = unit(np.cross(ctr, np.cross(unit(c10 + c11), unit(c00 + c01))))
    ec2 = unit(np.cross(ctr, np.cross(unit(c01 + c11), unit(c00 + c10))))

    cs = np.full((size - 1, size - 1, 9), BIG)
    cs[:, :, 5] = _cosang(c00, c10, c01)
    cs[:, :, 6] = -_cosang(c10, c00, c11)
    cs[:, :, 7] = _cosang(c11, c10, c01)
    cs[:, :, 8] = -_cosang(c01, c00, c11)
    cs[:, :, 0] = _cosang(unit(c00 + c01), pa, c01)
    cs[:, :, 1] = _cosang(unit(c00 + c10), c10, pa)
    cs[:, :, 2] = _cosang(unit(c10 + c11), pa, c10)
    cs[:, :, 3] = _cosang(unit(c01 + c11), c01, pa)
    ec1h, ec2h = ec1.copy(), ec2.copy()
    for sx in (slice(0, O), slice(-O, None)):
        for sy in (slice(0, O), slice(-O, None)):
            ec1h[sx, sy], ec2h[sx, sy] = BIG, BIG
    cs[:, :, 4] = (ec1h * ec2h).sum(axis=-1)
    cs[np.abs(1.0 - cs) < 1e-15] = 1.0
    sn = np.sqrt(np.clip(1.0 - cs ** 2, 0.0, None))
    sn[sn > 1.0] = 1.0

    # around a cube corner the outward-facing faces of the halo cells take the sine of the face they coincide with on the
    # neighbouring tile (geometry.py:222-236); written for the south-west corner, the others through mirrored views
    def view(a, fx, fy):
        a = a[::-1] if fx else a
        return a[:, ::-1] if fy else a

    def face(k, fx, fy):
        # supergrid faces 0 (west) 1 (south) 2 (east) 3 (north) exchange roles under mirrors
        if fx and k in (0, 2):
            k = 2 - k
        if fy and k in (1, 3):
            k = 4 - k
        return k

    for fx, fy in _mirrors(n):
        v = [view(sn[:, :, face(k, fx, fy)], fx, fy) for k in range(4)]
        v[2][O - 1, :O] = v[1][:O, O]
        v[3][:O, O - 1] = v[0][O, :O]
    # (kept from the reference, geometry.py:229: at the north-west corner this first adjustment takes the west-face sines of
    # rows N-2 .. N instead of the turned N .. N+2; sina_v / rsin_v / the divergence factors derived below see it, the
    # sin_sg fields themselves are set again, symmetrically, by the corner fix further down)
    sn[:O, -O, 1] = sn[O, -O - 2:size - 1 - O + 1, 0]

    # ---- cosa, sina at corners; face and centre values (calculate_trig_uv)
    cosa, sina = np.full((size, size), BIG), np.full((size, size), BIG)
    cosa[O:-O, O:-O] = 0.5 * (cs[O - 1:-O, O - 1:-O, 7] + cs[O:size - 1 - O + 1, O:size - 1 - O + 1, 5])
    sina[O:-O, O:-O] = 0.5 * (sn[O - 1:-O, O - 1:-O, 7] + sn[O:size - 1 - O + 1, O:size - 1 - O + 1, 5])
    cosa_u, sina_u, rsin_u = (np.full((size, size - 1), BIG) for _ in range(3))
    cosa_v, sina_v, rsin_v = (np.full((size - 1, size), BIG) for _ in range(3))
    cosa_u[1:-1] = 0.5 * (cs[:-1, :, 2] + cs[1:, :, 0])
    sina_u[1:-1] = 0.5 * (sn[:-1, :, 2] + sn[1:, :, 0])
    rsin_u[1:-1] = 1.0 / np.maximum(sina_u[1:-1] ** 2, TINY)
    cosa_v[:, 1:-1] = 0.5 * (cs[:, :-1, 3] + cs[:, 1:, 1])
    sina_v[:, 1:-1] = 0.5 * (sn[:, :-1, 3] + sn[:, 1:, 1])
    rsin_v[:, 1:-1] = 1.0 / np.maximum(sina_v[:, 1:-1] ** 2, TINY)
    cosa_s = cs[:, :, 4].copy()
    rsin2 = 1.0 / np.maximum(sn[:, :, 4] ** 2, TINY)
    for sx in (slice(0, O), slice(-O, None)):
        for sy in (slice(0, O), slice(-O, None)):
            cosa_s[sx, sy] = BIG
    rsina = 1.0 / np.maximum(sina[O:-O, O:-O] ** 2, TINY)
    # on a tile edge: 1 / sin instead of 1 / sin^2 on the faces, and no corner value at all
    def signed_floor(a):
        return np.where(np.abs(a) < TINY, TINY * np.sign(a), a)

    rsina[0, :], rsina[-1, :], rsina[:, 0], rsina[:, -1] = BIG, BIG, BIG, BIG
    rsin_u[O] = 1.0 / signed_floor(sina_u[O])
    rsin_u[-O - 1] = 1.0 / signed_floor(sina_u[-O - 1])
    rsin_v[:, O] = 1.0 / signed_floor(sina_v[:, O])
    rsin_v[:, -O - 1] = 1.0 / signed_floor(sina_v[:, -O - 1])

    # ---- supergrid_corner_fix: the corner halos are voided, then the two rows next to them take the turned neighbours
    for fx, fy in _mirrors(n):
        for arr, void in ((sn, TINY), (cs, BIG)):
            view(arr, fx, fy)[:O, :O, :] = void
            v = [view(arr[:, :, face(k, fx, fy)], fx, fy) for k in range(4)]
            v[2][O - 1, :O] = v[1][:O, O]
            v[3][:O, O - 1] = v[0][O, :O]

    def pad(a):
        out = np.zeros((size, size))
        out[:a.shape[0], :a.shape[1]] = a
        return out

    for k in range(1, 5):
        g[f"cos_sg{k}"], g[f"sin_sg{k}"] = pad(cs[:, :, k - 1]), pad(sn[:, :, k - 1])
    sin_sg5 = sn[:, :, 4]
    g["cosa"], g["sina"] = cosa, sina
    g["cosa_u"], g["sina_u"], g["rsin_u"] = pad(cosa_u), pad(sina_u), pad(rsin_u)
    g["cosa_v"], g["sina_v"], g["rsin_v"] = pad(cosa_v), pad(sina_v), pad(rsin_v)
    g["cosa_s"], g["rsin2"] = pad(cosa_s), pad(rsin2)
    rs = np.zeros((size, size))
    rs[O:-O, O:-O] = rsina
    g["rsina"] = rs

    # ---- divergence / del-6 factors (calculate_divg_del6): on a tile edge the face sine is the mean of the two cells' own
    dx, dy, dxc, dyc = g["dx"][:-1, :], g["dy"][:, :-1], g["dxc"][:, :-1], g["dyc"][:-1, :]
    with np.errstate(all="ignore"):
        divg_u, del6_u = sina_v * dyc / dx, sina_v * dx / dyc
        divg_v, del6_v = sina_u * dxc / dy, sina_u * dy / dxc
        for j, (ja, jb) in ((O, (O, O - 1)), (size - 1 - O, (size - 1 - O, size - 2 - O))):
            sm = 0.5 * (sn[:, ja, 1] + sn[:, jb, 3])
            divg_u[:, j], del6_u[:, j] = sm * dyc[:, j] / dx[:, j], sm * dx[:, j] / dyc[:, j]
        for i, (ia, ib) in ((O, (O, O - 1)), (size - 1 - O, (size - 1 - O, size - 2 - O))):
            sm = 0.5 * (sn[ia, :, 0] + sn[ib, :, 2])
            divg_v[i], del6_v[i] = sm * dxc[i] / dy[i], sm * dy[i] / dxc[i]
    g["divg_u"], g["del6_u"], g["divg_v"], g["del6_v"] = pad(divg_u), pad(del6_u), pad(divg_v), pad(del6_v)

    # ---- local -> lat-lon wind matrix at cell centres (calculate_grid_z / calculate_grid_a)
    lo, la = g["lon_agrid"][:-1, :-1], g["lat_agrid"][:-1, :-1]
    vlon = np.stack([-np.sin(lo), np.cos(lo), np.zeros_like(lo)], axis=-1)
    vlat = np.stack([-np.sin(la) * np.cos(lo), -np.sin(la) * np.sin(lo), np.cos(la)], axis=-1)
    z11, z12 = (ec1h * vlon).sum(axis=-1), (ec1h * vlat).sum(axis=-1)
    z21, z22 = (ec2h * vlon).sum(axis=-1), (ec2h * vlat).sum(axis=-1)
    g["a11"], g["a12"] = pad(0.5 * z22 / sin_sg5), pad(-0.5 * z12 / sin_sg5)
    g["a21"], g["a22"] = pad(-0.5 * z21 / sin_sg5), pad(0.5 * z11 / sin_sg5)

    # ---- unit vectors the initial state projects the zonal wind on (geometry.py:60-150,294-345): along x / y at the
    #      compute-domain corner points (ee1, ee2; one-sided on the tile edges), along the south faces (es1) and the west
    #      faces (ew2) at their midpoints
    cw = slice(O, e + 1)
    ax = np.cross(pd[O - 1:e, cw], pd[O + 1:e + 2, cw])
    ax[0], ax[-1] = np.cross(pd[O, cw], pd[O + 1, cw]), np.cross(pd[e - 1, cw], pd[e, cw])
    ay = np.cross(pd[cw, O - 1:e], pd[cw, O + 1:e + 2])
    ay[:, 0], ay[:, -1] = np.cross(pd[cw, O], pd[cw, O + 1]), np.cross(pd[cw, e - 1], pd[cw, e])
    ee1, ee2 = np.full((size, size, 3), np.nan), np.full((size, size, 3), np.nan)
    ee1[cw, cw], ee2[cw, cw] = unit(np.cross(ax, pd[cw, cw])), unit(np.cross(ay, pd[cw, cw]))
    es1, ew2 = np.full((size, size, 3), np.nan), np.full((size, size, 3), np.nan)
    mid = unit(pd[:-1, 1:-1] + pd[1:, 1:-1])
    es1[:-1, 1:-1] = unit(np.cross(np.cross(pd[:-1, 1:-1], pd[1:, 1:-1]), mid))
    mid = unit(pd[1:-1, :-1] + pd[1:-1, 1:])
    ew2[1:-1, :-1] = unit(np.cross(np.cross(pd[1:-1, :-1], pd[1:-1, 1:]), mid))
    g["ee1"], g["ee2"], g["es1"], g["ew2"] = ee1, ee2, es1, ew2

    # ---- A -> B interpolation weights on the four tile edges (edge_factors): for the corner points 1 .. N - 1 of an edge,
    #      d2 / (d1 + d2) with d1, d2 the arcs from the point to the midpoints (across the edge) of the two cell-centre pairs
    def edge_weights(lon_c, lat_c, lon_a0, lat_a0, lon_a1, lat_a1):
        m = to_lonlat(unit(to_xyz(lon_a0, lat_a0) + to_xyz(lon_a1, lat_a1)))   # midpoints for cells 0 .. N - 1
        d1 = arc(m[0][:-1], m[1][:-1], lon_c, lat_c)
        d2 = arc(m[0][1:], m[1][1:], lon_c, lat_c)
        return d2 / (d1 + d2)

    LA, TA, LO, LT = g["lon_agrid"], g["lat_agrid"], g["lon"], g["lat"]
    ci, cc = slice(O + 1, e), slice(O, e)  # corner points 1 .. N - 1, cells 0 .. N - 1
    ew, ee_, es_, en = (np.zeros(size) for _ in range(4))
    for a in (ew, ee_, es_, en):
        a[O:-O] = BIG  # corner points 0 and N of an edge have no weight
    ew[ci] = edge_weights(LO[O, ci], LT[O, ci], LA[O - 1, cc], TA[O - 1, cc], LA[O, cc], TA[O, cc])
    ee_[ci] = edge_weights(LO[e, ci], LT[e, ci], LA[e, cc], TA[e, cc], LA[e - 1, cc], TA[e - 1, cc])
    es_[ci] = edge_weights(LO[ci, O], LT[ci, O], LA[cc, O - 1], TA[cc, O - 1], LA[cc, O], TA[cc, O])
    en[ci] = edge_weights(LO[ci, e], LT[ci, e], LA[cc, e], TA[cc, e], LA[cc, e - 1], TA[cc, e - 1])
    g["edge_w"], g["edge_e"], g["edge_s"], g["edge_n"] = ew, ee_, es_, en
