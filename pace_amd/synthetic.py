"""Synthetic inputs for benchmarking and for HIP-vs-oracle tests at resolutions where no reference
fixture exists (C48 ... C384).

NOT the reference's grid generator (util/pace/util/grid/generation.py -- SURVEY.md section 8f rank 4,
"next").  This builds ONE equiangular gnomonic tile centred on (lon 180, lat 0) with its halo obtained
by extending the projection past the tile edge, and derives every metric term the acoustic step reads
from plain spherical geometry.  Magnitudes, smoothness and sign conventions match a real cubed-sphere
tile (cell areas, non-orthogonality angles up to ~30 degrees at the corners, dx != dy), which is what
the kernels' cost and branch behaviour depend on; values differ from the reference grid in the halo
(no neighbouring-face geometry) and at tile edges (no edge-specific averaging).

The model state is a smooth hydrostatically balanced atmosphere with a zonal jet plus a wave
perturbation -- the same kind of field as the Jablonowski-Williamson test (baroclinic.py:436) --
so PPM limiter branches, upwind selections and the vertical solver see realistic data.  Deterministic.
"""
import numpy as np

from .util import constants as c


def _norm(v):
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _dist(p, q, r):
    return r * np.arccos(np.clip(np.sum(p * q, axis=-1), -1.0, 1.0))


def _mid(p, q):
    return _norm(p + q)


def _tri_area(a, b, cc, r):
    # l'Huilier via spherical excess from vertex angles
    def ang(o, p, q):
        u = _norm(np.cross(o, p))
        v = _norm(np.cross(o, q))
        return np.arccos(np.clip(np.sum(u * v, axis=-1), -1.0, 1.0))

    return r * r * (ang(a, b, cc) + ang(b, cc, a) + ang(cc, a, b) - np.pi)


def _quad_area(p1, p2, p3, p4, r):
    return _tri_area(p1, p2, p3, r) + _tri_area(p1, p3, p4, r)


def _pad(a, shape):
    out = np.zeros(shape)
    s = tuple(slice(0, min(x, y)) for x, y in zip(a.shape, shape))
    out[s] = a[s]
    # replicate the last valid row/column into the spare points so no metric is zero
    for ax in range(len(shape)):
        n = a.shape[ax]
        if n < shape[ax]:
            idx = [slice(None)] * len(shape)
            src = [slice(None)] * len(shape)
            idx[ax] = slice(n, shape[ax])
            src[ax] = slice(n - 1, n)
            out[tuple(idx)] = out[tuple(src)]
    return out


def _angle_cos(pw, pe, ps, pn, p0):
    """cos of the angle between the local x and y directions at p0 (tangent-plane unit vectors)."""
    ex = pe - pw
    ex = ex - np.sum(ex * p0, axis=-1, keepdims=True) * p0
    ey = pn - ps
    ey = ey - np.sum(ey * p0, axis=-1, keepdims=True) * p0
    return np.sum(_norm(ex) * _norm(ey), axis=-1)


def hybrid_levels(nz, ptop=300.0, ps_ref=1.0e5):
    """Smooth hybrid sigma-pressure coefficients ak, bk (nz+1 interfaces), pure pressure above ~100 hPa.
    (The reference's tables, util/pace/util/grid/eta.py, exist only for 72/79/91 levels.)"""
    s = np.linspace(0.0, 1.0, nz + 1) ** 1.6
    p_ref = ptop + s * (ps_ref - ptop)
    sig = np.clip((p_ref - 1.0e4) / (ps_ref - 1.0e4), 0.0, 1.0)
    bk = sig ** 1.5
    ak = p_ref - bk * ps_ref
    ak[0] = ptop
    bk[0] = 0.0
    return ak, bk


def tile_metrics(n, nz=79, n_halo=3, radius=c.RADIUS):
    """dict name -> numpy array with the names of pace.util.grid.GridData / DampingCoefficients."""
    npt = n + 2 * n_halo + 1  # N+7 corner points
    shp = (npt, npt)
    d = (np.pi / 2.0) / n
    ang = -np.pi / 4.0 + (np.arange(npt) - n_halo) * d
    X, Y = np.meshgrid(np.tan(ang), np.tan(ang), indexing="ij")
    C = _norm(np.stack([-np.ones_like(X), -X, Y], axis=-1))  # face centred on lon = 180
    A = _norm(C[:-1, :-1] + C[1:, :-1] + C[:-1, 1:] + C[1:, 1:])  # cell centres (npt-1)^2
    W = _mid(C[:-1, :-1], C[:-1, 1:])  # west face mid of cell (i,j)  == east face of (i-1,j)
    W_all = _mid(C[:, :-1], C[:, 1:])  # (npt, npt-1): x-face midpoints at corner column i
    S_all = _mid(C[:-1, :], C[1:, :])  # (npt-1, npt): y-face midpoints at corner row j
    m = {}
    lon = np.mod(np.arctan2(C[..., 1], C[..., 0]), 2 * np.pi)
    lat = np.arcsin(C[..., 2])
    m["lon"], m["lat"] = lon, lat
    m["lon_agrid"] = _pad(np.mod(np.arctan2(A[..., 1], A[..., 0]), 2 * np.pi), shp)
    m["lat_agrid"] = _pad(np.arcsin(A[..., 2]), shp)
    m["dx"] = _pad(_dist(C[:-1, :], C[1:, :], radius), shp)  # along x at corner row j
    m["dy"] = _pad(_dist(C[:, :-1], C[:, 1:], radius), shp)
    m["dxa"] = _pad(_dist(W_all[:-1, :], W_all[1:, :], radius), shp)
    m["dya"] = _pad(_dist(S_all[:, :-1], S_all[:, 1:], radius), shp)
    dxc = np.zeros((npt - 1, npt - 1))
    dxc[1:, :] = _dist(A[:-1, :], A[1:, :], radius)
    dxc[0, :] = dxc[1, :]
    dyc = np.zeros((npt - 1, npt - 1))
    dyc[:, 1:] = _dist(A[:, :-1], A[:, 1:], radius)
    dyc[:, 0] = dyc[:, 1]
    m["dxc"], m["dyc"] = _pad(dxc, shp), _pad(dyc, shp)
    area = _quad_area(C[:-1, :-1], C[1:, :-1], C[1:, 1:], C[:-1, 1:], radius)
    m["area"] = _pad(area, shp)
    area_c = np.zeros((npt - 1, npt - 1))
    area_c[1:, 1:] = _quad_area(A[:-1, :-1], A[1:, :-1], A[1:, 1:], A[:-1, 1:], radius)
    area_c[0, :] = area_c[1, :]
    area_c[:, 0] = area_c[:, 1]
    m["area_c"] = _pad(area_c, shp)
    for nme in ("dx", "dy", "dxa", "dya", "dxc", "dyc", "area", "area_c"):
        m["r" + nme] = 1.0 / m[nme]
    # non-orthogonality angle at corners (B-grid), face midpoints and centres
    Cp = np.pad(C, ((1, 1), (1, 1), (0, 0)), mode="edge")
    cosa = _angle_cos(Cp[:-2, 1:-1], Cp[2:, 1:-1], Cp[1:-1, :-2], Cp[1:-1, 2:], C)
    sina = np.sqrt(1.0 - cosa ** 2)
    m["cosa"], m["sina"] = cosa, sina
    m["rsina"] = 1.0 / sina ** 2
    # west face of cell (i,j): between corners (i,j),(i,j+1)
    cu = _angle_cos(_padA(A)[:-1, 1:], _padA(A)[1:, 1:], C[:-1, :-1], C[:-1, 1:], W)
    cu = _pad(cu, shp)
    cu[0, :] = cu[1, :]  # the one-sided difference degenerates on the outermost halo column
    cv = _angle_cos(C[:-1, :-1], C[1:, :-1], _padA(A)[1:, :-1], _padA(A)[1:, 1:], S_all[:, :-1])
    cv = _pad(cv, shp)
    cv[:, 0] = cv[:, 1]
    cs = _angle_cos(W_all[:-1, :], W_all[1:, :], S_all[:, :-1], S_all[:, 1:], A)
    cs = _pad(cs, shp)
    m["cosa_u"], m["cosa_v"], m["cosa_s"] = cu, cv, cs
    m["sina_u"], m["sina_v"] = np.sqrt(1 - cu ** 2), np.sqrt(1 - cv ** 2)
    m["rsin_u"], m["rsin_v"], m["rsin2"] = 1.0 / (1 - cu ** 2), 1.0 / (1 - cv ** 2), 1.0 / (1 - cs ** 2)
    # supergrid sines / cosines at the 4 face midpoints of each cell: 1 west, 2 south, 3 east, 4 north
    m["cos_sg1"], m["cos_sg2"] = cu.copy(), cv.copy()
    m["cos_sg3"] = _pad(cu[1:, :], shp)
    m["cos_sg4"] = _pad(cv[:, 1:], shp)
    for t in "1234":
        m["sin_sg" + t] = np.sqrt(1.0 - m["cos_sg" + t] ** 2)
    m["divg_u"] = m["sina_v"] * m["dyc"] / m["dx"]
    m["divg_v"] = m["sina_u"] * m["dxc"] / m["dy"]
    m["del6_u"] = m["sina_v"] * m["dx"] / m["dyc"]
    m["del6_v"] = m["sina_u"] * m["dy"] / m["dxc"]
    m["fC"] = 2.0 * c.OMEGA * np.sin(lat)
    m["fC_agrid"] = 2.0 * c.OMEGA * np.sin(m["lat_agrid"])
    ew = np.full(npt, 1.0e8)
    lo, hi = n_halo + 1, n_halo + n
    ew[lo:hi] = 0.5 - 0.25 * np.cos(np.linspace(0, np.pi, hi - lo))
    m["edge_w"], m["edge_e"], m["edge_s"], m["edge_n"] = ew.copy(), ew[::-1].copy(), ew.copy(), ew[::-1].copy()
    cs_ = slice(n_halo, n_halo + n)
    m["da_min"] = float(m["area"][cs_, cs_].min())
    m["da_min_c"] = float(m["area_c"][n_halo : n_halo + n + 1, n_halo : n_halo + n + 1].min())
    ak, bk = hybrid_levels(nz)
    m["ak"], m["bk"], m["ptop"] = ak, bk, float(ak[0])
    p_int = ak + bk * 1.0e5
    m["dp_ref"] = p_int[1:] - p_int[:-1]
    m["p_ref"] = p_int
    m["p"] = (p_int[1:] - p_int[:-1]) / np.log(p_int[1:] / p_int[:-1])
    # local-to-lat-lon wind transform of CubedToLatLon (the reference derives a11 .. a22 from the unit vectors of the real
    # grid, util/pace/util/grid/generation.py; here a smooth, well-conditioned stand-in -- only used for timing)
    la, lo_ = m["lat_agrid"], m["lon_agrid"]
    m["a11"], m["a12"] = 1.0 + 0.05 * np.cos(la), 0.1 * np.sin(lo_) * np.cos(la)
    m["a21"], m["a22"] = -0.1 * np.sin(la) * np.cos(lo_), 1.0 + 0.05 * np.sin(la) ** 2
    return m


def _padA(A):
    return np.pad(A, ((1, 0), (1, 0), (0, 0)), mode="edge")


def acoustic_state(metrics, n, nz=79, n_halo=3, dt=3.571):
    """dict of (N+7, N+7, nz+1) numpy fields: a balanced state + the d_sw / riem_solver3 side inputs."""
    npt = n + 2 * n_halo + 1
    lat = metrics["lat_agrid"][:, :, None]
    lon = metrics["lon_agrid"][:, :, None]
    ak, bk = metrics["ak"], metrics["bk"]
    ps = 1.0e5 * (1.0 - 0.01 * np.cos(2 * lat) + 0.002 * np.sin(3 * lon))
    pe = ak[None, None, :] + bk[None, None, :] * ps  # interfaces (npt, npt, nz+1)
    s = {}
    z3 = lambda: np.zeros((npt, npt, nz + 1))  # noqa: E731
    delp = z3()
    delp[:, :, :nz] = pe[:, :, 1:] - pe[:, :, :-1]
    delp[:, :, nz] = delp[:, :, nz - 1]
    peln = np.log(pe)
    pm = delp[:, :, :nz] / (peln[:, :, 1:] - peln[:, :, :-1])
    eta = pm / 1.0e5
    temp = 288.0 * eta ** (c.RDGAS * 0.005 / c.GRAV) + 4.0 * np.cos(2 * lat) * np.sin(np.pi * eta)
    temp = np.maximum(temp, 200.0)
    q_con = z3()
    q_con[:, :, :nz] = 1.0e-4 * np.exp(-((eta - 0.8) ** 2) / 0.02) * (1 + 0.5 * np.cos(4 * lon))
    cappa = z3() + c.KAPPA
    cappa[:, :, :nz] = c.KAPPA * (1.0 - 0.1 * q_con[:, :, :nz] / 1.0e-4 * 0.01)
    pt = z3() + 1.0
    pt[:, :, :nz] = temp / np.exp(c.KAPPA * np.log(pm))
    delz = z3() - 100.0
    delz[:, :, :nz] = -(c.RDGAS / c.GRAV) * temp * (peln[:, :, 1:] - peln[:, :, :-1])
    zs = 500.0 * (1 + np.cos(2 * lon[:, :, 0])) * np.cos(lat[:, :, 0]) ** 2
    zh = z3()
    zh[:, :, nz] = zs
    for k in range(nz - 1, -1, -1):
        zh[:, :, k] = zh[:, :, k + 1] - delz[:, :, k]
    jet = 35.0 * np.cos(lat) ** 2 * np.sin(np.pi * eta) ** 2
    wave = 3.0 * np.sin(5 * lon + 2 * lat) * np.sin(np.pi * eta)
    u, v, w = z3(), z3(), z3()
    u[:, :, :nz] = jet + wave
    v[:, :, :nz] = 2.0 * np.cos(4 * lon - lat) * np.sin(np.pi * eta) + 0.3 * wave
    w[:, :, :nz] = 0.05 * np.sin(3 * lon) * np.cos(2 * lat) * np.sin(np.pi * eta)
    s.update(delp=delp, pt=pt, delz=delz, zh=zh, u=u, v=v, w=w, q_con=q_con, cappa=cappa)
    # C-grid / A-grid winds as c_sw would leave them (simple averages of the D-grid winds)
    s["ua"], s["va"] = u.copy(), v.copy()
    uc, vc = u.copy(), v.copy()
    uc[1:, :, :] = 0.5 * (u[:-1] + u[1:])
    vc[:, 1:, :] = 0.5 * (v[:, :-1] + v[:, 1:])
    s["uc"], s["vc"] = uc, vc
    divgd = z3()
    divgd[1:, 1:, :nz] = ((u[1:, 1:, :nz] - u[:-1, 1:, :nz]) + (v[1:, 1:, :nz] - v[1:, :-1, :nz])) / metrics["dx"][1:, 1:, None]
    s["divgd"] = divgd
    for name in ("delpc", "mfx", "mfy", "cx", "cy", "crx", "cry", "xfx", "yfx", "heat_source", "diss_est", "ppe", "pk3", "pk", "peln"):
        s[name] = z3()
    s["delpc"] = delp.copy()
    s["pe"] = pe.copy()
    s["pk3"][:] = 1.0e40
    s["zs"] = zs
    s["ws"] = 0.01 * np.sin(2 * lon[:, :, 0])
    s["dt"] = dt
    return s
