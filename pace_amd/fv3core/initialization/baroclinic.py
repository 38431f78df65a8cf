"""The Jablonowski-Williamson baroclinic-wave initial state on the cubed sphere, for all six tiles of a (1, 1) layout
(reference: fv3core/pace/fv3core/initialization/baroclinic.py:436-539 `init_baroclinic_state` with
baroclinic_jablonowski_williamson.py; JRMS2006 = Jablonowski & Williamson, QJRMS 132, 2006; DCMIP2016 test-case document).

    tiles = pace_amd.util.gridgen.metrics.generate(n, nz)           # metrics + the unit vectors ee1, ee2, es1, ew2
    states = init_baroclinic_state(tiles, n, nz)                    # six dicts of (N + 7, N + 7, nz + 1) numpy arrays

What the reference does and this reproduces: a reference surface pressure of 1000 hPa everywhere (the mountain that balances
the surface wind is in phis); the zonal wind of JRMS2006 eq. (2) plus the Gaussian perturbation of eq. (10) evaluated at both
ends and at the midpoint of every D-grid face and projected on the face's unit vector, averaged 1 : 2 : 1; temperature
(eq. 6) and surface geopotential (eq. 7) as nine-point cell averages (centre 1/4, face midpoints 1/8, corners 1/16); the
non-hydrostatic thickness delz from the hydrostatic relation, w = 0; specific humidity of DCMIP2016 eq. (18) and the
virtual-temperature adjustment; phis and (u, v) exchanged across tiles.
"""
import math

import numpy as np

from ...util import constants as c
from ...util.gridgen.metrics import arc, to_lonlat, to_xyz, unit
from ...util.gridgen.positions import exchange_scalar, exchange_vector

U0 = 35.0                       # maximum zonal wind (m/s)
U_PERT = 1.0                    # amplitude of the perturbation (m/s)
PERT_CENTRE = (math.pi / 9.0, 2.0 * math.pi / 9.0)  # 20 E, 40 N
PERT_RADIUS = c.RADIUS / 10.0
ETA_0, ETA_TROPOPAUSE = 0.252, 0.2
T_0, DELTA_T, LAPSE_RATE = 288.0, 480000.0, 0.005
P_SURFACE = 1.0e5
O = 3


def _zonal_wind(eta_v, lon, lat):
    """eq. (2) + eq. (10): (.., nz) array"""
    u = U0 * np.cos(eta_v) ** 1.5 * np.sin(2.0 * lat[..., None]) ** 2.0
    r = c.RADIUS * arc(PERT_CENTRE[0], PERT_CENTRE[1], lon, lat)
    near = (r / PERT_RADIUS) ** 2.0 < 40.0
    bump = U_PERT * np.exp(-((r / PERT_RADIUS) ** 2.0))
    return u + np.where(near, bump, 0.0)[..., None]


def _project(u, lon, vec):
    """component of a purely zonal wind along the unit vector `vec`"""
    return u * (vec[..., 1] * np.cos(lon) - vec[..., 0] * np.sin(lon))[..., None]


def _lat_function_temperature(eta, eta_v, t_mean, lat):
    lat = lat[..., None]
    a = -2.0 * np.sin(lat) ** 6.0 * (np.cos(lat) ** 2.0 + 1.0 / 3.0) + 10.0 / 63.0
    b = (8.0 / 5.0) * np.cos(lat) ** 3.0 * (np.sin(lat) ** 2.0 + 2.0 / 3.0) - math.pi / 4.0
    return t_mean + 0.75 * (eta * math.pi * U0 / c.RDGAS) * np.sin(eta_v) * np.sqrt(np.cos(eta_v)) * (
        a * 2.0 * U0 * np.cos(eta_v) ** 1.5 + b * c.RADIUS * c.OMEGA)


def _lat_function_phis(lat):
    ev = (1.0 - ETA_0) * math.pi * 0.5
    uc = U0 * np.cos(ev) ** 1.5
    a = -2.0 * np.sin(lat) ** 6.0 * (np.cos(lat) ** 2.0 + 1.0 / 3.0) + 10.0 / 63.0
    b = (8.0 / 5.0) * np.cos(lat) ** 3.0 * (np.sin(lat) ** 2.0 + 2.0 / 3.0) - math.pi / 4.0
    return uc * (a * uc + b * c.RADIUS * c.OMEGA)


def _nine_point(fn, lon, lat, lat_centre):
    """cell average of a function of latitude: centre, the four face midpoints, the four corners (lon / lat: corner arrays
    of (nx + 1, ny + 1); lat_centre (nx, ny))"""
    p = to_xyz(lon, lat)
    south = to_lonlat(unit(p[:-1, :] + p[1:, :]))[1]     # (nx, ny + 1): midpoints of the faces j = const
    west = to_lonlat(unit(p[:, :-1] + p[:, 1:]))[1]      # (nx + 1, ny)
    return (0.25 * fn(lat_centre) + 0.125 * (fn(south[:, :-1]) + fn(west[1:, :]) + fn(south[:, 1:]) + fn(west[:-1, :]))
            + 0.0625 * (fn(lat[:-1, :-1]) + fn(lat[1:, :-1]) + fn(lat[1:, 1:]) + fn(lat[:-1, 1:])))


def init_baroclinic_state(grid_data, quantity_factory, adiabatic: bool, hydrostatic: bool, moist_phys: bool, comm=None):
    """The reference's entry point (baroclinic.py:436-539): the DycoreState of the tile `grid_data` describes.  The state of
    all six tiles is evaluated on the host (the cross-tile halos of phis, u, v come from the neighbours' values), this
    rank's tile is uploaded; `comm` is accepted for signature parity and not used."""
    from ...util import gridgen
    from .dycore_state import DycoreState, _FIELDS

    s = quantity_factory.sizer
    n, nz = s.nx, s.nz
    tile = getattr(grid_data, "_tile", None)
    if tile is None:
        raise ValueError("grid_data must come from GridData.new_from_metric_terms (it knows which tile it is)")
    host = baroclinic_state_six_tiles(gridgen.tiles(n, nz), n, nz, adiabatic, hydrostatic, moist_phys)[tile]
    return DycoreState.init_from_numpy_arrays({k: v for k, v in host.items() if k in _FIELDS}, quantity_factory)


def baroclinic_state_six_tiles(tiles, n, nz, adiabatic=False, hydrostatic=False, moist_phys=True):
    """six dicts name -> numpy array ((N + 7, N + 7, nz + 1) or (N + 7, N + 7))"""
    if hydrostatic:
        raise NotImplementedError("the hydrostatic initial state is not implemented")
    size, K = n + 7, nz + 1
    e = O + n
    ak, bk, ptop = tiles[0]["ak"], tiles[0]["bk"], float(tiles[0]["ptop"])
    eta = 0.5 * ((ak[:-1] + ak[1:]) / P_SURFACE + bk[:-1] + bk[1:])
    eta_v = (eta - ETA_0) * math.pi * 0.5
    t_mean = T_0 * eta ** (c.RDGAS * LAPSE_RATE / c.GRAV)
    strat = ETA_TROPOPAUSE > eta
    t_mean[strat] = t_mean[strat] + DELTA_T * (ETA_TROPOPAUSE - eta[strat]) ** 5.0
    states = []
    cs = slice(O, e)
    for g in tiles:
        z3 = lambda fill=0.0: np.full((size, size, K), fill)  # noqa: E731
        s = dict(u=z3(), v=z3(), w=z3(1.0e30), ua=z3(1.0e35), va=z3(1.0e35), uc=z3(1.0e30), vc=z3(1.0e30), delp=z3(1.0e30),
                 delz=z3(1.0e25), pt=z3(1.0), pe=z3(), peln=z3(), pk=z3(), pkz=z3(), qvapor=z3(), q_con=z3(), omga=z3(),
                 mfxd=z3(), mfyd=z3(), cxd=z3(), cyd=z3(), diss_estd=z3())
        for sx in (slice(0, O), slice(O + n, None)):
            for sy in (slice(0, O), slice(O + n, None)):
                s["delp"][sx, sy] = 0.0
        s["phis"] = np.full((size, size), 1.0e25)
        s["ps"] = np.full((size, size), P_SURFACE)
        # pressures of the compute domain: the reference surface pressure is uniform
        delp = (ak[1:] - ak[:-1])[None, None, :] + P_SURFACE * (bk[1:] - bk[:-1])[None, None, :] + np.zeros((n, n, 1))
        pe = np.zeros((n, n, K))
        pe[:, :, 0] = ptop
        for k in range(1, K):
            pe[:, :, k] = pe[:, :, k - 1] + delp[:, :, k - 1]
        peln = np.zeros((n, n, K))
        peln[:, :, 0] = math.log(ptop)
        peln[:, :, 1:] = np.log(pe[:, :, 1:])
        pk = np.zeros((n, n, K))
        pk[:, :, 0] = ptop ** c.KAPPA
        pk[:, :, 1:] = np.exp(c.KAPPA * np.log(pe[:, :, 1:]))
        s["delp"][cs, cs, :nz], s["delp"][cs, cs, nz] = delp, 0.0
        s["pe"][cs, cs], s["peln"][cs, cs], s["pk"][cs, cs] = pe, peln, pk
        s["ps"][cs, cs] = pe[:, :, -1]

        lon, lat = g["lon"], g["lat"]
        # winds: v on the x-interfaces (i = 0 .. N, j = 0 .. N - 1), u on the y-interfaces
        cw = slice(O, e + 1)
        p = to_xyz(lon, lat)
        lo, la = lon[cw, cw], lat[cw, cw]
        mlon, mlat = to_lonlat(unit(p[cw, O:e] + p[cw, O + 1:e + 1]))            # midpoints of the west faces
        v1 = _project(_zonal_wind(eta_v, lo[:, 1:], la[:, 1:]), lo[:, 1:], g["ee2"][cw, O + 1:e + 1])
        v3 = _project(_zonal_wind(eta_v, lo[:, :-1], la[:, :-1]), lo[:, :-1], g["ee2"][cw, O:e])
        v2 = _project(_zonal_wind(eta_v, mlon, mlat), mlon, g["ew2"][cw, O:e])
        s["v"][cw, cs, :nz] = 0.25 * (v1 + 2.0 * v2 + v3)
        mlon, mlat = to_lonlat(unit(p[O:e, cw] + p[O + 1:e + 1, cw]))            # midpoints of the south faces
        u1 = _project(_zonal_wind(eta_v, lo[:-1, :], la[:-1, :]), lo[:-1, :], g["ee1"][O:e, cw])
        u3 = _project(_zonal_wind(eta_v, lo[1:, :], la[1:, :]), lo[1:, :], g["ee1"][O + 1:e + 1, cw])
        u2 = _project(_zonal_wind(eta_v, mlon, mlat), mlon, g["es1"][O:e, cw])
        s["u"][cs, cw, :nz] = 0.25 * (u1 + 2.0 * u2 + u3)

        latc = g["lat_agrid"][cs, cs]
        temp = _nine_point(lambda x: _lat_function_temperature(eta, eta_v, t_mean, x), lo, la, latc)
        s["phis"][cs, cs] = _nine_point(_lat_function_phis, lo, la, latc)
        s["w"][cs, cs] = 0.0
        dlnp = peln[:, :, 1:] - peln[:, :, :-1]
        if not adiabatic:
            ptmp = delp / dlnp - P_SURFACE
            qv = 0.021 * np.exp(-((latc[:, :, None] / PERT_CENTRE[1]) ** 4.0)) * np.exp(-((ptmp / 34000.0) ** 2.0))
            s["qvapor"][cs, cs, :nz] = qv
            temp = temp / (1.0 + c.ZVIR * qv)
        s["pt"][cs, cs, :nz] = temp
        delz = c.RDG * temp * dlnp
        s["delz"][cs, cs, :nz] = delz
        if moist_phys:
            s["pkz"][cs, cs, :nz] = np.exp(c.KAPPA * np.log(c.RDG * delp * temp * (1.0 + c.ZVIR * s["qvapor"][cs, cs, :nz]) / delz))
        else:
            s["pkz"][cs, cs, :nz] = np.exp(c.KAPPA * np.log(c.RDG * delp * temp / delz))
        states.append(s)
    exchange_scalar([s["phis"] for s in states], n)
    exchange_vector([s["u"] for s in states], [s["v"] for s in states], n, "d")
    return states
