"""ORACLE (test infrastructure) -- ctypes front end of oracle/omp/dsw_riem3.cpp, the C++ / OpenMP restatement of d_sw and
riem_solver3 at the reference's stencil granularity (bench.py's CPU baseline).  Same call signatures as oracle.dgrid_sw.d_sw /
oracle.vertical.riem_solver3 on the oracle's [i, j, k] arrays; the library works on [k][j][i] copies (i fastest).

Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this.  Build: `make -C oracle/omp`."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle_omp.so")

METRICS = ["cosa_u", "cosa_v", "rsin_u", "rsin_v", "sin_sg1", "sin_sg2", "sin_sg3", "sin_sg4", "rdxa", "rdya", "dx", "dy", "dxa",
           "dya", "rdx", "rdy", "area", "rarea", "del6_u", "del6_v", "cosa", "rsina", "fC_agrid", "rsin2", "cosa_s", "divg_u",
           "divg_v", "rarea_c", "sina_u", "sina_v", "dxc", "dyc", "lon", "lat", "lon_agrid", "lat_agrid"]
EDGES = ["edge_w", "edge_e", "edge_s", "edge_n"]
COLUMN = ["nord", "nord_v", "nord_w", "nord_t", "damp_vt", "damp_w", "damp_t", "d2_divg", "d_con", "ke_bg"]
DSW_FIELDS = ["delpc", "delp", "pt", "u", "v", "w", "uc", "vc", "ua", "va", "divgd", "mfx", "mfy", "cx", "cy", "crx", "cry", "xfx",
              "yfx", "q_con", "heat_source", "diss_est"]  # the reference's argument order without zh (unused by d_sw)
RIEM_FIELDS = ["cappa", "zs", "ws", "delz", "q_con", "delp", "pt", "zh", "pe", "ppe", "pk3", "pk", "peln", "w"]

_lib = None


def build():
    subprocess.run(["make", "-C", os.path.join(HERE, "omp")], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        try:
            _lib = C.CDLL(LIB)
        except OSError:  # (a stale or foreign build: make it again on this host)
            os.remove(LIB)
            build()
            _lib = C.CDLL(LIB)
        _lib.omp_port_threads.restype = C.c_int
    return _lib


def threads():
    return int(load().omp_port_threads())


def set_threads(n):
    load().omp_port_set_threads(int(n))


def to_kji(a):
    """[i, j, k] (or [i, j]) -> contiguous [k][j][i] (or [j][i])."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)


def from_kji(dst, src):
    dst[...] = src.T


def team_copy(dst, src):
    """dst[...] = src through the library's team, level by level (3-D [k][j][i] arrays): the pages of a level are first touched
    by the thread that works on it.  Anything else: a plain copy."""
    if os.environ.get("OMP_PORT_MASTER_TOUCH"):  # (as rounds 1-5 did: the calling thread copies; bench.py measures both)
        dst[...] = src
    elif dst.ndim == 3 and dst.flags.c_contiguous and src.flags.c_contiguous and dst.shape == src.shape:
        load().omp_port_copy_levels(C.c_void_p(dst.ctypes.data), C.c_void_p(src.ctypes.data), C.c_long(dst.shape[0]), C.c_long(dst.shape[1] * dst.shape[2]))
    else:
        dst[...] = src


def _fresh(src):
    """Untouched arrays of the shapes of `src`, filled by the team."""
    out = [np.empty_like(a) for a in src]
    for w, s in zip(out, src):
        team_copy(w, s)
    return out


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])


class Tile:
    """The grid of one tile in the library's layout (metric terms transposed once)."""

    def __init__(self, g):
        self.n, self.nk = int(g.n), int(g.nk)
        self.m = []
        for name in METRICS:
            a = getattr(g, name, None)
            self.m.append(to_kji(a) if a is not None else np.zeros((g.nj, g.ni)))
        for name in EDGES:
            a = np.asarray(getattr(g, name), dtype=np.float64)
            a = a[0, :] if (a.ndim == 2 and name in ("edge_w", "edge_e")) else a.reshape(-1)
            self.m.append(np.ascontiguousarray(a))
        self.mp = _ptrs(self.m)
        self.sc = (C.c_double * 2)(float(g.da_min), float(g.da_min_c))

    def dims(self, nk=None):
        return (C.c_int * 2)(self.n, self.nk if nk is None else nk)


def _col(col, nk):
    arrs = [np.ascontiguousarray(np.asarray(col[k], dtype=np.float64)[: nk + 1]) for k in COLUMN]
    return arrs, _ptrs(arrs)


class DswCall:
    """d_sw prepared for repeated timing: operands transposed once; run() is the library call alone."""

    def __init__(self, tile, col, cfg, ut, vt, fields, dt):
        self.tile, self.dt = tile, float(dt)
        self.colarr, self.colp = _col(col, tile.nk)
        self.icfg = (C.c_int * 6)(cfg["hord_dp"], cfg["hord_tm"], cfg["hord_vt"], cfg["hord_mt"], cfg["nord"], int(cfg.get("do_skeb", False)))
        self.dcfg = (C.c_double * 3)(cfg["dddmp"], cfg["d4_bg"], cfg["d_con"])
        self.src = [to_kji(ut), to_kji(vt)] + [to_kji(fields[k]) for k in DSW_FIELDS]
        self.work = _fresh(self.src)
        self.fp = _ptrs(self.work)

    def reset(self):
        for w, s in zip(self.work, self.src):
            team_copy(w, s)

    def run(self):
        load().omp_port_d_sw(self.tile.dims(), self.tile.mp, self.tile.sc, self.colp, self.icfg, self.dcfg, self.fp, C.c_double(self.dt))

    def outputs(self):
        return {"uc_contra": self.work[0].T, "vc_contra": self.work[1].T, **{k: self.work[2 + n].T for n, k in enumerate(DSW_FIELDS)}}


def d_sw(g, col, cfg, st, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source,
         diss_est, dt):
    """oracle.dgrid_sw.d_sw's signature; updates the arrays in place."""
    loc = dict(delpc=delpc, delp=delp, pt=pt, u=u, v=v, w=w, uc=uc, vc=vc, ua=ua, va=va, divgd=divgd, mfx=mfx, mfy=mfy, cx=cx, cy=cy,
               crx=crx, cry=cry, xfx=xfx, yfx=yfx, q_con=q_con, heat_source=heat_source, diss_est=diss_est)
    call = DswCall(Tile(g), col, cfg, st.uc_contra, st.vc_contra, loc, dt)
    call.run()
    out = call.outputs()
    from_kji(st.uc_contra, call.work[0])
    from_kji(st.vc_contra, call.work[1])
    for k in DSW_FIELDS:
        loc[k][...] = out[k]


class RiemCall:
    def __init__(self, tile, fields, last_call, dt, ptop, p_fac, beta=0.0, use_logp=False):
        self.tile = tile
        self.args = (int(bool(last_call)), C.c_double(dt), C.c_double(ptop), C.c_double(p_fac), C.c_double(beta), int(bool(use_logp)))
        self.src = [to_kji(fields[k]) for k in RIEM_FIELDS]
        self.work = _fresh(self.src)
        self.fp = _ptrs(self.work)

    def reset(self):
        for w, s in zip(self.work, self.src):
            team_copy(w, s)

    def run(self):
        load().omp_port_riem3(self.tile.dims(), self.tile.mp, self.tile.sc, self.fp, *self.args)

    def outputs(self):
        return {k: self.work[n].T for n, k in enumerate(RIEM_FIELDS)}


def riem_solver3(g, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w, p_fac, beta=0.0,
                 use_logp=False):
    """oracle.vertical.riem_solver3's signature; updates the arrays in place."""
    loc = dict(cappa=cappa, zs=zs, ws=ws, delz=delz, q_con=q_con, delp=delp, pt=pt, zh=zh, pe=pe, ppe=ppe, pk3=pk3, pk=pk, peln=peln, w=w)
    call = RiemCall(Tile(g), loc, last_call, dt, ptop, p_fac, beta, use_logp)
    call.run()
    out = call.outputs()
    for k in ("delz", "zh", "pe", "ppe", "pk3", "pk", "peln", "w"):
        loc[k][...] = out[k]


def fxadv(g, uc, vc, crx, cry, xfx, yfx, ut, vt, dt):
    t = Tile(g)
    arrs = [to_kji(a) for a in (uc, vc, crx, cry, xfx, yfx, ut, vt)]
    load().omp_port_fxadv(t.dims(), t.mp, t.sc, _ptrs(arrs), C.c_double(dt))
    for dst, src in zip((crx, cry, xfx, yfx, ut, vt), arrs[2:]):
        from_kji(dst, src)


def fvtp2d(g, q, crx, cry, xfx, yfx, fx, fy, hord, x_mass_flux=None, y_mass_flux=None, mass=None, nord_k=None, damp_c_k=None):
    t = Tile(g)
    arrs = [to_kji(a) if a is not None else None for a in (q, crx, cry, xfx, yfx, fx, fy, x_mass_flux, y_mass_flux, mass)]
    nk = t.nk
    nord = np.ascontiguousarray(np.asarray(nord_k, dtype=np.float64)[:nk]) if nord_k is not None else None
    damp = np.ascontiguousarray(np.asarray(damp_c_k, dtype=np.float64)[:nk]) if damp_c_k is not None else None
    load().omp_port_fvtp2d(t.dims(), t.mp, t.sc, _ptrs(arrs), int(hord), C.c_void_p(nord.ctypes.data if nord is not None else None),
                           C.c_void_p(damp.ctypes.data if damp is not None else None))
    from_kji(q, arrs[0])
    from_kji(fx, arrs[5])
    from_kji(fy, arrs[6])
