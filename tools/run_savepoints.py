"""Drop-in runner for the reference's savepoint data (the Fortran-serialised `<Name>-In.nc` / `<Name>-Out.nc` pairs of
Makefile.data_download:2-5, version 8.1.3) against the HIP operators -- the route by which parity can be pinned by the reference's
OWN golden data the day it is on a disk.  It never runs reference Python and imports nothing from oracle/.

    python tools/run_savepoints.py <dir> [--only D_SW,Riem_Solver3] [--device cuda] [--lib path.so] [--metrics grid.npz] [--rank-tile]

For every savepoint name it knows and finds in <dir>, and every (savepoint, rank) entry of the pair:
  * the input arrays are placed in (N + 7, N + 7, npz + 1) storages exactly as TranslateFortranData2Py.make_storage_data_input_vars
    does (stencils/pace/stencils/testing/translate.py:167-213: start indices from the variable's info dictionary, else from the
    array's shape, grid.py:425-436; `kaxis`, `serialname`),
  * the operator class of pace_amd with the reference's signature is called (the table below mirrors each Translate class's
    in_vars / parameters / out_vars and index windows, fv3core/tests/savepoint/translate/translate_*.py),
  * the outputs are sliced as slice_output does (translate.py:215-252, windows from grid.py:288-382) and compared with the `-Out`
    arrays in the reference's metric (util/pace/util/testing/comparison.py:6-68) against the class's max_error / the overrides
    of fv3core/tests/savepoint/translate/overrides/standard.yaml.

Files: NetCDF-4 through h5py / netCDF4 / xarray (whichever imports), NetCDF-3 classic through scipy.io (`nccopy -k classic`
converts), or `.npz` pairs with the same variable names and the same leading (savepoint, rank) axes (what
tests/test_savepoint_runner.py writes).  Grid: the metric terms of the rank's tile from pace_amd's own generator (checked against
the reference's MetricTerms to 3e-12, tests/test_gridgen.py) unless --metrics gives an .npz in pace_amd's names.
"""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HALO = 3


class SGrid:
    """The index vocabulary of stencils/pace/stencils/testing/grid.py for one rank of a (1, 1) layout."""

    def __init__(self, n, npz):
        self.n, self.npz = n, npz
        self.isd = self.jsd = 0
        self.is_ = self.js = HALO
        self.ie = self.je = HALO + n - 1
        self.ied = self.jed = 2 * HALO + n - 1
        self.nic = self.njc = n
        self.nid = self.njd = n + 2 * HALO

    def default_domain_dict(self):
        return {"istart": self.isd, "iend": self.ied, "jstart": self.jsd, "jend": self.jed, "kstart": 0, "kend": self.npz - 1}

    def compute_dict(self):
        return {"istart": self.is_, "iend": self.ie, "jstart": self.js, "jend": self.je, "kstart": 0, "kend": self.npz - 1}

    def _with(self, base, **kw):
        d = dict(base)
        d.update(kw)
        return d

    def default_dict_buffer_2d(self):
        return self._with(self.default_domain_dict(), iend=self.ied + 1, jend=self.jed + 1)

    def compute_dict_buffer_2d(self):
        return self._with(self.compute_dict(), iend=self.ie + 1, jend=self.je + 1)

    def default_buffer_k_dict(self):
        return self._with(self.default_domain_dict(), kend=self.npz)

    def compute_buffer_k_dict(self):
        return self._with(self.compute_dict(), kend=self.npz)

    def x3d_domain_dict(self):
        return self._with(self.default_domain_dict(), iend=self.ied + 1)

    def y3d_domain_dict(self):
        return self._with(self.default_domain_dict(), jend=self.jed + 1)

    def x3d_compute_dict(self):
        return self._with(self.default_domain_dict(), istart=self.is_, iend=self.ie + 1, jstart=self.js, jend=self.je)

    def y3d_compute_dict(self):
        return self._with(self.default_domain_dict(), istart=self.is_, iend=self.ie, jstart=self.js, jend=self.je + 1)

    def x3d_compute_domain_y_dict(self):
        return self._with(self.default_domain_dict(), istart=self.is_, iend=self.ie + 1)

    def y3d_compute_domain_x_dict(self):
        return self._with(self.default_domain_dict(), jstart=self.js, jend=self.je + 1)

    def horizontal_starts_from_shape(self, shape):  # grid.py:425-436
        n = self.n
        if tuple(shape[0:2]) in [(n, n), (n + 1, n), (n, n + 1), (n + 1, n + 1)]:
            return self.is_, self.js
        if tuple(shape[0:2]) == (n + 2, n + 2):
            return self.is_ - 1, self.js - 1
        return 0, 0


def place(array, info, grid):
    """make_storage_data_input_vars for one variable: a zero storage of (N + 7, N + 7, npz + 1) (K-only: npz + 1) with the
    serialised array at its start indices."""
    a = np.asarray(array, dtype=np.float64)
    if "kaxis" in info:
        a = np.moveaxis(a, info["kaxis"], 2)
    a = np.squeeze(a)
    if a.ndim == 0:
        return float(a)
    if a.ndim == 1:
        out = np.zeros(grid.npz + 1)
        out[: a.shape[0]] = a
        return out
    i0, j0 = grid.horizontal_starts_from_shape(a.shape)
    i0, j0, k0 = int(info.get("istart", i0)), int(info.get("jstart", j0)), int(info.get("kstart", 0))
    full = (grid.nid + 1, grid.njd + 1, grid.npz + 1)
    if a.ndim == 2:
        out = np.zeros(full[:2])
        out[i0:i0 + a.shape[0], j0:j0 + a.shape[1]] = a
        return out
    out = np.zeros(full)
    out[i0:i0 + a.shape[0], j0:j0 + a.shape[1], k0:k0 + a.shape[2]] = a
    return out


def slice_out(storage, info, grid):
    """slice_output for one variable (translate.py:215-252)."""
    ds = grid.default_domain_dict()
    ds.update({k: v for k, v in info.items() if k in ds})
    a = np.asarray(storage)
    if a.ndim == 3:
        a = a[ds["istart"]:ds["iend"] + 1, ds["jstart"]:ds["jend"] + 1, ds["kstart"]:ds["kend"] + 1]
    elif a.ndim == 2:
        a = a[ds["istart"]:ds["iend"] + 1, ds["jstart"]:ds["jend"] + 1]
    a = np.squeeze(a)
    if "kaxis" in info:
        a = np.moveaxis(a, 2, info["kaxis"])
    return a


def compare(a, b, near_zero=0.0):
    """util/pace/util/testing/comparison.py:6-68: 2 |a - b| / (|a| + |b|); NaN == NaN passes."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    denom = np.abs(a) + np.abs(b)
    with np.errstate(all="ignore"):
        rel = np.where(denom > 0, 2 * np.abs(a - b) / np.where(denom > 0, denom, 1.0), 0.0)
    rel[both_nan] = 0.0
    rel[np.isnan(rel)] = np.inf
    if near_zero > 0:
        rel[(np.abs(a) < near_zero) & (np.abs(b) < near_zero)] = 0.0
    return float(rel.max()) if rel.size else 0.0


# ---- the files --------------------------------------------------------------------------------------------------------------
def _open_nc(path):
    """{variable: array with leading (savepoint, rank) axes} of a NetCDF file, with whatever reader this Python has."""
    errors = []
    try:
        import h5py  # NetCDF-4 files are HDF5 files

        with h5py.File(path, "r") as f:
            return {k: np.asarray(v) for k, v in f.items() if hasattr(v, "shape") and v.ndim >= 2}
    except ImportError as e:
        errors.append(str(e))
    try:
        import netCDF4

        with netCDF4.Dataset(path) as f:
            return {k: np.asarray(v[:]) for k, v in f.variables.items()}
    except ImportError as e:
        errors.append(str(e))
    try:
        import xarray

        ds = xarray.open_dataset(path)
        return {k: ds[k].values for k in ds.data_vars}
    except ImportError as e:
        errors.append(str(e))
    try:
        from scipy.io import netcdf_file

        with netcdf_file(path, "r", mmap=False) as f:  # NetCDF-3 classic only
            return {k: np.array(v[:]) for k, v in f.variables.items()}
    except Exception as e:  # noqa: BLE001
        errors.append(f"scipy.io.netcdf_file: {e}")
    raise RuntimeError(f"cannot read {path}: this Python has neither h5py, netCDF4 nor xarray (NetCDF-4 = HDF5), and scipy reads NetCDF-3 "
                       f"classic only -- install h5py, or convert with `nccopy -k classic`, or save the pair as .npz.  ({'; '.join(errors)})")


def read_pair(directory, name):
    out = []
    for kind in ("In", "Out"):
        base = os.path.join(directory, f"{name}-{kind}")
        if os.path.exists(base + ".npz"):
            out.append(dict(np.load(base + ".npz")))
        elif os.path.exists(base + ".nc"):
            out.append(_open_nc(base + ".nc"))
        else:
            return None
    return out


# ---- the savepoints (one entry per Translate class; `info` dictionaries as there) -----------------------------------------------
def _named(info, serialname):
    d = dict(info)
    d["serialname"] = serialname
    return d


class Spec:
    def __init__(self, in_vars, parameters, out_vars, max_error, run, near_zero=0.0, ignore_near_zero=None, index_parameters=()):
        self.in_vars, self.parameters, self.out_vars, self.max_error, self.run = in_vars, parameters, out_vars, max_error, run
        self.near_zero, self.ignore_near_zero, self.index_parameters = near_zero, ignore_near_zero or {}, tuple(index_parameters)
        # (Every variable is compared over the window its Translate class names -- the whole storage where that is `{}`.  Rounds 1-5
        # compared some on the compute domain only: the halo state the reference's in-place corner fills and work-domain writes leave
        # behind was not replayed.  Since round 6 it is: D_SW, C_SW, D2A2C_Vect, DivergenceDamping.)


def spec_d_sw(g):  # translate_d_sw.py:12-65
    iv = {"uc": g.x3d_domain_dict(), "vc": g.y3d_domain_dict(), "w": {}, "delpc": {}, "delp": {}, "u": g.y3d_domain_dict(),
          "v": g.x3d_domain_dict(), "xfx": g.x3d_compute_domain_y_dict(), "crx": g.x3d_compute_domain_y_dict(),
          "yfx": g.y3d_compute_domain_x_dict(), "cry": g.y3d_compute_domain_x_dict(), "mfx": g.x3d_compute_dict(),
          "mfy": g.y3d_compute_dict(), "cx": g.x3d_compute_domain_y_dict(), "cy": g.y3d_compute_domain_x_dict(), "heat_source": {},
          "diss_est": {}, "q_con": {}, "pt": {}, "ua": {}, "va": {}, "zh": {}, "divgd": g.default_dict_buffer_2d()}
    iv = {k: _named(v, k + "d") for k, v in iv.items()}
    ov = {k: v for k, v in iv.items() if k != "zh"}

    def run(env, f, p):
        from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
        from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist
        from pace_amd.tile import DSW_ARGS

        cfg = DGridShallowWaterLagrangianDynamicsConfig(**env.namelist.get("d_sw", {}))
        op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, get_column_namelist(cfg, env.qf),
                                                nested=False, stretched_grid=False, config=cfg)
        op(*[f[k] for k in DSW_ARGS], p["dt"])
        return f

    return Spec(iv, ["dt"], ov, 3.2e-10, run)


def spec_riem_solver3(g):  # translate_riem_solver3.py:8-82; bound: overrides/standard.yaml:49-61
    iv = {"cappa": {}, "zs": {}, "w": {}, "delz": {}, "q_con": {}, "delp": {}, "pt": {}, "zh": {},
          "p": {"istart": g.is_ - 1, "jstart": g.js - 1, "kaxis": 1, "serialname": "pe"}, "ppe": {}, "pk3": {}, "pk": {},
          "log_p_interface": {"istart": g.is_, "jstart": g.js, "kaxis": 1, "serialname": "peln"},
          "ws": {"istart": g.is_, "jstart": g.js, "serialname": "wsd"}}
    ov = {"zh": {"kend": g.npz}, "w": {},
          "p": {"istart": g.is_ - 1, "iend": g.ie + 1, "jstart": g.js - 1, "jend": g.je + 1, "kend": g.npz, "kaxis": 1, "serialname": "pe"},
          "log_p_interface": {"istart": g.is_, "iend": g.ie, "jstart": g.js, "jend": g.je, "kend": g.npz, "kaxis": 1, "serialname": "peln"},
          "ppe": {"kend": g.npz}, "delz": {}, "pk": g.compute_buffer_k_dict(), "pk3": g.default_buffer_k_dict()}

    def run(env, f, p):
        from pace_amd.fv3core import RiemannConfig
        from pace_amd.fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver

        op = NonhydrostaticVerticalSolver(env.stencil_factory, env.qf, RiemannConfig(**env.namelist.get("riemann", {})))
        op(bool(p["last_call"]), p["dt"], f["cappa"], p["ptop"], f["zs"], f["ws"], f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"], f["p"],
           f["ppe"], f["pk3"], f["pk"], f["log_p_interface"], f["w"])
        return f

    return Spec(iv, ["dt", "ptop", "last_call"], ov, 5e-6, run)


def spec_fvtp2d(g):  # translate_fvtp2d.py:8-70
    iv = {"q": {}, "mass": {}, "damp_c": {}, "nord": {"serialname": "nord_column"}, "crx": {"istart": g.is_}, "cry": {"jstart": g.js},
          "x_area_flux": {"istart": g.is_, "serialname": "xfx"}, "y_area_flux": {"jstart": g.js, "serialname": "yfx"},
          "x_mass_flux": _named(g.x3d_compute_dict(), "mfx"), "y_mass_flux": _named(g.y3d_compute_dict(), "mfy")}
    ov = {"q": {}, "q_x_flux": _named(g.x3d_compute_dict(), "fx"), "q_y_flux": _named(g.y3d_compute_dict(), "fy")}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport

        nord, damp = env.kq(np.asarray(f.pop("nord_host"))), env.kq(np.asarray(f.pop("damp_c_host")))
        f["q_x_flux"], f["q_y_flux"] = env.q3(), env.q3()
        op = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, grid_type=0, hord=int(p["hord"]), nord=nord,
                                   damp_c=damp)
        op(f["q"], f["crx"], f["cry"], f["x_area_flux"], f["y_area_flux"], f["q_x_flux"], f["q_y_flux"], x_mass_flux=f.get("x_mass_flux"),
           y_mass_flux=f.get("y_mass_flux"), mass=f.get("mass"))
        return f

    return Spec(iv, ["hord"], ov, 1e-14, run)


def spec_riem_solver_c(g):  # translate_riem_solver_c.py:8-34
    iv = {k: {} for k in ("cappa", "hs", "w3", "ptc", "q_con", "delpc", "gz", "pef", "ws")}
    ov = {"pef": {"kend": g.npz}, "gz": {"kend": g.npz}}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.riem_solver_c import NonhydrostaticVerticalSolverCGrid

        op = NonhydrostaticVerticalSolverCGrid(env.stencil_factory, env.qf, p_fac=env.namelist.get("riemann", {}).get("p_fac", 0.05))
        op(p["dt2"], f["cappa"], p["ptop"], f["hs"], f["ws"], f["ptc"], f["q_con"], f["delpc"], f["gz"], f["pef"], f["w3"])
        return f

    return Spec(iv, ["dt2", "ptop"], ov, 5e-14, run)


def spec_nh_p_grad(g):  # translate_nh_p_grad.py:8-46
    iv = {k: {} for k in ("u", "v", "pp", "gz", "pk3", "delp")}
    ov = {"u": g.y3d_domain_dict(), "v": g.x3d_domain_dict(), "pp": {"kend": g.npz + 1}, "gz": {"kend": g.npz + 1}, "pk3": {"kend": g.npz + 1},
          "delp": {}}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.nh_p_grad import NonHydrostaticPressureGradient

        op = NonHydrostaticPressureGradient(env.stencil_factory, env.qf, env.grid_data, 0)
        op(f["u"], f["v"], f["pp"], f["gz"], f["pk3"], f["delp"], p["dt"], p["ptop"], p["akap"])
        return f

    return Spec(iv, ["dt", "ptop", "akap"], ov, 5e-10, run)


def spec_fxadv(g):  # translate_fxadv.py:8-72 (uc_contra / vc_contra are compared on the compute domain + 2: `_subset`)
    ut, vt = _named(g.x3d_domain_dict(), "ut"), _named(g.y3d_domain_dict(), "vt")
    iv = {"uc": {}, "vc": {}, "uc_contra": ut, "vc_contra": vt, "x_area_flux": _named(g.x3d_compute_domain_y_dict(), "xfx_adv"),
          "crx": _named(g.x3d_compute_domain_y_dict(), "crx_adv"), "y_area_flux": _named(g.y3d_compute_domain_x_dict(), "yfx_adv"),
          "cry": _named(g.y3d_compute_domain_x_dict(), "cry_adv")}
    sub = {"istart": g.is_ - 2, "iend": g.ie + 2, "jstart": g.js - 2, "jend": g.je + 2}
    ov = {"uc_contra": dict(ut, **dict(sub, iend=g.ie + 3)), "vc_contra": dict(vt, **dict(sub, jend=g.je + 3)),
          "x_area_flux": iv["x_area_flux"], "crx": iv["crx"], "y_area_flux": iv["y_area_flux"], "cry": iv["cry"]}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.fxadv import FiniteVolumeFluxPrep

        FiniteVolumeFluxPrep(env.stencil_factory, env.grid_data, quantity_factory=env.qf)(
            f["uc"], f["vc"], f["crx"], f["cry"], f["x_area_flux"], f["y_area_flux"], f["uc_contra"], f["vc_contra"], p["dt"])
        return f

    return Spec(iv, ["dt"], ov, 1e-14, run)


def spec_c_sw(g):  # translate_c_sw.py:73-113
    iv = {"delp": {}, "pt": {}, "u": {"jend": g.jed + 1}, "v": {"iend": g.ied + 1}, "w": {}, "uc": {"iend": g.ied + 1}, "vc": {"jend": g.jed + 1},
          "ua": {}, "va": {}, "ut": {}, "vt": {}, "omga": {}, "divgd": {"iend": g.ied + 1, "jend": g.jed + 1}}
    iv = {k: _named(v, k + "d") for k, v in iv.items()}
    ov = dict(iv)
    ov["delpcd"], ov["ptcd"] = {}, {}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.c_sw import CGridShallowWaterDynamics

        nord = int(env.namelist.get("d_sw", {}).get("nord", env.namelist.get("nord", 3)))
        op = CGridShallowWaterDynamics(env.stencil_factory, env.qf, env.grid_data, nested=False, grid_type=0, nord=nord)
        f["delpcd"], f["ptcd"] = op(f["delp"], f["pt"], f["u"], f["v"], f["w"], f["uc"], f["vc"], f["ua"], f["va"], f["ut"], f["vt"], f["divgd"],
                                    f["omga"], p["dt2"])
        return f

    return Spec(iv, ["dt2"], ov, 2e-10, run)


def spec_updatedzc(g):  # translate_updatedzc.py:10-70 (gz and ws are compared on the compute domain: `_subset`)
    iv = {"zs": {}, "ut": {"serialname": "utc"}, "vt": {"serialname": "vtc"}, "gz": {}, "ws": {}}
    cd = g.compute_dict()
    ov = {"gz": dict(cd, kend=g.npz), "ws": {k: v for k, v in cd.items() if k[0] != "k"}}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.updatedzc import UpdateGeopotentialHeightOnCGrid

        op = UpdateGeopotentialHeightOnCGrid(env.stencil_factory, env.qf, area=env.grid_data.area, dp_ref=env.grid_data.dp_ref,
                                             grid_data=env.grid_data)
        op(f["zs"], f["ut"], f["vt"], f["gz"], f["ws"], p["dt2"])
        return f

    return Spec(iv, ["dt2"], ov, 1e-14, run)


def spec_updatedzd(g):  # translate_updatedzd.py:12-87 (height compared on the compute domain, `ws` = wsd there; near-zero values ignored)
    iv = {"surface_height": {"serialname": "zs"}, "height": {"kend": g.npz + 1, "serialname": "zh"},
          "courant_number_x": _named(g.x3d_compute_domain_y_dict(), "crx"), "courant_number_y": _named(g.y3d_compute_domain_x_dict(), "cry"),
          "x_area_flux": _named(g.x3d_compute_domain_y_dict(), "xfx"), "y_area_flux": _named(g.y3d_compute_domain_x_dict(), "yfx"),
          "ws": _named({k: v for k, v in g.compute_dict().items() if k[0] != "k"}, "wsd")}
    ov = {k: iv[k] for k in ("courant_number_x", "courant_number_y", "x_area_flux", "y_area_flux", "ws")}
    ov["height"] = dict(g.compute_dict(), kend=g.npz, serialname="zh")

    def run(env, f, p):
        from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
        from pace_amd.fv3core.stencils.d_sw import get_column_namelist
        from pace_amd.fv3core.stencils.updatedzd import UpdateHeightOnDGrid

        cfg = DGridShallowWaterLagrangianDynamicsConfig(**env.namelist.get("d_sw", {}))
        op = UpdateHeightOnDGrid(env.stencil_factory, env.qf, env.damping, env.grid_data, 0, int(env.namelist.get("hord_tm", cfg.hord_tm)),
                                 column_namelist=get_column_namelist(cfg, env.qf))
        op(f["surface_height"], f["height"], f["courant_number_x"], f["courant_number_y"], f["x_area_flux"], f["y_area_flux"], f["ws"], p["dt"])
        return f

    return Spec(iv, ["dt"], ov, 1e-14, run, ignore_near_zero={"height": 1e-30, "ws": 1e-30})


def spec_d2a2c_vect(g):  # translate_d2a2c_vect.py:8-47
    iv = {k: {} for k in ("uc", "vc", "u", "v", "ua", "va", "utc", "vtc")}
    ov = {"uc": g.x3d_domain_dict(), "vc": g.y3d_domain_dict(), "ua": {}, "va": {}, "utc": {}, "vtc": {}}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.d2a2c_vect import DGrid2AGrid2CGridVectors

        DGrid2AGrid2CGridVectors(env.stencil_factory, env.qf, env.grid_data, False, 0, True)(
            f["uc"], f["vc"], f["u"], f["v"], f["ua"], f["va"], f["utc"], f["vtc"])
        return f

    return Spec(iv, [], ov, 2e-10, run)


def spec_divergence_damping(g):  # translate_divergencedamping.py:11-76 (ke on the B-grid domain, delpc)
    iv = {"u": {}, "v": {}, "va": {}, "damped_rel_vort_bgrid": {"serialname": "vort"}, "ua": {}, "divg_d": {}, "vc": {}, "uc": {}, "delpc": {},
          "ke": {}, "rel_vort_agrid": {"serialname": "wk"}, "nord_col": {}, "d2_bg": {}}
    ov = {"ke": {"iend": g.ied + 1, "jend": g.jed + 1}, "delpc": {}}

    def run(env, f, p):
        from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
        from pace_amd.fv3core.stencils.divergence_damping import DivergenceDamping

        cfg = DGridShallowWaterLagrangianDynamicsConfig(**env.namelist.get("d_sw", {}))
        nord_col, d2_bg = env.kq(np.asarray(f.pop("nord_col_host"))), env.kq(np.asarray(f.pop("d2_bg_host")))
        op = DivergenceDamping(env.stencil_factory, env.qf, env.grid_data, env.damping, False, False, cfg.dddmp, cfg.d4_bg, cfg.nord, 0, nord_col,
                               d2_bg)
        op(f["u"], f["v"], f["va"], f["damped_rel_vort_bgrid"], f["ua"], f["divg_d"], f["vc"], f["uc"], f["delpc"], f["ke"], f["rel_vort_agrid"],
           p["dt"])
        return f

    return Spec(iv, ["dt"], ov, 1.4e-10, run)


def spec_delnflux(g):  # translate_delnflux.py:8-47 (DelnFlux_2: the same without `mass`)
    iv = {"q": {}, "fx": g.x3d_compute_dict(), "fy": g.y3d_compute_dict(), "damp_c": {}, "nord_column": {}, "mass": {}}
    ov = {"fx": g.x3d_compute_dict(), "fy": g.y3d_compute_dict()}

    def run(env, f, p):
        from pace_amd.fv3core.stencils.delnflux import DelnFlux

        nord, damp = env.kq(np.asarray(f.pop("nord_column_host"))), env.kq(np.asarray(f.pop("damp_c_host")))
        op = DelnFlux(env.stencil_factory, env.qf, env.damping, env.grid_data.rarea, nord, damp, grid_data=env.grid_data)
        op(f["q"], f["fx"], f["fy"], mass=f.get("mass"))
        return f

    return Spec(iv, [], ov, 1e-14, run)


def _spec_ppm(g, axis):  # translate_xppm.py:8-58 / translate_yppm.py:8-60 (XPPM: rows jfirst .. jlast; YPPM: columns ifirst .. ilast)
    first, last = ("jfirst", "jlast") if axis == 0 else ("ifirst", "ilast")
    if axis == 0:
        iv = {"q": {"serialname": "qx", "jstart": first}, "c": {"serialname": "cx", "istart": g.is_}}
        ov = {"xflux": {"istart": g.is_, "iend": g.ie + 1, "jstart": first, "jend": last}}
    else:
        iv = {"q": {"istart": first}, "c": {"jstart": g.js}}
        ov = {"flux": {"istart": first, "iend": last, "jstart": g.js, "jend": g.je + 1}}
    out = "xflux" if axis == 0 else "flux"
    order = "iord" if axis == 0 else "jord"

    def run(env, f, p):
        from pace_amd.fv3core.stencils.xppm import XPiecewiseParabolic
        from pace_amd.fv3core.stencils.yppm import YPiecewiseParabolic

        a, b = int(p[first]), int(p[last])
        f[out] = env.q3()
        if axis == 0:
            op = XPiecewiseParabolic(env.stencil_factory, env.grid_data.dxa, 0, int(p[order]), (g.is_, a, 0), (g.n + 1, b - a + 1, g.npz))
        else:
            op = YPiecewiseParabolic(env.stencil_factory, env.grid_data.dya, 0, int(p[order]), (a, g.js, 0), (b - a + 1, g.n + 1, g.npz))
        op(f["q"], f["c"], f[out])
        return f

    return Spec(iv, [order, first, last], ov, 1e-14, run, index_parameters=(first, last))


def spec_xppm(g):
    return _spec_ppm(g, 0)


def spec_yppm(g):
    return _spec_ppm(g, 1)


SAVEPOINTS = {"D_SW": spec_d_sw, "Riem_Solver3": spec_riem_solver3, "FvTp2d": spec_fvtp2d, "Riem_Solver_C": spec_riem_solver_c,
              "NH_P_Grad": spec_nh_p_grad, "FxAdv": spec_fxadv, "C_SW": spec_c_sw, "UpdateDzC": spec_updatedzc, "UpdateDzD": spec_updatedzd,
              "D2A2C_Vect": spec_d2a2c_vect, "DivergenceDamping": spec_divergence_damping, "DelnFlux": spec_delnflux,
              "XPPM": spec_xppm, "YPPM": spec_yppm}
# ---- DynCore (translate_dyncore.py:13-200): the WHOLE AcousticDynamics call, a ParallelTranslate: all ranks of a savepoint run together
# and exchange their halos (six ranks = the six tiles of the cubed sphere, one tile per rank: pace_amd.util.run_tiles) -------------
def dyncore_vars(g):
    cd = g.compute_dict()
    col = {"istart": g.is_, "iend": g.ie, "jstart": g.js, "jend": g.je, "kend": g.npz + 1}
    iv = {"cappa": {}, "u": g.y3d_domain_dict(), "v": g.x3d_domain_dict(), "w": {}, "delz": {}, "delp": {}, "pt": {},
          "pe": {"istart": g.is_ - 1, "iend": g.ie + 1, "jstart": g.js - 1, "jend": g.je + 1, "kend": g.npz + 1, "kaxis": 1},
          "pk": dict(col), "phis": {"kstart": 0, "kend": 0}, "wsd": {k: v for k, v in cd.items() if k[0] != "k"}, "omga": {}, "ua": {}, "va": {},
          "uc": g.x3d_domain_dict(), "vc": g.y3d_domain_dict(), "mfxd": g.x3d_compute_dict(), "mfyd": g.y3d_compute_dict(),
          "cxd": g.x3d_compute_domain_y_dict(), "cyd": g.y3d_compute_domain_x_dict(), "pkz": cd, "peln": dict(col, kaxis=1), "q_con": {},
          "ak": {}, "bk": {}, "diss_estd": {}}
    ov = {k: v for k, v in iv.items() if k not in ("ak", "bk", "phis", "pkz")}
    return iv, ov


def run_dyncore(pair, args, lib):
    """Returns (ok, bound, worst) like run_one.  The namelist's `acoustic` entry: keyword arguments of AcousticDynamicsConfig
    (n_split, k_split, nord, d_con, rf_fast, rf_cutoff, tau, p_fac, hord_tm, delt_max, ...; its d_grid_shallow_water / riemann parts from
    the `d_sw` / `riemann` entries); default: baroclinic_c12.yaml.  Metrics: `--metrics` may hold `{rank}`."""
    from pace_amd.fv3core import AcousticDynamicsConfig, DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.dyn_core import AcousticDynamics
    from pace_amd.tile import Env
    from pace_amd.util import CubedSphereCommunicator, run_tiles

    ins, outs = pair
    some = next(v for v in ins.values() if np.asarray(v).ndim >= 5)
    n_sp, n_rank = some.shape[0], some.shape[1]
    if n_rank != 6:
        raise SystemExit(f"DynCore: {n_rank} ranks in the data; this runner drives one tile per rank (6)")
    w0 = max(np.squeeze(np.asarray(v)[0, 0]).shape[0] for v in ins.values() if np.squeeze(np.asarray(v)[0, 0]).ndim == 3)
    n = w0 - 2 * HALO - (1 if (w0 - 2 * HALO) % 2 else 0)
    npz = int(np.squeeze(np.asarray(ins["delp"])[0, 0]).shape[2])
    grid = SGrid(n, npz)
    iv, ov = dyncore_vars(grid)
    nl = getattr(args, "namelist", None) or {}
    ac = dict(nl.get("acoustic", {}))
    dsw = DGridShallowWaterLagrangianDynamicsConfig(**nl.get("d_sw", {}))
    cfg = AcousticDynamicsConfig(**{**dict(n_split=6, k_split=1, nord=dsw.nord, d_con=dsw.d_con, rf_fast=True, rf_cutoff=3000.0, tau=10.0, p_fac=0.05,
                                           hord_tm=dsw.hord_tm, delt_max=0.002), **ac},
                                 d_grid_shallow_water=dsw, riemann=RiemannConfig(**{**dict(p_fac=0.05), **nl.get("riemann", {})}))
    # ignore_near_zero_errors: wsd 1e-18 (translate_dyncore.py:121) and, for the baroclinic test case, the work-field leftovers uc / vc
    # and the accumulators whose entries on a tile's symmetry line are rounding residue (overrides/baroclinic.yaml:13-20); the
    # namelist's `near_zero` entry replaces them for other data
    near_zero = dict(nl.get("near_zero", {"wsd": 1e-18, "uc": 1e-13, "vc": 1e-13, "mfxd": 1e-3, "mfyd": 1e-3, "cxd": 1e-3, "cyd": 1e-3}))
    worst = {}
    for sp in range(n_sp):
        def program(comm, sp=sp):
            rank = comm.Get_rank()
            one_in = {k: np.asarray(v)[sp, rank] for k, v in ins.items() if np.asarray(v).ndim >= 2}
            mpath = args.metrics.format(rank=rank) if args.metrics else None
            metrics = dict(metrics_for(n, npz, rank, mpath))
            placed = {}
            for var, info in iv.items():
                if var in one_in:
                    placed[var] = place(one_in[var], info, grid)
            for k in ("ak", "bk"):  # the vertical coordinate travels with the data (translate_dyncore.py:126-140)
                if k in placed:
                    metrics[k] = placed[k]
            metrics["ptop"] = float(np.squeeze(one_in["ptop"]))
            env = Env(lib, args.device, metrics, n, npz)
            cube = CubedSphereCommunicator(comm, device=args.device, lib=lib)
            state = DycoreState.init_from_numpy_arrays({k: v for k, v in placed.items() if k not in ("cappa", "wsd", "ak", "bk")}, env.qf)
            wsd = env.q2(placed.get("wsd"))
            dyn = AcousticDynamics(cube, env.stencil_factory, env.qf, env.grid_data, env.damping, 0, False, False, cfg, state.phis, wsd, state)
            dyn.cappa.set(placed["cappa"])
            dyn(state, timestep=float(np.squeeze(one_in["mdt"])), n_map=int(np.squeeze(one_in["n_map"])))
            if args.device != "cpu":
                import torch

                torch.cuda.synchronize()
            res = {k: getattr(state, k).numpy() for k in ov if k not in ("cappa", "wsd")}
            res["cappa"], res["wsd"] = dyn.cappa.numpy(), wsd.numpy()
            return res

        results = run_tiles(6, program)
        for rank, res in enumerate(results):
            for var, info in ov.items():
                if var not in outs:
                    continue
                ref = np.squeeze(np.asarray(outs[var])[sp, rank])
                got = slice_out(res[var], info, grid)
                worst[var] = max(worst.get(var, 0.0), compare(ref, got, near_zero=near_zero.get(var, 0.0)))
    bound = 2e-6  # translate_dyncore.py:120
    return all(e <= bound for e in worst.values()), bound, worst


def metrics_for(n, npz, tile, path=None):
    if path:
        return dict(np.load(path))
    from pace_amd.util import gridgen

    return gridgen.tiles(n, npz)[tile]


def run_one(name, pair, args, lib):
    from pace_amd.tile import Env

    ins, outs = pair
    some = next(v for v in ins.values() if np.asarray(v).ndim >= 5)
    n_sp, n_rank = some.shape[0], some.shape[1]
    worst = {}
    for sp in range(n_sp):
        for rank in range(n_rank):
            one_in = {k: np.asarray(v)[sp, rank] for k, v in ins.items() if np.asarray(v).ndim >= 2}
            one_out = {k: np.asarray(v)[sp, rank] for k, v in outs.items() if np.asarray(v).ndim >= 2}
            # the grid's size: the widest 3-D variable spans N + 6 (or, staggered, N + 7) points; the model's levels are the
            # shortest third axis among the variables of that width (interface variables have npz + 1)
            wide = [np.squeeze(v).shape for v in one_in.values() if np.squeeze(v).ndim == 3]
            w0 = max(s_[0] for s_ in wide)
            n = w0 - 2 * HALO - (1 if (w0 - 2 * HALO) % 2 else 0)
            npz = min(s_[2] for s_ in wide if s_[0] >= n + 2 * HALO)
            grid = SGrid(n, npz)
            spec = SAVEPOINTS[name](grid)
            env = Env(lib, args.device, metrics_for(n, npz, rank % 6 if args.rank_tile else 0, args.metrics), n, npz)
            env.namelist = getattr(args, "namelist", None) or {}
            fields, params = {}, {}
            for pname in spec.parameters:
                params[pname] = float(np.squeeze(one_in[pname]))
            # window entries given by NAME are parameters of the savepoint holding Fortran indices of the model's global grid
            # (TranslateXPPM.jvars: + fpy_model_index_offset = 2, then global_to_local: the same number on a 1 x 1 layout)
            for pname in spec.index_parameters:
                params[pname] = int(params[pname]) + 2

            def resolved(info):
                return {k: (int(params[v]) if isinstance(v, str) and k != "serialname" else v) for k, v in info.items()}

            for var, info in spec.in_vars.items():
                sname = info.get("serialname", var)
                if sname not in one_in:
                    continue
                st = place(one_in[sname], resolved(info), grid)
                if isinstance(st, float):
                    params[var] = st
                elif st.ndim == 1:
                    fields[var + "_host"] = st
                elif st.ndim == 2:
                    fields[var] = env.q2(st)
                else:
                    fields[var] = env.q3(st)
            res = spec.run(env, fields, params)
            if args.device != "cpu":
                import torch

                torch.cuda.synchronize()
            for var, info in spec.out_vars.items():
                info = resolved(info)
                sname = info.get("serialname", var)
                if sname not in one_out:
                    continue
                got = slice_out(res[var].numpy(), info, grid)
                ref = np.squeeze(one_out[sname])
                nz_ = spec.ignore_near_zero.get(var, spec.near_zero)
                worst[var] = max(worst.get(var, 0.0), compare(ref, got, near_zero=nz_))
    ok = all(e <= spec.max_error for e in worst.values())
    return ok, spec.max_error, worst


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("directory")
    ap.add_argument("--only", default="")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--lib", default=None, help="library to load (default: pace_amd/libpace_hip.so; the CPU emulation build for --device cpu)")
    ap.add_argument("--metrics", default=None, help=".npz of metric terms in pace_amd's names (default: generated for the rank's tile)")
    ap.add_argument("--namelist", default=None,
                    help="YAML / JSON file with the configuration the data were made with, as keyword arguments of pace_amd's config classes: "
                         "{d_sw: {hord_dp: 6, ...}, riemann: {p_fac: 0.05, ...}} (default: the classes' defaults = baroclinic_c12.yaml)")
    ap.add_argument("--rank-tile", action="store_true", help="rank r is tile r of the cubed sphere (6-rank data); default: every rank is tile 0")
    args = ap.parse_args()
    from pace_amd import _lib

    if args.namelist:
        import yaml

        args.namelist = yaml.safe_load(open(args.namelist))
    lib = _lib.Library(args.lib) if args.lib else _lib.load()
    names = [s for s in args.only.split(",") if s] or sorted(SAVEPOINTS) + ["DynCore"]
    found = {os.path.basename(p).rsplit("-In.", 1)[0] for p in glob.glob(os.path.join(args.directory, "*-In.*"))}
    failed = 0
    for name in names:
        if name == "DynCore" and name in found:
            ok, bound, worst = run_dyncore(read_pair(args.directory, name), args, lib)
            print(f"DynCore: {'PASS' if ok else 'FAIL'}  bound {bound:g} (six ranks, halo updates included)")
            print("   the reference's windows: " + "  ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())))
            failed += 0 if ok else 1
            continue
        if name not in SAVEPOINTS:
            print(f"{name}: not in this runner's table")
            failed += 1
            continue
        if name not in found:
            print(f"{name}: no {name}-In.nc / .npz in {args.directory}")
            continue
        ok, bound, worst = run_one(name, read_pair(args.directory, name), args, lib)
        print(f"{name}: {'PASS' if ok else 'FAIL'}  bound {bound:g}")
        print("   the reference's windows: " + "  ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())))
        failed += 0 if ok else 1
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
