"""Per-stencil device implementations behind ``FrozenStencil`` (the L3 boundary, dsl/pace/dsl/stencil.py:395-434).

A ``FrozenStencil`` here is a lookup of the stencil's identity in this registry, not a compilation.  Registered are the
definition functions of the acoustic path that are launched as stencils of their own by the reference's classes and tests
and whose semantics one device entry point provides:

* the dyn_core one-liners (``zero_data``, ``gz_from_surface_height_and_thicknesses``,
  ``interface_pressure_from_toa_pressure_and_thickness``, ``compute_geopotential``, ``p_grad_c_stencil``), ``copy_defn``,
  ``edge_pe``, ``apply_diffusive_heating`` -- each kernel implements the ONE launch window the reference constructs the stencil
  with (dyn_core.py:480-587); a FrozenStencil built with another origin / domain is refused at construction, not mis-run;
* ``compute_x_flux`` / ``compute_y_flux`` (xppm.py:269-287, yppm.py) -- any origin / domain, order from the ``mord`` external.

Identities are matched on ``module.name`` with the reference's module path, on pace_amd's own mirror of it, and on the bare
function name (the translate tests define some stencils in their own modules).
"""
import ctypes as C

from .stencil import register_stencil


def _ptr(x):
    t = x.data if hasattr(x, "dims") else x
    return t.data_ptr()


def _geom(st):
    from ..util.grid import geom_struct

    f = st._factory
    if getattr(f, "_geom_cache", None) is None:
        qf = getattr(f, "quantity_factory", None)
        if qf is None:
            raise RuntimeError("this StencilFactory was built without a quantity_factory: device stencils need the field layout "
                               "(StencilFactory(config, grid_indexing, quantity_factory=...))")
        f._geom_cache = geom_struct(qf)
    return f._geom_cache


def _stream(st):
    import torch

    f = st._factory
    if f.quantity_factory.device.type == "cpu":
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _window(st, origin, domain, what):
    """The device kernel covers exactly the window the reference launches this stencil on."""
    o = st.origin if isinstance(st.origin, tuple) else st.origin.get("_all_")
    if tuple(o) != tuple(origin) or tuple(st.domain) != tuple(domain):
        raise NotImplementedError(f"{st.name}: the device kernel implements the launch window of {what} "
                                  f"(origin {tuple(origin)}, domain {tuple(domain)}); requested origin {tuple(o)}, domain {st.domain}")


def _both(mod, name):
    return (f"pace.fv3core.stencils.{mod}.{name}", f"pace_amd.fv3core.stencils.{mod}.{name}")


class _Impl:
    """An implementation = a window check at construction + the call."""

    def __init__(self, check, call):
        self.check, self.call = check, call

    def __call__(self, st, **named):
        self.call(st, **named)


def _register(paths, check, call):
    impl = _Impl(check, call)
    for p in paths:
        register_stencil(p)(impl)
    register_stencil(paths[0].rsplit(".", 1)[1])(impl)
    return impl


def _gi(st):
    return st._factory.grid_indexing


# ---- dyn_core.py:51-171 ----
_register(_both("dyn_core", "zero_data"),
          lambda st: _window(st, _gi(st).origin_full(), _gi(st).domain_full(), "dyn_core.py:539-543"),
          lambda st, mfxd, mfyd, cxd, cyd, heat_source, diss_estd, first_timestep: st._factory.lib.call(
              "pace_zero_data", C.byref(_geom(st)), _ptr(mfxd), _ptr(mfyd), _ptr(cxd), _ptr(cyd), _ptr(heat_source), _ptr(diss_estd),
              int(bool(first_timestep)), _stream(st)))
_register(_both("dyn_core", "gz_from_surface_height_and_thicknesses"),
          lambda st: _window(st, _gi(st).origin_compute(), _gi(st).domain_compute(add=(0, 0, 1)), "dyn_core.py:510-514"),
          lambda st, zs, delz, gz: st._factory.lib.call("pace_gz_from_surface_height_and_thicknesses", C.byref(_geom(st)), _ptr(zs),
                                                        _ptr(delz), _ptr(gz), _stream(st)))
_register(_both("dyn_core", "interface_pressure_from_toa_pressure_and_thickness"),
          lambda st: _window(st, _gi(st).origin_compute(add=(-1, -1, 0)), _gi(st).domain_compute(add=(2, 2, 0)), "dyn_core.py:515-521"),
          lambda st, delp, pem, ptop: st._factory.lib.call("pace_interface_pressure_from_toa_pressure_and_thickness",
                                                           C.byref(_geom(st)), _ptr(delp), _ptr(pem), float(ptop), _stream(st)))


def _geo_window(st):
    from ..util.constants import X_DIM, Y_DIM, Z_INTERFACE_DIM

    o, d = _gi(st).get_origin_domain([X_DIM, Y_DIM, Z_INTERFACE_DIM], halos=(2, 2))
    _window(st, o, d, "dyn_core.py:480-487")


_register(_both("dyn_core", "compute_geopotential"), _geo_window,
          lambda st, zh, gz: st._factory.lib.call("pace_compute_geopotential", C.byref(_geom(st)), _ptr(zh), _ptr(gz), _stream(st)))


def _p_grad_c_check(st):
    _window(st, _gi(st).origin_compute(), _gi(st).domain_compute(add=(1, 1, 0)), "dyn_core.py:523-528")
    if st.externals.get("hydrostatic", False):
        raise NotImplementedError("p_grad_c_stencil: hydrostatic = True is not implemented")


def _p_grad_c_call(st, rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2):
    gd = getattr(rdxc, "_grid_data", None)
    if gd is None:
        raise ValueError("p_grad_c_stencil: pass grid_data.rdxc / rdyc of a pace_amd GridData")
    st._factory.lib.call("pace_p_grad_c", C.byref(_geom(st)), C.byref(gd.c_struct()), _ptr(uc), _ptr(vc), _ptr(delpc), _ptr(pkc),
                         _ptr(gz), float(dt2), _stream(st))


_register(_both("dyn_core", "p_grad_c_stencil"), _p_grad_c_check, _p_grad_c_call)

# ---- basic_operations.py:7, pe_halo.py:6-34, temperature_adjust.py:8-43 ----
_register(_both("basic_operations", "copy_defn"),
          lambda st: _window(st, _gi(st).origin_full(), _gi(st).domain_full(add=(0, 0, 1)), "dyn_core.py:583-587"),
          lambda st, q_in, q_out: st._factory.lib.call("pace_copy", C.byref(_geom(st)), _ptr(q_in), _ptr(q_out), _stream(st)))
_register(_both("pe_halo", "edge_pe"),
          lambda st: _window(st, _gi(st).origin_full(), _gi(st).domain_full(add=(0, 0, 1)), "dyn_core.py:548-554"),
          lambda st, pe, delp, ptop: st._factory.lib.call("pace_edge_pe", C.byref(_geom(st)), _ptr(pe), _ptr(delp), float(ptop),
                                                          _stream(st)))


def _heating_check(st):
    o, d = _gi(st).origin_compute(), _gi(st).domain_compute()
    got_o = st.origin if isinstance(st.origin, tuple) else st.origin.get("_all_")
    if tuple(got_o) != tuple(o) or tuple(st.domain[:2]) != tuple(d[:2]) or not (0 < st.domain[2] <= d[2]):
        raise NotImplementedError(f"{st.name}: the device kernel covers the compute domain, top nk levels (dyn_core.py:575-581)")


_register(_both("temperature_adjust", "apply_diffusive_heating"), _heating_check,
          lambda st, delp, delz, cappa, heat_source, pt, delt_time_factor: st._factory.lib.call(
              "pace_apply_diffusive_heating", C.byref(_geom(st)), _ptr(delp), _ptr(delz), _ptr(cappa), _ptr(heat_source), _ptr(pt),
              float(delt_time_factor), int(st.domain[2]), _stream(st)))


# ---- xppm.py:269-287 / yppm.py: any window ----
def _ppm(axis):
    def check(st):
        if abs(int(st.externals.get("mord", st.externals.get("iord", 0)))) not in (5, 6, 8):
            raise NotImplementedError(f"{st.name}: mord must be 5, 6 or 8")

    def call(st, q, courant, dxa, xflux):
        gd = getattr(dxa, "_grid_data", None)
        if gd is None:
            raise ValueError(f"{st.name}: pass grid_data.dxa / dya of a pace_amd GridData")
        o = st.origin if isinstance(st.origin, tuple) else st.origin.get("_all_")
        st._factory.lib.call("pace_ppm", C.byref(_geom(st)), C.byref(gd.c_struct()), axis, int(st.externals.get("mord", 6)), _ptr(q),
                             _ptr(courant), _ptr(xflux), int(o[0]), int(o[1]), int(o[2]), *[int(x) for x in st.domain], _stream(st))

    return check, call


_register(_both("xppm", "compute_x_flux"), *_ppm(0))
_register(_both("yppm", "compute_y_flux"), *_ppm(1))


# ---- definitions the reference's Translate tests launch as stencils of their own (csrc/k_stencils.hip, pace_stencil) ----
# ids of include/pace_hip.h
ST_FLUX_CAPACITOR, ST_HEAT_DISS, ST_APPLY_FLUXES, ST_UBKE, ST_VBKE = 1, 2, 3, 4, 5
ST_COPY_CORNERS_X, ST_COPY_CORNERS_Y, ST_FILL_CORNERS_BGRID_X, ST_FILL_CORNERS_BGRID_Y = 6, 7, 8, 9
ST_FILL_CORNERS_DGRID, ST_FILL_CORNERS_2CELLS_X, ST_FILL_CORNERS_2CELLS_Y = 10, 11, 12
ST_XTP_U, ST_YTP_V, ST_MOIST_PT_LAST_STEP, ST_MOIST_PKZ, ST_MOIST_PT = 13, 14, 15, 16, 17


def _origin(st):
    """The launch origin.  A per-field origin dictionary (stencil.py:436-470) is honoured only if every entry equals "_all_": these
    kernels take ONE window for all their fields, and a different origin for one of them would silently be ignored."""
    if isinstance(st.origin, tuple):
        return tuple(st.origin)
    base = tuple(st.origin.get("_all_"))
    for name, o in st.origin.items():
        if name != "_all_" and tuple(o)[: len(base)] != base[: len(tuple(o))]:
            raise NotImplementedError(f"{st.name}: per-field origin {name}={tuple(o)} differs from _all_={base}: the device stencil takes one window")
    return base


def _call_stencil(st, ident, fields, scalars=(), metrics_from=None):
    """Any launch window: the kernels honour origin / domain (and refuse windows whose offset reads would leave the storage)."""
    from .. import _lib

    gd = None
    for m in (metrics_from or ()):
        gd = getattr(m, "_grid_data", None)
        if gd is not None:
            break
    if metrics_from is not None and gd is None:
        raise ValueError(f"{st.name}: pass the metric arguments (rarea, cosa, ...) of a pace_amd GridData")
    met = gd.c_struct() if gd is not None else _lib.Metrics()  # (definitions without metric arguments never read it)
    ptrs = (C.c_void_p * len(fields))(*[_ptr(f) for f in fields])
    sc = (C.c_double * max(1, len(scalars)))(*[float(x) for x in scalars])
    o, d = _origin(st), tuple(st.domain)
    st._factory.lib.call("pace_stencil", C.byref(_geom(st)), C.byref(met), int(ident), ptrs, len(fields), sc, len(scalars),
                         (C.c_int * 3)(*[int(x) for x in o]), (C.c_int * 3)(*[int(x) for x in d]), _stream(st))


def _no_check(st):
    return None


_register(_both("d_sw", "flux_capacitor"), _no_check,
          lambda st, cx, cy, xflux, yflux, crx_adv, cry_adv, fx, fy: _call_stencil(st, ST_FLUX_CAPACITOR,
                                                                                   [cx, cy, xflux, yflux, crx_adv, cry_adv, fx, fy]))
_register(_both("d_sw", "heat_diss"), _no_check,
          lambda st, fx2, fy2, w, rarea, heat_source, diss_est, dw, damp_w, ke_bg, dt: _call_stencil(
              st, ST_HEAT_DISS, [fx2, fy2, w, heat_source, diss_est, dw, damp_w, ke_bg], [dt], metrics_from=[rarea]))
_register(_both("d_sw", "apply_fluxes"), _no_check,
          lambda st, q, delp, gx, gy, rarea: _call_stencil(st, ST_APPLY_FLUXES, [q, delp, gx, gy], metrics_from=[rarea]))
# (the reference defines ubke / vbke in its test module, tests/savepoint/translate/translate_d_sw.py:67-81,118-133: matched on
# the bare name)
_register(("pace_amd.fv3core.stencils.d_sw.ubke",), _no_check,
          lambda st, uc, vc, cosa, rsina, ut, ub, dt4, dt5: _call_stencil(st, ST_UBKE, [uc, vc, ut, ub], [dt5], metrics_from=[cosa, rsina]))
_register(("pace_amd.fv3core.stencils.d_sw.vbke",), _no_check,
          lambda st, vc, uc, cosa, rsina, vt, vb, dt4, dt5: _call_stencil(st, ST_VBKE, [uc, vc, vt, vb], [dt5], metrics_from=[cosa, rsina]))


def _corner_paths(name):
    return (f"pace.stencils.corners.{name}", f"pace_amd.stencils.corners.{name}")


_register(_corner_paths("copy_corners_x_stencil_defn"), _no_check, lambda st, q_in, q_out: _call_stencil(st, ST_COPY_CORNERS_X, [q_in, q_out]))
_register(_corner_paths("copy_corners_y_stencil_defn"), _no_check, lambda st, q_in, q_out: _call_stencil(st, ST_COPY_CORNERS_Y, [q_in, q_out]))
_register(_corner_paths("fill_corners_bgrid_x_defn"), _no_check, lambda st, q_in, q_out: _call_stencil(st, ST_FILL_CORNERS_BGRID_X, [q_in, q_out]))
_register(_corner_paths("fill_corners_bgrid_y_defn"), _no_check, lambda st, q_in, q_out: _call_stencil(st, ST_FILL_CORNERS_BGRID_Y, [q_in, q_out]))
_register(_corner_paths("fill_corners_dgrid_defn"), _no_check,
          lambda st, x_in, x_out, y_in, y_out, mysign: _call_stencil(st, ST_FILL_CORNERS_DGRID, [x_in, x_out, y_in, y_out], [mysign]))
_register(_corner_paths("fill_corners_2cells_x_stencil"), _no_check,
          lambda st, q_out, q_in: _call_stencil(st, ST_FILL_CORNERS_2CELLS_X, [q_out, q_in]))
_register(_corner_paths("fill_corners_2cells_y_stencil"), _no_check,
          lambda st, q_out, q_in: _call_stencil(st, ST_FILL_CORNERS_2CELLS_Y, [q_out, q_in]))


# xtp_u_stencil_defn / ytp_v_stencil_defn: defined in the reference's test modules (tests/savepoint/translate/translate_xtp_u.py:13-23,
# translate_ytp_v.py), matched on the bare name; externals iord (5, 6, 7: the orders d_sw runs hord_mt with)
def _xtp_check(st):
    if int(st.externals.get("iord", 0)) not in (5, 6, 7):
        raise NotImplementedError(f"{st.name}: iord must be 5, 6 or 7")


_register(("pace_amd.fv3core.stencils.xtp_u.xtp_u_stencil_defn",), _xtp_check,
          lambda st, ub_contra_times_dt, u, updated_u, dx, dxa, rdx: _call_stencil(
              st, ST_XTP_U, [ub_contra_times_dt, u, updated_u], [int(st.externals["iord"])], metrics_from=[dx, dxa, rdx]))
_register(("pace_amd.fv3core.stencils.ytp_v.ytp_v_stencil_defn",), _xtp_check,
          lambda st, vb_contra_times_dt, v, updated_v, dy, dya, rdy: _call_stencil(
              st, ST_YTP_V, [vb_contra_times_dt, v, updated_v], [int(st.externals["iord"])], metrics_from=[dy, dya, rdy]))


# moist_pt_last_step (moist_cv.py:84-118; translate_last_step.py:16 launches it as a stencil of its own)
_register(_both("moist_cv", "moist_pt_last_step"), _no_check,
          lambda st, qvapor, qliquid, qrain, qsnow, qice, qgraupel, gz, pt, pkz, dtmp, r_vir: _call_stencil(
              st, ST_MOIST_PT_LAST_STEP, [qvapor, qliquid, qrain, qsnow, qice, qgraupel, gz, pt, pkz], [dtmp, r_vir]))


# moist_pkz (moist_cv.py:130-172; translate_moistcvpluspkz_2d.py:19 builds it on a one-row window)
_register(_both("moist_cv", "moist_pkz"), _no_check,
          lambda st, qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, gz, cvm, pkz, pt, cappa, delp, delz, r_vir: _call_stencil(
              st, ST_MOIST_PKZ, [qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, gz, cvm, pkz, pt, cappa, delp, delz], [r_vir]))
# moist_pt: the stencil the reference's test module wraps around moist_cv.moist_pt_func (translate_moistcvpluspt_2d.py:9-50); the
# same definition is offered as pace_amd.fv3core.stencils.moist_cv.moist_pt
_register(("translate_moistcvpluspt_2d.moist_pt", "pace_amd.fv3core.stencils.moist_cv.moist_pt"), _no_check,
          lambda st, qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, pt, cappa, delp, delz, r_vir: _call_stencil(
              st, ST_MOIST_PT, [qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, pt, cappa, delp, delz], [r_vir]))
