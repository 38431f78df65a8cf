"""LagrangianToEulerian -- Fortran Lagrangian_to_Eulerian (reference: fv3core/pace/fv3core/stencils/remapping.py:286-695).

The saturation adjustment (saturation_adjustment.py, 1100 lines of microphysics) is not part of pace_amd: a configuration
with ``do_sat_adj = True`` is refused at construction."""
from typing import Dict

from ...util.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM
from .._config import RemappingConfig
from ._common import Operator, check_layout, dptr
from .fillz import pointer_table
from .map_single import MapSingle
from .mapn_tracer import MapNTracer

# TODO(reference): "Should this be set here or in global_constants?" (remapping.py:38-39)
CONSV_MIN = 0.001
_WATER = ("qvapor", "qliquid", "qrain", "qsnow", "qice", "qgraupel")


class LagrangianToEulerian(Operator):
    """Remap the deformed Lagrangian surfaces onto the reference, or "Eulerian", coordinate levels."""

    def __init__(self, stencil_factory, quantity_factory, config: RemappingConfig, area_64, nq, pfull, tracers: Dict[str, object],
                 checkpointer=None):
        super().__init__(stencil_factory, quantity_factory)
        if config.kord_tm >= 0:
            raise NotImplementedError("map ppm, untested mode where kord_tm >= 0")
        if config.hydrostatic:
            raise NotImplementedError("Hydrostatic is not implemented")
        if config.do_sat_adj:
            raise NotImplementedError("the saturation adjustment (do_sat_adj) is outside the scope of pace_amd")
        self._t_min = 184.0
        self._nq = nq
        qf = quantity_factory
        self._pe1 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        self._pe2 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        self._pe3 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        self._dp2 = qf.zeros([X_DIM, Y_DIM, Z_DIM], units="Pa")
        self._pn2 = qf.zeros([X_DIM, Y_DIM, Z_DIM], units="Pa")
        self._pe0 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        self._kord_tm = abs(config.kord_tm)
        self._kord_wz = config.kord_wz
        self._kord_mt = config.kord_mt
        self._do_sat_adjust = config.do_sat_adj
        sf = stencil_factory
        self._map_single_pt = MapSingle(sf, qf, self._kord_tm, 1, dims=[X_DIM, Y_DIM, Z_DIM])
        self._mapn_tracer = MapNTracer(sf, qf, abs(config.kord_tr), nq, fill=config.fill, tracers=tracers)
        self._map_single_w = MapSingle(sf, qf, self._kord_wz, -2, dims=[X_DIM, Y_DIM, Z_DIM])
        self._map_single_delz = MapSingle(sf, qf, self._kord_wz, 1, dims=[X_DIM, Y_DIM, Z_DIM])
        self._map_single_u = MapSingle(sf, qf, self._kord_mt, -1, dims=[X_DIM, Y_INTERFACE_DIM, Z_DIM])
        self._map_single_v = MapSingle(sf, qf, self._kord_mt, -1, dims=[X_INTERFACE_DIM, Y_DIM, Z_DIM])

    def __call__(self, tracers, pt, delp, delz, peln, u, v, w, cappa, q_con, q_cld, pkz, pk, pe, hs, ps, wsd, ak, bk, dp1,
                 ptop: float, akap: float, zvir: float, last_step: bool, consv_te: float, mdt: float):
        """Same arguments as the reference (remapping.py:485-563): tracers, pt, delp, delz, peln, u, v, w, cappa (inout);
        q_con, pk, ps (out); pkz, pe (inout); wsd, ak, bk (in); q_cld, hs, dp1, mdt are only used by the saturation
        adjustment."""
        check_layout(self._geom, pt, delp, delz, peln, u, v, w, cappa, q_con, pkz, pk, pe)
        if last_step:
            if consv_te > CONSV_MIN:
                raise NotImplementedError("We do not support consv_te > 0.001 because that would trigger an allReduce")
            elif consv_te < -CONSV_MIN:
                raise NotImplementedError("Unimplemented/untested case consv(" + str(consv_te) + ")  < -CONSV_MIN("
                                          + str(-CONSV_MIN) + ")")
        water = pointer_table([tracers[name] for name in _WATER])
        st = self.stream
        self.call("pace_l2e_prepare", water, dptr(q_con), dptr(pt), dptr(cappa), dptr(delp), dptr(delz), dptr(pe),
                  dptr(self._pe1), dptr(self._pe2), dptr(ak), dptr(bk), dptr(self._dp2), dptr(ps), dptr(self._pn2), dptr(peln),
                  dptr(pk), float(ptop), float(akap), float(zvir), st())
        # now that we have the pressure profiles, we can start remapping
        self._map_single_pt(pt, peln, self._pn2, qmin=self._t_min)
        self._mapn_tracer(self._pe1, self._pe2, self._dp2, tracers)
        self._map_single_w(w, self._pe1, self._pe2, qs=wsd)
        self._map_single_delz(delz, self._pe1, self._pe2)
        self.call("pace_l2e_post", water, dptr(q_con), dptr(pkz), dptr(pt), dptr(cappa), dptr(delp), dptr(delz), dptr(peln),
                  dptr(self._pe0), dptr(self._pn2), float(zvir), st())
        self.call("pace_l2e_pressures", 0, dptr(pe), dptr(self._pe1), dptr(ak), dptr(bk), dptr(self._pe0), dptr(self._pe3), st())
        self._map_single_u(u, self._pe0, self._pe3)
        self.call("pace_l2e_pressures", 1, dptr(pe), dptr(self._pe1), dptr(ak), dptr(bk), dptr(self._pe0), dptr(self._pe3), st())
        self._map_single_v(v, self._pe0, self._pe3)
        # on the last step, we need the regular temperature to send to the physics, but if we're staying in dynamics we
        # need to keep it as the virtual potential temperature
        self.call("pace_l2e_finish", water, dptr(pe), dptr(self._pe2), dptr(pt), dptr(pkz), float(zvir), int(bool(last_step)), st())
