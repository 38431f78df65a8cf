"""DycoreState: the fields the acoustic dynamics read and write, with the reference's names and dims
(fv3core/pace/fv3core/initialization/dycore_state.py:11-340), and the tracer species that the tracer advection, the
remapping and neg_adj3 work on."""
import dataclasses

from ...util.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM

_C = [X_DIM, Y_DIM, Z_DIM]
_FIELDS = {
    "u": ([X_DIM, Y_INTERFACE_DIM, Z_DIM], "m/s"), "v": ([X_INTERFACE_DIM, Y_DIM, Z_DIM], "m/s"), "w": (_C, "m/s"),
    "ua": (_C, "m/s"), "va": (_C, "m/s"), "uc": ([X_INTERFACE_DIM, Y_DIM, Z_DIM], "m/s"),
    "vc": ([X_DIM, Y_INTERFACE_DIM, Z_DIM], "m/s"), "delp": (_C, "Pa"), "delz": (_C, "m"), "ps": ([X_DIM, Y_DIM], "Pa"),
    "pe": ([X_DIM, Y_DIM, Z_INTERFACE_DIM], "Pa"), "pt": (_C, "degK"), "peln": ([X_DIM, Y_DIM, Z_INTERFACE_DIM], "ln(Pa)"),
    "pk": ([X_DIM, Y_DIM, Z_INTERFACE_DIM], "unknown"), "pkz": (_C, "unknown"), "q_con": (_C, "kg/kg"), "omga": (_C, "Pa/s"),
    "mfxd": ([X_INTERFACE_DIM, Y_DIM, Z_DIM], "unknown"), "mfyd": ([X_DIM, Y_INTERFACE_DIM, Z_DIM], "unknown"),
    "cxd": ([X_INTERFACE_DIM, Y_DIM, Z_DIM], ""), "cyd": ([X_DIM, Y_INTERFACE_DIM, Z_DIM], ""), "diss_estd": (_C, "unknown"),
    "phis": ([X_DIM, Y_DIM], "m^2 s^-2"),
    "qvapor": (_C, "kg/kg"), "qliquid": (_C, "kg/kg"), "qrain": (_C, "kg/kg"), "qice": (_C, "kg/kg"), "qsnow": (_C, "kg/kg"),
    "qgraupel": (_C, "kg/kg"), "qo3mr": (_C, "kg/kg"), "qsgs_tke": (_C, "m**2/s**2"), "qcld": (_C, ""),
}


@dataclasses.dataclass
class DycoreState:
    u: object = None
    v: object = None
    w: object = None
    ua: object = None
    va: object = None
    uc: object = None
    vc: object = None
    delp: object = None
    delz: object = None
    ps: object = None
    pe: object = None
    pt: object = None
    peln: object = None
    pk: object = None
    pkz: object = None
    q_con: object = None
    omga: object = None
    mfxd: object = None
    mfyd: object = None
    cxd: object = None
    cyd: object = None
    diss_estd: object = None
    phis: object = None
    qvapor: object = None
    qliquid: object = None
    qrain: object = None
    qice: object = None
    qsnow: object = None
    qgraupel: object = None
    qo3mr: object = None
    qsgs_tke: object = None
    qcld: object = None

    @classmethod
    def init_zeros(cls, quantity_factory):
        return cls(**{name: quantity_factory.zeros(dims, units, dtype=float) for name, (dims, units) in _FIELDS.items()})

    @classmethod
    def init_from_numpy_arrays(cls, dict_of_numpy_arrays, quantity_factory):
        for name in dict_of_numpy_arrays:
            if name not in _FIELDS:
                raise KeyError(name + " is provided, but not part of the dycore state")
        state = cls.init_zeros(quantity_factory)
        for name, a in dict_of_numpy_arrays.items():
            getattr(state, name).set(a)
        return state
