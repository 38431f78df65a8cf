"""Namelist-style configuration dataclasses: the subset of fv3core/pace/fv3core/_config.py:59-476 the
acoustic step reads, with the reference's field names and defaults for the baroclinic test case."""
import dataclasses
from typing import Tuple


@dataclasses.dataclass
class RiemannConfig:
    p_fac: float = 0.05
    a_imp: float = 1.0
    use_logp: bool = False
    beta: float = 0.0


@dataclasses.dataclass
class DGridShallowWaterLagrangianDynamicsConfig:
    dddmp: float = 0.5
    d2_bg: float = 0.0
    d2_bg_k1: float = 0.2
    d2_bg_k2: float = 0.1
    d4_bg: float = 0.15
    ke_bg: float = 0.0
    nord: int = 3
    n_sponge: int = 48
    grid_type: int = 0
    d_ext: float = 0.0
    inline_q: bool = False
    hord_dp: int = 6
    hord_tm: int = 6
    hord_mt: int = 6
    hord_vt: int = 6
    do_f3d: bool = False
    do_skeb: bool = False
    d_con: float = 1.0
    vtdm4: float = 0.06
    do_vort_damp: bool = True
    hydrostatic: bool = False
    convert_ke: bool = False


@dataclasses.dataclass
class AcousticDynamicsConfig:
    n_split: int = 1
    k_split: int = 1
    nord: int = 3
    d_con: float = 1.0
    d_ext: float = 0.0
    beta: float = 0.0
    use_logp: bool = False
    hydrostatic: bool = False
    rf_fast: bool = True
    rf_cutoff: float = 3000.0
    tau: float = 10.0
    p_fac: float = 0.05
    hord_tm: int = 6
    grid_type: int = 0
    delt_max: float = 0.002
    breed_vortex_inline: bool = False
    use_old_omega: bool = True
    d_grid_shallow_water: DGridShallowWaterLagrangianDynamicsConfig = dataclasses.field(
        default_factory=DGridShallowWaterLagrangianDynamicsConfig
    )
    riemann: RiemannConfig = dataclasses.field(default_factory=RiemannConfig)


@dataclasses.dataclass
class DynamicalCoreConfig:
    layout: Tuple[int, int] = (1, 1)
    npx: int = 13
    npy: int = 13
    npz: int = 79
    dt_atmos: float = 225.0
    k_split: int = 1
    n_split: int = 1
    acoustic_dynamics: AcousticDynamicsConfig = dataclasses.field(default_factory=AcousticDynamicsConfig)
    # what DynamicalCore itself reads (fv3core/pace/fv3core/_config.py:150-420; defaults = util/pace/util/namelist.py:10-70
    # with the baseline case's values for the ones it sets)
    nwat: int = 6
    moist_phys: bool = True
    hydrostatic: bool = False
    z_tracer: bool = True
    inline_q: bool = False
    adiabatic: bool = False
    check_negative: bool = False
    consv_te: float = 0.0
    rf_fast: bool = True
    tau: float = 10.0
    grid_type: int = 0
    hord_tr: int = 8
    c2l_ord: int = 4
    nf_omega: int = 1
    fill: bool = True
    kord_tm: int = -9
    kord_tr: int = 9
    kord_wz: int = 9
    kord_mt: int = 9
    do_sat_adj: bool = False

    @property
    def remapping(self):
        return RemappingConfig(fill=self.fill, kord_tm=self.kord_tm, kord_tr=self.kord_tr, kord_wz=self.kord_wz,
                               kord_mt=self.kord_mt, do_sat_adj=self.do_sat_adj, hydrostatic=self.hydrostatic)

    @property
    def d_grid_shallow_water(self):
        return self.acoustic_dynamics.d_grid_shallow_water

    @property
    def riemann(self):
        return self.acoustic_dynamics.riemann


@dataclasses.dataclass(frozen=True)
class RemappingConfig:
    """fv3core/pace/fv3core/_config.py:42-56 (without the saturation-adjustment namelist, which is out of scope)."""

    fill: bool = True
    kord_tm: int = -9
    kord_tr: int = 9
    kord_wz: int = 9
    kord_mt: int = 9
    do_sat_adj: bool = False
    hydrostatic: bool = False
