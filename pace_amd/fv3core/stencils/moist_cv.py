"""Identity keys of the moist_cv stencil definitions a FrozenStencil can be built from (dsl/device_stencils.py holds the
implementations): moist_pt_last_step (moist_cv.py:84-118), moist_pkz (moist_cv.py:130-172) and moist_pt -- the stencil the
reference's test module wraps around moist_cv.moist_pt_func (tests/savepoint/translate/translate_moistcvpluspt_2d.py:9-37)."""


def moist_pt_last_step(qvapor, qliquid, qrain, qsnow, qice, qgraupel, gz, pt, pkz, dtmp, r_vir):
    raise TypeError("a stencil definition: build it with StencilFactory.from_origin_domain")


def moist_pkz(qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, gz, cvm, pkz, pt, cappa, delp, delz, r_vir):
    raise TypeError("a stencil definition: build it with StencilFactory.from_origin_domain")


def moist_pt(qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, pt, cappa, delp, delz, r_vir):
    raise TypeError("a stencil definition: build it with StencilFactory.from_origin_domain")
