"""C-ABI library: loads, exports every symbol include/pace_hip.h declares (no compute calls -- no GPU
here); host-side mirror of the reference's dsl/util API.  CPU only."""
import os
import re

import numpy as np
import pytest

from helpers import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pace_hip.h")).read()
    return sorted(set(re.findall(r"\b(pace_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from pace_amd import _lib

    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_declared_symbols():
    from pace_amd import _lib

    if not os.path.exists(_lib.DEFAULT_LIB):
        import __graft_entry__

        __graft_entry__.build()
    import ctypes

    dll = ctypes.CDLL(_lib.DEFAULT_LIB)
    for sym in _declared_symbols():
        assert hasattr(dll, sym), sym
    dll.pace_version.restype = ctypes.c_char_p
    assert b"gfx950" in dll.pace_version()


def test_missing_library_fails_loudly(tmp_path):
    from pace_amd import _lib

    with pytest.raises(_lib.PaceError, match="no CPU fallback"):
        _lib.Library(str(tmp_path / "libpace_hip.so"))


def test_grid_indexing_matches_reference_semantics():
    """dsl/pace/dsl/stencil.py:629-716 and tests/main/dsl/test_stencil_factory.py."""
    from pace_amd.dsl import GridIndexing

    gi = GridIndexing((12, 12, 79), 3, True, True, True, True)
    assert (gi.isc, gi.iec, gi.jsc, gi.jec, gi.isd, gi.ied) == (3, 14, 3, 14, 0, 17)
    assert gi.domain_full() == (18, 18, 79)
    assert gi.domain_compute(add=(1, 1, 0)) == (13, 13, 79)
    assert gi.origin_full(add=(1, 1, 0)) == (1, 1, 0)
    assert gi.max_shape == (19, 19, 80)
    o, d = gi.get_origin_domain(["x_interface", "y", "z"], halos=(1, 2))
    assert o == (2, 1, 0) and d == (15, 16, 79)
    r = gi.restrict_vertical(k_start=3)
    assert r.origin == (3, 3, 3) and r.domain == (12, 12, 76)
    with pytest.raises(ValueError):
        gi.restrict_vertical(k_start=80)
    with pytest.raises(ValueError):
        gi.restrict_vertical(k_start=1, nk=79)
    ax = gi.axis_offsets((0, 0, 0), (18, 18, 79))
    assert ax["i_start"] == 3 and ax["local_ie"] == 14


def test_quantity_layout_and_views():
    from pace_amd.util import QuantityFactory, SubtileGridSizer

    sizer = SubtileGridSizer.from_tile_params(nx_tile=12, ny_tile=12, nz=79, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device="cpu")
    q = qf.zeros(["x", "y", "z"], "m")
    assert q.shape == (19, 19, 80) and q.origin == (3, 3, 0) and q.extent == (12, 12, 79)
    assert tuple(q.data.stride()) == (1, 32, 32 * 19)  # i fastest, rows padded to 128 B
    q.data[4, 5, 6] = 7.0
    assert q._base[6, 5, 4] == 7.0
    q.view[:] = np.ones((12, 12, 79))
    assert float(q.data.sum()) == 12 * 12 * 79 - 1 + 7.0 or float(q.data.sum()) == 12 * 12 * 79
    qi = qf.zeros(["x_interface", "y_interface", "z_interface"], "")
    assert qi.extent == (13, 13, 80) and qi.shape == (19, 19, 80)
    with pytest.raises(NotImplementedError):
        SubtileGridSizer.from_tile_params(12, 12, 79, 3, {}, layout=(3, 3))


def test_frozen_stencil_unregistered_raises():
    from pace_amd import _lib
    from pace_amd.dsl import CompilationConfig, GridIndexing, StencilConfig, StencilFactory

    def some_gtscript_stencil(q_in, q_out):
        pass

    class _Lib:  # no library needed to test the lookup
        def version(self):
            return "x"

    sf = StencilFactory(StencilConfig(compilation_config=CompilationConfig()), GridIndexing((12, 12, 79), 3, True, True, True, True), lib=_Lib())
    with pytest.raises(NotImplementedError, match="no HIP implementation"):
        sf.from_origin_domain(some_gtscript_stencil, (0, 0, 0), (18, 18, 79))
    with pytest.raises(ValueError):
        CompilationConfig(backend="numpy")
    assert isinstance(_lib.PaceError("x"), RuntimeError)


def test_column_namelist_matches_reference_values():
    """d_sw.get_column_namelist (d_sw.py:633-683) vs the values the reference computed (fixture)."""
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import get_column_namelist
    from pace_amd.util import QuantityFactory, SubtileGridSizer

    from helpers import golden

    sizer = SubtileGridSizer.from_tile_params(nx_tile=12, ny_tile=12, nz=79, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    col = get_column_namelist(DGridShallowWaterLagrangianDynamicsConfig(), QuantityFactory(sizer, device="cpu"))
    ref = golden("column_namelist_c12.npz")
    for k, v in ref.items():
        np.testing.assert_array_equal(col[k].numpy()[:79], v[:79], err_msg=k)


def test_entry_points_reject_inconsistent_geometry():
    """Every entry validates pace_geom_t before anything is launched (the kernels address fields with 32-bit byte offsets):
    strides that do not cover the storage -> PACE_ERR_ARG, a field of 4 GB or more -> PACE_ERR_UNSUPPORTED.  Checked on the
    emulation build (argument checking is the same translation unit, csrc/capi.hip); nothing is computed."""
    import ctypes as C

    from helpers import build_emu
    from pace_amd import _lib

    lib = _lib.Library(build_emu())
    buf = (C.c_double * 8)()
    p = C.cast(buf, C.c_void_p).value

    def copy_rc(n, nk, sj, sk):
        g = _lib.Geom(n, nk, sj, 0, sk)
        return lib.cdll.pace_copy(C.byref(g), p, p, None)

    assert copy_rc(12, 79, 16, 32 * 19) == -1            # row stride shorter than N + 7
    assert copy_rc(12, 79, 32, 32 * 18) == -1            # level stride shorter than sj * (N + 7)
    assert copy_rc(0, 79, 32, 32 * 19) == -1 and copy_rc(12, 0, 32, 32 * 19) == -1
    assert copy_rc(3000, 79, 3008, 3008 * 3007) == -3    # (nk + 1) * sk * 8 B >= 4 GB
    with pytest.raises(_lib.PaceError, match="pace_copy failed"):
        lib.call("pace_copy", C.byref(_lib.Geom(12, 79, 16, 0, 32 * 19)), p, p, None)


def test_frozen_stencils_of_the_registry():
    """The per-stencil boundary (dsl/pace/dsl/stencil.py:395-434) is real for the stencils the registry lists: built through
    StencilFactory.from_origin_domain / from_dims_halo from a definition function with the reference's name and arguments,
    called with positional / keyword Quantities, refused at CONSTRUCTION for a window the device kernel does not implement,
    TypeError for origin= / domain= at call (tests/main/dsl/test_stencil_wrapper.py).  Emulation build, CPU."""
    from helpers import Env, build_emu, golden
    from pace_amd import _lib
    from pace_amd.dsl import get_stencils_with_varied_bounds
    from pace_amd.fv3core.stencils import basic_operations, dyn_core, xppm

    lib = _lib.Library(build_emu())
    n, nz = 12, 6
    env = Env(lib, "cpu", golden("grid_c12_tile0.npz"), n, nz)
    sf, gi = env.stencil_factory, env.grid_indexing
    rng = np.random.default_rng(0)
    a = rng.random((n + 7, n + 7, nz + 1))
    src, dst = env.q3(a), env.q3()
    copy = sf.from_origin_domain(basic_operations.copy_defn, origin=gi.origin_full(), domain=gi.domain_full(add=(0, 0, 1)))
    copy(src, q_out=dst)
    assert np.array_equal(dst.numpy()[:-1, :-1, :], a[:-1, :-1, :]) and not dst.numpy()[-1].any()
    with pytest.raises(TypeError, match="origin"):
        copy(src, dst, origin=(0, 0, 0))
    with pytest.raises(NotImplementedError, match="launch window"):
        sf.from_origin_domain(basic_operations.copy_defn, origin=gi.origin_compute(), domain=gi.domain_compute())
    geo = sf.from_dims_halo(dyn_core.compute_geopotential, compute_dims=["x", "y", "z_interface"], compute_halos=(2, 2))
    gz = env.q3()
    geo(src, gz)
    w = (slice(1, n + 5), slice(1, n + 5))
    assert np.array_equal(gz.numpy()[w], a[w] * 9.80665)
    # a stencil that honours ANY window: the PPM flux (xppm.py:269-287) on two windows at once
    q, c = env.q3(a), env.q3(0.3 * (a - 0.5))
    outs = [env.q3(), env.q3()]
    origins, domains = [(3, 3, 0), (5, 4, 1)], [(n + 1, n, nz), (3, 2, 2)]
    sts = get_stencils_with_varied_bounds(xppm.compute_x_flux, origins, domains, sf, externals={"iord": 6, "mord": 6, "xt_minmax": True})
    for st, o in zip(sts, outs):
        st(q, c, env.grid_data.dxa, o)
    big, small = outs[0].numpy(), outs[1].numpy()
    ws = tuple(slice(o, o + d) for o, d in zip(origins[1], domains[1]))
    assert np.array_equal(small[ws], big[ws]) and np.count_nonzero(small) == np.count_nonzero(small[ws]) > 0
    with pytest.raises(NotImplementedError, match="no HIP implementation"):
        sf.from_origin_domain(lambda q_in, q_out: None, (0, 0, 0), (3, 3, 3))


def test_timer_and_kernel_times():
    """pace.util.Timer's contract (util/pace/util/_timing.py:9-98; util/tests/test_timer.py) and the per-entry-point collector
    behind StencilFactory.exec_report()."""
    import time

    from pace_amd.util import NullTimer, Timer

    t = Timer()
    with t.clock("a"):
        time.sleep(0.01)
    t.start("b")
    with pytest.raises(ValueError, match="already started"):
        t.start("b")
    with pytest.warns(RuntimeWarning):
        assert "b" not in t.times
    t.stop("b")
    with t.clock("a"):
        pass
    assert t.hits == {"a": 2, "b": 1} and t.times["a"] >= 0.01
    t.disable()
    with t.clock("c"):
        pass
    assert "c" not in t.times and not t.enabled
    t.enable()
    t.reset()
    assert t.times == {} and t.hits == {}
    n = NullTimer()
    with n.clock("x"):
        pass
    assert n.times == {}
    with pytest.raises(NotImplementedError):
        n.enable()
    # the step_dynamics timer argument accumulates the reference's three phase names (fv_dynamics.py:505-545)
    from helpers import build_emu
    from pace_amd import _lib

    lib = _lib.Library(build_emu())
    from test_emu_kernels import _lone_dycore

    core, state, _ = _lone_dycore(lib)
    timer = Timer()
    core.step_dynamics(state, timer)
    assert set(timer.hits) == {"DynCore", "TracerAdvection", "Remapping"} and all(v == 1 for v in timer.hits.values())
