"""ctypes binding of the C ABI in include/pace_hip.h.

The product path is ``load()``: it loads ``pace_amd/libpace_hip.so`` (built in-tree by
``make`` / ``__graft_entry__.build()``) and raises if it is missing -- there is no CPU fallback.
``Library(path)`` with an explicit path exists so the test-suite can also bind
``tests/emu/libpace_emu.so`` (the same kernel sources compiled for the CPU; test infrastructure).
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libpace_hip.so")
DEFAULT_LIB_F32 = os.path.join(_HERE, "libpace_hip_f32.so")  # the same sources with float32 storage (make f32)

c_dp = C.c_void_p  # device pointer to double


class Geom(C.Structure):
    _fields_ = [("n", C.c_int32), ("nk", C.c_int32), ("sj", C.c_int32), ("pad_", C.c_int32), ("sk", C.c_int64)]


METRIC_2D = (
    "area rarea rarea_c dx dy dxa dya dxc dyc rdx rdy rdxa rdya rdxc rdyc cosa rsina cosa_u cosa_v cosa_s "
    "sina_u sina_v rsin_u rsin_v rsin2 sin_sg1 sin_sg2 sin_sg3 sin_sg4 cos_sg1 cos_sg2 cos_sg3 cos_sg4 "
    "del6_u del6_v divg_u divg_v fC fC_agrid"
).split()
METRIC_1D = ["edge_w", "edge_e", "edge_s", "edge_n"]


class Metrics(C.Structure):
    _fields_ = (
        [(n, c_dp) for n in METRIC_2D]
        + [(n, c_dp) for n in METRIC_1D]
        + [("a2b_corner_w", (C.c_double * 3) * 4), ("da_min", C.c_double), ("da_min_c", C.c_double)]
    )


COLUMN_FIELDS = "nord nord_v nord_w nord_t damp_vt damp_w damp_t d2_divg d_con ke_bg fac_vt fac_t fac_vt_c fac_w_c".split()


class Column(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in COLUMN_FIELDS]


DSW_SKIP_DEAD_OUTPUTS = 1  # include/pace_hip.h PACE_DSW_SKIP_DEAD_OUTPUTS


class DswConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("flags", C.c_int32),
        ("hord_dp", C.c_int32), ("hord_tm", C.c_int32), ("hord_vt", C.c_int32), ("hord_mt", C.c_int32),
        ("nord", C.c_int32), ("do_skeb", C.c_int32), ("dddmp", C.c_double), ("d4_bg", C.c_double), ("d_con", C.c_double),
        # optional separate outputs of the four transported scalars (include/pace_hip.h): all four or none
        ("delp_out", C.c_void_p), ("pt_out", C.c_void_p), ("w_out", C.c_void_p), ("q_con_out", C.c_void_p),
        # ... and of the D-grid winds (both or none, with the four above, whole-d_sw calls only)
        ("u_out", C.c_void_p), ("v_out", C.c_void_p),
    ]


class UpdatedzdK(C.Structure):
    _fields_ = [
        ("gk", c_dp), ("beta", c_dp), ("gamma", c_dp),
        ("xt1_top", C.c_double), ("a_bot", C.c_double), ("xt1_bot", C.c_double), ("xt2_bot", C.c_double),
        ("damp", c_dp), ("nord", c_dp), ("nmax", C.c_int32), ("pad_", C.c_int32),
    ]


class HaloDesc(C.Structure):
    _fields_ = [
        ("field", c_dp), ("buf", c_dp),
        ("i0", C.c_int32), ("j0", C.c_int32), ("di_a", C.c_int32), ("dj_a", C.c_int32), ("di_b", C.c_int32), ("dj_b", C.c_int32),
        ("na", C.c_int32), ("nb", C.c_int32), ("nk", C.c_int32), ("pad_", C.c_int32), ("sign", C.c_double),
    ]


class PaceError(RuntimeError):
    pass


_ERR = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported configuration"}

_P = C.POINTER
_PROTOS = {
    "pace_version": (C.c_char_p, []),
    "pace_real_bytes": (C.c_int, []),
    "pace_last_error": (C.c_char_p, []),
    "pace_fxadv": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 8 + [C.c_double, C.c_void_p]),
    "pace_fvtp2d": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 9 + [C.c_int, C.c_int, C.c_void_p]),
    "pace_fvtp2d_update": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 10 + [C.c_int, c_dp, C.c_int, C.c_int, C.c_void_p]),
    "pace_delnflux_nosg": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 5 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pace_delnflux": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "pace_a2b_ord4": (C.c_int, [_P(Geom), _P(Metrics), c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pace_d_sw_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_d_sw_prepare": (C.c_int, [_P(Geom), _P(Column), c_dp, C.c_void_p]),
    "pace_d_sw_pingpong_supported": (C.c_int, [_P(Geom), _P(DswConfig)]),
    "pace_d_sw_wind_outputs_supported": (C.c_int, [_P(Geom), _P(DswConfig)]),
    "pace_d_sw_outputs_supported": (C.c_int, [_P(Geom), _P(Column), _P(DswConfig)]),
    "pace_d_sw": (C.c_int, [_P(Geom), _P(Metrics), _P(Column), _P(DswConfig), c_dp] + [c_dp] * 23 + [C.c_double, C.c_void_p]),
    "pace_d_sw_transport": (C.c_int, [_P(Geom), _P(Metrics), _P(Column), _P(DswConfig), c_dp] + [c_dp] * 23 + [C.c_double, C.c_void_p]),
    "pace_d_sw_winds": (C.c_int, [_P(Geom), _P(Metrics), _P(Column), _P(DswConfig), c_dp] + [c_dp] * 23 + [C.c_double, C.c_void_p]),
    "pace_d_sw_phases": (C.c_int, [C.c_int, _P(Geom), _P(Metrics), _P(Column), _P(DswConfig), c_dp] + [c_dp] * 23 + [C.c_double, C.c_void_p]),
    "pace_d_sw_overlapped": (C.c_int, [C.c_int, _P(Geom), _P(Metrics), _P(Column), _P(DswConfig), c_dp] + [c_dp] * 23
                             + [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pace_riem_solver3_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_riem_solver3": (
        C.c_int,
        [_P(Geom), c_dp, C.c_int, C.c_double, c_dp, C.c_double] + [c_dp] * 13 + [C.c_double, C.c_void_p],
    ),
    "pace_c_sw_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_c_sw": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 16 + [C.c_double, C.c_int, C.c_void_p]),
    "pace_c_sw_part": (C.c_int, [C.c_int, _P(Geom), _P(Metrics)] + [c_dp] * 16 + [C.c_double, C.c_int, C.c_void_p]),
    "pace_d2a2c_vect": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 9 + [C.c_void_p]),
    "pace_riem_solver_c_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_riem_solver_c": (C.c_int, [_P(Geom), c_dp, C.c_double, c_dp, C.c_double] + [c_dp] * 8 + [C.c_double, C.c_void_p]),
    "pace_updatedzc_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_ppm": (C.c_int, [_P(Geom), _P(Metrics), C.c_int, C.c_int, c_dp, c_dp, c_dp] + [C.c_int] * 6 + [C.c_void_p]),
    "pace_sim1_solver_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_sim1_solver": (C.c_int, [_P(Geom), C.c_void_p, C.c_int, C.c_double, C.c_double] + [c_dp] * 10 + [C.c_void_p]),
    "pace_divergence_damping_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_divergence_damping": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 12 + [C.c_double, _P(C.c_double), c_dp, C.c_double,
                                                                                   C.c_double, C.c_int, C.c_void_p]),
    "pace_updatedzc": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 7 + [C.c_double, C.c_void_p]),
    "pace_updatedzd_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_updatedzd": (C.c_int, [_P(Geom), _P(Metrics), c_dp, _P(UpdatedzdK)] + [c_dp] * 7 + [C.c_double, C.c_int, C.c_void_p]),
    "pace_zero_data": (C.c_int, [_P(Geom)] + [c_dp] * 6 + [C.c_int, C.c_void_p]),
    "pace_interface_pressure_from_toa_pressure_and_thickness": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_double, C.c_void_p]),
    "pace_gz_from_surface_height_and_thicknesses": (C.c_int, [_P(Geom), c_dp, c_dp, c_dp, C.c_void_p]),
    "pace_compute_geopotential": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_void_p]),
    "pace_copy": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_void_p]),
    "pace_p_grad_c": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 5 + [C.c_double, C.c_void_p]),
    "pace_nh_p_grad_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_nh_p_grad": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 7 + [C.c_double] * 3 + [C.c_void_p]),
    "pace_edge_pe": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_double, C.c_void_p]),
    "pace_pk3_halo": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_double, C.c_double, C.c_void_p]),
    "pace_ray_fast": (C.c_int, [_P(Geom), c_dp, c_dp, c_dp, _P(C.c_double), _P(C.c_double)] + [C.c_double] * 4 + [C.c_int, C.c_void_p]),
    "pace_del2cubed_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_del2cubed": (C.c_int, [_P(Geom), _P(Metrics), c_dp, c_dp, C.c_double, C.c_int, C.c_void_p]),
    "pace_apply_diffusive_heating": (C.c_int, [_P(Geom)] + [c_dp] * 5 + [C.c_double, C.c_int, C.c_void_p]),
    "pace_tracer_flux_compute": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 4 + [C.c_void_p]),
    "pace_tracer_divide_fluxes": (C.c_int, [_P(Geom)] + [c_dp] * 6 + [C.c_int, C.c_void_p]),
    "pace_apply_mass_flux": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 4 + [C.c_void_p]),
    "pace_apply_tracer_flux": (C.c_int, [_P(Geom), _P(Metrics)] + [c_dp] * 5 + [C.c_void_p]),
    "pace_swap_dp": (C.c_int, [_P(Geom), c_dp, c_dp, C.c_void_p]),
    "pace_map_single_workspace_bytes": (C.c_int64, [_P(Geom)]),
    "pace_map_single": (C.c_int, [_P(Geom), c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double] + [C.c_int] * 4 + [C.c_void_p]),
    "pace_mapn_tracer_workspace_bytes": (C.c_int64, [_P(Geom), C.c_int]),
    "pace_mapn_tracer": (C.c_int, [_P(Geom), c_dp, _P(C.c_void_p), C.c_int, c_dp, c_dp, C.c_int, C.c_void_p]),
    "pace_fillz": (C.c_int, [_P(Geom), _P(C.c_void_p), C.c_int, c_dp, C.c_void_p]),
    "pace_l2e_prepare": (C.c_int, [_P(Geom), _P(C.c_void_p)] + [c_dp] * 15 + [C.c_double] * 3 + [C.c_void_p]),
    "pace_l2e_post": (C.c_int, [_P(Geom), _P(C.c_void_p)] + [c_dp] * 9 + [C.c_double, C.c_void_p]),
    "pace_l2e_pressures": (C.c_int, [_P(Geom), C.c_int] + [c_dp] * 6 + [C.c_void_p]),
    "pace_l2e_finish": (C.c_int, [_P(Geom), _P(C.c_void_p)] + [c_dp] * 4 + [C.c_double, C.c_int, C.c_void_p]),
    "pace_fv_setup_pt": (C.c_int, [_P(Geom), _P(C.c_void_p)] + [c_dp] * 7 + [C.c_void_p]),
    "pace_omega_from_w": (C.c_int, [_P(Geom)] + [c_dp] * 4 + [C.c_void_p]),
    "pace_neg_adj3": (C.c_int, [_P(Geom), _P(C.c_void_p)] + [c_dp] * 3 + [C.c_void_p]),
    "pace_c2l_ord": (C.c_int, [_P(Geom), _P(Metrics), C.c_int] + [c_dp] * 8 + [C.c_void_p]),
    "pace_stencil": (C.c_int, [_P(Geom), _P(Metrics), C.c_int, _P(C.c_void_p), C.c_int, _P(C.c_double), C.c_int, _P(C.c_int), _P(C.c_int),
                               C.c_void_p]),
    "pace_halo_pack": (C.c_int, [_P(Geom), _P(HaloDesc), C.c_int, C.c_void_p]),
    "pace_halo_unpack": (C.c_int, [_P(Geom), _P(HaloDesc), C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


class Library:
    def __init__(self, path):
        if not os.path.exists(path):
            raise PaceError(
                f"{path} not found: build it with `make` (or __graft_entry__.build()). "
                "pace_amd has no CPU fallback -- the HIP library is required."
            )
        self.path = path
        if "emu" not in os.path.basename(path):
            # One HIP runtime per process: PyTorch bundles its own libamdhip64/libhsa-runtime64 and owns the
            # device memory and streams we are handed, so it must be loaded first; our DT_NEEDED
            # libamdhip64.so.7 then binds to that same runtime instead of a second copy under /opt/rocm.
            import torch  # noqa: F401
        self.cdll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        self.real_bytes = int(self.cdll.pace_real_bytes())  # 8: float64 fields; 4: the float32-storage build

    timing = None  # a pace_amd.util.KernelTimes while per-entry-point device times are being collected

    # PACE_TRACE_CALLS=1: name every entry point on stderr and wait for the device after it, so that an asynchronous device
    # fault is reported right after the call that caused it (debugging aid; serialises everything)
    _trace = bool(os.environ.get("PACE_TRACE_CALLS"))

    def call(self, name, *args):
        end = None
        if self.timing is not None:
            # every entry point's last argument is its stream (pace_d_sw_overlapped: followed by the side stream and three events)
            last = (args[-5] if name == "pace_d_sw_overlapped" else args[-1]) if args else None
            ptr = getattr(last, "value", None) if isinstance(last, C.c_void_p) else None
            end = self.timing.bracket(name, ptr)
        if self._trace:
            # (the rank, and the leading integer argument of the entry points that have one -- the `phases` mask of
            # pace_d_sw_phases: tests/test_halo.py reads the order of a multi-rank step from these lines)
            lead = f" {args[0]}" if args and isinstance(args[0], int) else ""
            sys.stderr.write(f"[pace r{os.environ.get('RANK', '0')}] {name}{lead}\n")
            sys.stderr.flush()
        rc = getattr(self.cdll, name)(*args)
        if self._trace and "emu" not in os.path.basename(self.path):
            import torch

            torch.cuda.synchronize()
        if end is not None:
            end.record()
        if rc != 0:
            detail = self.cdll.pace_last_error().decode() if rc == -2 else ""
            raise PaceError(f"{name} failed: {_ERR.get(rc, rc)} {detail}".rstrip())

    def version(self):
        return self.cdll.pace_version().decode()


_default = {}


def load(precision: int = 64):
    """The product library (gfx950): float64 storage, or float32 with precision=32.  Raises PaceError if it has not been built."""
    if precision not in (64, 32):
        raise ValueError("precision is 64 or 32")
    if precision not in _default:
        path = DEFAULT_LIB if precision == 64 else DEFAULT_LIB_F32
        if precision == 64 and os.environ.get("PACE_HIP_LIB"):  # another build of the same sources (tools/, experiments)
            path = os.environ["PACE_HIP_LIB"]
        _default[precision] = Library(path)
    return _default[precision]
