"""DGridShallowWaterLagrangianDynamics + get_column_namelist
(reference: fv3core/pace/fv3core/stencils/d_sw.py:614-683,726-1237)."""
import ctypes as C
import os

import numpy as np
import torch

from ... import _lib
from ...util.constants import Z_DIM
from .._config import DGridShallowWaterLagrangianDynamicsConfig
from ._common import Operator, check_layout, dptr, host_column

dcon_threshold = 1e-5


_COLUMN_NAMES = ["ke_bg", "d_con", "nord", "nord_v", "nord_w", "nord_t", "damp_vt", "damp_w", "damp_t", "d2_divg"]


def column_namelist_arrays(config: DGridShallowWaterLagrangianDynamicsConfig, nz: int):
    """The values of d_sw.py:633-683 as host arrays of nz + 1 entries (the last one is the allocator's zero)."""
    names = _COLUMN_NAMES
    col = {n: np.zeros(nz + 1) for n in names}
    v = {n: col[n][:nz] for n in names}  # the reference's .view (compute levels)
    for n in ("ke_bg", "d_con", "nord"):
        v[n][:] = getattr(config, n)
    v["d2_divg"][:] = min(0.2, config.d2_bg)
    v["nord_v"][:] = min(2, v["nord"][0])
    v["nord_w"][:] = v["nord_v"][0]
    v["nord_t"][:] = v["nord_v"][0]
    v["damp_vt"][:] = config.vtdm4 if config.do_vort_damp else 0
    v["damp_w"][:] = v["damp_vt"][0]
    v["damp_t"][:] = v["damp_vt"][0]

    def set_low(k):
        for n in ("nord", "nord_w", "d_con"):
            v[n][k] = 0
        v["damp_w"][k] = v["d2_divg"][k]

    def vort_opt(k):
        if config.do_vort_damp:
            v["nord_v"][k] = 0
            v["damp_vt"][k] = 0.5 * v["d2_divg"][k]

    if nz == 1 or config.n_sponge < 0:
        v["d2_divg"][0] = config.d2_bg
    else:
        v["d2_divg"][0] = max(0.01, config.d2_bg, config.d2_bg_k1)
        set_low(0)
        vort_opt(0)
        if config.d2_bg_k2 > 0.01:
            v["d2_divg"][1] = max(config.d2_bg, config.d2_bg_k2)
            set_low(1)
            vort_opt(1)
        if config.d2_bg_k2 > 0.05:
            v["d2_divg"][2] = max(config.d2_bg, 0.2 * config.d2_bg_k2)
            set_low(2)
    return col


def get_column_namelist(config: DGridShallowWaterLagrangianDynamicsConfig, quantity_factory):
    """d_sw.py:633-683: dictionary of K-Quantities describing how nord/damp vary with level."""
    col = column_namelist_arrays(config, quantity_factory.sizer.nz)
    out = {}
    for n in _COLUMN_NAMES:
        q = quantity_factory.zeros([Z_DIM], units="unknown")
        q.set(col[n])
        out[n] = q
    return out


class DGridShallowWaterLagrangianDynamics(Operator):
    """Fortran d_sw.  One call = the 29-launch HIP sequence in csrc/k_dsw.hip::launch_d_sw instead of
    the reference's 177 stencil launches."""

    def __init__(self, stencil_factory, quantity_factory, grid_data, damping_coefficients, column_namelist, nested: bool,
                 stretched_grid: bool, config: DGridShallowWaterLagrangianDynamicsConfig, *, swap_scalar_storage: bool = False):
        """The keyword-only argument is an extension (default: the reference's contract, d_sw.py:726-1237).

        swap_scalar_storage: delp, pt, w, q_con are written to spare buffers this object owns and swapped into the caller's
        Quantities (``Quantity.swap_storage``): the Quantity objects look updated in place, but their STORAGE changes identity at
        every call -- tensors taken from ``.data`` before the call, cached device pointers and captured HIP graphs go stale
        (``Quantity.generation`` counts the swaps; holders of cached pointers check it).  Off: the library writes to its workspace
        and copies back (+ one copy kernel).  AcousticDynamics, which owns its state between halo updates, turns it on."""
        super().__init__(stencil_factory, quantity_factory, grid_data)
        assert config.grid_type < 3, "ubke and vbke only implemented for grid_type < 3"
        assert not config.inline_q, "inline_q not yet implemented"
        assert config.d_ext <= 0, "untested d_ext > 0. need to call a2b_ord2, not yet implemented"
        assert not nested, "nested not implemented"
        if stretched_grid:
            raise NotImplementedError("stretched_grid")
        if config.do_f3d:
            raise NotImplementedError("do_f3d is not implemented")
        nz = self.grid_indexing.domain[2]
        col = {k: host_column(v, nz) for k, v in column_namelist.items()}
        assert (col["damp_vt"] > dcon_threshold).all()
        assert (col["damp_w"] > dcon_threshold).all()
        da_min, da_min_c = damping_coefficients.da_min, damping_coefficients.da_min_c
        # calc_damp exactly where the reference evaluates it (DelnFlux.__init__ with da_min; d_sw.py:924-933 with da_min_c)
        col["fac_vt"] = (col["damp_vt"] * da_min) ** (col["nord_v"] + 1)
        col["fac_t"] = (col["damp_t"] * da_min) ** (col["nord_t"] + 1)
        col["fac_vt_c"] = (col["damp_vt"] * da_min_c) ** (col["nord_v"] + 1)
        col["fac_w_c"] = (col["damp_w"] * da_min_c) ** (col["nord_w"] + 1)
        self._col_host = {k: np.ascontiguousarray(v) for k, v in col.items()}
        self._col = _lib.Column()
        for k in _lib.COLUMN_FIELDS:
            setattr(self._col, k, self._col_host[k].ctypes.data_as(C.POINTER(C.c_double)))
        self._cfg = _lib.DswConfig(C.sizeof(_lib.DswConfig), 0, config.hord_dp,
                                   config.hord_tm, config.hord_vt, config.hord_mt, config.nord, int(config.do_skeb), config.dddmp,
                                   config.d4_bg, config.d_con)
        nbytes = self.lib.cdll.pace_d_sw_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)
        self.call("pace_d_sw_prepare", C.byref(self._col), self._workspace.data_ptr(), self.stream())
        # The four scalars d_sw transports are written to buffers of their own where the library supports it (the fused scalar
        # kernel of the production tilings, include/pace_hip.h pace_dsw_config_t) and swapped into the caller's Quantities.
        # (pace_d_sw_outputs_supported: the one predicate of what the launcher accepts, the column namelist's damping orders included)
        accepts = int(self.lib.cdll.pace_d_sw_outputs_supported(C.byref(self._geom), C.byref(self._col), C.byref(self._cfg)))
        self._pingpong = bool(swap_scalar_storage) and not os.environ.get("PACE_DSW_INPLACE") and bool(accepts & 1)
        # ... and the winds, where the library updates them in the kernel that transports the scalars
        self._wind_outputs = self._pingpong and bool(accepts & 2)
        self._quantity_factory = quantity_factory
        self._spares = None

    def _outputs_for(self, fields, winds: bool):
        """Point the config at the spare buffers (allocated at the first call as copies of the fields, so that the storage line
        beyond the halo, which no kernel writes, holds what the fields hold).  fields: delp, pt, w, q_con, u, v; `winds`: the call
        runs the whole of d_sw (the winds have outputs of their own only then).  Returns the (field, spare) pairs to swap."""
        cfg = self._cfg
        cfg.delp_out = cfg.pt_out = cfg.w_out = cfg.q_con_out = cfg.u_out = cfg.v_out = None
        if not self._pingpong or not all(hasattr(f, "swap_storage") for f in fields):
            return []
        if self._spares is None:
            self._spares = []
            for f in fields[:4] + (tuple(fields[4:]) if self._wind_outputs else ()):  # (the winds' spares only where they are taken)
                sp = self._quantity_factory.empty(f.dims, f.units)
                sp.data[...] = f.data
                self._spares.append(sp)
        sp = self._spares
        cfg.delp_out, cfg.pt_out, cfg.w_out, cfg.q_con_out = (dptr(x) for x in sp[:4])
        pairs = list(zip(fields[:4], sp[:4]))
        if winds and self._wind_outputs:
            cfg.u_out, cfg.v_out = dptr(sp[4]), dptr(sp[5])
            pairs += list(zip(fields[4:], sp[4:]))
        return pairs

    @staticmethod
    def _swap_in(pairs):
        for f, sp in pairs:
            f.swap_storage(sp)

    def _args(self, fields, dt):
        check_layout(self._geom, *fields)
        return (C.byref(self._met), C.byref(self._col), C.byref(self._cfg), self._workspace.data_ptr(), *[dptr(f) for f in fields],
                float(dt))

    def start_flux_preparation(self, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con,
                               zh, heat_source, diss_est, dt):
        """An extension for overlapping the uc / vc halo exchange with compute (dyn_core.py:817-820: `uc__vc.start()` ...
        `uc__vc.wait()` right before d_sw): the part of FiniteVolumeFluxPrep that reads no halo value of uc / vc -- the box
        [is+2, ie-1] x [js+2, je-1], 91 % of the points at C192 -- is launched now; the following ``__call__`` (same arguments,
        after the wait) then computes only the frame of the flux preparation before it goes on.  Same results bit for bit."""
        fields = (delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source,
                  diss_est)
        self._cfg.delp_out = self._cfg.pt_out = self._cfg.w_out = self._cfg.q_con_out = self._cfg.u_out = self._cfg.v_out = None
        self.lib.call("pace_d_sw_phases", 16, C.byref(self._geom), *self._args(fields, dt), self.stream())
        self._prep_started = True

    _prep_started = False

    def __call__(self, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh,
                 heat_source, diss_est, dt, overlap_winds: bool = False, skip_dead_outputs: bool = False):
        """skip_dead_outputs=True (an extension): delpc, divgd, uc, vc are left unspecified (include/pace_hip.h
        PACE_DSW_SKIP_DEAD_OUTPUTS): they are work fields of the divergence damping that c_sw recomputes before anything reads
        them again (dyn_core.py:720-852) -- AcousticDynamics asks for it in every substep but the last, whose leftovers in uc / vc
        the reference's TranslateDynCore compares (translate_dyncore.py:84-85).

        overlap_winds=True (an extension; the default is the reference's behaviour): the wind update of d_sw, which
        nothing before nh_p_grad reads, is launched on this object's side stream: after return it runs concurrently with the caller's next
        launches (halo exchange, updatedzd, riem_solver3).  The caller MUST call
        ``join()`` before touching u, v, uc, vc, heat_source, diss_est, delpc or divgd again."""
        fields = (delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh,
                  heat_source, diss_est)
        late_winds = bool(overlap_winds and not self._emu and os.environ.get("PACE_DSW_LATE_WINDS"))  # (phases in separate calls)
        pairs = self._outputs_for((delp, pt, w, q_con, u, v), winds=not late_winds)
        self._cfg.flags = _lib.DSW_SKIP_DEAD_OUTPUTS if skip_dead_outputs else 0
        args = self._args(fields, dt)
        # flux preparation: everything (1), or only its frame (32) if start_flux_preparation did the interior box (16)
        prep = 32 if self._prep_started else 1
        self._prep_started = False

        def phases(mask, stream_ptr):
            self.lib.call("pace_d_sw_phases", mask, C.byref(self._geom), *args, stream_ptr)

        if not overlap_winds or self._emu:
            if prep == 1:
                self.call("pace_d_sw", *args, self.stream())
            else:
                phases(prep | 14, self.stream())
            self._swap_in(pairs)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self._workspace.device)
            self._ev_prep, self._ev_scalars, self._done = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        main, side = torch.cuda.current_stream(), self._side
        side_ptr = C.c_void_p(side.cuda_stream)
        # Measured at C192 x 79: running winds A (mask 4) concurrently with the scalar transport (mask 2) gains nothing -- both
        # saturate the SIMDs -- while the winds next to the bandwidth-shaped column solver do (-10 % per substep).
        # (Measured in round 2 as well: only the kinetic energy + vorticity (mask 64, bandwidth-shaped) next to the scalar
        # transports and the rest of the winds (128 | 8) next to the column solver: no difference either.)
        # Winds A (kinetic energy, vorticity, divergence damping, vorticity transport: they need only the flux preparation) run
        # on the side stream NEXT TO the scalar transports, winds B (heating, final wind update: they need the new delp) after
        # them, next to whatever the caller launches next (the column solver).  Measured in round 3 (bench.py, C192 x 79, three
        # alternating runs each): 1.137 ms against 1.191 ms for the round-2 order (all winds after the scalars), which
        # PACE_DSW_LATE_WINDS=1 restores.  (Round 2 had measured no difference: the transport kernels have changed since.)
        if not os.environ.get("PACE_DSW_LATE_WINDS"):
            # (Round 3, rejected: kinetic energy + vorticity started already after the first half of the flux preparation,
            # next to its streaming second half: 1.160 ms against 1.141 ms, four alternating runs -- profiles/r03_experiments/x14.)
            # One call does the choreography (flux preparation, event, winds A on the side stream, scalars, event, winds B on the
            # side stream, event) instead of four calls and three event operations from here.  (The host's 105 - 120 us per
            # substep did not change with it: they are the runtime's dozen kernel launches, not this layer.)
            if self._ev_handles is None:
                for e in (self._ev_prep, self._ev_scalars, self._done):
                    e.record(main)  # (torch creates the underlying event at its first record)
                self._ev_handles = tuple(C.c_void_p(e.cuda_event) for e in (self._ev_prep, self._ev_scalars, self._done))
            self.lib.call("pace_d_sw_overlapped", prep, C.byref(self._geom), *args, self.stream(), side_ptr, *self._ev_handles)
            self._pending = True
            self._swap_in(pairs)
            return
        phases(prep | 2, self.stream())  # flux preparation + scalar transport on the calling stream
        self._ev_scalars.record(main)
        side.wait_event(self._ev_scalars)
        phases(12, side_ptr)            # the whole wind update on the side stream
        self._done.record(side)
        self._pending = True
        self._swap_in(pairs)

    _side = None
    _ev_handles = None
    _pending = False

    def join(self):
        """Make the calling stream wait for an overlapped wind update (no-op otherwise)."""
        if self._pending:
            torch.cuda.current_stream().wait_event(self._done)
            self._pending = False
