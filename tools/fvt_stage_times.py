"""Where does a workgroup of the scalar-phase kernel (k_fvt_scalars) spend its time?  Development tool: needs
tools/build_prof.sh (build/var/prof/libpace_hip.so, stage stamps compiled in).  C192 x 79, synthetic state."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig  # noqa: E402
from pace_amd.fv3core.stencils._common import dptr  # noqa: E402
from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist  # noqa: E402
from pace_amd.tile import DSW_ARGS, Env  # noqa: E402

PARTS = ["footprint -> LDS", "damping pass 1 + inner sweep", "outer sweep + damping pass 2", "face fluxes + cell update"]  # (the resident layout, round 6)
SCALARS = ["delp", "w", "q_con", "pt", "winds"]


def main():
    n, nz = 192, 79
    lib = _lib.Library(os.path.join(ROOT, "build", "var", "prof", "libpace_hip.so"))
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg,
                                              swap_scalar_storage=True)
    dsw._cfg.flags = _lib.DSW_SKIP_DEAD_OUTPUTS
    copies = [{k: env.q3(s[k]) for k in DSW_ARGS} for _ in range(6)]
    fused = bool(dsw._wind_outputs)  # the winds are the fifth pass of the scalar-phase kernel

    def phase(mask, f):
        lib.call("pace_d_sw_phases", mask, C.byref(dsw._geom), C.byref(dsw._met), C.byref(dsw._col), C.byref(dsw._cfg),
                 dsw._workspace.data_ptr(), *[dptr(f[k]) for k in DSW_ARGS], float(s["dt"]), None)

    host = (C.c_longlong * (4 * 128 * 32))()
    host_arr = (C.c_longlong * (4 * 128 * 64))()
    arrive = [[], [], [], []]
    rows = [[], [], [], []]
    nstamp = 21 if fused else 17
    for rep in range(6):
        f = copies[rep]
        phase(1, f)
        if fused:  # the kernel alone (256), on the kinetic energy / vorticities of phase 4, with outputs of its own
            phase(4, f)
            torch.cuda.synchronize()
            dsw._outputs_for(tuple(f[k] for k in ("delp", "pt", "w", "q_con", "u", "v")), winds=True)
            phase(256, f)
            dsw._cfg.delp_out = dsw._cfg.pt_out = dsw._cfg.w_out = dsw._cfg.q_con_out = dsw._cfg.u_out = dsw._cfg.v_out = None
        else:
            torch.cuda.synchronize()
            phase(2, f)
        torch.cuda.synchronize()
        assert lib.cdll.pace_debug_fvt_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(4, 128, 32)[:, :nz, :nstamp].astype(float)
        if rep >= 1:
            for w in range(4):
                rows[w].append(np.diff(a[w], axis=1))
            if hasattr(lib.cdll, "pace_debug_fvt_arrive") and lib.cdll.pace_debug_fvt_arrive(host_arr) == 0:
                b = np.frombuffer(host_arr, dtype=np.int64).reshape(4, 128, 32, 2)[:, :nz].astype(float)
                for w in range(4):
                    # arrival at barrier q of pass s_, both roles, relative to the stamp that opens the stage (4 s_ + q)
                    arrive[w].append(b[w][:, :nstamp - 1, :] - a[w][:, :nstamp - 1, None])
    # the single-scalar kernel as d_sw launches it: the vorticity transport (phases: flux preparation, ke / vorticity, damping + transport)
    rows1 = [[], [], [], []]
    for rep in range(6):
        f = copies[rep]
        for mask in (1, 64, 128):
            phase(mask, f)
            torch.cuda.synchronize()
        assert lib.cdll.pace_debug_fvt_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(4, 128, 32)[:, :nz, 20:27].astype(float)
        if rep >= 1:
            for w in range(4):
                rows1[w].append(np.diff(a[w], axis=1))
    parts1 = ["footprint -> LDS", "damping, faces, + f", "inner sweeps (operand loads)", "outer x + u epilogue", "outer y + v epilogue", "tail"]
    for w, label in enumerate(("interior", "corner", "west-edge", "south-edge")):
        med = np.median(np.concatenate(rows1[w]), axis=0)
        print(f"{label} workgroup of k_fvt<6,0,0> (vorticity transport): {med.sum():.0f} cycles | " +
              "  ".join(f"{parts1[p]} {med[p]:.0f}" for p in range(6)))
    for w, label in enumerate(("interior", "corner", "west-edge", "south-edge")):
        d = np.concatenate(rows[w])
        med = np.median(d, axis=0)
        print(f"{label} workgroup of k_fvt_scalars: {med.sum():.0f} cycles")
        for sc in range(5 if fused else 4):
            line = "  ".join(f"{PARTS[p][:24]:24s} {med[4 * sc + p]:7.0f}" for p in range(4))
            print(f"   {SCALARS[sc]:6s} {med[4 * sc:4 * sc + 4].sum():7.0f} | {line}")
        if arrive[w]:
            am = np.median(np.concatenate(arrive[w]), axis=0)  # [stage, role]
            print("   when the first x-run wave / the first y-run wave reach the barrier that ends a stage (cycles after the stage opened):")
            for sc in range(5 if fused else 4):
                print(f"   {SCALARS[sc]:6s} " + "  ".join(
                    f"{PARTS[p_][:18]:18s} " + (f"x {am[4 * sc + p_, 0]:6.0f} y {am[4 * sc + p_, 1]:6.0f}" if abs(am[4 * sc + p_, 0]) < 1e9 else "(no barrier)     ")
                    for p_ in range(4)))


if __name__ == "__main__":
    main()
