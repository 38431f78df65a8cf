"""A/B measurements of library variants on the GPU box (experiments; tools/build_variant.sh builds them).

    python tools/exp_variants.py base=pace_amd/libpace_hip.so pf1=build/var/pf1/libpace_hip.so ... [--n 192] [--reps 20]

Every library is measured in a process of its own (round 3: two libraries in one process share the runtime's hardware queues --
each brings its two side streams -- and the one loaded second measured 3 - 5 % slower in the multi-stream phases whichever it
was; profiles/r03_experiments/x19); the outputs travel through a file for the bit-for-bit comparison with the first library's.

For every library: the fused transport kernel alone (pace_fvtp2d_update, operands rotated through distinct copies so that
nothing is cache-warm), the scalar phase of d_sw (mask 2), the wind phase (mask 12), riem_solver3 and the whole substep on one
stream; outputs are compared bit for bit with the first library's.  A library built with -DFV_PROF also prints the stage
times of one interior and one corner workgroup per level.
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig  # noqa: E402
from pace_amd.fv3core.stencils._common import dptr, host_column  # noqa: E402
from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist  # noqa: E402
from pace_amd.fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver  # noqa: E402
from pace_amd.tile import DSW_ARGS, Env  # noqa: E402

STAGES = ["load q footprint", "fused del-n damping", "inner y PPM (+cry)", "q_i (+yfx, area)", "inner x PPM (+crx)",
          "q_j (+xfx, area)", "outer x PPM (+x mass flux, store)", "outer y PPM (+y mass flux, store)", "fluxes -> LDS",
          "cell update (+rarea, q, delp)"]


def timed(fn, reps, before=None):
    ts = []
    for r in range(reps):
        if before is not None:
            before(r)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(r)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--copies", type=int, default=8)
    ap.add_argument("--json", default=None)
    ap.add_argument("--child", default=None, help="(internal) measure the one library given, save its outputs / results under this prefix")
    args = ap.parse_args()
    if args.child is None:
        import subprocess
        import tempfile

        tmp = tempfile.mkdtemp(prefix="pace_exp_")
        first, results = None, {}
        for n_, spec in enumerate(args.libs):
            pre = os.path.join(tmp, f"v{n_}")
            cmd = [sys.executable, os.path.abspath(__file__), spec, "--n", str(args.n), "--nz", str(args.nz), "--reps", str(args.reps),
                   "--copies", str(args.copies), "--child", pre]
            p = subprocess.run(cmd, capture_output=True, text=True)
            out = [ln for ln in p.stdout.splitlines() if ln.strip()]
            if p.returncode != 0:
                print(f"[{spec}] FAILED: {p.stderr[-400:]}")
                continue
            res = json.load(open(pre + ".json"))
            cur = np.load(pre + ".npz")
            if first is None:
                first, worst = cur, {}
            else:
                worst = {}
                for k in cur.files:
                    if not np.array_equal(cur[k], first[k], equal_nan=True):
                        d = np.abs(cur[k] - first[k])
                        worst[k] = float(np.nanmax(d) / (np.nanmax(np.abs(first[k])) + 1e-300))
            for ln in out:
                print(ln.replace("@@DIFF@@", str(worst) if worst else "no (bit-identical)"), flush=True)
            res["differs_from_first"] = worst
            results[spec.split("=", 1)[0]] = res
        if args.json:
            with open(args.json, "w") as f:
                json.dump(results, f, indent=1)
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)
        return
    n, nz = args.n, args.nz
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    results = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        lib = _lib.Library(os.path.join(ROOT, path))
        env = Env(lib, "cuda", m, n, nz)
        cfg = DGridShallowWaterLagrangianDynamicsConfig()
        col = get_column_namelist(cfg, env.qf)
        dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg)
        # as the acoustic loop (and bench.py) call d_sw in every substep but the last: the divergence damping's work fields are not
        # brought to their final state (PACE_EXP_DSW_FLAGS=0: the full contract)
        dsw._cfg.flags = int(os.environ.get("PACE_EXP_DSW_FLAGS", str(_lib.DSW_SKIP_DEAD_OUTPUTS)))
        riem = NonhydrostaticVerticalSolver(env.stencil_factory, env.qf, RiemannConfig())
        names = list(DSW_ARGS) + ["cappa", "delz", "pe", "ppe", "pk3", "pk", "peln"]
        copies = [{k: env.q3(s[k]) for k in names} for _ in range(args.copies)]
        zs, ws = env.q2(s["zs"]), env.q2(s["ws"])
        geom, met = dsw._geom, dsw._met

        def restore(r):
            f = copies[r % args.copies]
            for k in names:
                f[k].set(s[k])

        def phase(mask):
            def run(r):
                f = copies[r % args.copies]
                lib.call("pace_d_sw_phases", mask, C.byref(geom), C.byref(met), C.byref(dsw._col), C.byref(dsw._cfg),
                         dsw._workspace.data_ptr(), *[dptr(f[k]) for k in DSW_ARGS], float(s["dt"]), None)
            return run

        def run_riem(r):
            f = copies[r % args.copies]
            riem(False, s["dt"], f["cappa"], m["ptop"], zs, ws, f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"], f["pe"], f["ppe"],
                 f["pk3"], f["pk"], f["peln"], f["w"])

        def step(r):
            phase(15)(r)
            run_riem(r)

        da_min = env.damping.da_min
        nord_t, damp_t = host_column(col["nord_t"], nz), host_column(col["damp_t"], nz)
        kdev = torch.as_tensor(np.concatenate([(damp_t * da_min) ** (nord_t + 1), nord_t]), dtype=env.qf.real, device="cuda")
        out = env.q3()
        # flux preparation once per copy so that crx .. yfx are what the transport sees inside d_sw
        for r in range(args.copies):
            phase(1)(r)
        torch.cuda.synchronize()

        def kernel(r):
            b = copies[r % args.copies]
            lib.call("pace_fvtp2d_update", C.byref(geom), C.byref(met), b["pt"].ptr, b["crx"].ptr, b["cry"].ptr, b["xfx"].ptr,
                     b["yfx"].ptr, b["mfx"].ptr, b["mfy"].ptr, b["delp"].ptr, kdev.data_ptr(), kdev.data_ptr() + lib.real_bytes * nz,
                     int(nord_t.max()), out.ptr, 6, nz, None)

        res = {}
        kernel(0)
        torch.cuda.synchronize()
        outs = {"fvtp2d_update": out.numpy().copy()}
        res["fvtp2d_update"] = timed(kernel, args.reps)
        if hasattr(lib.cdll, "pace_debug_fv_prof"):
            host = (C.c_longlong * (256 * 16 * 4))()
            rows = [[], [], [], []]
            for rep in range(8):
                kernel(rep)
                torch.cuda.synchronize()
                assert lib.cdll.pace_debug_fv_prof(host) == 0
                a = np.frombuffer(host, dtype=np.int64).reshape(4, 256, 16)[:, :nz, :11].astype(float)
                if rep >= 2:
                    for w in range(4):
                        rows[w].append(np.diff(a[w], axis=1))
            for w, label in enumerate(("interior", "corner", "west-edge", "south-edge")):
                d = np.concatenate(rows[w])
                med = np.median(d, axis=0)
                print(f"[{name}] {label} workgroup of k_fvtp2d<6,2,1>: {med.sum():.0f} cycles")
                for i, v in enumerate(med):
                    print(f"    {STAGES[i]:42s} {v:8.0f}  ({100 * v / med.sum():4.1f} %)")
                res[f"stages_{label}"] = [float(v) for v in med]
        restore(0)
        phase(1)(0)
        res["dsw_scalars"] = timed(phase(2), args.reps, before=lambda r: (restore(r), phase(1)(r)))
        res["dsw_winds"] = timed(phase(12), args.reps, before=lambda r: (restore(r), phase(3)(r)))
        res["riem3"] = timed(run_riem, args.reps, before=restore)
        res["fxadv"] = timed(phase(1), args.reps, before=restore)
        res["step"] = timed(step, args.reps, before=restore)
        restore(0)
        step(0)
        torch.cuda.synchronize()
        for k in DSW_ARGS + ["pe", "ppe", "pk3", "delz"]:
            outs[k] = copies[0][k].numpy().copy()
        np.savez(args.child + ".npz", **outs)
        with open(args.child + ".json", "w") as f:
            json.dump(res, f)
        results[name] = res
        line = "  ".join(f"{k} {v[0]:7.1f}" for k, v in res.items() if isinstance(v, tuple))
        print(f"[{name}] us (median): {line}   differs: @@DIFF@@", flush=True)
        del copies, dsw, riem, env, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
