"""Where does a wave of the column solver (k_riem_column, riem_solver3 instance) spend its time?  Development tool: needs
tools/build_prof.sh (build/var/prof/libpace_hip.so, stage stamps compiled in).  C192 x 79, synthetic state."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import Env, run_riem3  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402

PARTS = ["loads + interface pressures (2 scans)", "precompute: 3 log + 1 exp per level", "pk3 = exp(kappa log p), stores",
         "system 1 (pp): rows, pivots scan, 2 sweeps", "aa", "system 2 (w): pivots scan, 2 sweeps", "pe scan",
         "p1 backwards", "dz: exp / log per level", "height rebuild, stores"]


def main():
    n, nz = 192, 79
    lib = _lib.Library(os.path.join(ROOT, "build", "var", "prof", "libpace_hip.so"))
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": s["q_con"], "delp": s["delp"],
           "pt": s["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
           "log_p_interface": s["peln"], "w": s["w"]}
    host = (C.c_longlong * (256 * 16))()
    rows = []
    for rep in range(6):
        run_riem3(env, inp, False, s["dt"], m["ptop"])
        torch.cuda.synchronize()
        assert lib.cdll.pace_debug_riem_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(256, 16)[:n, :11].astype(float)
        if rep >= 1:
            rows.append(np.diff(a, axis=1))
    d = np.concatenate(rows)
    med = np.median(d, axis=0)
    print(f"one wave of k_riem_column<0,5>: {med.sum():.0f} cycles (median over {d.shape[0]} waves)")
    for p, name in enumerate(PARTS):
        print(f"   {name:46s} {med[p]:8.0f}  {100 * med[p] / med.sum():5.1f} %")


if __name__ == "__main__":
    main()
