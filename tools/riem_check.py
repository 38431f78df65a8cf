"""riem_solver3 on the GPU against the oracle on the d_sw-updated synthetic state (the setting of
tests/test_gpu_parity.py::test_d_sw_and_riem3_match_oracle): error table per variable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import DSW_ARGS, DSW_CFG, Env, compare, oracle_grid, run_d_sw, run_riem3, window  # noqa: E402

from oracle import dgrid_sw, vertical  # noqa: E402
from pace_amd import _lib, synthetic  # noqa: E402

n, nz = int(sys.argv[1]), 79
lib = _lib.load()
metrics = synthetic.tile_metrics(n, nz)
s = synthetic.acoustic_state(metrics, n, nz)
col = {k: np.ascontiguousarray(v[:nz]) for k, v in dict(np.load(os.path.join(ROOT, "tests/golden/column_namelist_c12.npz"))).items()}
env = Env(lib, "cuda", metrics, n, nz)
g = oracle_grid(metrics, n, nz)
st = dgrid_sw.DSWState(s["u"].shape)
a = {k: s[k].copy() for k in DSW_ARGS}
dgrid_sw.d_sw(g, col, DSW_CFG, st, *[a[k] for k in DSW_ARGS], s["dt"])
inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": a["q_con"], "delp": a["delp"],
       "pt": a["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
       "log_p_interface": s["peln"], "w": a["w"]}
got = run_riem3(env, inp, False, s["dt"], metrics["ptop"])
b = {k: v.copy() for k, v in inp.items()}
vertical.riem_solver3(g, False, s["dt"], b["cappa"], metrics["ptop"], b["zs"], b["ws"], b["delz"], b["q_con"], b["delp"], b["pt"],
                      b["zh"], b["p"], b["ppe"], b["pk3"], b["pk"], b["log_p_interface"], b["w"], p_fac=0.05)
for k in ("delz", "zh", "ppe", "pk3", "w"):
    nk = nz if k in ("delz", "w") else nz + 1
    W = window(n, 0, 0, nk)
    r, o = b[k][W], got[k][W]
    scale = float(np.abs(r).max())
    e = compare(r, o, near_zero=1e-9 * scale)
    with np.errstate(all="ignore"):
        rel = 2 * np.abs(r - o) / (np.abs(r) + np.abs(o))
    rel[np.isnan(rel)] = 0
    rel[(np.abs(r) < 1e-9 * scale) & (np.abs(o) < 1e-9 * scale)] = 0
    wh = np.unravel_index(np.argmax(rel), rel.shape)
    print(k, f"err {e:.2e} scale {scale:.3e} worst at {wh}: ref {r[wh]:.6e} got {o[wh]:.6e}  abs err / scale {np.abs(r - o).max() / scale:.2e}")
