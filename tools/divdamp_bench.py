"""DivergenceDamping on its own at C192 x 79 (HIP events on the launch stream).  PACE_LEGACY_DIVERGENCE_DAMPING=1 for the
per-pass kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core.stencils.divergence_damping import DivergenceDamping  # noqa: E402
from pace_amd.fv3core.stencils.d_sw import column_namelist_arrays  # noqa: E402
from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig  # noqa: E402
from pace_amd.tile import DSW_CFG, Env  # noqa: E402

n, nz = int(os.environ.get("N", 192)), 79
lib = _lib.load()
m = synthetic.tile_metrics(n, nz)
s = synthetic.acoustic_state(m, n, nz)
env = Env(lib, "cuda", m, n, nz)
col = column_namelist_arrays(DGridShallowWaterLagrangianDynamicsConfig(), nz)
f = {k: env.q3(s[k]) for k in ("u", "v", "va", "ua", "divgd", "vc", "uc")}
f["vort_b"], f["delpc"], f["ke"], f["wk"] = env.q3(), env.q3(), env.q3(0.5 * s["u"] ** 2), env.q3(1e-5 * s["pt"])
op = DivergenceDamping(env.stencil_factory, env.qf, env.grid_data, env.damping, False, False, DSW_CFG["dddmp"], DSW_CFG["d4_bg"],
                       DSW_CFG["nord"], 0, env.kq(col["nord"]), env.kq(col["d2_divg"]))
run = lambda: op(f["u"], f["v"], f["va"], f["vort_b"], f["ua"], f["divgd"], f["vc"], f["uc"], f["delpc"], f["ke"], f["wk"], float(s["dt"]))
for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
a.record()
for _ in range(reps):
    run()
b.record()
torch.cuda.synchronize()
print(f"DivergenceDamping C{n}: {a.elapsed_time(b) / reps * 1e3:.1f} us  (legacy={bool(os.environ.get('PACE_LEGACY_DIVERGENCE_DAMPING'))})")
