"""Extract the reference's ak / bk tables (util/pace/util/grid/eta.py: 79, 91 and 72 layers) into a data file
(pace_amd/util/gridgen/eta_tables.npz).  Dev container only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402
from pace.util.grid.eta import set_hybrid_pressure_coefficients  # noqa: E402

out = {}
for km in (79, 91, 72):
    pc = set_hybrid_pressure_coefficients(km)
    out[f"ak{km}"], out[f"bk{km}"] = np.asarray(pc.ak, dtype=float), np.asarray(pc.bk, dtype=float)
    print(km, pc.ptop, out[f"ak{km}"][:3])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pace_amd", "util", "gridgen", "eta_tables.npz"), **out)
