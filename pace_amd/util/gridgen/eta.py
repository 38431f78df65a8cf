"""Hybrid sigma-pressure coefficients ak, bk.  The reference tabulates them for 79, 91 and 72 layers only
(util/pace/util/grid/eta.py:38-573, `set_hybrid_pressure_coefficients`); the tables are DATA of the model configuration and are
shipped here as a fixture (pace_amd/util/gridgen/eta_tables.npz, extracted once by tools/make_eta_tables.py).  Other layer counts
get a smooth generic distribution (not a reference configuration)."""
import os

import numpy as np

_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eta_tables.npz")


def hybrid_coefficients(nz):
    if os.path.exists(_TABLES):
        d = np.load(_TABLES)
        if f"ak{nz}" in d.files:
            ak, bk = d[f"ak{nz}"].astype(float), d[f"bk{nz}"].astype(float)
            return ak, bk, float(ak[0])
    # generic: pure pressure above ~100 hPa, sigma-like below
    s = np.linspace(0.0, 1.0, nz + 1) ** 1.6
    p_ref = 300.0 + s * (1.0e5 - 300.0)
    sig = np.clip((p_ref - 1.0e4) / (1.0e5 - 1.0e4), 0.0, 1.0)
    bk = sig ** 1.5
    ak = p_ref - bk * 1.0e5
    ak[0], bk[0] = 300.0, 0.0
    return ak, bk, float(ak[0])
