"""GPU parity on the REAL cubed sphere and at BASELINE's headline / largest sizes (round 3; VERDICT round 2, "next round" 1a-1c).

The operands are the baroclinic test case's own: grid from pace_amd.util.gridgen (edge-averaged metrics, six distinct tiles),
Jablonowski-Williamson initial state, six tiles stepped together on the device, one tile's fields captured at the reference's
"D_SW-In" checkpoint of the second acoustic substep (pace_amd.tile.baroclinic_substep_inputs -- what bench.py times).  The
oracle (numpy restatement of the reference) runs on the SAME inputs on the host:

* full-field d_sw + riem_solver3 at C48 (polar tile), C96 (equatorial tile) and C192 x 79 (the tile bench.py measures):
  every argument of d_sw on the window and at the bound of the reference's Translate test, 3.2e-10 (translate_d_sw.py:19,
  36-65); riem_solver3 at 5e-6 relative (overrides/standard.yaml:49-61) and 1e-10 of the field's magnitude absolute;
* every operator of the acoustic loop body on the sphere's metrics (tests/opchain.py) at C48 and C96;
* BASELINE configuration 5, C384 x 91 with float32 fields: d_sw on a slab of levels x the full plane, riem_solver3 and MapSingle
  (kord 9, 10) on sampled blocks of columns, against the float64 oracle at float32-storage accuracy, and one whole
  DynamicalCore step of one tile (finite, bitwise reproducible).
"""
import numpy as np
import pytest

from helpers import DSW_ARGS, DSW_CFG, Env, compare, dsw_window, golden, oracle_grid, run_d_sw, run_riem3, window

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pace_amd import _lib

    return _lib.load()


@pytest.fixture(scope="module")
def sphere(lib, tmp_path_factory):
    """(n, nz, tile) -> (metrics, fields, dt): the operands of the second acoustic substep of the baroclinic case."""
    from pace_amd.tile import baroclinic_substep_inputs

    cache = str(tmp_path_factory.mktemp("baroclinic"))

    def get(n, nz, tile):
        metrics, s, sc = baroclinic_substep_inputs(lib, "cuda", n, nz, tile, cache_dir=cache)
        s = dict(s)
        s["dt"] = sc["dt"]
        return metrics, s

    return get


def full_column(nz):
    return {k: np.ascontiguousarray(v[:nz]) for k, v in golden("column_namelist_c12.npz").items()}


def riem_inputs(s, a=None):
    a = a or s
    return {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": a["q_con"], "delp": a["delp"], "pt": a["pt"],
            "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"], "log_p_interface": s["peln"], "w": a["w"]}


def check_d_sw_and_riem3(lib, metrics, s, n, nz, report=None):
    from oracle import dgrid_sw, vertical

    col = full_column(nz)
    env = Env(lib, "cuda", metrics, n, nz)
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    g = oracle_grid(metrics, n, nz)
    st = dgrid_sw.DSWState(s["u"].shape)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, DSW_CFG, st, *[a[k] for k in DSW_ARGS], s["dt"])
    errs = {}
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nz)
        assert np.isfinite(a[k][W]).all(), k
        scale = float(np.abs(a[k][W]).max())
        err = compare(a[k][W], out[k][W], near_zero=1e-12 * max(scale, 1e-300))
        errs["d_sw." + k] = err
        assert err < 3.2e-10, (k, err)  # translate_d_sw.py:19
    inp = riem_inputs(s, a)
    got = run_riem3(env, inp, False, s["dt"], float(metrics["ptop"]))
    b = {k: v.copy() for k, v in inp.items()}
    vertical.riem_solver3(g, False, s["dt"], b["cappa"], float(metrics["ptop"]), b["zs"], b["ws"], b["delz"], b["q_con"], b["delp"], b["pt"],
                          b["zh"], b["p"], b["ppe"], b["pk3"], b["pk"], b["log_p_interface"], b["w"], p_fac=0.05)
    for k in ("delz", "zh", "ppe", "pk3", "w"):
        nk = nz if k in ("delz", "w") else nz + 1
        W = window(n, 0, 0, nk)
        scale = float(np.abs(b[k][W]).max())
        # (relative metric of the reference, 5e-6, overrides/standard.yaml:49-61; entries below 1e-5 of the magnitude of ppe / w --
        # zero crossings -- are held to the absolute bound only)
        err = compare(b[k][W], got[k][W], near_zero=(1e-5 if k in ("ppe", "w") else 1e-9) * scale)
        errs["riem_solver3." + k] = err
        assert err < 5e-6, (k, err)
        ab = float(np.abs(b[k][W] - got[k][W]).max()) / scale
        errs["riem_solver3." + k + ".abs_over_scale"] = ab
        assert ab < 1e-10, (k, ab)
    if report is not None:
        report.update(errs)
    return errs


@pytest.mark.parametrize("n,tile", [(48, 2), (96, 4), (192, 0)])
def test_d_sw_and_riem3_full_field_on_the_sphere(lib, sphere, n, tile):
    """Full-field parity at the headline size (C192 x 79, the very operands bench.py times) and at C48 / C96 on other tiles of
    the cube (tile 2 holds the north pole)."""
    import json
    import os

    nz = 79
    metrics, s = sphere(n, nz, tile)
    errs = check_d_sw_and_riem3(lib, metrics, s, n, nz)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(errs, open(os.path.join(out_dir, f"sphere_dsw_riem3_c{n}_tile{tile}_errors.json"), "w"), indent=1)


@pytest.mark.parametrize("n,tile", [(48, 5), (96, 1)])
def test_every_operator_of_the_loop_on_the_sphere(lib, sphere, n, tile):
    """tests/opchain.py's per-operator chain (Translate windows and tolerances) started from the baroclinic case's state on the
    generated grid instead of the synthetic single tile: the seven operators up to updatedzd (opchain.SPHERE_CHAIN says why the
    chain stops there; riem_solver3 has its full-field test above, the rest the six-tile loop on the sphere)."""
    import json
    import os

    from opchain import Chain, ProductOps, check_case

    metrics, s = sphere(n, 79, tile)
    chain = Chain(n, 79, metrics=metrics, state=s)
    ops = ProductOps(lib, "cuda", chain)
    report, seen, failed = {}, [], []
    import opchain

    with np.errstate(all="ignore"):
        cases = list(chain.cases(only=opchain.SPHERE_CHAIN))
    for case in cases:
        try:
            check_case(ops, case, report=report, sphere=True)
        except AssertionError as e:  # (collect everything first: the report is the artefact)
            failed.append(str(e))
        seen.append(case.name)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, f"sphere_operator_errors_c{n}_tile{tile}.json"), "w"), indent=1)
    assert not failed, failed
    assert len(seen) == len(opchain.SPHERE_CHAIN)


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configuration 5: C384 x 91, float32 fields
# ----------------------------------------------------------------------------------------------------------------------
C384 = dict(n=384, nz=91)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(a).max() + 1e-300))


def test_c384x91_f32_d_sw_level_slab_matches_oracle():
    """d_sw of the float32 library at C384 x 91 (levels are independent in d_sw): the device run on all 91 levels, the float64
    oracle on a slab of levels spanning every level class (the two sponge levels, the third, and levels from the middle and the
    bottom) x the FULL plane -- every tile seam and edge form of the 12 x 16 workgroup tiling.  Bound: float32 storage of every
    field and metric (6e-8 per rounding): 2e-5 of the field's magnitude, 5e-3 for the dissipative-heating differences."""
    from oracle import dgrid_sw
    from pace_amd import _lib, synthetic

    n, nz = C384["n"], C384["nz"]
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    lib32 = _lib.load(32)
    env = Env(lib32, "cuda", metrics, n, nz)
    col = {k: np.concatenate([v[:79], np.repeat(v[78:79], nz - 79 + 1)])[:nz] for k, v in golden("column_namelist_c12.npz").items()}
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    assert out["delp"].dtype == np.float32
    k_sel = np.array([0, 1, 2, 3, 45, 46, 89, 90])
    nk = len(k_sel)
    g = oracle_grid(metrics, n, nk)
    colk = {k: np.concatenate([v[k_sel], v[k_sel][-1:]]) for k, v in col.items()}
    pick = lambda a_: np.ascontiguousarray(np.concatenate([a_[:, :, k_sel], a_[:, :, k_sel[-1:]]], axis=2))  # noqa: E731
    # the oracle sees what the device saw: the float32-rounded inputs
    a = {k: pick(s[k]).astype(np.float32).astype(np.float64) for k in DSW_ARGS}
    st = dgrid_sw.DSWState(a["u"].shape)
    dgrid_sw.d_sw(g, colk, DSW_CFG, st, *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nk)
        ref, got = a[k][W], out[k][:, :, k_sel][W[0], W[1]]
        assert np.isfinite(got).all(), k
        e = _rel(ref, got)
        assert e < (5e-3 if k in ("heat_source", "diss_est") else 2e-5), (k, e)


def test_c384x91_f32_d_sw_on_a_generated_sphere_tile_matches_oracle():
    """The same on the cubed sphere: the metric terms of one C384 tile from pace_amd's own generator (gnomonic grid, every edge
    and corner form with its real metrics; checked against the reference's MetricTerms at C12 in tests/test_gridgen.py), the
    state built on them, all of d_sw in float32 storage -- the fused scalar + wind kernel, 12 x 16 tiles of 32 x 24 -- against
    the float64 oracle on a slab of levels x the full plane."""
    from oracle import dgrid_sw
    from pace_amd import _lib, synthetic
    from pace_amd.util import gridgen

    n, nz = C384["n"], C384["nz"]
    metrics = gridgen.tiles(n, nz)[2]
    with np.errstate(all="ignore"):
        s = synthetic.acoustic_state(metrics, n, nz)
    s = {k: (np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0) if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    lib32 = _lib.load(32)
    env = Env(lib32, "cuda", metrics, n, nz)
    col = {k: np.concatenate([v[:79], np.repeat(v[78:79], nz - 79 + 1)])[:nz] for k, v in golden("column_namelist_c12.npz").items()}
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    assert out["delp"].dtype == np.float32
    k_sel = np.array([0, 2, 46, 90])
    nk = len(k_sel)
    r32 = lambda a_: a_.astype(np.float32).astype(np.float64)  # noqa: E731
    g = oracle_grid({k: (r32(v) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in metrics.items()}, n, nk)
    colk = {k: np.concatenate([v[k_sel], v[k_sel][-1:]]) for k, v in col.items()}
    pick = lambda a_: np.ascontiguousarray(np.concatenate([a_[:, :, k_sel], a_[:, :, k_sel[-1:]]], axis=2))  # noqa: E731
    a = {k: r32(pick(s[k])) for k in DSW_ARGS}  # the oracle sees what the device saw: float32-rounded inputs and metrics
    dgrid_sw.d_sw(g, colk, DSW_CFG, dgrid_sw.DSWState(a["u"].shape), *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nk)
        ref, got = a[k][W], out[k][:, :, k_sel][W[0], W[1]]
        assert np.isfinite(got).all(), k
        e = _rel(ref, got)
        assert e < (5e-3 if k in ("heat_source", "diss_est") else 2e-4 if k in ("uc", "vc", "divgd", "delpc") else 5e-5), (k, e)


def test_c384x91_f32_riem_solver3_sampled_columns_match_oracle():
    """riem_solver3 of the float32 library at C384 x 91 -- the six-levels-per-lane instance of the 16-lanes-per-column kernel
    (91 levels; 79 use five) -- on the whole tile; the float64 oracle on 12 x 12 blocks of columns (corners, an edge, the centre)
    with the float32-rounded inputs.  The solver amplifies input rounding (differences of large pressures): bounds are those of
    tests/test_f32.py (C96): 2e-5 of the magnitude, 5e-3 for the perturbation pressure and w."""
    from oracle import vertical
    from pace_amd import _lib, synthetic

    n, nz = C384["n"], C384["nz"]
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(_lib.load(32), "cuda", metrics, n, nz)
    inp = riem_inputs(s)
    got = run_riem3(env, inp, True, s["dt"], float(metrics["ptop"]))
    again = run_riem3(env, inp, True, s["dt"], float(metrics["ptop"]))
    outs = ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w")
    for k in outs:
        nk = nz if k in ("delz", "w") else nz + 1
        a = got[k][window(n, 0, 0, nk)]
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, again[k][window(n, 0, 0, nk)]), k
    m12 = synthetic.tile_metrics(12, nz)
    g12 = oracle_grid(m12, 12, nz)
    for (bi, bj) in ((0, 0), (372, 0), (0, 372), (372, 372), (186, 0), (186, 186)):
        sub = {}
        for k, v in inp.items():
            blk = v[3 + bi:3 + bi + 12, 3 + bj:3 + bj + 12].astype(np.float32).astype(np.float64)
            full = np.zeros((19, 19) + v.shape[2:])
            full[3:15, 3:15] = blk
            sub[k] = full
        vertical.riem_solver3(g12, True, s["dt"], sub["cappa"], float(metrics["ptop"]), sub["zs"], sub["ws"], sub["delz"], sub["q_con"],
                              sub["delp"], sub["pt"], sub["zh"], sub["p"], sub["ppe"], sub["pk3"], sub["pk"], sub["log_p_interface"],
                              sub["w"], p_fac=0.05)
        for k in outs:
            nk = nz if k in ("delz", "w") else nz + 1
            ref = sub[k][3:15, 3:15, :nk]
            dev = got[k][3 + bi:3 + bi + 12, 3 + bj:3 + bj + 12, :nk]
            e = _rel(ref, dev)
            assert e < (5e-3 if k in ("ppe", "w") else 2e-5), (bi, bj, k, e)


@pytest.mark.parametrize("kord,iv", [(9, 1), (10, 1), (9, -2)])
def test_c384x91_f32_map_single_sampled_columns_match_oracle(kord, iv):
    """MapSingle of the float32 library at C384 x 91 on the whole tile; the float64 oracle (oracle/remapping.py) on sampled rows of
    columns with the float32-rounded inputs.  The remapped field is a bounded average of its inputs: 2e-5 of the magnitude
    (the float64 library is bit-exact against the oracle, test_map_single_matches_oracle_c48)."""
    import torch

    from oracle import remapping
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core.stencils.map_single import MapSingle
    from test_gpu_parity import _remap_columns

    n, km = C384["n"], C384["nz"]
    env = Env(_lib.load(32), "cuda", synthetic.tile_metrics(n, km), n, km)
    q, pe1, pe2 = _remap_columns(n, km, seed=11 + kord + iv, deform=1.2)
    qs = 0.1 * q[:, :, km - 1]
    qmin = 200.0 if iv == 1 else 0.0
    fq, f1, f2, fs = env.q3(q), env.q3(pe1), env.q3(pe2), env.q2(qs)
    MapSingle(env.stencil_factory, env.qf, kord, iv, ["x", "y", "z"])(fq, f1, f2, qs=fs if iv == -2 else None, qmin=qmin)
    torch.cuda.synchronize()
    got = fq.numpy()
    assert got.dtype == np.float32
    assert np.isfinite(got[3:3 + n, 3:3 + n, :km]).all()
    r32 = lambda a_: a_.astype(np.float32).astype(np.float64)  # noqa: E731
    for j in (3, 3 + n // 2, 3 + n - 1):
        w = (slice(3, 3 + n), slice(j, j + 1))
        ref = r32(q[w])
        remapping.map_single(ref, r32(pe1[w]), r32(pe2[w]), km, kord, iv, qs=r32(qs[w]) if iv == -2 else None, qmin=qmin)
        d = np.abs(ref[:, :, :km] - got[w][:, :, :km]) / (np.abs(ref[:, :, :km]).max() + 1e-300)
        if kord == 9:
            assert d.max() < 2e-5, (kord, iv, j, float(d.max()))
        else:
            # kord 10 selects between reconstructions by comparisons of second differences (remap_profile.py:430-560): float32
            # rounding of the inputs flips the choice in isolated cells (measured: up to 5e-3 of the magnitude in ~1 % of the
            # cells; the float64 library is bit-exact against the oracle).  Bound the bulk tightly and the flipped cells loosely.
            frac = float((d > 2e-5).mean())
            assert frac < 0.03 and d.max() < 2e-2, (kord, iv, j, frac, float(d.max()))


def test_c384x91_f32_dynamical_core_step_one_tile_is_finite_and_reproducible(tmp_path):
    """One whole DynamicalCore.step_dynamics of one C384 x 91 tile with float32 fields (lone-rank loopback halo exchange,
    n_split = 2): every prognostic field finite, and a second run from the same state bitwise identical."""
    from helpers import run_in_child

    a, b = run_in_child("dycore_c384_f32", tmp_path)
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), k
