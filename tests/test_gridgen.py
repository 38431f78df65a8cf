"""Grid + initial-state generation against what the reference's own MetricTerms / init_baroclinic_state
produced at C12 (captured with tools/capture.py into tests/golden/acoustic_c12_tile*.npz [grid_*, in_*] and
dycore_c12_tile*.npz [in_pt, in_qvapor, in_ps]).  CPU only."""
import numpy as np
import pytest

from helpers import golden

N, NZ = 12, 79
# worst relative error of each metric over the WHOLE storage incl. halos and corner fills, all six tiles (measured: 3e-12)
TOL = 2e-11


@pytest.fixture(scope="module")
def tiles():
    from pace_amd.util import gridgen

    return gridgen.tiles(N, NZ)


def test_metric_terms_match_reference_capture(tiles):
    names = None
    for t in range(6):
        ref = {k[5:]: v for k, v in golden(f"acoustic_c12_tile{t}.npz").items() if k.startswith("grid_")}
        names = sorted(set(ref) - {"area_64"})
        for k in names:
            r, g = np.asarray(ref[k]), np.asarray(tiles[t][k])
            assert r.shape == g.shape, (k, r.shape, g.shape)
            if r.ndim == 0:
                assert abs(float(g) - float(r)) <= 1e-13 * abs(float(r)), (t, k)
                continue
            marker = np.abs(r) >= 1.0e7  # the reference's "no value here" markers (1e8, -1e8) must sit at the same places
            assert np.array_equal(marker, np.abs(g) >= 1.0e7), (t, k)
            with np.errstate(invalid="ignore"):
                err = np.abs(np.where(marker, 0.0, g - r))
            if marker.all():
                continue
            scale = float(np.abs(r[~marker]).max())
            assert float(np.nanmax(err)) <= TOL * scale, (t, k, float(np.nanmax(err)) / scale)
    assert {"area", "dx", "dxc", "cos_sg4", "rsin2", "del6_v", "edge_n", "a22", "fC_agrid", "da_min_c", "dp_ref"} <= set(names)


def test_baroclinic_state_matches_reference_capture(tiles):
    from pace_amd.fv3core.initialization.baroclinic import baroclinic_state_six_tiles

    states = baroclinic_state_six_tiles(tiles, N, NZ)
    for t in range(6):
        a, d = golden(f"acoustic_c12_tile{t}.npz"), golden(f"dycore_c12_tile{t}.npz")
        s = states[t]
        c = (slice(3, 3 + N), slice(3, 3 + N))
        for k in ("delp", "pe", "pk", "peln", "delz", "w"):
            nk = NZ + 1 if k in ("pe", "pk", "peln") else NZ
            np.testing.assert_allclose(s[k][c][:, :, :nk], a["in_" + k][c][:, :, :nk], rtol=1e-14, atol=0, err_msg=f"{t} {k}")
        # winds and surface geopotential incl. the halos they receive from the neighbouring tiles (scale: the jet, 35 m/s)
        for k in ("u", "v"):
            ok = np.abs(a["in_" + k]) < 1e20
            assert float(np.abs(np.where(ok, s[k] - a["in_" + k], 0.0))[:, :, :NZ].max()) < 2e-12, (t, k)
        ok = np.abs(a["in_phis"]) < 1e20
        assert float(np.abs(np.where(ok, s["phis"] - a["in_phis"], 0.0)).max()) < 1e-11 * float(np.abs(a["in_phis"][c]).max()), t
        np.testing.assert_allclose(s["pt"][c][:, :, :NZ], d["in_pt"][:, :, :NZ], rtol=1e-14, atol=0)
        np.testing.assert_allclose(s["qvapor"][c][:, :, :NZ], d["in_qvapor"][:, :, :NZ], rtol=1e-13, atol=1e-30)
        np.testing.assert_allclose(s["ps"][c], d["in_ps"][c], rtol=1e-15)


def test_reference_entry_points():
    """MetricTerms(quantity_factory, communicator) -> GridData / DampingCoefficients.new_from_metric_terms ->
    init_baroclinic_state(grid_data, quantity_factory, adiabatic, hydrostatic, moist_phys, comm): the reference's call chain
    (tests/main/fv3core/test_dycore_call.py:64-101), tile 3 of six, on the CPU."""
    from pace_amd.fv3core.initialization.baroclinic import init_baroclinic_state
    from pace_amd.util import NullComm, QuantityFactory, SubtileGridSizer
    from pace_amd.util.grid import DampingCoefficients, GridData
    from pace_amd.util.gridgen import MetricTerms

    sizer = SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device="cpu")
    mt = MetricTerms(qf, NullComm(rank=3, total_ranks=6))
    gd = GridData.new_from_metric_terms(mt)
    dc = DampingCoefficients.new_from_metric_terms(mt, gd)
    ref = golden("acoustic_c12_tile3.npz")
    np.testing.assert_allclose(gd.area.numpy()[3:15, 3:15], ref["grid_area"][3:15, 3:15], rtol=1e-12)
    assert abs(dc.da_min - float(ref["grid_da_min"])) < 1e-12 * dc.da_min and gd.ptop == 300.0
    state = init_baroclinic_state(gd, qf, adiabatic=False, hydrostatic=False, moist_phys=True, comm=None)
    assert float(np.abs(state.u.numpy()[3:15, 3:16, :NZ] - ref["in_u"][3:15, 3:16, :NZ]).max()) < 2e-12
    with pytest.raises(NotImplementedError):
        init_baroclinic_state(gd, qf, adiabatic=False, hydrostatic=True, moist_phys=True)


def test_generated_grid_is_consistent_at_c48():
    """Properties at a size with no capture: the six tiles cover the sphere (areas sum to 4 pi R^2), metrics are positive and
    symmetric under the tile's own mirror symmetries, halos agree with the neighbours' interiors."""
    from pace_amd.util import constants as c
    from pace_amd.util import gridgen

    n = 48
    g = gridgen.tiles(n, 79)
    total = sum(float(t["area"][3:3 + n, 3:3 + n].sum()) for t in g)
    assert abs(total / (4.0 * np.pi * c.RADIUS ** 2) - 1.0) < 1e-10
    for t in g:
        a = t["area"][3:3 + n, 3:3 + n]
        assert (a > 0).all() and np.allclose(a, a[::-1, :], rtol=1e-10) and np.allclose(a, a.T, rtol=1e-10)
        assert (t["dx"][3:3 + n, 3:4 + n] > 0).all() and (t["sin_sg1"][:-1, :-1] > 0).all()
    # east halo of tile 0 = west interior of tile 1 (no rotation across that edge)
    assert np.allclose(g[0]["area"][3 + n:6 + n, 3:3 + n], g[1]["area"][3:6, 3:3 + n], rtol=0, atol=0)


def test_substep_inputs_captured_from_the_baroclinic_case(tmp_path):
    """bench.py --state baroclinic: the six generated tiles stepped together (emulation library here), tile 4 captured at the
    D_SW-In checkpoint of the second acoustic substep; every operand of d_sw + riem_solver3 is there, finite on the compute
    domain, moving (non-zero Courant numbers), and a second call is served from the cache."""
    from helpers import DSW_ARGS, build_emu

    from pace_amd import _lib
    from pace_amd.tile import RIEM_ONLY, baroclinic_substep_inputs

    lib = _lib.Library(build_emu())
    metrics, fields, sc = baroclinic_substep_inputs(lib, "cpu", 12, 10, 4, cache_dir=str(tmp_path))
    for k in list(DSW_ARGS) + list(RIEM_ONLY) + ["zs", "ws"]:
        assert np.isfinite(fields[k][3:15, 3:15]).all(), k
    assert np.abs(fields["crx"][3:15, 3:15, :10]).max() > 1e-4 and np.abs(fields["w"][3:15, 3:15, :10]).max() > 0.0
    assert (fields["delp"][3:15, 3:15, :10] > 0).all() and sc["ptop"] == float(metrics["ptop"])
    m2, f2, _ = baroclinic_substep_inputs(lib, "cpu", 12, 10, 4, cache_dir=str(tmp_path))
    assert all(np.array_equal(fields[k], f2[k]) for k in fields) and set(m2) == set(metrics)
