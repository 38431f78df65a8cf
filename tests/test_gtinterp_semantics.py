"""Known-answer tests of tools/gtinterp.py -- the interpreter that EXECUTES the reference's gtscript source to make the fixtures
under tests/golden/ (GT4Py itself cannot be installed here).  Each case is a tiny stencil whose result follows by hand from
GT4Py's documented execution model; the expected arrays below are written out independently with plain numpy loops/slices.

What is pinned here (hand-derived) and in tools/run_reference_dsl_tests.py (the reference's own literal-valued tests, run in the
dev container): temporaries are computed on whatever extent later offset reads need; statements of a computation run one after
the other over the whole domain; API fields are written inside origin .. origin + domain only; `horizontal(region[...])` bounds
are relative to the compute domain the axis offsets describe, not to the launch window; interval bounds (negative = from the
end); FORWARD / BACKWARD run level by level and see the levels already done; temporaries persist across levels; gtscript
functions are inlined with their arguments' offsets composed; K- and IJ-fields; per-column `while` with a data-dependent k index.
What stays assumed: DESIGN.md section 5.  CPU only.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gtinterp  # noqa: E402
from gtinterp import IJ, K, Field, function, stencil  # noqa: E402

F3 = Field[float]
F2 = Field[IJ, float]
FK = Field[K, float]

NI, NJ, NK = 9, 8, 6


def field(seed):
    return np.random.default_rng(seed).random((NI, NJ, NK)) + 0.5


def run(defn, origin, domain, externals=None, **args):
    stencil(definition=defn, externals=externals or {})(origin=origin, domain=domain, **args)


# 1 ---------------------------------------------------------------------------------------------------------------------
def _temp_extent(a: F3, b: F3):
    with computation(PARALLEL), interval(...):  # noqa: F821
        tmp = a * 2.0
        b = tmp[-1, 0, 0] + tmp[1, 0, 0] + tmp[0, -2, 0]


def test_temporaries_are_computed_where_later_offset_reads_need_them():
    a, b = field(1), np.zeros((NI, NJ, NK))
    run(_temp_extent, (2, 2, 0), (5, 4, NK), a=a, b=b)
    exp = np.zeros_like(b)
    for i in range(2, 7):
        for j in range(2, 6):
            exp[i, j] = 2 * a[i - 1, j] + 2 * a[i + 1, j] + 2 * a[i, j - 2]  # incl. i = 2, 6 and j = 2, 3: tmp outside the domain
    np.testing.assert_array_equal(b, exp)


# 2 ---------------------------------------------------------------------------------------------------------------------
def _sequential(a: F3, b: F3, c: F3):
    with computation(PARALLEL), interval(...):  # noqa: F821
        t = a
        b = t + 1.0
        t = b * 3.0          # second statement sees the first one's result everywhere
        c = t[1, 0, 0] - b   # ... also at a neighbour: the statement before has finished on the whole (extended) domain


def test_statements_of_a_parallel_computation_complete_one_after_the_other():
    a, b, c = field(2), np.zeros((NI, NJ, NK)), np.zeros((NI, NJ, NK))
    run(_sequential, (1, 1, 0), (6, 5, NK), a=a, b=b, c=c)
    W = (slice(1, 7), slice(1, 6))
    np.testing.assert_array_equal(b[W], a[W] + 1.0)
    # t[1,0,0] at the last column of the window lies outside it: t is a temporary, computed there from b's NEW value where b was
    # written (inside the window) -- outside the window b keeps the caller's value (0.0): (0.0) * 3
    exp = np.zeros_like(c)
    bb = np.zeros_like(b)
    bb[W] = a[W] + 1.0
    exp[W] = (bb * 3.0)[2:8, 1:6] - bb[W]
    np.testing.assert_array_equal(c, exp)
    assert not b[0].any() and not b[7:].any() and not c[:, 0].any()  # nothing written outside origin .. origin + domain


# 3 ---------------------------------------------------------------------------------------------------------------------
def _regions(a: F3, b: F3):
    from __externals__ import i_end, i_start, j_start

    with computation(PARALLEL), interval(...):  # noqa: F821
        b = a
        with horizontal(region[i_start, :], region[i_end, j_start + 1 :]):  # noqa: F821
            b = a + 10.0
        with horizontal(region[i_start - 1, j_start]):  # noqa: F821
            b = -1.0


@pytest.mark.parametrize("origin,domain", [((2, 1, 0), (5, 6, NK)), ((3, 2, 0), (3, 3, NK)), ((1, 1, 0), (7, 6, NK))])
def test_region_bounds_are_those_of_the_axis_offsets_not_of_the_launch_window(origin, domain):
    """The reference passes i_start ... j_end = the tile's compute-domain edges in the coordinates of the launch origin
    (GridIndexing.axis_offsets, dsl/pace/dsl/stencil.py:717-759); a stencil launched on a sub-window or on a window with
    halo points must still apply its edge statements at the tile edge, and drop them where the window does not reach it."""
    isc, iec, jsc = 2, 6, 1  # the "tile" compute domain inside the 9 x 8 storage: i 2..6, j 1..6
    ext = dict(i_start=gtinterp.I[0] + (isc - origin[0]), i_end=gtinterp.I[0] + (iec - origin[0]), j_start=gtinterp.J[0] + (jsc - origin[1]))
    a, b = field(3), np.full((NI, NJ, NK), 7.0)
    run(_regions, origin, domain, externals=ext, a=a, b=b)
    exp = np.full_like(b, 7.0)
    for i in range(origin[0], origin[0] + domain[0]):
        for j in range(origin[1], origin[1] + domain[1]):
            v = a[i, j].copy()
            if i == isc or (i == iec and j >= jsc + 1):
                v = a[i, j] + 10.0
            if i == isc - 1 and j == jsc:
                v = np.full(NK, -1.0)
            exp[i, j] = v
    np.testing.assert_array_equal(b, exp)


# 4 ---------------------------------------------------------------------------------------------------------------------
def _forward_backward(d: F3, top: float, p: F3, q: F3, r: F3):
    with computation(FORWARD):  # noqa: F821
        with interval(0, 1):  # noqa: F821
            p = top
        with interval(1, None):  # noqa: F821
            p = p[0, 0, -1] + d[0, 0, -1]
    with computation(BACKWARD):  # noqa: F821
        with interval(-1, None):  # noqa: F821
            q = p
        with interval(0, -1):  # noqa: F821
            q = q[0, 0, 1] * 0.5 + p
    with computation(PARALLEL):  # noqa: F821
        with interval(0, 2):  # noqa: F821
            r = 1.0
        with interval(2, -1):  # noqa: F821
            r = p[0, 0, 1] - p
        with interval(-1, None):  # noqa: F821
            r = -2.0


def test_interval_bounds_and_sequential_vertical_sweeps():
    d = field(4)
    p, q, r = (np.zeros((NI, NJ, NK)) for _ in range(3))
    run(_forward_backward, (0, 0, 0), (NI, NJ, NK), d=d, top=3.0, p=p, q=q, r=r)
    ep = np.zeros_like(p)
    ep[:, :, 0] = 3.0
    for k in range(1, NK):
        ep[:, :, k] = ep[:, :, k - 1] + d[:, :, k - 1]
    eq = np.zeros_like(q)
    eq[:, :, NK - 1] = ep[:, :, NK - 1]
    for k in range(NK - 2, -1, -1):
        eq[:, :, k] = eq[:, :, k + 1] * 0.5 + ep[:, :, k]
    er = np.zeros_like(r)
    er[:, :, 0:2] = 1.0
    er[:, :, 2:NK - 1] = ep[:, :, 3:NK] - ep[:, :, 2:NK - 1]
    er[:, :, NK - 1] = -2.0
    np.testing.assert_array_equal(p, ep)
    np.testing.assert_array_equal(q, eq)
    np.testing.assert_array_equal(r, er)


def test_vertical_window_of_a_launch_is_origin_k_plus_domain_k():
    """restrict_vertical (dsl/pace/dsl/stencil.py:819-855): interval(0, 1) is the FIRST LEVEL OF THE LAUNCH, not of the storage."""
    d = field(5)
    p, q, r = (np.full((NI, NJ, NK), 9.0) for _ in range(3))
    run(_forward_backward, (0, 0, 2), (NI, NJ, 3), d=d, top=1.0, p=p, q=q, r=r)
    assert (p[:, :, :2] == 9.0).all() and (p[:, :, 5:] == 9.0).all()
    np.testing.assert_array_equal(p[:, :, 2], np.full((NI, NJ), 1.0))
    np.testing.assert_array_equal(p[:, :, 4], 1.0 + d[:, :, 2] + d[:, :, 3])
    np.testing.assert_array_equal(q[:, :, 4], p[:, :, 4])
    np.testing.assert_array_equal(q[:, :, 2], (q[:, :, 4] * 0.5 + p[:, :, 3]) * 0.5 + p[:, :, 2])
    np.testing.assert_array_equal(r[:, :, 2:4], np.ones((NI, NJ, 2)))  # interval(0, 2) of a three-level launch; (2, -1) is empty
    np.testing.assert_array_equal(r[:, :, 4], np.full((NI, NJ), -2.0))


# 5 ---------------------------------------------------------------------------------------------------------------------
def _temp_across_levels(a: F3, b: F3):
    with computation(FORWARD):  # noqa: F821
        with interval(0, 1):  # noqa: F821
            acc = a
            b = acc
        with interval(1, None):  # noqa: F821
            acc = acc[0, 0, -1] * 0.5 + a[1, 0, 0]
            b = acc + acc[0, 0, -1]


def test_temporaries_keep_their_levels_in_sequential_computations():
    a, b = field(6), np.zeros((NI, NJ, NK))
    run(_temp_across_levels, (0, 0, 0), (NI - 1, NJ, NK), a=a, b=b)
    acc = np.zeros((NI - 1, NJ, NK))
    acc[:, :, 0] = a[:-1, :, 0]
    exp = np.zeros_like(b)
    exp[:-1, :, 0] = acc[:, :, 0]
    for k in range(1, NK):
        acc[:, :, k] = acc[:, :, k - 1] * 0.5 + a[1:, :, k]
        exp[:-1, :, k] = acc[:, :, k] + acc[:, :, k - 1]
    np.testing.assert_array_equal(b, exp)


# 6 ---------------------------------------------------------------------------------------------------------------------
@function
def _centred(q, w: float):
    return w * (q[-1, 0, 0] + q[1, 0, 0])


@function
def _pair(q):
    lo = q[0, -1, 0]
    hi = q[0, 1, 0]
    return lo, hi


def _inlining(a: F3, b: F3, c: F3):
    with computation(PARALLEL), interval(...):  # noqa: F821
        b = _centred(a[1, 0, 0], 0.25) + _centred(a, 2.0)[0, 1, 0]
        lo, hi = _pair(a[-1, 0, 0])
        c = hi - lo


def test_gtscript_functions_are_inlined_with_composed_offsets():
    a, b, c = field(7), np.zeros((NI, NJ, NK)), np.zeros((NI, NJ, NK))
    run(_inlining, (2, 2, 0), (5, 4, NK), a=a, b=b, c=c)
    W = (slice(2, 7), slice(2, 6))
    np.testing.assert_array_equal(b[W], 0.25 * (a[2:7, 2:6] + a[4:9, 2:6]) + 2.0 * (a[1:6, 3:7] + a[3:8, 3:7]))
    np.testing.assert_array_equal(c[W], a[1:6, 3:7] - a[1:6, 1:5])


# 7 ---------------------------------------------------------------------------------------------------------------------
def _low_rank(a: F3, m: F2, col: FK, s: F2, b: F3):
    with computation(PARALLEL), interval(...):  # noqa: F821
        b = a * m[1, 0] + col + col[1] * 0.0
    with computation(FORWARD), interval(...):  # noqa: F821
        s = a + m  # an IJ field written in a sequential sweep ends as the LAST level's value


def test_ij_and_k_fields():
    a, b = field(8), np.zeros((NI, NJ, NK))
    m = np.random.default_rng(9).random((NI, NJ))
    col = np.arange(NK + 1, dtype=float)
    s = np.zeros((NI, NJ))
    run(_low_rank, (0, 0, 0), (NI - 1, NJ, NK), a=a, m=m, col=col, s=s, b=b)
    np.testing.assert_array_equal(b[:-1], a[:-1] * m[1:, :, None] + col[None, None, :NK])
    np.testing.assert_array_equal(s[:-1], a[:-1, :, NK - 1] + m[:-1])
    assert not s[-1].any()


# 8 ---------------------------------------------------------------------------------------------------------------------
def _column_loops(pe: F3, nsteps: F2, lev: F2, cnt: F3, acc: F3, out: F3):
    with computation(FORWARD), interval(...):  # noqa: F821
        # map_single.py:45-81 in miniature: a per-column while loop (columns leave it one by one) and a data-dependent k offset
        cnt = 0.0
        acc = 0.0
        while cnt < nsteps:
            acc = acc + pe
            cnt = cnt + 1.0
        out = pe[0, 0, lev] + acc


def test_per_column_while_and_data_dependent_k_offset():
    rng = np.random.default_rng(10)
    pe = rng.random((NI, NJ, NK + 1))
    nsteps = rng.integers(0, 5, (NI, NJ)).astype(float)
    lev = rng.integers(0, 2, (NI, NJ)).astype(float)  # offset relative to the current level: pe[k + lev]
    cnt, acc, out = (np.zeros((NI, NJ, NK + 1)) for _ in range(3))
    run(_column_loops, (1, 0, 0), (NI - 1, NJ, NK), pe=pe, nsteps=nsteps, lev=lev, cnt=cnt, acc=acc, out=out)
    exp = np.zeros_like(out)
    for i in range(1, NI):
        for j in range(NJ):
            for k in range(NK):
                exp[i, j, k] = pe[i, j, k + int(lev[i, j])] + sum(pe[i, j, k] for _ in range(int(nsteps[i, j])))
    np.testing.assert_allclose(out, exp, rtol=1e-15, atol=0)
    np.testing.assert_array_equal(cnt[1:, :, :NK], np.broadcast_to(nsteps[1:, :, None], (NI - 1, NJ, NK)))
    assert not out[0].any()  # the column outside the launch window never took part


# 9 ---------------------------------------------------------------------------------------------------------------------
def _masks(a: F3, b: F3, flag: bool):
    from __externals__ import MODE

    with computation(PARALLEL), interval(...):  # noqa: F821
        if __INLINED(MODE == 6):  # noqa: F821
            t = a - 1.0
        else:
            t = a + 100.0
        if a > 1.0:
            b = t * 2.0
        elif a > 0.75:
            b = -t
        else:
            b = 0.0
        if flag:
            b = b + 0.5
        b = b if t[1, 0, 0] > 0.0 else b - 4.0


def test_field_valued_and_compile_time_conditionals():
    a, b = field(11), np.zeros((NI, NJ, NK))
    run(_masks, (0, 0, 0), (NI - 1, NJ, NK), externals={"MODE": 6}, a=a, b=b, flag=True)
    t = a - 1.0
    e = np.where(a > 1.0, t * 2.0, np.where(a > 0.75, -t, 0.0)) + 0.5
    e = np.where(t[1:] > 0.0, e[:-1], e[:-1] - 4.0)
    np.testing.assert_array_equal(b[:-1], e)
