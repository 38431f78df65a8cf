"""A small numpy interpreter for gtscript stencil *definitions* (dev-container tool).

Purpose
-------
The reference (ai2cm/pace) expresses all of its numerics as gtscript functions that
only the third-party GT4Py compiler can execute; GT4Py is not installable here.  This
module implements just enough of the gtscript execution model (PARALLEL / FORWARD /
BACKWARD computations, vertical intervals, horizontal regions, field offsets,
temporaries, inlined gtscript functions, compile-time externals) to *run the
reference's own stencil source* on numpy arrays inside this container.

It is used ONLY by ``tools/make_golden.py`` (to generate the fixtures committed under
``tests/golden/``) and by ``tools/crosscheck_oracle.py``.  It never ships to the GPU
box in any role other than source text of this repo, never imports at test time, and
is not part of the product or the oracle.  It is our own restatement of GT4Py's
documented semantics -- it is not GT4Py -- so fixtures produced through it are
labelled "reference source executed under tools/gtinterp" in DESIGN.md.

Execution model implemented
---------------------------
* every 3-D value is evaluated on the *whole* storage in (i, j) (NaN where an offset
  read falls outside the storage), so temporaries automatically have whatever
  horizontal extent later offset reads need;
* writes to API (argument) fields are restricted to ``origin .. origin+domain`` in
  (i, j) and to the active vertical interval; temporaries are written everywhere in
  (i, j), in the active interval only;
* ``with horizontal(region[...])`` and field-valued ``if`` become boolean masks;
* FORWARD / BACKWARD computations are executed level by level.
"""
import ast
import enum
import inspect
import math
import textwrap
import types

import numpy as np


# --------------------------------------------------------------------------- axes
class AxisIndex:
    """gtscript.I[n] + m : a position relative to the start/end of the compute domain."""

    def __init__(self, axis, index=None, offset=0):
        self.axis = axis
        self.index = index
        self.offset = offset

    def __getitem__(self, index):
        return AxisIndex(self.axis, int(index), 0)

    def __add__(self, n):
        return AxisIndex(self.axis, self.index, self.offset + int(n))

    __radd__ = __add__

    def __sub__(self, n):
        return AxisIndex(self.axis, self.index, self.offset - int(n))

    def resolve(self, start, end):
        if self.index is None:
            raise ValueError("bare axis used as a bound")
        base = start + self.index if self.index >= 0 else end + self.index
        return base + self.offset

    def __repr__(self):
        return f"{self.axis}[{self.index}]{self.offset:+d}"


def _resolve_bound(b, start, end, default):
    if b is None:
        return default
    if isinstance(b, AxisIndex):
        return b.resolve(start, end)
    b = int(b)
    return start + b if b >= 0 else end + b


class _AxesMarker:
    def __init__(self, axes):
        self.axes = tuple(axes)


class FieldType:
    def __init__(self, axes, dtype):
        self.axes = tuple(axes)
        self.dtype = dtype


class _FieldMeta(type):
    def __getitem__(cls, item):
        if isinstance(item, tuple):
            axes, dtype = item
        else:
            axes, dtype = IJK, item
        if isinstance(axes, AxisIndex):
            axes = _AxesMarker([axes.axis])
        return FieldType(axes.axes, dtype)


class Field(metaclass=_FieldMeta):
    pass


I = AxisIndex("I")  # noqa: E741
J = AxisIndex("J")
K = _AxesMarker(["K"])
IJK = _AxesMarker(["I", "J", "K"])
IJ = _AxesMarker(["I", "J"])
IK = _AxesMarker(["I", "K"])
JK = _AxesMarker(["J", "K"])

PARALLEL, FORWARD, BACKWARD = "PARALLEL", "FORWARD", "BACKWARD"


class AccessKind(enum.IntFlag):
    NONE = 0
    READ = 1
    WRITE = 2
    READ_WRITE = 3


class FieldInfo:
    def __init__(self, axes, access):
        self.axes = tuple(axes)
        self.access = access


class GtFunction:
    """Result of @gtscript.function: remembered so calls are inlined by the interpreter."""

    def __init__(self, fn):
        self.fn = fn
        self.__name__ = fn.__name__
        self._tree = None

    @property
    def tree(self):
        if self._tree is None:
            self._tree = _parse(self.fn)
        return self._tree

    def __call__(self, *a, **k):
        raise RuntimeError("gtscript function called outside a stencil")


def function(fn):
    return GtFunction(fn)


def _parse(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src).body[0]
    assert isinstance(tree, ast.FunctionDef)
    return tree


# --------------------------------------------------------------------------- values
class Ref:
    """Lazy reference to a storage (3-D, IJ or K) with an accumulated offset."""

    __slots__ = ("arr", "axes", "off")

    def __init__(self, arr, axes, off=(0, 0, 0)):
        self.arr = arr
        self.axes = axes
        self.off = off

    def shifted(self, idx):
        di, dj, dk = self.off
        if self.axes == ("I", "J", "K"):
            a, b, c = idx
            return Ref(self.arr, self.axes, (di + a, dj + b, dk + c))
        if self.axes == ("I", "J"):
            a, b = idx[0], idx[1]
            return Ref(self.arr, self.axes, (di + a, dj + b, dk))
        if self.axes == ("K",):
            return Ref(self.arr, self.axes, (di, dj, dk + idx[0]))
        if self.axes == ("I",):
            return Ref(self.arr, self.axes, (di + idx[0], dj, dk))
        if self.axes == ("J",):
            return Ref(self.arr, self.axes, (di, dj + idx[0], dk))
        raise NotImplementedError(self.axes)


def _shift3(arr, di, dj, k0, k1, dk):
    NI, NJ, NK = arr.shape
    if di == 0 and dj == 0 and 0 <= k0 + dk and k1 + dk <= NK:
        return arr[:, :, k0 + dk : k1 + dk]
    out = np.full((NI, NJ, k1 - k0), np.nan)
    si0, si1 = max(0, di), min(NI, NI + di)
    sj0, sj1 = max(0, dj), min(NJ, NJ + dj)
    sk0, sk1 = max(0, k0 + dk), min(NK, k1 + dk)
    if si0 < si1 and sj0 < sj1 and sk0 < sk1:
        out[si0 - di : si1 - di, sj0 - dj : sj1 - dj, sk0 - dk - k0 : sk1 - dk - k0] = arr[
            si0:si1, sj0:sj1, sk0:sk1
        ]
    return out


def _shift_slab(slab, di, dj):
    if di == 0 and dj == 0:
        return slab
    slab = np.asarray(slab, dtype=float)
    NI, NJ = slab.shape[0], slab.shape[1]
    if NI == 1 and NJ == 1:
        return slab
    out = np.full(slab.shape, np.nan)
    si0, si1 = (max(0, di), min(NI, NI + di)) if NI > 1 else (0, 1)
    sj0, sj1 = (max(0, dj), min(NJ, NJ + dj)) if NJ > 1 else (0, 1)
    ddi = di if NI > 1 else 0
    ddj = dj if NJ > 1 else 0
    if si0 < si1 and sj0 < sj1:
        out[si0 - ddi : si1 - ddi, sj0 - ddj : sj1 - ddj] = slab[si0:si1, sj0:sj1]
    return out


# Audit of the one execution-model assumption that the reference-held tests do not pin: set to a list
# and every read, at a non-zero horizontal offset, of an API field that the SAME stencil call has already written is recorded
# as (stencil, offset).  tools/semantics_audit.py runs the whole reference DynamicalCore with it switched on.
AUDIT = None
# Differential execution: set to a dict and every stencil call is executed twice from the same inputs -- once as usual
# (statements see the writes of earlier statements of the call) and once with every horizontally shifted read of an API field
# served from the field's contents AT ENTRY -- and the two results are compared; {stencil: [calls, calls that differ]}.
DIFFERENTIAL = None
# Audit of region writes: set to a dict and every statement under `with horizontal(region[...])` that assigns to an API field
# records {(stencil, field): [calls, points of the region mask that lay OUTSIDE origin .. origin + domain and were therefore not
# written]}.  tools/region_write_audit.py runs the whole reference DynamicalCore with it.
REGION_AUDIT = None


class Ctx:
    stencil_name = "?"
    entry_snapshot = None

    def __init__(self, shape, origin, domain, externals):
        self.written_api = set()
        self.shape = shape
        self.origin = origin
        self.domain = domain
        self.externals = externals
        NI, NJ, NK = shape
        self.I = np.arange(NI).reshape(NI, 1, 1)
        self.J = np.arange(NJ).reshape(1, NJ, 1)
        self.dom_mask = (
            (self.I >= origin[0])
            & (self.I < origin[0] + domain[0])
            & (self.J >= origin[1])
            & (self.J < origin[1] + domain[1])
        )
        self.k0 = 0
        self.k1 = 0

    def materialize(self, v):
        if isinstance(v, Ref):
            di, dj, dk = v.off
            if AUDIT is not None and (di or dj) and id(v.arr) in self.written_api:
                AUDIT.append((self.stencil_name, (di, dj, dk), tuple(self.origin[:2]), tuple(self.domain[:2]), tuple(self.shape[:2])))
            if v.axes == ("I", "J", "K"):
                if self.entry_snapshot is not None and (di or dj) and id(v.arr) in self.entry_snapshot:
                    return _shift3(self.entry_snapshot[id(v.arr)], di, dj, self.k0, self.k1, dk)
                return _shift3(v.arr, di, dj, self.k0, self.k1, dk)
            if v.axes == ("I", "J"):
                a = np.asarray(v.arr, dtype=float)[:, :, None]
                return _shift_slab(a, di, dj)
            if v.axes == ("K",):
                a = np.asarray(v.arr, dtype=float).reshape(1, 1, -1)
                return _shift3(a, 0, 0, self.k0, self.k1, dk)
            if v.axes == ("I",):
                a = np.asarray(v.arr, dtype=float).reshape(-1, 1, 1)
                return _shift_slab(np.broadcast_to(a, (a.shape[0], 2, 1)), di, 0)[:, :1]
            if v.axes == ("J",):
                a = np.asarray(v.arr, dtype=float).reshape(1, -1, 1)
                return _shift_slab(np.broadcast_to(a, (2, a.shape[1], 1)), 0, dj)[:1]
            raise NotImplementedError(v.axes)
        return v

    def region_mask(self, specs, open_bounds_clipped=False):
        """open_bounds_clipped (the region audit): open-ended slice bounds (`:`, `a:`, `:b`) end at the launch window instead of
        running over the whole storage, so that what is left outside the window comes from EXPLICIT bounds only."""
        total = None
        oi, oj = self.origin[0], self.origin[1]
        ei, ej = oi + self.domain[0], oj + self.domain[1]
        for si, sj in specs:
            mi = _axis_mask(si, self.I, oi, ei, open_bounds_clipped)
            mj = _axis_mask(sj, self.J, oj, ej, open_bounds_clipped)
            m = mi & mj
            total = m if total is None else (total | m)
        return total


def _axis_mask(spec, idx, start, end, open_bounds_clipped=False):
    if isinstance(spec, slice):
        lo = _resolve_bound(spec.start, start, end, start if open_bounds_clipped else -(10 ** 9))
        hi = _resolve_bound(spec.stop, start, end, end if open_bounds_clipped else 10 ** 9)
        return (idx >= lo) & (idx < hi)
    p = _resolve_bound(spec, start, end, None)
    return idx == p


_MATH = {
    "abs": np.abs,
    "min": np.minimum,
    "max": np.maximum,
    "sqrt": np.sqrt,
    "exp": np.exp,
    "log": np.log,
    "sin": np.sin,
    "cos": np.cos,
    "tan": np.tan,
    "asin": np.arcsin,
    "acos": np.arccos,
    "atan": np.arctan,
    "floor": np.floor,
    "ceil": np.ceil,
    "trunc": np.trunc,
    "isnan": np.isnan,
}


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class Scope:
    def __init__(self, glob, is_function):
        self.vars = {}
        self.glob = glob
        self.is_function = is_function
        self.api = {}  # stencil level: name -> (storage, axes)


class Interp:
    def __init__(self, ctx):
        self.ctx = ctx

    # ------------------------------------------------------------------ names
    def lookup(self, name, scope):
        if name in scope.vars:
            return scope.vars[name]
        if name in self.ctx.externals:
            return self.ctx.externals[name]
        if name in scope.glob:
            return scope.glob[name]
        if name in _MATH:
            return _MATH[name]
        if name in ("True", "False"):
            return name == "True"
        import builtins

        if hasattr(builtins, name):
            return getattr(builtins, name)
        raise NameError(f"gtinterp: unknown name {name}")

    # ------------------------------------------------------------------ expressions
    def ev(self, node, scope):
        m = getattr(self, "ev_" + type(node).__name__, None)
        if m is None:
            raise NotImplementedError(ast.dump(node))
        return m(node, scope)

    def ev_Constant(self, node, scope):
        return node.value

    def ev_Name(self, node, scope):
        return self.lookup(node.id, scope)

    def ev_Attribute(self, node, scope):
        return getattr(self.ev(node.value, scope), node.attr)

    def ev_Tuple(self, node, scope):
        return tuple(self.ev(e, scope) for e in node.elts)

    def _num(self, v):
        return self.ctx.materialize(v)

    def ev_BinOp(self, node, scope):
        a = self.ev(node.left, scope)
        b = self.ev(node.right, scope)
        if isinstance(a, AxisIndex) or isinstance(b, AxisIndex):
            if isinstance(node.op, ast.Add):
                return a + b
            if isinstance(node.op, ast.Sub):
                return a - b
            raise NotImplementedError
        a, b = self._num(a), self._num(b)
        op = type(node.op)
        if op is ast.Add:
            return a + b
        if op is ast.Sub:
            return a - b
        if op is ast.Mult:
            return a * b
        if op is ast.Div:
            return a / b
        if op is ast.Pow:
            return a ** b
        if op is ast.Mod:
            return a % b
        if op is ast.FloorDiv:
            return a // b
        raise NotImplementedError(op)

    def ev_UnaryOp(self, node, scope):
        v = self.ev(node.operand, scope)
        if isinstance(node.op, ast.USub):
            if isinstance(v, (int, float)):
                return -v
            return -self._num(v)
        if isinstance(node.op, ast.UAdd):
            return self._num(v)
        if isinstance(node.op, ast.Not):
            v = self._num(v)
            if isinstance(v, (bool, np.bool_)):
                return not v
            return np.logical_not(v)
        raise NotImplementedError

    def ev_Compare(self, node, scope):
        assert len(node.ops) == 1
        a = self._num(self.ev(node.left, scope))
        b = self._num(self.ev(node.comparators[0], scope))
        op = type(node.ops[0])
        return {
            ast.Lt: lambda: a < b,
            ast.LtE: lambda: a <= b,
            ast.Gt: lambda: a > b,
            ast.GtE: lambda: a >= b,
            ast.Eq: lambda: a == b,
            ast.NotEq: lambda: a != b,
        }[op]()

    def ev_BoolOp(self, node, scope):
        vals = [self._num(self.ev(v, scope)) for v in node.values]
        out = vals[0]
        for v in vals[1:]:
            if isinstance(out, (bool, np.bool_)) and isinstance(v, (bool, np.bool_)):
                out = (out and v) if isinstance(node.op, ast.And) else (out or v)
            else:
                out = np.logical_and(out, v) if isinstance(node.op, ast.And) else np.logical_or(out, v)
        return out

    def ev_IfExp(self, node, scope):
        c = self._num(self.ev(node.test, scope))
        if isinstance(c, (bool, np.bool_)):
            return self._num(self.ev(node.body if c else node.orelse, scope))
        a = self._num(self.ev(node.body, scope))
        b = self._num(self.ev(node.orelse, scope))
        return np.where(c, a, b)

    def ev_Subscript(self, node, scope):
        base = self.ev(node.value, scope)
        sl = node.slice
        if isinstance(sl, ast.Index):  # py<3.9
            sl = sl.value
        if isinstance(base, AxisIndex):
            return base[self.ev(sl, scope)]
        idx = self.ev(sl, scope)
        if not isinstance(idx, tuple):
            idx = (idx,)
        if isinstance(base, Ref) and base.axes == ("I", "J", "K") and len(idx) == 3 and not isinstance(
                idx[2], (int, np.integer)):
            # variable (per-column) k offset, relative to the current level: field[0, 0, lev]
            assert int(idx[0]) == 0 and int(idx[1]) == 0, "variable k offset with a horizontal shift"
            ctx = self.ctx
            assert ctx.k1 - ctx.k0 == 1, "variable k offset outside a sequential computation"
            kv = np.asarray(self._num(idx[2]), dtype=float)
            kv = np.where(np.isfinite(kv), kv, 0.0).astype(np.int64)
            arr = base.arr
            NI, NJ, NK = arr.shape
            kk = np.broadcast_to(kv, (NI, NJ, 1)) + (ctx.k0 + base.off[2])
            bad = (kk < 0) | (kk >= NK)
            got = np.take_along_axis(arr, np.clip(kk, 0, NK - 1), axis=2).astype(float)
            got[bad] = np.nan
            if base.off[0] or base.off[1]:
                got = _shift_slab(got, base.off[0], base.off[1])
            return got
        idx = tuple(int(x) for x in idx)
        if isinstance(base, Ref):
            return base.shifted(idx)
        if isinstance(base, (int, float)):
            return base
        # slab value (function local / expression): only horizontal shifts
        if len(idx) == 3:
            if idx[2] != 0:
                raise NotImplementedError("k-offset on a non-field value")
            return _shift_slab(base, idx[0], idx[1])
        if len(idx) == 2:
            return _shift_slab(base, idx[0], idx[1])
        if len(idx) == 1:
            if idx[0] != 0:
                raise NotImplementedError("k-offset on a non-field value")
            return base
        raise NotImplementedError

    def ev_Call(self, node, scope):
        if isinstance(node.func, ast.Name) and node.func.id == "__INLINED":
            return self._static(node.args[0], scope)
        if isinstance(node.func, ast.Name) and node.func.id == "compile_assert":
            assert self._static(node.args[0], scope), "compile_assert failed"
            return None
        f = self.ev(node.func, scope)
        if isinstance(f, GtFunction):
            return self.call_function(f, node, scope)
        args = [self._num(self.ev(a, scope)) for a in node.args]
        if f in (np.minimum, np.maximum) and len(args) > 2:
            out = args[0]
            for a in args[1:]:
                out = f(out, a)
            return out
        if f is float or f is int or f is bool:
            return f(*args)
        if f in _MATH.values() or isinstance(f, np.ufunc):
            return f(*args)
        if f in (math.log, math.exp, math.sqrt):
            return getattr(np, f.__name__)(*args)
        raise NotImplementedError(f"call to {f}")

    def _static(self, node, scope):
        v = self.ev(node, scope)
        v = self._num(v)
        return bool(v)

    def call_function(self, f, node, scope):
        tree = f.tree
        params = [a.arg for a in tree.args.args + tree.args.kwonlyargs]
        fscope = Scope(f.fn.__globals__, True)
        vals = [self.ev(a, scope) for a in node.args]
        for p, v in zip(params, vals):
            fscope.vars[p] = v
        for kw in node.keywords:
            fscope.vars[kw.arg] = self.ev(kw.value, scope)
        # defaults are not used by the reference's gtscript functions
        try:
            self.exec_block(tree.body, fscope, None)
        except _Return as r:
            return r.value
        return None

    # ------------------------------------------------------------------ statements
    def exec_block(self, body, scope, mask):
        for st in body:
            self.exec_stmt(st, scope, mask)

    def exec_stmt(self, st, scope, mask):
        if isinstance(st, ast.Expr):
            if isinstance(st.value, ast.Constant):
                return  # docstring
            self.ev(st.value, scope)
            return
        if isinstance(st, ast.ImportFrom):
            if st.module == "__externals__":
                for a in st.names:
                    scope.vars[a.asname or a.name] = self.ctx.externals[a.name]
                return
            raise NotImplementedError("import in stencil")
        if isinstance(st, ast.Pass):
            return
        if isinstance(st, ast.Return):
            v = self.ev(st.value, scope)
            if mask is not None:
                raise NotImplementedError("return under mask")
            raise _Return(v)
        if isinstance(st, ast.Assign):
            assert len(st.targets) == 1
            self.assign(st.targets[0], self.ev(st.value, scope), scope, mask)
            return
        if isinstance(st, ast.AnnAssign):
            self.assign(st.target, self.ev(st.value, scope), scope, mask)
            return
        if isinstance(st, ast.AugAssign):
            cur = self._num(self.ev(st.target, scope))
            rhs = self._num(self.ev(st.value, scope))
            op = type(st.op)
            if op is ast.Add:
                v = cur + rhs
            elif op is ast.Sub:
                v = cur - rhs
            elif op is ast.Mult:
                v = cur * rhs
            elif op is ast.Div:
                v = cur / rhs
            else:
                raise NotImplementedError
            self.assign(st.target, v, scope, mask)
            return
        if isinstance(st, ast.If):
            c = self.ev(st.test, scope)
            c = self._num(c)
            if isinstance(c, (bool, np.bool_, int)):
                self.exec_block(st.body if c else st.orelse, scope, mask)
                return
            c = np.asarray(c, dtype=bool)
            mt = c if mask is None else (mask & c)
            mf = ~c if mask is None else (mask & ~c)
            self.exec_block(st.body, scope, mt)
            if st.orelse:
                self.exec_block(st.orelse, scope, mf)
            return
        if isinstance(st, ast.While):
            # per-column loop: columns leave the loop one by one; only columns of the compute domain take part (writes
            # outside it are dropped, so a column out there could never change its own condition)
            dom = self.ctx.dom_mask
            for _ in range(100000):
                c = np.asarray(self._num(self.ev(st.test, scope)), dtype=bool)
                m = (c & dom) if mask is None else (mask & c & dom)
                if not np.any(m):
                    break
                self.exec_block(st.body, scope, m)
            else:
                raise RuntimeError("while loop did not terminate")
            return
        if isinstance(st, ast.With):
            call = st.items[0].context_expr
            name = call.func.id
            if name == "horizontal":
                specs = []
                for a in call.args:
                    sl = a.slice
                    if isinstance(sl, ast.Index):
                        sl = sl.value
                    assert isinstance(sl, ast.Tuple) and len(sl.elts) == 2
                    specs.append(tuple(self._region_item(e, scope) for e in sl.elts))
                rm = self.ctx.region_mask(specs)
                m = rm if mask is None else (mask & rm)
                if REGION_AUDIT is not None:
                    stack = getattr(self.ctx, "region_stack", None)
                    if stack is None:
                        stack = self.ctx.region_stack = []
                    stack.append((rm, self.ctx.region_mask(specs, open_bounds_clipped=True)))
                    try:
                        self.exec_block(st.body, scope, m)
                    finally:
                        stack.pop()
                    return
                self.exec_block(st.body, scope, m)
                return
            raise NotImplementedError(f"with {name} inside a computation")
        raise NotImplementedError(ast.dump(st))

    def _region_item(self, e, scope):
        if isinstance(e, ast.Slice):
            lo = None if e.lower is None else self.ev(e.lower, scope)
            hi = None if e.upper is None else self.ev(e.upper, scope)
            return slice(lo, hi)
        return self.ev(e, scope)

    def assign(self, target, value, scope, mask):
        if isinstance(target, ast.Tuple):
            assert isinstance(value, tuple) and len(value) == len(target.elts)
            for t, v in zip(target.elts, value):
                self.assign(t, v, scope, mask)
            return
        if isinstance(target, ast.Subscript):
            # writes with explicit zero offset, e.g. w[0, 0, 0] = ...
            target = target.value
        assert isinstance(target, ast.Name)
        name = target.id
        ctx = self.ctx
        val = self._num(value)
        if scope.is_function:
            if isinstance(val, (int, float, bool, np.bool_)) and mask is None:
                scope.vars[name] = val
                return
            if mask is None:
                scope.vars[name] = val
                return
            old = scope.vars.get(name, None)
            old = np.nan if old is None else self._num(old)
            scope.vars[name] = np.where(mask, val, old)
            return
        # stencil level: API field or temporary storage
        if name in scope.api:
            arr, axes = scope.api[name]
            is_api = True
        else:
            if name not in scope.vars or not isinstance(scope.vars[name], Ref):
                arr = np.full(ctx.shape, np.nan)
                scope.vars[name] = Ref(arr, ("I", "J", "K"))
            arr = scope.vars[name].arr
            axes = ("I", "J", "K")
            is_api = False
        k0, k1 = ctx.k0, ctx.k1
        m = mask
        if is_api:
            if REGION_AUDIT is not None and getattr(ctx, "region_stack", None):
                rm_, ex_ = ctx.region_stack[-1]
                for r_, e_ in ctx.region_stack[:-1]:
                    rm_, ex_ = rm_ & r_, ex_ & e_
                outside = np.broadcast_to(rm_, ctx.dom_mask.shape) & ~ctx.dom_mask
                explicit = np.broadcast_to(ex_, ctx.dom_mask.shape) & ~ctx.dom_mask
                rec = REGION_AUDIT.setdefault((ctx.stencil_name, name), [0, 0, 0])
                rec[0] += 1
                rec[1] += int(outside.sum())
                rec[2] += int(explicit.sum())
            m = ctx.dom_mask if m is None else (m & ctx.dom_mask)
            ctx.written_api.add(id(arr))
        if axes == ("I", "J", "K"):
            tgt = arr[:, :, k0:k1]
            full = np.broadcast_to(val, tgt.shape) if not np.isscalar(val) else val
            if m is None:
                tgt[...] = full
            else:
                mm = np.broadcast_to(m, tgt.shape)
                if np.isscalar(full):
                    tgt[mm] = full
                else:
                    tgt[mm] = full[mm]
        elif axes == ("I", "J"):
            v2 = val if np.isscalar(val) else np.broadcast_to(val, (arr.shape[0], arr.shape[1], k1 - k0))[:, :, -1]
            mm = np.broadcast_to(m, (arr.shape[0], arr.shape[1], k1 - k0))[:, :, -1]
            if np.isscalar(v2):
                arr[mm] = v2
            else:
                arr[mm] = v2[mm]
        else:
            raise NotImplementedError(f"write to field with axes {axes}")


class StencilObject:
    def __init__(self, definition, externals=None, name=None, **_ignored):
        self.definition = definition
        self.externals = dict(externals or {})
        self.name = name or definition.__name__
        self.tree = _parse(definition)
        self.params = [a.arg for a in self.tree.args.args + self.tree.args.kwonlyargs]
        ann = definition.__annotations__
        written = set()
        for n in ast.walk(self.tree):
            tg = []
            if isinstance(n, ast.Assign):
                tg = n.targets
            elif isinstance(n, (ast.AugAssign, ast.AnnAssign)):
                tg = [n.target]
            for t in tg:
                for e in t.elts if isinstance(t, ast.Tuple) else [t]:
                    if isinstance(e, ast.Subscript):
                        e = e.value
                    if isinstance(e, ast.Name):
                        written.add(e.id)
        self.field_info = {}
        self.kinds = {}
        for p in self.params:
            a = ann.get(p)
            if isinstance(a, FieldType):
                acc = AccessKind.READ_WRITE if p in written else AccessKind.READ
                self.field_info[p] = FieldInfo(a.axes, acc)
                self.kinds[p] = a.axes
            else:
                self.field_info[p] = None
                self.kinds[p] = None

    # gt4py call forms -----------------------------------------------------
    def __call__(self, *args, origin=None, domain=None, validate_args=True, exec_info=None, **kwargs):
        named = dict(zip(self.params, args))
        named.update(kwargs)
        self._timed(named, origin, domain, exec_info)

    def run(self, _origin_=None, _domain_=None, exec_info=None, **kwargs):
        self._timed(kwargs, _origin_, _domain_, exec_info)

    def _timed(self, named, origin, domain, exec_info):
        """gt4py's exec_info protocol (StencilObject._call_run): per stencil class name, call counts and accumulated times."""
        import time

        t0 = time.perf_counter()
        self._execute(named, origin, domain)
        t1 = time.perf_counter()
        if exec_info is not None:
            rec = exec_info.setdefault(type(self).__name__, {"ncalls": 0, "total_call_time": 0.0, "total_run_time": 0.0})
            rec["ncalls"] += 1
            rec["call_start_time"], rec["call_end_time"] = t0, t1
            rec["run_start_time"], rec["run_end_time"] = t0, t1
            rec["total_call_time"] += t1 - t0
            rec["total_run_time"] += t1 - t0

    def _execute(self, named, origin, domain):
        if DIFFERENTIAL is None:
            return self._execute_once(named, origin, domain, False)
        arrays = {p: named[p] for p in self.params if self.kinds[p] is not None and isinstance(named[p], np.ndarray)}
        before = {p: a.copy() for p, a in arrays.items()}
        self._execute_once(named, origin, domain, True)
        alt = {p: a.copy() for p, a in arrays.items()}
        for p, a in arrays.items():
            a[...] = before[p]
        self._execute_once(named, origin, domain, False)
        name = f"{self.definition.__module__}.{self.definition.__name__}"
        rec = DIFFERENTIAL.setdefault(name, [0, 0])
        rec[0] += 1
        if any(not np.array_equal(alt[p], arrays[p], equal_nan=True) for p in arrays):
            rec[1] += 1

    def _execute_once(self, named, origin, domain, from_entry):
        if isinstance(origin, dict):
            o3 = origin.get("_all_")
            if o3 is None:
                o3 = max((tuple(v) for v in origin.values()), key=len)
        else:
            o3 = origin
        o3 = tuple(int(x) for x in o3)
        domain = tuple(int(x) for x in domain)
        shape = None
        for p in self.params:
            if self.kinds[p] == ("I", "J", "K"):
                arr = named[p]
                if shape is None:
                    shape = arr.shape
                elif arr.shape != shape:
                    raise ValueError(f"{self.name}: field {p} shape {arr.shape} != {shape}")
        if shape is None:
            raise NotImplementedError(f"{self.name}: no 3-D field argument")
        ctx = Ctx(shape, o3, domain, self.externals)
        ctx.stencil_name = f"{self.definition.__module__}.{self.definition.__name__}"
        if from_entry:
            ctx.entry_snapshot = {id(named[p]): named[p].copy() for p in self.params
                                  if self.kinds[p] == ("I", "J", "K") and isinstance(named[p], np.ndarray)}
        interp = Interp(ctx)
        scope = Scope(self.definition.__globals__, False)
        for p in self.params:
            kind = self.kinds[p]
            v = named[p]
            if kind is None:
                if isinstance(v, np.ndarray) and v.ndim == 0:
                    v = v.item()
                scope.vars[p] = v
            else:
                if not isinstance(v, np.ndarray):
                    v = np.asarray(v)
                    if not v.flags.writeable:
                        raise ValueError("non-ndarray field argument")
                scope.vars[p] = Ref(v, kind)
                scope.api[p] = (v, kind)
        kb, ke = o3[2], o3[2] + domain[2]
        with np.errstate(all="ignore"):
            for st in self.tree.body:
                if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant):
                    continue
                if isinstance(st, ast.ImportFrom):
                    interp.exec_stmt(st, scope, None)
                    continue
                assert isinstance(st, ast.With), ast.dump(st)
                self._run_computation(interp, st, scope, kb, ke)

    def _run_computation(self, interp, st, scope, kb, ke):
        items = [it.context_expr for it in st.items]
        assert items[0].func.id == "computation"
        order = items[0].args[0].id
        blocks = []
        if len(items) == 2:
            blocks.append((self._interval(items[1], kb, ke, interp, scope), st.body))
        else:
            for sub in st.body:
                assert isinstance(sub, ast.With) and sub.items[0].context_expr.func.id == "interval"
                blocks.append((self._interval(sub.items[0].context_expr, kb, ke, interp, scope), sub.body))
        ctx = interp.ctx
        if order == "PARALLEL":
            for (ka, kz), body in blocks:
                if kz <= ka:
                    continue
                ctx.k0, ctx.k1 = ka, kz
                interp.exec_block(body, scope, None)
            return
        lo = min(b[0][0] for b in blocks)
        hi = max(b[0][1] for b in blocks)
        ks = range(lo, hi) if order == "FORWARD" else range(hi - 1, lo - 1, -1)
        for k in ks:
            for (ka, kz), body in blocks:
                if ka <= k < kz:
                    ctx.k0, ctx.k1 = k, k + 1
                    interp.exec_block(body, scope, None)

    @staticmethod
    def _interval(call, kb, ke, interp, scope):
        args = call.args
        if len(args) == 1 and isinstance(args[0], ast.Constant) and args[0].value is Ellipsis:
            return kb, ke
        a = interp.ev(args[0], scope)
        b = interp.ev(args[1], scope)
        lo = kb if a is None else (kb + a if a >= 0 else ke + a)
        hi = ke if b is None else (kb + b if b >= 0 else ke + b)
        return lo, hi


def _make(definition, externals, name, build_info):
    """Like gt4py, every stencil is an instance of its own class, named after the definition + an id."""
    import hashlib
    import time

    t0 = time.perf_counter()
    src = inspect.getsource(definition) + repr(sorted((externals or {}).items(), key=lambda kv: kv[0]))
    gt_id = hashlib.sha256(src.encode()).hexdigest()[:10]
    cls = type(f"{definition.__name__}____gtinterp_{gt_id}", (StencilObject,), {"_gt_id_": gt_id})
    obj = cls(definition, externals=externals, name=name)
    if build_info is not None:
        dt = time.perf_counter() - t0
        build_info.update(parse_time=dt, module_time=0.0, codegen_time=0.0, build_time=0.0, load_time=0.0)
    return obj


def stencil(backend=None, definition=None, externals=None, name=None, build_info=None, **kw):
    if definition is None:
        return lambda fn: _make(fn, externals, name, build_info)
    return _make(definition, externals, name, build_info)


lazy_stencil = stencil


def build_modules():
    """Return {module name: module} providing the slice of the gt4py API the reference uses."""
    gts = types.ModuleType("gt4py.cartesian.gtscript")
    for k, v in dict(
        I=I, J=J, K=K, IJK=IJK, IJ=IJ, IK=IK, JK=JK, Field=Field, PARALLEL=PARALLEL, FORWARD=FORWARD,
        BACKWARD=BACKWARD, function=function, stencil=stencil, lazy_stencil=lazy_stencil,
        computation=None, interval=None, horizontal=None, region=None, __INLINED=None, compile_assert=None,
        exp=np.exp, log=np.log, sqrt=np.sqrt, sin=np.sin, cos=np.cos, tan=np.tan, asin=np.arcsin,
        acos=np.arccos, atan=np.arctan, floor=np.floor, ceil=np.ceil, trunc=np.trunc, isnan=np.isnan,
        abs=np.abs, min=np.minimum, max=np.maximum, mod=np.mod,
    ).items():
        setattr(gts, k, v)
    defs = types.ModuleType("gt4py.cartesian.definitions")
    defs.AccessKind = AccessKind
    return {"gt4py.cartesian.gtscript": gts, "gt4py.cartesian.definitions": defs}
