"""Make the read-only reference at /root/reference importable in THIS container.

Dev-container tool (never runs on the GPU box, never imported by tests/bench/product).
It (1) fabricates permissive placeholder modules for third-party packages the image
lacks (dace, cftime, f90nml, xarray, netCDF4, zarr, dacite, mpi4py and the parts of gt4py
that only matter to real compilation), (2) installs ``tools/gtinterp`` as
``gt4py.cartesian.gtscript`` so that the reference's gtscript stencil definitions are
*executed* (by our numpy interpreter) instead of compiled, (3) restores numpy-1 aliases
the reference relies on, (4) puts the reference packages on ``sys.path``.

Used by tools/make_golden.py and tools/crosscheck_oracle.py.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np


REFERENCE_ROOT = os.environ.get("PACE_REFERENCE_ROOT", "/root/reference")


class _Placeholder:
    """Callable / subscriptable / decorator-safe stand-in for anything."""

    def __init__(self, name="placeholder"):
        self._name = name

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _Placeholder) and not k:
            return a[0]
        if len(a) == 1 and isinstance(a[0], range) and not k:
            return a[0]  # dace nounroll(range(n)) must still iterate
        return _Placeholder(self._name + "()")

    def __getitem__(self, key):
        return _Placeholder(self._name + "[]")

    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        return _Placeholder(self._name + "." + key)

    def _binary(self, other=None):
        return _Placeholder(self._name)

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __neg__ = _binary
    __or__ = __ror__ = __and__ = __rand__ = _binary

    def __iter__(self):
        return iter(())

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __mro_entries__(self, bases):
        return (object,)

    def __bool__(self):
        return False


class _PlaceholderModule(types.ModuleType):
    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        return _Placeholder(self.__name__ + "." + key)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("gt4py", "dace", "cftime", "f90nml", "xarray", "netCDF4", "zarr", "dacite")

    def __init__(self, real):
        self.real = real

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name in self.real:
            m = self.real[spec.name]
        else:
            m = _PlaceholderModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import gtinterp

    real = gtinterp.build_modules()
    sys.meta_path.insert(0, _Finder(real))

    import numpy.lib._index_tricks_impl as iti

    sys.modules["numpy.lib.index_tricks"] = iti
    for alias, target in [
        ("float_", np.float64),
        ("int_", np.int64),
        ("product", np.prod),
        ("bool", np.bool_),
        ("float", float),
        ("int", int),
    ]:
        if not hasattr(np, alias):
            setattr(np, alias, target)

    for pkg in ("util", "dsl", "stencils", "fv3core", "physics", "driver"):
        p = os.path.join(REFERENCE_ROOT, pkg)
        if p not in sys.path:
            sys.path.insert(0, p)

    import gt4py.cartesian as cart
    import gt4py.cartesian.definitions  # noqa: F401
    import gt4py.cartesian.gtscript as gts
    import gt4py.storage as st

    cart.gtscript = gts
    cart.definitions = sys.modules["gt4py.cartesian.definitions"]

    def alloc(fn):
        return lambda shape, dtype=float, **kw: fn(shape, dtype=dtype)

    st.zeros, st.ones, st.empty = alloc(np.zeros), alloc(np.ones), alloc(np.empty)
    st.from_array = lambda data, dtype=None, **kw: np.array(data, dtype=dtype)

    import pace.dsl.gt4py_utils as u

    # gt4py's backend registry answers this from the backend's storage_info: "gpu" for cuda / gt:gpu / dace:gpu
    gpu_backends = ("cuda", "gt:gpu", "dace:gpu")
    u.is_gpu_backend = lambda backend: backend in gpu_backends

    # compiler-pass selection is meaningless under the interpreter
    import pace.dsl.stencil_config as sc

    sc.StencilConfig._get_oir_pipeline = classmethod(lambda cls, skip_passes: None)
    sc.is_gpu_backend = u.is_gpu_backend
