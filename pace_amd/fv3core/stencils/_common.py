"""Shared plumbing for the class-level operator bindings."""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ...util.grid import geom_struct


def dptr(x):
    """Device address of element (0,0,0) of a Quantity / tensor, or None."""
    if x is None:
        return None
    t = x.data if hasattr(x, "dims") else x
    return t.data_ptr()


def check_layout(geom, *fields):
    for f in fields:
        if f is None:
            continue
        t = f.data if hasattr(f, "dims") else f
        if t.dim() == 3 and tuple(t.stride()) != (1, geom.sj, geom.sk):
            raise ValueError(f"field layout {tuple(t.stride())} does not match (1, {geom.sj}, {geom.sk}); "
                             "allocate fields with pace_amd.util.QuantityFactory")
        if t.dtype != getattr(geom, "_real", torch.float64):
            raise ValueError(f"field dtype {t.dtype} does not match the library's storage type "
                             f"{getattr(geom, '_real', torch.float64)} (dsl/pace/dsl/typing.py:24)")


class Operator:
    """Base: remembers the library, geometry struct, metrics struct and the stream to launch on."""

    def __init__(self, stencil_factory, quantity_factory, grid_data=None):
        self.lib = stencil_factory.lib
        self.grid_indexing = stencil_factory.grid_indexing
        self._qf = quantity_factory
        self._geom = geom_struct(quantity_factory)
        if self.lib.real_bytes != quantity_factory.itemsize:
            raise _lib.PaceError(f"the library stores {self.lib.real_bytes}-byte reals, the quantity factory allocates "
                                 f"{quantity_factory.real}: load the matching build (pace_amd._lib.load(precision=...))")
        self._grid_data = grid_data
        self._met = grid_data.c_struct() if grid_data is not None else None
        self._emu = quantity_factory.device.type == "cpu"
        if self._emu and "emulation" not in self.lib.version():
            raise _lib.PaceError("CPU tensors can only be used with the emulation test library")
        if (not self._emu) and "emulation" in self.lib.version():
            raise _lib.PaceError("the emulation test library cannot run on device tensors")

    def stream(self):
        if self._emu:
            return None
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(self, name, *args):
        self.lib.call(name, C.byref(self._geom), *args)


def host_column(values, nk):
    """Contiguous float64 host array of one value per layer."""
    if hasattr(values, "dims"):
        values = values.numpy()
    elif torch.is_tensor(values):
        values = values.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64)[:nk])
    return a


def expand_externals(values, nk):
    """The reference bakes (level 0, 1, 2, >=3) values into externals nord0..nord3
    (delnflux.py:41-82,1129-1134); expand to one value per level."""
    v = host_column(values, max(4, min(nk, len(values))))
    out = np.full(nk, v[3] if len(v) > 3 else v[-1], dtype=np.float64)
    out[: min(3, nk)] = v[: min(3, nk)]
    return out
