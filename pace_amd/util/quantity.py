"""Quantity / QuantityFactory / SubtileGridSizer over device memory.

Mirrors the reference's container API (util/pace/util/quantity.py:259-615,
util/pace/util/initialization/allocator.py:31-155, sizer.py:33-155): a Quantity has ``data``
(indexed [i, j, k], halo included), ``dims``, ``units``, ``origin``, ``extent`` and ``view`` (compute
domain).  What differs is the storage: ``data`` is a strided torch view over an i-fastest buffer
[k][j][i] whose row stride is padded to 128 B, which is the layout libpace_hip.so expects.
"""
from typing import Sequence

import numpy as np
import torch

from . import constants as c

ROW_ALIGN = 16  # doubles (128 B; 32 elements of a float32 field)


def row_stride(ni: int, itemsize: int = 8) -> int:
    align = ROW_ALIGN * 8 // itemsize
    return (ni + align - 1) // align * align


class SubtileGridSizer:
    """sizer.py:33-155 for one rank of a (1, 1) layout tile."""

    def __init__(self, nx: int, ny: int, nz: int, n_halo: int = c.N_HALO_DEFAULT, extra_dim_lengths=None):
        self.nx, self.ny, self.nz, self.n_halo = nx, ny, nz, n_halo
        self.extra_dim_lengths = dict(extra_dim_lengths or {})

    @classmethod
    def from_tile_params(cls, nx_tile, ny_tile, nz, n_halo, extra_dim_lengths=None, layout=(1, 1), tile_partitioner=None,
                         tile_rank=0):
        if tuple(layout) != (1, 1):
            raise NotImplementedError("pace_amd maps one cubed-sphere tile per device: layout must be (1, 1)")
        return cls(nx_tile, ny_tile, nz, n_halo, extra_dim_lengths)

    def get_origin(self, dims: Sequence[str]):
        return tuple(self.n_halo if d in c.HORIZONTAL_DIMS else 0 for d in dims)

    def get_extent(self, dims: Sequence[str]):
        ext = {
            c.X_DIM: self.nx, c.X_INTERFACE_DIM: self.nx + 1, c.Y_DIM: self.ny, c.Y_INTERFACE_DIM: self.ny + 1,
            c.Z_DIM: self.nz, c.Z_INTERFACE_DIM: self.nz + 1,
        }
        ext.update(self.extra_dim_lengths)
        return tuple(ext[d] for d in dims)

    def get_shape(self, dims: Sequence[str]):
        # sizer.py:132-155: every horizontal axis is nx + 1 + 2*halo long, the vertical nz + 1
        out = []
        for d in dims:
            if d in c.X_DIMS:
                out.append(self.nx + 1 + 2 * self.n_halo)
            elif d in c.Y_DIMS:
                out.append(self.ny + 1 + 2 * self.n_halo)
            elif d in c.Z_DIMS:
                out.append(self.nz + 1)
            else:
                out.append(self.extra_dim_lengths[d])
        return tuple(out)


class _View:
    def __init__(self, q):
        self._q = q

    def _slices(self):
        return tuple(slice(o, o + e) for o, e in zip(self._q.origin, self._q.extent))

    def __getitem__(self, idx):
        return self._q.data[self._slices()][idx]

    def __setitem__(self, idx, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(np.asarray(value), dtype=self._q.data.dtype, device=self._q.data.device)
        self._q.data[self._slices()][idx] = value


class Quantity:
    def __init__(self, data: torch.Tensor, dims, units, origin=None, extent=None, base=None):
        self._data = data
        self._base = base if base is not None else data
        self.dims = tuple(dims)
        self.units = units
        self.origin = tuple(origin) if origin is not None else (0,) * len(self.dims)
        self.extent = tuple(extent) if extent is not None else tuple(s - o for s, o in zip(data.shape, self.origin))
        self.view = _View(self)

    @property
    def data(self) -> torch.Tensor:
        return self._data

    @property
    def ptr(self) -> int:
        """Device address of element (0, 0, 0) -- what the C ABI takes."""
        return self._base.data_ptr()

    @property
    def shape(self):
        return tuple(self._data.shape)

    def numpy(self) -> np.ndarray:
        return self._data.detach().cpu().numpy()

    def set(self, array):
        self._data[...] = torch.as_tensor(np.asarray(array), dtype=self._data.dtype, device=self._data.device)

    def swap_storage(self, other: "Quantity"):
        """Exchange the device buffers of two Quantities of the same layout (an extension: d_sw writes the four scalars it
        transports to buffers of their own -- include/pace_hip.h pace_dsw_config_t -- and the operator swaps them in, so that the
        caller's Quantity objects hold the new values as if they had been updated in place).  Tensors taken from ``data`` BEFORE
        the swap keep pointing at the old buffer."""
        if self._data.shape != other._data.shape or self._data.stride() != other._data.stride() or self._data.dtype != other._data.dtype:
            raise ValueError("swap_storage needs two quantities of the same layout")
        self._data, other._data = other._data, self._data
        self._base, other._base = other._base, self._base
        self._generation = self.generation + 1
        other._generation = other.generation + 1

    _generation = 0

    @property
    def generation(self) -> int:
        """How often ``swap_storage`` has replaced this Quantity's storage.  Whoever keeps something derived from the storage
        across calls (a device pointer, a tensor taken from ``data``, a captured graph) records the generation with it and
        checks it before use: ``HaloUpdater.wait`` does, for the fields it was started on."""
        return self._generation

    def transpose(self, target_dims: Sequence[str]) -> "Quantity":
        """quantity.py:518-560: the same storage seen with its dimensions in another order (a view; what the reference's
        checkpoint calls use to hand [x, z, y] "Fortran data" to the checkpointer)."""
        target_dims = tuple(target_dims)
        if sorted(target_dims) != sorted(self.dims):
            raise ValueError(f"cannot transpose dims {self.dims} to {target_dims}")
        order = [self.dims.index(d) for d in target_dims]
        return Quantity(self._data.permute(*order), target_dims, self.units, origin=[self.origin[n] for n in order],
                        extent=[self.extent[n] for n in order], base=self._base)

    def __repr__(self):
        return f"Quantity(dims={self.dims}, units={self.units!r}, shape={self.shape}, origin={self.origin}, extent={self.extent})"


class QuantityFactory:
    def __init__(self, sizer: SubtileGridSizer, device="cuda", dtype=torch.float64):
        """``dtype``: storage type of every float field -- torch.float64 (libpace_hip.so) or torch.float32 (libpace_hip_f32.so;
        an extension: the checked-out reference has no 32-bit mode, dsl/pace/dsl/typing.py:24)."""
        self.sizer = sizer
        self.device = torch.device(device)
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("fields are float64 or float32")
        self.real = dtype
        self.itemsize = 8 if dtype == torch.float64 else 4

    @classmethod
    def from_backend(cls, sizer, backend: str, dtype=torch.float64):
        """Reference signature (allocator.py:42-51); the only backend is the HIP one."""
        device = "cpu" if backend in ("emu", "cpu-emulation") else "cuda"
        return cls(sizer, device, dtype)

    def _allocate(self, fill, dims, units, dtype):
        dims = tuple(dims)
        shape = self.sizer.get_shape(dims)
        tdtype = {float: self.real, int: torch.int64, bool: torch.bool}.get(dtype, dtype)
        if len(dims) == 3:
            ni, nj, nk = shape
            base = torch.full((nk, nj, row_stride(ni, self.itemsize)), fill, dtype=tdtype, device=self.device)
            data = base.permute(2, 1, 0)[:ni]
        elif len(dims) == 2:
            ni, nj = shape
            base = torch.full((nj, row_stride(ni, self.itemsize)), fill, dtype=tdtype, device=self.device)
            data = base.permute(1, 0)[:ni]
        elif len(dims) == 1:
            base = torch.full(shape, fill, dtype=tdtype, device=self.device)
            data = base
        else:
            raise NotImplementedError(dims)
        return Quantity(data, dims, units, origin=self.sizer.get_origin(dims), extent=self.sizer.get_extent(dims), base=base)

    def zeros(self, dims, units, dtype=float):
        return self._allocate(0, dims, units, dtype)

    def empty(self, dims, units, dtype=float):
        return self._allocate(0, dims, units, dtype)

    def ones(self, dims, units, dtype=float):
        return self._allocate(1, dims, units, dtype)

    def from_array(self, data, dims, units):
        q = self.empty(dims, units, dtype=float)
        q.set(data)
        return q

    def get_quantity_halo_spec(self, dims, n_halo=None, dtype=float):
        """allocator.py:132-155: the memory description a HaloUpdater is built from."""
        from .halo import QuantityHaloSpec

        return QuantityHaloSpec(self.sizer.n_halo if n_halo is None else n_halo, (1, self.row_stride, self.level_stride), self.itemsize,
                                self.sizer.get_shape(dims), self.sizer.get_origin(dims), self.sizer.get_extent(dims), tuple(dims),
                                None, self.real)

    @property
    def row_stride(self):
        return row_stride(self.sizer.nx + 1 + 2 * self.sizer.n_halo, self.itemsize)

    @property
    def level_stride(self):
        return self.row_stride * (self.sizer.ny + 1 + 2 * self.sizer.n_halo)
