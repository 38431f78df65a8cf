"""Guard-page allocations for the emulated kernels (test infrastructure).

GPU sanitizers are not available, and an out-of-bounds READ of a kernel is silent on the CPU as long as it lands in mapped
memory -- on the GPU it is a memory access fault whenever the array happens to end at the end of an allocator segment.
`guarded()` makes every CPU tensor the host layer allocates (QuantityFactory fields, metric terms, workspaces, per-level
tables) end exactly at an inaccessible page (mode "over") or start right after one (mode "under"), so such an access faults
under emulation too; with PACE_EMU_GUARD=1 the emulator reports the kernel, block and thread (tests/emu/hip_emu.cpp).
"""
import contextlib
import ctypes
import mmap

import numpy as np

_PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_keep = []  # the mappings live as long as the process (a test child)


def guarded_array(shape, dtype, mode):
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    body = max(_PAGE, -(-nbytes // _PAGE) * _PAGE)
    mm = mmap.mmap(-1, body + 2 * _PAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    for off in (0, _PAGE + body):
        if _libc.mprotect(base + off, _PAGE, 0) != 0:  # PROT_NONE
            raise OSError(ctypes.get_errno(), "mprotect")
    offset = _PAGE + (body - nbytes if mode == "over" else 0)
    _keep.append(mm)
    return np.frombuffer(mm, dtype=dtype, count=nbytes // dtype.itemsize, offset=offset).reshape(shape)


@contextlib.contextmanager
def guarded(mode="over"):
    """Within the block torch.full / torch.zeros / torch.as_tensor make guarded CPU tensors."""
    import torch

    real_full, real_zeros, real_as_tensor = torch.full, torch.zeros, torch.as_tensor
    np_dtype = {torch.float64: np.float64, torch.int64: np.int64, torch.bool: np.bool_, torch.float32: np.float32,
                torch.int32: np.int32}

    def on_cpu(device):
        return device is None or torch.device(device).type == "cpu"

    def full(size, fill_value, *, dtype=None, device=None, **kw):
        if not on_cpu(device) or kw:
            return real_full(size, fill_value, dtype=dtype, device=device, **kw)
        a = guarded_array(tuple(size), np_dtype[dtype or torch.float64], mode)
        a[...] = fill_value
        return torch.from_numpy(a)

    def zeros(*size, dtype=None, device=None, **kw):
        if not on_cpu(device) or kw:
            return real_zeros(*size, dtype=dtype, device=device, **kw)
        if len(size) == 1 and not isinstance(size[0], int):
            size = tuple(size[0])
        return full(size, 0, dtype=dtype, device=device)

    def as_tensor(data, dtype=None, device=None):
        t = real_as_tensor(data, dtype=dtype, device=device)
        if t.device.type != "cpu" or t.dtype not in np_dtype or t.dim() == 0:
            return t
        a = guarded_array(tuple(t.shape), np_dtype[t.dtype], mode)
        a[...] = t.numpy()
        return torch.from_numpy(a)

    torch.full, torch.zeros, torch.as_tensor = full, zeros, as_tensor
    try:
        yield
    finally:
        torch.full, torch.zeros, torch.as_tensor = real_full, real_zeros, real_as_tensor
