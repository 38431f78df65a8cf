"""Cubed-sphere halo exchange between tiles that each live on their own device.

Mirrors the reference's L2 API (util/pace/util/halo_updater.py:29-303,359-536, communicator.py:331-560,
quantity.py:55-66, dsl/pace/dsl/dace/wrapped_halo_exchange.py:9-73): ``QuantityHaloSpec``,
``CubedSphereCommunicator.get_scalar_halo_updater / get_vector_halo_updater / halo_update / vector_halo_update /
synchronize_vector_interfaces``, ``HaloUpdater.start / wait / update``, ``WrappedHaloUpdater``.

What is different underneath: the reference slices the boundary (``_boundary_utils.py:58-95``), rotates it on the host
side of the message (``rotate.py:4-50``, ``halo_data_transformer.py:387-461``) and sends one buffer per neighbour
through mpi4py.  Here every strip is described once, at construction, by an affine index map in the RECEIVER's
orientation (derived below by pushing index grids through the same slice-and-rotate steps, then checked to be
exactly affine); one HIP launch packs every strip of every field of the updater (the rotation happens in the pack),
one grouped RCCL send/recv moves the four messages, one launch unpacks.  Nothing synchronises the device.
"""
import ctypes as C
import dataclasses
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from . import constants as c
from .partitioner import EDGES, CubedSpherePartitioner, facing_edge, tile_neighbour

WEST, EAST, NORTH, SOUTH = c.WEST, c.EAST, c.NORTH, c.SOUTH


@dataclasses.dataclass
class QuantityHaloSpec:
    """quantity.py:55-66."""

    n_points: int
    strides: tuple
    itemsize: int
    shape: tuple
    origin: tuple
    extent: tuple
    dims: tuple
    numpy_module: object = None
    dtype: object = float


def spec_of(quantity, n_points: int) -> QuantityHaloSpec:
    return QuantityHaloSpec(n_points, tuple(quantity.data.stride()), quantity.data.element_size(), tuple(quantity.data.shape),
                            quantity.origin, quantity.extent, quantity.dims, None, quantity.data.dtype)


# ---------------------------------------------------------------------------------------------------------------
# geometry: boundary slices and rotations on index grids
# ---------------------------------------------------------------------------------------------------------------
def _stagger(dims):
    return (1 if c.X_INTERFACE_DIM in dims else 0), (1 if c.Y_INTERFACE_DIM in dims else 0)


def _boundary_slices(n, edge, n_pts, interior, xi, yi, n_halo):
    """_boundary_utils.get_boundary_slice for an edge boundary of a tile-sized field with x extent n + xi and y
    extent n + yi: along the edge the compute extent, across it n_pts points -- inside the tile (skipping the
    shared interface row of staggered fields) when ``interior``, else in the halo."""
    o = n_halo

    def across(ext, overlap, at_start):
        if at_start:
            return (o + overlap, o + overlap + n_pts) if interior else (o - n_pts, o)
        e = o + ext
        return (e - overlap - n_pts, e - overlap) if interior else (e, e + n_pts)

    ex, ey = n + xi, n + yi
    if edge == WEST:
        return across(ex, xi, True), (o, o + ey)
    if edge == EAST:
        return across(ex, xi, False), (o, o + ey)
    if edge == SOUTH:
        return (o, o + ex), across(ey, yi, True)
    return (o, o + ex), across(ey, yi, False)


def _rot(a, nrot):
    """rotate.rotate_scalar_data for arrays whose axes are (x, y)."""
    nrot %= 4
    if nrot == 1:
        return np.rot90(a, axes=(1, 0))
    if nrot == 3:
        return np.rot90(a, axes=(0, 1))
    if nrot == 2:
        return a[::-1, ::-1]
    return a


@dataclasses.dataclass
class _Strip:
    """One field's strip of one message, in receiver orientation."""

    field: int  # index into the concatenated quantity list given to start()
    i0: int
    j0: int
    di_a: int
    dj_a: int
    di_b: int
    dj_b: int
    na: int
    nb: int
    nk: int
    sign: float
    offset: int = 0  # doubles from the start of the message

    @property
    def size(self):
        return self.na * self.nb * self.nk


def _affine(si, sj):
    """Fit (i, j) = (i0, j0) + a * (di_a, dj_a) + b * (di_b, dj_b) to index grids and verify it is exact."""
    na, nb = si.shape
    i0, j0 = int(si[0, 0]), int(sj[0, 0])
    di_a = int(si[1, 0] - si[0, 0]) if na > 1 else 0
    dj_a = int(sj[1, 0] - sj[0, 0]) if na > 1 else 0
    di_b = int(si[0, 1] - si[0, 0]) if nb > 1 else 0
    dj_b = int(sj[0, 1] - sj[0, 0]) if nb > 1 else 0
    a, b = np.meshgrid(np.arange(na), np.arange(nb), indexing="ij")
    assert (si == i0 + a * di_a + b * di_b).all() and (sj == j0 + a * dj_a + b * dj_b).all(), "halo map is not affine"
    return i0, j0, di_a, dj_a, di_b, dj_b


def _send_strip(n, n_halo, ni, edge, rot, to_edge, n_pts, src_dims, dst_dims, nk, field, sign):
    """Strip of the message to the neighbour across ``edge``: read from a local field with staggering src_dims,
    laid out for the neighbour's halo (its edge ``to_edge``) of a field with staggering dst_dims."""
    sxi, syi = _stagger(src_dims)
    (x0, x1), (y0, y1) = _boundary_slices(n, edge, n_pts, True, sxi, syi, n_halo)
    ii, jj = np.meshgrid(np.arange(ni), np.arange(ni), indexing="ij")
    si, sj = _rot(ii[x0:x1, y0:y1], -rot), _rot(jj[x0:x1, y0:y1], -rot)
    dxi, dyi = _stagger(dst_dims)
    (rx0, rx1), (ry0, ry1) = _boundary_slices(n, to_edge, n_pts, False, dxi, dyi, n_halo)
    if si.shape != (rx1 - rx0, ry1 - ry0):
        raise ValueError(f"halo strip {si.shape} does not fit the neighbour's {(rx1 - rx0, ry1 - ry0)} (dims {src_dims} -> {dst_dims})")
    return _Strip(field, *_affine(si, sj), si.shape[0], si.shape[1], nk, sign)


def _recv_strip(n, n_halo, edge, n_pts, dims, nk, field):
    xi, yi = _stagger(dims)
    (rx0, rx1), (ry0, ry1) = _boundary_slices(n, edge, n_pts, False, xi, yi, n_halo)
    return _Strip(field, rx0, ry0, 1, 0, 0, 1, rx1 - rx0, ry1 - ry0, nk, 1.0)


def _nk_of(spec):
    for d, e in zip(spec.dims, spec.extent):
        if d in c.Z_DIMS:
            return int(e)
    return 1


class _Messages:
    """Send and receive strips of one exchange pattern, their buffers and ctypes descriptor tables."""

    def __init__(self, device, peers_send, peers_recv, dtype=torch.float64):
        self.device = device
        self.dtype = dtype if isinstance(dtype, torch.dtype) else torch.float64
        self.itemsize = 8 if self.dtype == torch.float64 else 4
        self.send = {p: [] for p in peers_send}  # peer rank -> [strips]
        self.recv = {p: [] for p in peers_recv}
        self._ready = False

    def finalize(self):
        self.sendbuf, self.recvbuf = {}, {}
        for table, bufs in ((self.send, self.sendbuf), (self.recv, self.recvbuf)):
            for peer, strips in table.items():
                off = 0
                for s in strips:
                    s.offset = off
                    off += s.size
                bufs[peer] = torch.zeros(max(off, 1), dtype=self.dtype, device=self.device)
        self._desc_cache = {}
        self._ready = True

    def descriptors(self, ptrs):
        """(pack table, n, unpack table, n) for the given tuple of field base addresses."""
        hit = self._desc_cache.get(ptrs)
        if hit is not None:
            return hit
        out = []
        for table, bufs in ((self.send, self.sendbuf), (self.recv, self.recvbuf)):
            strips = [(peer, s) for peer, ss in table.items() for s in ss]
            arr = (_lib.HaloDesc * max(len(strips), 1))()
            for d, (peer, s) in zip(arr, strips):
                d.field = ptrs[s.field]
                d.buf = bufs[peer].data_ptr() + self.itemsize * s.offset
                d.i0, d.j0, d.di_a, d.dj_a, d.di_b, d.dj_b = s.i0, s.j0, s.di_a, s.dj_a, s.di_b, s.dj_b
                d.na, d.nb, d.nk, d.sign = s.na, s.nb, s.nk, s.sign
            out += [arr, len(strips)]
        self._desc_cache[ptrs] = tuple(out)
        return self._desc_cache[ptrs]


class HaloUpdater:
    """halo_updater.py:29-303.  Built once per group of fields, reused every substep."""

    def __init__(self, communicator: "CubedSphereCommunicator", tag: int, specs_x: Sequence[QuantityHaloSpec],
                 specs_y: Optional[Sequence[QuantityHaloSpec]] = None):
        self._cube = communicator
        self._tag = tag
        self._inflight = None
        specs_x, specs_y = list(specs_x), list(specs_y or [])
        vector = len(specs_y) > 0
        if vector and len(specs_x) != len(specs_y):
            raise ValueError("vector halo update needs as many y as x quantities")
        self._n_x, self._n_y = len(specs_x), len(specs_y)
        first = specs_x[0]
        xi0, yi0 = _stagger(first.dims)
        n = first.extent[0] - xi0
        n_halo = first.origin[0]
        ni = first.shape[0]
        self._geom = _lib.Geom(n, first.shape[2] - 1, first.strides[1], 0, first.strides[2])
        tile = communicator.rank
        topo = communicator.partitioner
        nbrs = {e: topo.neighbour(tile, e) for e in EDGES}
        msgs = _Messages(communicator.device, [nbrs[e][0] for e in EDGES], [nbrs[e][0] for e in EDGES], first.dtype)
        # One message per peer.  Its strips are ordered by the SENDER's edges, so the receiving side walks the peer's
        # edges too (on the cubed sphere two tiles share one edge and the order is moot; in a two-tile ring they share four).
        recv_edges = {}  # (peer, position in the peer's message) -> my edge
        for peer in set(nb[0] for nb in nbrs.values()):
            order = [topo.arrival_edge(peer, e2) for e2 in EDGES if topo.neighbour(peer, e2)[0] == tile]
            recv_edges[peer] = order
        sent = {peer: 0 for peer in recv_edges}
        for e in EDGES:
            to, rot = nbrs[e]
            to_edge = topo.arrival_edge(tile, e)
            k = (-rot) % 4
            re = recv_edges[to][sent[to]]  # my edge whose halo the matching strip of the peer's message fills
            sent[to] += 1
            for f, sx in enumerate(specs_x):
                nk = _nk_of(sx)
                if not vector:
                    msgs.send[to].append(_send_strip(n, n_halo, ni, e, rot, to_edge, sx.n_points, sx.dims, sx.dims, nk, f, 1.0))
                    msgs.recv[to].append(_recv_strip(n, n_halo, re, sx.n_points, sx.dims, nk, f))
                    continue
                sy = specs_y[f]
                fx, fy = f, self._n_x + f
                # rotate_vector_data (rotate.py:30-50): what arrives as the x (y) component is +-x or +-y of the sender
                x_src, x_sign = {0: (fx, 1.0), 1: (fy, 1.0), 2: (fx, -1.0), 3: (fy, -1.0)}[k]
                y_src, y_sign = {0: (fy, 1.0), 1: (fx, -1.0), 2: (fy, -1.0), 3: (fx, 1.0)}[k]
                dims_of = {fx: sx.dims, fy: sy.dims}
                msgs.send[to].append(_send_strip(n, n_halo, ni, e, rot, to_edge, sx.n_points, dims_of[x_src], sx.dims, nk, x_src, x_sign))
                msgs.send[to].append(_send_strip(n, n_halo, ni, e, rot, to_edge, sy.n_points, dims_of[y_src], sy.dims, nk, y_src, y_sign))
                msgs.recv[to].append(_recv_strip(n, n_halo, re, sx.n_points, sx.dims, nk, fx))
                msgs.recv[to].append(_recv_strip(n, n_halo, re, sy.n_points, sy.dims, nk, fy))
        msgs.finalize()
        self._msgs = msgs

    def force_finalize_on_wait(self):
        pass

    def message_bytes(self):
        """{peer rank: bytes of the ONE message this updater sends to it per update} (each of the up to four neighbours gets its
        strips of all the group's fields in one buffer: halo_updater.py:217-303 sends one per field and edge)."""
        return {int(peer): int(sum(s.size for s in strips) * self._msgs.itemsize) for peer, strips in self._msgs.send.items()}

    def start(self, quantities_x: List, quantities_y: Optional[List] = None):
        if self._inflight is not None:
            raise RuntimeError("Previous exchange hasn't been properly finished."
                               "E.g. previous start() call didn't have a wait() call.")
        qs = list(quantities_x) + list(quantities_y or [])
        if len(qs) != self._n_x + self._n_y:
            raise ValueError(f"updater was built for {self._n_x}+{self._n_y} quantities, got {len(qs)}")
        ptrs = tuple(q.ptr for q in qs)
        pack, npack, unpack, nunpack = self._msgs.descriptors(ptrs)
        cube = self._cube
        cube.lib.call("pace_halo_pack", C.byref(self._geom), pack, npack, cube.stream())
        req = cube.comm.exchange([(b, p) for p, b in self._msgs.sendbuf.items()], [(b, p) for p, b in self._msgs.recvbuf.items()],
                                 tag=self._tag)
        self._inflight = (req, unpack, nunpack, [(q, getattr(q, "generation", 0)) for q in qs])

    def wait(self):
        if self._inflight is None:
            raise RuntimeError('Halo update "wait" call before "start"')
        req, unpack, nunpack, started_on = self._inflight
        for q, gen in started_on:  # the unpack table holds the addresses the fields had at start()
            if getattr(q, "generation", 0) != gen:
                raise RuntimeError("the storage of a Quantity was swapped (Quantity.swap_storage) between start() and wait() of its "
                                   "halo update: the received halo would be written to the buffer it no longer owns")
        req.wait()
        self._cube.lib.call("pace_halo_unpack", C.byref(self._geom), unpack, nunpack, self._cube.stream())
        self._inflight = None

    def update(self, quantities_x: List, quantities_y: Optional[List] = None):
        self.start(quantities_x, quantities_y)
        self.wait()


class VectorInterfaceHaloUpdater:
    """halo_updater.py:359-536: the south row of x and the west column of y overwrite the copies of those shared
    interface points held by the neighbouring tiles (their north row / east column)."""

    def __init__(self, communicator: "CubedSphereCommunicator", tag: int, spec_x: QuantityHaloSpec, spec_y: QuantityHaloSpec):
        self._cube = communicator
        self._tag = tag
        if _stagger(spec_x.dims) != (0, 1) or _stagger(spec_y.dims) != (1, 0):
            raise ValueError("x must be on (x, y_interface) and y on (x_interface, y)")
        n, o = spec_x.extent[0], spec_x.origin[0]
        nk = _nk_of(spec_x)
        self._geom = _lib.Geom(n, spec_x.shape[2] - 1, spec_x.strides[1], 0, spec_x.strides[2])
        tile = communicator.rank
        if isinstance(communicator.partitioner, CubedSpherePartitioner):
            nb = tile_neighbour
        else:  # (the ring of tiles bench.py uses at rank counts that cannot form a cube: no rotations, both go to the previous tile)
            nb = communicator.partitioner.neighbour
        (to_s, rot_s), (to_w, rot_w) = nb(tile, SOUTH), nb(tile, WEST)
        (from_n, _), (from_e, _) = nb(tile, NORTH), nb(tile, EAST)
        msgs = _Messages(communicator.device, [to_s, to_w], [from_n, from_e], spec_x.dtype)
        # south row of x (field 0): reversed if the neighbour's axis runs the other way, sign from the vector rotation
        rev = (-rot_s) % 4 == 1
        sign = -1.0 if rot_s in (3, 2) else 1.0
        msgs.send[to_s].append(_Strip(0, o + n - 1 if rev else o, o, -1 if rev else 1, 0, 0, 0, n, 1, nk, sign))
        rev = (-rot_w) % 4 == 3
        sign = -1.0 if rot_w in (1, 2) else 1.0
        msgs.send[to_w].append(_Strip(1, o, o + n - 1 if rev else o, 0, -1 if rev else 1, 0, 0, n, 1, nk, sign))
        msgs.recv[from_n].append(_Strip(0, o, o + n, 1, 0, 0, 0, n, 1, nk, 1.0))
        msgs.recv[from_e].append(_Strip(1, o + n, o, 0, 1, 0, 0, n, 1, nk, 1.0))
        msgs.finalize()
        self._msgs = msgs

    def update(self, x_quantity, y_quantity):
        cube = self._cube
        pack, npack, unpack, nunpack = self._msgs.descriptors((x_quantity.ptr, y_quantity.ptr))
        cube.lib.call("pace_halo_pack", C.byref(self._geom), pack, npack, cube.stream())
        req = cube.comm.exchange([(b, p) for p, b in self._msgs.sendbuf.items()], [(b, p) for p, b in self._msgs.recvbuf.items()],
                                 tag=self._tag)
        req.wait()
        cube.lib.call("pace_halo_unpack", C.byref(self._geom), unpack, nunpack, cube.stream())


class _TileView:
    rank = 0


class CubedSphereCommunicator:
    """communicator.py:680-760 for a (1, 1) layout: rank == tile index."""

    def __init__(self, comm, partitioner: Optional[CubedSpherePartitioner] = None, device=None, lib=None, force_cpu=False,
                 timer=None):
        self.comm = comm
        self.partitioner = partitioner or CubedSpherePartitioner()
        if comm.Get_size() != self.partitioner.total_ranks:
            raise ValueError(f"was given a partitioner for {self.partitioner.total_ranks} ranks but a comm object with only "
                             f"{comm.Get_size()} ranks, are we running with mpi and the correct number of ranks?")
        self.device = torch.device(device if device is not None else "cuda")
        self.lib = lib if lib is not None else _lib.load()
        self.tile = _TileView()
        self._last_tag = 0
        self._interface_updaters = {}
        self._adhoc = {}

    @property
    def rank(self) -> int:
        return self.comm.Get_rank()

    def stream(self):
        if self.device.type == "cpu":
            return None
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _get_halo_tag(self) -> int:
        self._last_tag += 1
        return self._last_tag

    def get_scalar_halo_updater(self, specifications: List[QuantityHaloSpec]) -> HaloUpdater:
        if len(specifications) == 0:
            raise RuntimeError("Cannot create updater with specifications list")
        if specifications[0].n_points == 0:
            raise ValueError("cannot perform a halo update on zero halo points")
        return HaloUpdater(self, self._get_halo_tag(), specifications)

    def get_vector_halo_updater(self, specifications_x: List[QuantityHaloSpec], specifications_y: List[QuantityHaloSpec]) -> HaloUpdater:
        if len(specifications_x) == 0 and len(specifications_y) == 0:
            raise RuntimeError("Cannot create updater with empty specifications list")
        if specifications_x[0].n_points == 0 and specifications_y[0].n_points == 0:
            raise ValueError("Cannot perform a halo update on zero halo points")
        return HaloUpdater(self, self._get_halo_tag(), specifications_x, specifications_y)

    # one-shot forms (communicator.py:331-519); the updater is cached per field group
    def start_halo_update(self, quantity, n_points: int) -> HaloUpdater:
        qs = list(quantity) if isinstance(quantity, (list, tuple)) else [quantity]
        key = ("s", n_points) + tuple((q.dims, q.shape) for q in qs)
        if key not in self._adhoc:
            self._adhoc[key] = self.get_scalar_halo_updater([spec_of(q, n_points) for q in qs])
        self._adhoc[key].start(qs)
        return self._adhoc[key]

    def halo_update(self, quantity, n_points: int):
        self.start_halo_update(quantity, n_points).wait()

    def start_vector_halo_update(self, x_quantity, y_quantity, n_points: int) -> HaloUpdater:
        xs = list(x_quantity) if isinstance(x_quantity, (list, tuple)) else [x_quantity]
        ys = list(y_quantity) if isinstance(y_quantity, (list, tuple)) else [y_quantity]
        key = ("v", n_points) + tuple((q.dims, q.shape) for q in xs + ys)
        if key not in self._adhoc:
            self._adhoc[key] = self.get_vector_halo_updater([spec_of(q, n_points) for q in xs], [spec_of(q, n_points) for q in ys])
        self._adhoc[key].start(xs, ys)
        return self._adhoc[key]

    def vector_halo_update(self, x_quantity, y_quantity, n_points: int):
        self.start_vector_halo_update(x_quantity, y_quantity, n_points).wait()

    def synchronize_vector_interfaces(self, x_quantity, y_quantity):
        key = (x_quantity.dims, x_quantity.shape, y_quantity.dims)
        if key not in self._interface_updaters:
            self._interface_updaters[key] = VectorInterfaceHaloUpdater(self, self._get_halo_tag(), spec_of(x_quantity, 1),
                                                                      spec_of(y_quantity, 1))
        self._interface_updaters[key].update(x_quantity, y_quantity)


class WrappedHaloUpdater:
    """wrapped_halo_exchange.py:9-73: looks the quantities up by name in a state object / dict at call time."""

    def __init__(self, updater: Optional[HaloUpdater], state, qty_x_names: List[str], qty_y_names: Optional[List[str]] = None,
                 comm: Optional[CubedSphereCommunicator] = None):
        self._updater = updater
        self._state = state
        self._qtx_x_names = qty_x_names
        self._qtx_y_names = qty_y_names
        self._comm = comm

    def _get(self, names):
        if isinstance(self._state, dict):
            return [self._state[x] for x in names]
        return [getattr(self._state, x) for x in names]

    def start(self):
        if self._qtx_y_names is None:
            self._updater.start(self._get(self._qtx_x_names))
        else:
            self._updater.start(self._get(self._qtx_x_names), self._get(self._qtx_y_names))

    def wait(self):
        self._updater.wait()

    def update(self):
        self.start()
        self.wait()

    def interface(self):
        assert len(self._qtx_x_names) == 1
        assert len(self._qtx_y_names) == 1
        self._comm.synchronize_vector_interfaces(self._get(self._qtx_x_names)[0], self._get(self._qtx_y_names)[0])
