"""Cubed-sphere tile topology for a (1, 1) layout: one tile per rank, six ranks
(util/pace/util/partitioner.py:31-38,365-606).  Only edge neighbours exist; tile corners are never exchanged
(partitioner.py:525-590)."""
from .constants import EAST, NORTH, SOUTH, WEST

EDGES = (WEST, EAST, NORTH, SOUTH)


def tile_neighbour(tile: int, edge: int):
    """(neighbour tile, n_clockwise_rotations of the neighbour's axes relative to ours) -- partitioner.py:425-523."""
    if tile % 2 == 0:
        return {WEST: ((tile - 2) % 6, 1), EAST: ((tile + 1) % 6, 0), NORTH: ((tile + 2) % 6, 3), SOUTH: ((tile - 1) % 6, 0)}[edge]
    return {WEST: ((tile - 1) % 6, 0), EAST: ((tile + 2) % 6, 1), NORTH: ((tile + 1) % 6, 0), SOUTH: ((tile - 2) % 6, 3)}[edge]


def facing_edge(tile: int, to_tile: int) -> int:
    """The edge of ``tile`` that borders ``to_tile``."""
    for e in EDGES:
        if tile_neighbour(tile, e)[0] == to_tile:
            return e
    raise ValueError(f"tiles {tile} and {to_tile} are not neighbours")


class TilePartitioner:
    def __init__(self, layout=(1, 1)):
        if tuple(layout) != (1, 1):
            raise NotImplementedError("pace_amd maps one cubed-sphere tile per device: layout must be (1, 1)")
        self.layout = (1, 1)

    @property
    def total_ranks(self):
        return 1


class CubedSpherePartitioner:
    def __init__(self, tile: TilePartitioner = None):
        self.tile = tile or TilePartitioner()
        self.layout = self.tile.layout

    @property
    def total_ranks(self):
        return 6

    def tile_index(self, rank: int) -> int:
        return rank

    def neighbour(self, rank: int, edge: int):
        """(neighbour rank, n_clockwise_rotations) across ``edge``."""
        return tile_neighbour(rank, edge)

    def arrival_edge(self, rank: int, edge: int) -> int:
        """The neighbour's edge at which what ``rank`` sends across ``edge`` arrives."""
        return facing_edge(tile_neighbour(rank, edge)[0], rank)

    def boundary(self, boundary_type: int, rank: int):
        to, rot = tile_neighbour(rank, boundary_type)
        return _Boundary(rank, to, rot, boundary_type)


class RingPartitioner:
    """NOT a reference topology: ``n`` tiles in a periodic ring, every tile exchanging its four edges without rotation
    (west <-> east and south <-> north with the previous / next tile).  The strips, message sizes and the pack /
    exchange / unpack path are exactly those of the cubed sphere; bench.py uses it to keep the halo exchange inside the
    measured step at rank counts that cannot form a cube (2, 4, 8), and the tests use it for world-size-2 runs."""

    _OPPOSITE = {WEST: EAST, EAST: WEST, SOUTH: NORTH, NORTH: SOUTH}

    def __init__(self, n: int):
        if n < 2:
            raise ValueError("a ring needs at least two tiles")
        self.n = n
        self.layout = (1, 1)

    @property
    def total_ranks(self):
        return self.n

    def tile_index(self, rank: int) -> int:
        return rank

    def neighbour(self, rank: int, edge: int):
        return ((rank - 1) % self.n if edge in (WEST, SOUTH) else (rank + 1) % self.n), 0

    def arrival_edge(self, rank: int, edge: int) -> int:
        return self._OPPOSITE[edge]

    def boundary(self, boundary_type: int, rank: int):
        to, rot = self.neighbour(rank, boundary_type)
        return _Boundary(rank, to, rot, boundary_type)


class _Boundary:
    def __init__(self, from_rank, to_rank, n_clockwise_rotations, boundary_type):
        self.from_rank, self.to_rank = from_rank, to_rank
        self.n_clockwise_rotations, self.boundary_type = n_clockwise_rotations, boundary_type
