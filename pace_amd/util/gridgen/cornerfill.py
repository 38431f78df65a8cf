"""The 3 x 3 corner halos of a cubed-sphere tile have no neighbouring tile: the reference fills them by turning the adjacent
edge halo around the cube corner (stencils/pace/stencils/corners.py:755-985, the numpy helpers `fill_corners_2d`,
`fill_corners_agrid / dgrid / cgrid`).  Here: the south-west rule of each flavour written once, the other three corners
obtained by mirroring the index space (a mirrored view of the arrays turns any corner into the south-west one).

Conventions: arrays are (N + 7)^2 storages with halo 3; `o` = 3 = first compute index, cells 0 .. N-1 and corner points
0 .. N relative to `o`.  For a mirrored axis a CELL-centred index c maps to N - 1 - c and a CORNER / interface index to N - c.
"""
import numpy as np


class _View:
    """q seen through optional mirrors of the x / y axis; `stag` = (sx, sy): 1 where the axis is an interface axis."""

    def __init__(self, q, n, mx, my, stag, o=3):
        self.q, self.n, self.mx, self.my, self.stag, self.o = q, n, mx, my, stag, o

    def _ij(self, i, j):
        if self.mx:
            i = self.n - 1 + self.stag[0] - i
        if self.my:
            j = self.n - 1 + self.stag[1] - j
        return self.o + i, self.o + j

    def get(self, i, j):
        return self.q[self._ij(i, j)]

    def set(self, i, j, v):
        self.q[self._ij(i, j)] = v


def _corners():
    return ((False, False), (True, False), (False, True), (True, True))


def fill_scalar(q, n, grid, direction, halo=3):
    """fill_corners_2d: grid 'A' (cell centres) or 'B' (corner points), direction 'x' or 'y'."""
    s = 0 if grid == "A" else 1
    for mx, my in _corners():
        v = _View(q, n, mx, my, (s, s))
        for i in range(1, 1 + halo):
            for j in range(1, 1 + halo):
                if direction == "x":
                    v.set(-i, -j, v.get(-j, i - 1 + s))
                else:
                    v.set(-j, -i, v.get(i - 1 + s, -j))


def fill_pair(x, y, n, grid, halo=3):
    """fill_corners_agrid / dgrid / cgrid with vector = False: x lives on the x-family, y on the y-family of
    grid 'a' (both cell centred), 'd' (x on (X, Y_INTERFACE), y on (X_INTERFACE, Y)) or 'c' (x on (X_INTERFACE, Y),
    y on (X, Y_INTERFACE)).  A corner-halo entry of x takes the entry of y it becomes when turned around the cube corner."""
    sx, sy = {"a": ((0, 0), (0, 0)), "d": ((0, 1), (1, 0)), "c": ((1, 0), (0, 1))}[grid]
    for mx, my in _corners():
        swap = mx != my  # a single mirror exchanges the roles the reference's tables give to x and y at that corner
        vx, vy = _View(x, n, mx, my, sx), _View(y, n, mx, my, sy)
        for i in range(1, 1 + halo):
            for j in range(1, 1 + halo):
                if grid == "a":
                    vx.set(-i, -j, vy.get(-j, i - 1))
                    vy.set(-j, -i, vx.get(i - 1, -j))
                elif grid == "d":
                    vx.set(-i, -j, vy.get(-j, i - 1))
                    vy.set(-i, -j, vx.get(j - 1, -i))
                else:
                    vx.set(-i, -j, vy.get(j - 1, -i))
                    vy.set(-i, -j, vx.get(-j, i - 1))
        del swap
