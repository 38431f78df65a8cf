"""Cubed-sphere grid metrics and the baroclinic initial state at any resolution, generated on the host with numpy
(reference: util/pace/util/grid/generation.py `MetricTerms`,
fv3core/pace/fv3core/initialization/baroclinic.py `init_baroclinic_state`).

The reference builds its grid with one `MetricTerms` per rank and a dozen halo exchanges between the ranks; for a (1, 1)
layout every quantity is a function of (N, tile) alone, so here EVERY process generates all six tiles on the host (a few
hundred (N + 7)^2 numpy arrays: 0.2 s at C48, 3 s at C192) and keeps its own -- no communication, no dependence on
which ranks exist.  `tiles(n, nz)` caches the result per process.
"""
import functools

from .metrics import generate
from .positions import corner_positions, exchange_scalar, exchange_vector, exchange_vector_unsigned  # noqa: F401


@functools.lru_cache(maxsize=4)
def tiles(n: int, nz: int):
    """list of six dicts: the metric terms of every tile (names of GridData / DampingCoefficients) + the unit vectors ee1,
    ee2, es1, ew2 the initial state needs"""
    return generate(n, nz)


class MetricTerms:
    """The reference's entry point (generation.py:200-357): metric terms of the tile this rank owns.

    MetricTerms(quantity_factory, communicator) or MetricTerms.from_tile_sizing(npx, npy, npz, communicator, backend).
    Attribute access gives numpy arrays / floats under the reference's names (`.area`, `.dx`, `.cos_sg1`, `.da_min`, ...);
    `GridData.new_from_metric_terms(mt)` and `DampingCoefficients.new_from_metric_terms(mt)` make the device containers."""

    def __init__(self, quantity_factory, communicator, grid_type: int = 0):
        if grid_type >= 3:
            raise NotImplementedError("grid_type >= 3")
        s = quantity_factory.sizer
        if s.nx != s.ny:
            raise ValueError("tiles are square")
        self.quantity_factory = quantity_factory
        self._tile = int(communicator.rank) % 6 if communicator is not None else 0
        self._terms = tiles(s.nx, s.nz)[self._tile]

    @classmethod
    def from_tile_sizing(cls, npx: int, npy: int, npz: int, communicator, backend: str = "hip:gfx950", grid_type: int = 0,
                         device="cuda"):
        from ..quantity import QuantityFactory, SubtileGridSizer

        sizer = SubtileGridSizer.from_tile_params(nx_tile=npx - 1, ny_tile=npy - 1, nz=npz, n_halo=3, extra_dim_lengths={},
                                                  layout=(1, 1))
        return cls(QuantityFactory(sizer, device=device), communicator, grid_type)

    @property
    def terms(self) -> dict:
        return self._terms

    def __getattr__(self, name):
        terms = self.__dict__.get("_terms")
        if terms is not None and name in terms:
            return terms[name]
        raise AttributeError(name)
