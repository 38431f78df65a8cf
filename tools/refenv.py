"""Build per-rank reference environments (grid, factories, baroclinic state) on 6 threads.

Dev-container tool used by make_golden.py / crosscheck_oracle.py.
"""
import refshim

refshim.install()

import pace.util  # noqa: E402
from pace.dsl.stencil import GridIndexing, StencilFactory  # noqa: E402
from pace.dsl.stencil_config import CompilationConfig, StencilConfig  # noqa: E402
from pace.util import (  # noqa: E402
    CubedSphereCommunicator,
    CubedSpherePartitioner,
    QuantityFactory,
    SubtileGridSizer,
    TilePartitioner,
)
from pace.util.grid import DampingCoefficients, GridData, MetricTerms  # noqa: E402

from threadcomm import run_ranks  # noqa: E402


class RankEnv:
    pass


def build_rank(comm, nx, nz, with_state=True):
    env = RankEnv()
    part = CubedSpherePartitioner(TilePartitioner((1, 1)))
    cube = CubedSphereCommunicator(comm, part)
    sizer = SubtileGridSizer.from_tile_params(
        nx_tile=nx, ny_tile=nx, nz=nz, n_halo=3, extra_dim_lengths={}, layout=(1, 1),
        tile_partitioner=part.tile, tile_rank=cube.tile.rank,
    )
    qf = QuantityFactory.from_backend(sizer, "numpy")
    mt = MetricTerms(quantity_factory=qf, communicator=cube)
    env.comm, env.cube, env.sizer, env.qf, env.mt = comm, cube, sizer, qf, mt
    env.grid_data = GridData.new_from_metric_terms(mt)
    env.damping = DampingCoefficients.new_from_metric_terms(mt)
    cfg = StencilConfig(compilation_config=CompilationConfig(backend="numpy", rebuild=False, validate_args=True))
    env.grid_indexing = GridIndexing.from_sizer_and_communicator(sizer, cube)
    env.stencil_factory = StencilFactory(cfg, env.grid_indexing)
    if with_state:
        from pace.fv3core.initialization.baroclinic import init_baroclinic_state

        env.state = init_baroclinic_state(
            env.grid_data, quantity_factory=qf, adiabatic=False, hydrostatic=False,
            moist_phys=True, comm=cube,
        )
    return env


def build_all(nx=12, nz=79, with_state=True):
    return run_ranks(6, lambda c: build_rank(c, nx, nz, with_state))
