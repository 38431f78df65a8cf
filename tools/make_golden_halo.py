"""Golden vectors of the cubed-sphere halo exchange from the reference's OWN pace.util, running natively (no gtscript, hence
no interpreter: CubedSphereCommunicator / HaloUpdater / HaloDataTransformer are plain numpy code): six ranks on threads
(tools/threadcomm.py stands in for mpi4py), layout (1, 1), C12, one level.

    python tools/make_golden_halo.py        ->  tests/golden/halo_native_c12.npz

Inputs are integer-coded (tile, kind, i, j, k) so that every halo value names the cell it came from; the file stores the inputs
and what the reference leaves in every array after
  halo_update(n_points = 3) of a cell-centred and of a corner (B-grid) field, halo_update(n_points = 2) of a z-interface field,
  vector_halo_update of a D-grid pair (u on y-interfaces, v on x-interfaces) and of a C-grid pair,
  synchronize_vector_interfaces of a D-grid pair.
tests/test_oracle_golden.py holds oracle/halo.py to it, tests/test_halo.py (emulated kernels) and tests/test_gpu_parity.py the
product's pack / exchange / unpack.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import pace.util  # noqa: E402
from threadcomm import run_ranks  # noqa: E402

N, NZ = 12, 1
DIMS = {"c": ["x", "y", "z"], "xi": ["x_interface", "y", "z"], "yi": ["x", "y_interface", "z"],
        "b": ["x_interface", "y_interface", "z"], "zi": ["x", "y", "z_interface"]}


def base_fields():
    i, j, k = np.meshgrid(np.arange(N + 7), np.arange(N + 7), np.arange(NZ + 1), indexing="ij")
    out = {}
    for kind_id, kind in enumerate(DIMS):
        out[kind] = [(t + 1) * 1.0e6 + kind_id * 1.0e5 + i * 1.0e3 + j * 10.0 + k + 0.5 for t in range(6)]
    return out


def main():
    base = base_fields()

    def rank(comm):
        part = pace.util.CubedSpherePartitioner(pace.util.TilePartitioner((1, 1)))
        cube = pace.util.CubedSphereCommunicator(comm, part)
        sizer = pace.util.SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={},
                                                            layout=(1, 1), tile_partitioner=part.tile, tile_rank=0)
        qf = pace.util.QuantityFactory.from_backend(sizer, "numpy")
        r = comm.Get_rank()
        out = {}

        def q(key):
            x = qf.zeros(DIMS[key], "")
            x.data[:] = base[key][r]
            return x

        s = q("c"); cube.halo_update(s, n_points=3); out["c"] = s.data.copy()
        s = q("b"); cube.halo_update(s, n_points=3); out["b"] = s.data.copy()
        s = q("zi"); cube.halo_update(s, n_points=2); out["zi"] = s.data.copy()
        u, v = q("yi"), q("xi"); cube.vector_halo_update(u, v, n_points=3); out["du"], out["dv"] = u.data.copy(), v.data.copy()
        u, v = q("xi"), q("yi"); cube.vector_halo_update(u, v, n_points=3); out["cu"], out["cv"] = u.data.copy(), v.data.copy()
        u, v = q("yi"), q("xi"); cube.synchronize_vector_interfaces(u, v); out["su"], out["sv"] = u.data.copy(), v.data.copy()
        return out

    ref = run_ranks(6, rank)
    data = {"n": N, "nz": NZ}
    for kind, tiles in base.items():
        data["in_" + kind] = np.stack(tiles)
    for name in ref[0]:
        data["out_" + name] = np.stack([ref[t][name] for t in range(6)])
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "halo_native_c12.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
