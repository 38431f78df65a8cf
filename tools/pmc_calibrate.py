"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters with THIS code's access width (one double = 8 B per lane).

MI355X_MICROARCH.md calibrates "FETCH_SIZE reports half of the bytes" only for 16 B-per-lane streaming reads and says other
widths are uncalibrated.  The kernels here load doubles, one per lane.  This script launches the library's plain copy kernel
(k_scale_copy: dst[c] = src[c], one double per thread, rows of N + 6 contiguous doubles) on `pairs` distinct source /
destination fields at C384 x 79 (98 MB each way per launch, 2.3 GB in total: far beyond the 256 MB Infinity Cache) and prints
the exact byte counts per launch.  Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
passes), then `python tools/pmc_calibrate.py --report <fetch dir> <write dir>` gives counter-to-byte factors.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_fetch -- python3 tools/pmc_calibrate.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/cal_write -- python3 tools/pmc_calibrate.py
    python3 tools/pmc_calibrate.py --report gpurun_out/cal_fetch gpurun_out/cal_write > profiles/r02_pmc_calibration.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, NZ, PAIRS = 384, 79, 12


def bytes_per_launch():
    return (N + 6) * (N + 6) * (NZ + 1) * 8


def run():
    import ctypes as C

    import torch

    from pace_amd import _lib, synthetic  # noqa: F401
    from pace_amd.util import QuantityFactory, SubtileGridSizer
    from pace_amd.util.grid import geom_struct

    lib = _lib.load()
    sizer = SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device="cuda")
    geom = geom_struct(qf)
    src = [qf.ones(["x", "y", "z"], "") for _ in range(PAIRS)]
    dst = [qf.zeros(["x", "y", "z"], "") for _ in range(PAIRS)]
    torch.cuda.synchronize()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        for a, b in zip(src, dst):
            lib.call("pace_copy", C.byref(geom), a.data.data_ptr(), b.data.data_ptr(), st)
    torch.cuda.synchronize()
    assert float(dst[-1].data[5, 5, 5]) == 1.0
    print(json.dumps({"kernel": "k_scale_copy", "launches": 3 * PAIRS, "read_bytes_per_launch": bytes_per_launch(),
                      "write_bytes_per_launch": bytes_per_launch()}))


def report(dirs):
    vals = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_scale_copy" in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {"access": "8 B per lane (one double), rows of N + 6 contiguous doubles, C384 x 79", "bytes_per_launch_each_way": bytes_per_launch()}
    for name, v in vals.items():
        v = v[PAIRS:]  # skip the first sweep (page faults / first touch)
        avg = sum(v) / len(v)
        out[name] = {"launches": len(v), "counter_avg": avg, "counter_x_1024": avg * 1024,
                     "true_bytes_over_counter_x_1024": bytes_per_launch() / (avg * 1024)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--report":
        report(sys.argv[2:])
    else:
        run()
