"""Where the HOST time of an acoustic substep goes (single tile, no exchanges): cProfile of the Python layer while the device
runs asynchronously.  python tools/host_profile.py [--n 96]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import acoustic_bench

    ops, _, _ = acoustic_bench.build_ops(args.n, 79)
    for _, fn in ops:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        for _, fn in ops:
            fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"C{args.n}: host issue time per substep {1e3 * t_issue / args.reps:.3f} ms, with device completion {1e3 * t_all / args.reps:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.reps):
        for _, fn in ops:
            fn()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
