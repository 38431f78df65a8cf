"""Average duration of the dispatches of one kernel that ran ALONE (no other kernel dispatch overlapping in time) in a
rocprofv3 --kernel-trace run (rocpd SQLite output).  bench.py's step overlaps its kernels on up to four streams, so the plain
per-kernel average of `--stats` mixes in-step (slowed-down) launches with the stand-alone launches of the roofline loop; this
is the number `roofline.us_per_launch` has to agree with.

    python tools/rocprof_isolated.py results.db k_fvtp2dILi6ELi2ELi1E
"""
import sqlite3
import sys


def main(path, needle):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
                               join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
    alone = []
    for n, (name, a, b) in enumerate(rows):
        if needle not in name:
            continue
        overlap = False
        for m in range(max(0, n - 8), min(len(rows), n + 9)):
            if m != n and rows[m][1] < b and rows[m][2] > a:
                overlap = True
                break
        if not overlap:
            alone.append((b - a) / 1e3)
    every = [(b - a) / 1e3 for name, a, b in rows if needle in name]
    print(f"kernel containing {needle!r}: {len(every)} dispatches, average {sum(every) / max(1, len(every)):.2f} us")
    if alone:
        alone.sort()
        print(f"  of which ran alone (no overlapping dispatch): {len(alone)}, average {sum(alone) / len(alone):.2f} us, "
              f"median {alone[len(alone) // 2]:.2f} us, min {alone[0]:.2f}, max {alone[-1]:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
