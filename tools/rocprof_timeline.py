"""Timeline of one bench step out of a rocprofv3 --kernel-trace run (rocpd SQLite): kernel, stream/queue, start and end
relative to the first kernel of the step, so that overlap between the two streams of a step and idle gaps are visible.

    python tools/rocprof_timeline.py results.db [step_index_from_end=2]
"""
import sqlite3
import sys


def main(path, back=2):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = f"""select s.kernel_name, d.start, d.end, {qcol or 0} from rocpd_kernel_dispatch d
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
    rows = [(n.split("(")[0], a, b, qq) for n, a, b, qq in cur.execute(q)]
    # first kernel of a d_sw: the flux preparation (one launch, k_fxadv_fused; PACE_FXADV_SPLIT=1: k_fxadv_frame is its first)
    starts = [i for i, r in enumerate(rows) if "k_fxadv_frame" in r[0] or "k_fxadv_fused" in r[0]]
    i0 = starts[-back]
    i1 = starts[-back + 1] if back > 1 else len(rows)
    t0 = rows[i0][1]
    print(f"step of {(rows[i1 - 1][2] - t0) / 1e3:.1f} us, {i1 - i0} kernels")
    for n, a, b, qq in rows[i0:i1]:
        print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f}  q{qq}  {n[:70]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
