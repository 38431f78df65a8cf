"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd SQLite output) into a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,sgpr,lds_bytes")
    for r in rows:
        print(f"\"{r[0]}\",{r[1]},{r[2]/1e3:.1f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},{100*r[2]/tot:.2f},{r[6]},{r[7]},{r[8]}")


if __name__ == "__main__":
    main(sys.argv[1])
