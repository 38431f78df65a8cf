"""Average duration per kernel from a rocprofv3 --kernel-trace --stats --output-format csv directory (top N)."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.1f} us")
