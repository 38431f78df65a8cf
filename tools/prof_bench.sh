#!/bin/bash
# Per-kernel times of one bench.py configuration under the kernel trace (GPU box):
#   bash tools/prof_bench.sh <tag> [bench.py arguments]     ->  gpurun_out/<tag>/kernel_stats.csv, bench.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
D=/tmp/prof_$TAG; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-contract --no-traffic "$@" > "$O/bench.json" 2> "$O/trace.err"
DB=$(find $D -name '*.db' | head -1)
cd "$R"
[ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" > "$O/kernel_stats.csv"
rm -rf $D
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['avg_us']) * int(r['calls']))
for r in rows[:16]:
    print(f"  {r['kernel'][:70]:70s} calls {r['calls']:>4s} avg {float(r['avg_us']):8.1f} min {float(r['min_us']):8.1f}")
PY
tail -1 "$O/bench.json" | cut -c1-200
