#!/bin/bash
# Per-kernel times (and, with "pmc" as first argument, counted HBM bytes) of the whole acoustic loop body, tools/acoustic_bench.py,
# on the GPU box:   bash tools/loop_kernel_times.sh [pmc] [--n 96]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
PMC=0; if [ "${1:-}" = "pmc" ]; then PMC=1; shift; fi
cd /tmp
D=/tmp/lkt; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/t -o loop -- python3 $R/tools/acoustic_bench.py --reps 3 "$@" > /dev/null 2> $D/err
python3 $R/tools/rocprof_summary.py $(find $D/t -name '*.db' | head -1) | python3 -c "
import sys,csv
for r in csv.DictReader(sys.stdin):
    if r['kernel'].startswith('_Z') and 'at6native' not in r['kernel']:
        print(f\"  {r['kernel'][:52]:52s} calls {r['calls']:>4s} avg {float(r['avg_us']):8.1f} vgpr {r['vgpr']:>4s} lds {r['lds_bytes']:>6s}\")
"
if [ $PMC = 1 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/pmc_$c -- python3 $R/tools/acoustic_bench.py --reps 3 "$@" > /dev/null 2>> $D/err
  done
  python3 $R/tools/pmc_summary.py $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict) and 'hbm_bytes_per_launch' in v: print(f\"  {k[:40]:40s} {v['hbm_bytes_per_launch']/1e6:8.1f} MB (read {v['read_bytes_corrected']/1e6:7.1f} write {v['write_bytes_raw']/1e6:7.1f})\")
"
fi
rm -rf $D
