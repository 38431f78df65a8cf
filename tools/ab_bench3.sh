#!/bin/bash
# like ab_bench.sh for any number of builds: bash tools/ab_bench3.sh <tag> <rounds> <lib> [<lib> ...]   ("default" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd "$R"
for r in $(seq 1 $N); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset PACE_HIP_LIB; else export PACE_HIP_LIB=$R/$L; fi
    python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$L', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_us %.1f' % (r.get('us_per_launch') or 0), 'other %.4f' % ((d.get('other_contract') or {}).get('ms_per_step') or 0), 'verified', d.get('verified'))
" | tee -a "$O/ab.txt"
  done
done
