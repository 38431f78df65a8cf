#!/bin/bash
# A/B of settings of ONE environment variable on the whole bench, alternating:  bash tools/ab_envs.sh <tag> <rounds> VAR value [value ...]
# ("-" = unset)  -> gpurun_out/<tag>/ab_env.txt; also prints the average time of kernels whose name contains $KERNEL (default divdamp_fused)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=$2; VAR=$3; shift 3
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for r in $(seq 1 $N); do
  for V in "$@"; do
    if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
    python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$V', 'ms_per_step %.4f' % d['ms_per_step'], 'other %.4f' % ((d.get('other_contract') or {}).get('ms_per_step') or 0))
" | tee -a "$O/ab_env.txt"
  done
done
