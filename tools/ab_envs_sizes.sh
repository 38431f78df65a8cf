#!/bin/bash
# like ab_envs.sh over several tile sizes:  bash tools/ab_envs_sizes.sh <tag> <rounds> "<sizes>" VAR value [value ...]   ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=$2; SIZES=$3; VAR=$4; shift 4
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
for n in $SIZES; do
for r in $(seq 1 $N); do
  for V in "$@"; do
    if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
    python bench.py --tile-size $n --no-cpu-baseline --no-traffic --no-other-contract 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C$n $VAR=$V', 'ms_per_step %.4f' % d['ms_per_step'], 'verified', d.get('verified'))
" | tee -a "$O/ab_env.txt"
  done
done
done
