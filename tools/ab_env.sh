#!/bin/bash
# A/B of an environment switch on the whole bench:  tools/ab_env.sh VAR [rounds]   -> ms_per_step with VAR unset / VAR=1, alternating
VAR=$1; R=${2:-3}
for r in $(seq $R); do
  a=$(python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  b=$(env $VAR=1 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  echo "round $r: default $a   $VAR=1 $b"
done
