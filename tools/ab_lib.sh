#!/bin/bash
# A/B of two builds of the library on the whole loop and on the bench (GPU box; swaps pace_amd/libpace_hip.so in the scratch copy):
#   tools/ab_lib.sh build/var/<name>/libpace_hip.so [rounds]
VAR=$1; R=${2:-2}
cp pace_amd/libpace_hip.so /tmp/prod.so
run() {
  python tools/acoustic_bench.py --reps 10 2>/dev/null | grep -E "^(c_sw|d_sw|updatedzd|riem_solver3|nh_p_grad|riem_solver_c|updatedzc|p_grad_c|total)" | tr '\n' ' ' | sed 's/  */ /g'
  python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(" | bench ms", round(json.loads(sys.stdin.readlines()[-1])["ms_per_step"],4))'
}
for r in $(seq $R); do
  cp /tmp/prod.so pace_amd/libpace_hip.so; echo -n "[prod] "; run
  cp $VAR pace_amd/libpace_hip.so; echo -n "[var ] "; run
done
cp /tmp/prod.so pace_amd/libpace_hip.so
