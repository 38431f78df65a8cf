# bench.py (ms per substep) and the whole loop for several allocation skews (PACE_ALLOC_SKEW_BYTES); GPU box
for r in 1 2; do
  for s in 0 256 4352 69888 1114112; do
    a=$(PACE_ALLOC_SKEW_BYTES=$s python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(round(json.loads(sys.stdin.readlines()[-1])["ms_per_step"],4))')
    b=$(PACE_ALLOC_SKEW_BYTES=$s python tools/acoustic_bench.py --reps 10 2>/dev/null | grep -E "^(c_sw|d_sw|riem_solver3|total)" | awk '{printf "%s %s  ", $1, $2}')
    echo "round $r skew $s: bench $a ms | $b"
  done
done
