for r in 1 2 3; do
  a=$(python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  b=$(GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  c=$(GPU_MAX_HW_QUEUES=2 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  echo "round $r: default $a   8 queues $b   2 queues $c"
done
