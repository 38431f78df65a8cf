export PACE_BENCH_CACHE=/tmp
timeout 300 python bench.py --no-cpu-baseline --no-traffic > /dev/null 2>&1
for v in tiled point tiled point; do
  if [ $v = point ]; then export PACE_DSW_POINT_KE=1; else unset PACE_DSW_POINT_KE; fi
  timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
unset PACE_DSW_POINT_KE
cd /tmp; rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats -d /tmp/tr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-traffic > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find /tmp/tr -name "*.db" | head -1) | grep "^\"_Z" | grep -v at6native | awk -F, '{print substr($1,1,40), $2, $4, $5, $8, $10}' | head -10
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
