mkdir -p gpurun_out/r05g; export PACE_BENCH_CACHE=/tmp
timeout 300 python bench.py --no-cpu-baseline --no-traffic > /dev/null 2>&1
for v in defer prep defer prep; do
  if [ $v = prep ]; then export PACE_DSW_FLUXES_IN_PREP=1; else unset PACE_DSW_FLUXES_IN_PREP; fi
  timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
for v in defer prep; do
  if [ $v = prep ]; then export PACE_DSW_FLUXES_IN_PREP=1; else unset PACE_DSW_FLUXES_IN_PREP; fi
  cd /tmp; rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats -d /tmp/tr -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-traffic > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  echo "== $v"; python tools/rocprof_summary.py $(find /tmp/tr -name "*.db" | head -1) | grep "^\"_Z" | grep -v at6native | awk -F, '{print substr($1,1,40), $2, $4, $5}' | head -12
done
