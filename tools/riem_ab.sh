#!/bin/bash
# riem_solver3's kernel under the kernel trace, for each library given: bash tools/riem_ab.sh <tag> <lib> [<lib> ...]  ("default" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do
for L in "$@"; do
  if [ "$L" = default ]; then unset PACE_HIP_LIB; else export PACE_HIP_LIB=$R/$L; fi
  rm -rf "$O/tr"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/tr" -o b -- python3 "$R/bench.py" --no-cpu-baseline --no-other-contract --no-traffic > "$O/last.json" 2>> "$O/err.txt"
  DB=$(find "$O/tr" -name '*.db' | head -1)
  echo "$L $(tail -1 $O/last.json | python3 -c 'import json,sys; print("step %.4f" % json.loads(sys.stdin.read())["ms_per_step"])') $(python3 $R/tools/rocprof_summary.py $DB | grep -E 'k_riem_column' | awk -F, '{print "riem avg_us", $4, "min", $5}')" | tee -a "$O/ab.txt"
done
done
rm -rf "$O/tr"
