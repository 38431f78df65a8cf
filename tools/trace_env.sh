#!/bin/bash
# the step's timeline under the kernel trace for settings of one environment variable: bash tools/trace_env.sh <tag> <size> VAR value [value ...]  ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; NSZ=$2; VAR=$3; shift 3
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
  rm -rf "$O/tr"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/tr" -o b -- python3 "$R/bench.py" --tile-size $NSZ --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> "$O/err.txt"
  DB=$(find "$O/tr" -name '*.db' | head -1)
  echo "== C$NSZ $VAR=$V" | tee -a "$O/timelines.txt"; python3 $R/tools/rocprof_timeline.py $DB | cut -c1-110 | tee -a "$O/timelines.txt"
done
rm -rf "$O/tr"
