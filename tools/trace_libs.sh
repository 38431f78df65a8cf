#!/bin/bash
# the step's timeline under the kernel trace for several builds: bash tools/trace_libs.sh <tag> <size> <grep pattern> <lib> [<lib> ...]  ("default" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; NSZ=$2; PAT=$3; shift 3
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = default ]; then unset PACE_HIP_LIB; else export PACE_HIP_LIB=$R/$L; fi
  rm -rf "$O/tr"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/tr" -o b -- python3 "$R/bench.py" --tile-size $NSZ --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> "$O/err.txt"
  DB=$(find "$O/tr" -name '*.db' | head -1)
  echo "== C$NSZ $L" | tee -a "$O/timelines.txt"; python3 $R/tools/rocprof_timeline.py $DB | cut -c1-110 | grep -E "$PAT" | tee -a "$O/timelines.txt"
done
rm -rf "$O/tr"
