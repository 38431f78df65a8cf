#!/bin/bash
# Per-kernel averages of arbitrary counter sets (one rocprofv3 pass per quoted set) over tools/exp_variants.py for one library:
#   bash tools/pmc_sets.sh name=path.so "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
spec=$1; shift
name=${spec%%=*}
cd /tmp
dirs=""
i=0
for set in "$@"; do
  i=$((i+1))
  D=/tmp/pmcs_${name}_$i; rm -rf $D; mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D -- python3 $R/tools/exp_variants.py $spec --reps 4 --n ${KT_N:-192} --child $D/out > $D/log 2>&1 || { echo "pass '$set' failed:" >&2; tail -3 $D/log >&2; continue; }
  dirs="$dirs $D"
done
python3 $R/tools/pmc_summary.py $dirs
