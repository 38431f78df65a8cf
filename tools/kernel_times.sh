#!/bin/bash
# Per-kernel average times of tools/exp_variants.py for the given libraries (rocprofv3 --kernel-trace --stats), on the GPU box:
#   [KT_N=96] bash tools/kernel_times.sh name=path.so [name=path.so ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for spec in "$@"; do
  name=${spec%%=*}
  D=/tmp/kt_$name; rm -rf $D
  mkdir -p $D
  # (the measuring process itself under the profiler: the parent mode of exp_variants.py does no GPU work)
  rocprofv3 --kernel-trace --stats -d $D -o t -- python3 $R/tools/exp_variants.py $spec --reps 10 --n ${KT_N:-192} --child $D/out > /dev/null 2>&1
  echo "== $name"
  python3 $R/tools/rocprof_summary.py $(find $D -name "*.db" | head -1) | python3 -c "
import sys,csv,re
for r in csv.DictReader(sys.stdin):
    k=r['kernel']
    if k.startswith('_Z') and ('fvtp2d' in k or 'k_' in k):
        m=re.match(r'_Z\d+(k_\w+?)(I[\w]*?E)?Ev', k)
        short=k[:46]
        print(f\"  {short:46s} calls {r['calls']:>4s} avg {float(r['avg_us']):8.1f} min {float(r['min_us']):8.1f}\")
" | head -24
  rm -rf $D
done
