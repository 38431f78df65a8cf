#!/bin/bash
# Per-kernel HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of tools/exp_variants.py for one library, on the GPU box:
#   bash tools/pmc_kernels.sh name=path.so  > gpurun_out/pmc_name.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
spec=$1
name=${spec%%=*}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmck_${name}_$c; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D -- python3 $R/tools/exp_variants.py $spec --reps 4 --child $D/out > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/pmck_${name}_FETCH_SIZE /tmp/pmck_${name}_WRITE_SIZE
