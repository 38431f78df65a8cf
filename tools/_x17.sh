export TMPDIR=/tmp PACE_BENCH_CACHE=/tmp
R=$(pwd); O=$R/gpurun_out/x17; mkdir -p $O
for i in 1 2; do timeout 300 python tools/acoustic_bench.py 2>/dev/null | grep -v amdgpu | sed -n 3,3p; done
PACE_CSW_ONE_STREAM=1 timeout 300 python tools/acoustic_bench.py 2>/dev/null | grep -v amdgpu | sed -n 3,3p
cd /tmp
PACE_CSW_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o loop -- python3 $R/tools/acoustic_bench.py --reps 3 > /dev/null 2> $O/t.err
DB=$(find $O/t -name '*.db' | head -1); python $R/tools/rocprof_summary.py $DB > $O/stats.csv; rm -rf $O/t
grep -E "csw|d2a2c" $O/stats.csv | awk -F, '{print $1, $2, $3, $4}' | cut -c1-110
