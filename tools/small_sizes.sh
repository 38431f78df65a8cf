R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O
cd $R
for n in 48 96 192; do for r in 1 2; do python bench.py --tile-size $n --no-cpu-baseline --no-traffic --no-other-contract 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($n, d['ms_per_step'])"; done; done | tee $O/sizes.txt
cd /tmp; export TMPDIR=/tmp
for n in 48 96; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$n -o b -- python3 $R/bench.py --tile-size $n --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> $O/trace.err
DB=$(find $O/trace$n -name '*.db' | head -1); python $R/tools/rocprof_timeline.py $DB | tee $O/timeline_c$n.txt; rm -rf $O/trace$n
done
