# The dynamic instruction mix of every kernel of the bench step (rocprofv3 --pmc, separate passes of three counters each):
#   bash tools/collect_valu_mix.sh    -> gpurun_out/valu_mix.json
set -u
R=$(pwd); O=$R/gpurun_out/mix; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PACE_BENCH_CACHE=/tmp
python3 $R/bench.py --no-traffic --no-cpu-baseline --steps 2 > /dev/null 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> $O/err.txt
done
cd $R
python tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > gpurun_out/valu_mix.json
rm -rf $O
python - <<'PY'
import json
d=json.load(open('gpurun_out/valu_mix.json'))
for k,v in d.items():
    w=v.get("SQ_WAVES_avg",0) or 1
    print(k[:36].ljust(38), " ".join(f"{x[9:-4]}={v[x]/w:.0f}" for x in sorted(v) if x.endswith("_avg") and x!="SQ_WAVES_avg"))
PY
