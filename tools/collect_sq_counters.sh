set -u
R=$(pwd); O=$R/gpurun_out/sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PACE_BENCH_CACHE=/tmp
python3 $R/bench.py --no-traffic --no-cpu-baseline --steps 2 > /dev/null 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2>> $O/err.txt
done
cd $R
python tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/pmc_sq.json
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
python - <<'PY'
import json
d=json.load(open('gpurun_out/sq/pmc_sq.json'))
for k in ("k_fvtp2d<6, 2, 1>","k_divdamp_fused","k_riem_column<0, 5>","k_kinetic_energy<6>","k_fxadv_edges"):
    v=d.get(k,{})
    if v: print(k, {x:round(v[x]) for x in v if x.endswith("_avg")})
PY
