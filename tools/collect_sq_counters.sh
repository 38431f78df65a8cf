# SQ counters of the step's kernels (bench.py, C192 x 79), one group of counters per pass (MI355X_MICROARCH.md: counters in their
# own runs, kernel trace only):  gpurun -- 'bash tools/collect_sq_counters.sh [tag]'  ->  gpurun_out/<tag>/sq_counters.json
set -u
TAG=${1:-r06}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PACE_BENCH_CACHE=/tmp
python3 $R/bench.py --no-traffic --no-cpu-baseline --steps 2 > /dev/null 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sqp$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> $O/sq_err.txt
done
cd $R
python tools/pmc_summary.py $O/sqp1 $O/sqp2 $O/sqp3 $O/sqp4 $O/sqp5 > $O/sq_counters_raw.json
rm -rf $O/sqp1 $O/sqp2 $O/sqp3 $O/sqp4 $O/sqp5
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
d = json.load(open(O + "/sq_counters_raw.json"))
out = {}
for k, v in d.items():
    g = lambda n: v.get(n + "_avg")  # noqa: E731
    e = {n[:-4]: round(x) for n, x in v.items() if n.endswith("_avg")}
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        e["valu_instructions_per_wave"] = round(g("SQ_INSTS_VALU") / g("SQ_WAVES"))
        e["salu_instructions_per_wave"] = round((g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES"))
        e["lds_instructions_per_wave"] = round((g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES"))
        f64 = sum(g(n) or 0 for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        e["fp64_share_of_valu"] = round(f64 / g("SQ_INSTS_VALU"), 3)
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_* count cycles (x4: per-SIMD quad cycles on this family) in which an instruction of the class is in flight;
        # the ratios to SQ_BUSY_CYCLES are what matters, compared between kernels
        e["active_valu_over_busy"] = round(g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES"), 3)
        e["active_lds_over_busy"] = round((g("SQ_ACTIVE_INST_LDS") or 0) / g("SQ_BUSY_CYCLES"), 3)
        e["active_any_over_busy"] = round((g("SQ_ACTIVE_INST_ANY") or 0) / g("SQ_BUSY_CYCLES"), 3)
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
        e["wait_inst_any_over_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
        e["wait_any_over_wave_cycles"] = round((g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES"), 3)
    out[k] = e
json.dump(out, open(O + "/sq_counters.json", "w"), indent=1, sort_keys=True)
for k, e in out.items():
    print(k[:40], {x: e[x] for x in e if "_per_wave" in x or "_over_" in x or x == "fp64_share_of_valu"})
PY
