#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into
# profiles/).  Run from the repo root:  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
# rocprofv3: the program itself after `--`, counters in their own passes, no tracing domain besides the kernel trace.
set -u
TAG=${1:-r06}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
export PACE_BENCH_CACHE=/tmp
cd "$R"
timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"
timeout 600 python bench.py --full-outputs --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > "$O/bench_full_outputs.json"
# (each size twice, the second line kept: the first run of a size on a fresh box pays for its set-up -- caches of the state, clocks)
[ -x build/ubench_streams ] && ./build/ubench_streams > "$O/ubench_streams.txt" 2>&1
timeout 600 python bench.py --precision 32 --tile-size 384 --nz 91 --no-cpu-baseline --state synthetic 2>/dev/null | tail -1 > "$O/f32_bench.jsonl"
timeout 300 python tools/acoustic_bench.py --n 48 2>/dev/null | grep -v amdgpu > "$O/acoustic_bench_c48.txt"
for n in 48 96 384; do timeout 300 python bench.py --tile-size $n --no-cpu-baseline --no-traffic > /dev/null 2>&1; timeout 300 python bench.py --tile-size $n --no-cpu-baseline --no-traffic 2>/dev/null | tail -1; done > "$O/bench_sizes.jsonl"
for n in 48 96; do timeout 300 python bench.py --tile-size $n --graph on --no-cpu-baseline --no-traffic 2>/dev/null | tail -1; done > "$O/bench_sizes_graph.jsonl"
timeout 300 python tools/acoustic_bench.py 2>/dev/null | grep -v amdgpu > "$O/acoustic_bench.txt"
timeout 300 python tools/acoustic_bench.py --n 96 2>/dev/null | grep -v amdgpu > "$O/acoustic_bench_c96.txt"   # BASELINE configuration 3
# stage times of the scalar-phase kernel's workgroups (interior / corner / edge tiles): tools/build_prof.sh builds the stamped library
[ -f build/var/prof/libpace_hip.so ] && timeout 300 python tools/fvt_stage_times.py 2>/dev/null | grep -v amdgpu > "$O/scalar_phase_stage_times.txt"
[ -f build/var/prof/libpace_hip.so ] && timeout 300 python tools/dd_stage_times.py 2>/dev/null | grep -v amdgpu | tail -9 > "$O/divdamp_workgroup_timeline.txt"
[ -f build/var/prof/libpace_hip.so ] && timeout 300 python tools/csw_stage_times.py 2>/dev/null | grep -v amdgpu | tail -3 > "$O/csw_tile_workgroup_timeline.txt"
[ -f build/var/prof/libpace_hip.so ] && timeout 300 python tools/riem_stage_times.py 2>/dev/null | grep -v amdgpu > "$O/riem_stage_times.txt"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-other-contract --no-traffic > "$O/bench_under_rocprof.json" 2> "$O/trace.err"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace_loop" -o loop -- python3 "$R/tools/acoustic_bench.py" --reps 3 > /dev/null 2> "$O/trace_loop.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> "$O/pmc.err"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_loop_$c" -- python3 "$R/tools/acoustic_bench.py" --reps 3 > /dev/null 2>> "$O/pmc.err"
done
cd "$R"
DB=$(find "$O/trace" -name '*.db' | head -1)
DBL=$(find "$O/trace_loop" -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" > "$O/kernel_stats.csv" && python tools/rocprof_timeline.py "$DB" > "$O/timeline.txt"
[ -n "$DB" ] && python tools/rocprof_isolated.py "$DB" k_fvt_scalars > "$O/dominant_kernel_alone.txt"
# the C48 step's timeline (launch- and round-trip-bound: where the gaps are)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/trace48" -o b48 -- python3 "$R/bench.py" --tile-size 48 --no-cpu-baseline --no-other-contract --no-traffic > /dev/null 2>> "$O/trace.err"
cd "$R"
DB48=$(find "$O/trace48" -name '*.db' | head -1)
[ -n "$DB48" ] && python tools/rocprof_timeline.py "$DB48" > "$O/timeline_c48.txt"
rm -rf "$O/trace48"
[ -n "$DBL" ] && python tools/rocprof_summary.py "$DBL" > "$O/whole_loop_kernel_stats.csv"
python tools/pmc_summary.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" > "$O/pmc_traffic.json"
python tools/step_table.py "$O/kernel_stats.csv" "$O/pmc_traffic.json" > "$O/step_table.json"
python tools/pmc_summary.py "$O/pmc_loop_FETCH_SIZE" "$O/pmc_loop_WRITE_SIZE" > "$O/whole_loop_pmc_traffic.json"
rm -rf "$O"/trace "$O"/trace_loop "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_loop_FETCH_SIZE "$O"/pmc_loop_WRITE_SIZE
ls -la "$O"
tail -1 "$O/bench.json" | cut -c1-400
