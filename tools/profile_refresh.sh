#!/bin/bash
# Regenerates the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun from the repo root).
# Every rocprofv3 call has python3 directly after "--"; counters are collected in their own passes.
set -u
R=${1:-r01}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
B="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
P="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
python3 bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
finddb() { find "$1" -name '*_results.db' | head -1; }
for wl in franka_shelf_1024x32 franka_shelf_4096x32 planar7_1024x32; do
  rocprofv3 --kernel-trace --stats -d "$OUT/kt_$wl" -- python3 $B --workload $wl > "$OUT/kt_$wl.log" 2>&1
  python3 tools/rocprof_summary.py stats "$(finddb "$OUT/kt_$wl")" > "$OUT/stats_$wl.txt"
  tail -1 "$OUT/kt_$wl.log" | grep '^{' > "$OUT/bench_profiled_$wl.json"
  rm -rf "$OUT/kt_$wl"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" -- python3 $P > "$OUT/pmc_$c.log" 2>&1
  python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_$c")" > "$OUT/pmc_$c.txt"
  rm -rf "$OUT/pmc_$c"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/pmc_sq" -- python3 $P > "$OUT/pmc_sq.log" 2>&1
python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_sq")" > "$OUT/pmc_sq.txt"
rm -rf "$OUT/pmc_sq"
ls -la "$OUT"
