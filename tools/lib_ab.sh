#!/bin/bash
# same-box A/B of two builds of the library: per-kernel times of the all-fp32 step (rocprofv3 --kernel-trace --stats), alternating, twice.
# gpurun -- 'bash tools/lib_ab.sh libomds_hip_base.so libomds_hip.so [workload]'   (names under optimalmodulationds_amd/csrc/)
set -u
A=$1; B=$2; WL=${3:-franka_shelf_1024x32}
OUT=$PWD/gpurun_out/lib_ab
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for round in 1 2; do
  for lib in $A $B; do
    export OMDS_LIB=$GRAFT_REPO_ROOT/optimalmodulationds_amd/csrc/$lib
    rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary --path fp32 --workload $WL > "$OUT/kt_$lib.log" 2>&1
    echo "== $lib (round $round)"
    python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats "$(find "$OUT/kt" -name '*_results.db' | head -1)" | head -5
    rm -rf "$OUT/kt"
  done
done
