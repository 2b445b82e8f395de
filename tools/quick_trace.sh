#!/bin/bash
# per-kernel times of both steps of one workload (rocprofv3 --kernel-trace --stats): gpurun -- 'bash tools/quick_trace.sh [workload]'
set -u
WL=${1:-franka_shelf_1024x32}
OUT=$PWD/gpurun_out/qt
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for path in fp32 screened; do
  rocprofv3 --kernel-trace --stats -d "$OUT/kt_$path" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary --path $path --workload $WL > "$OUT/kt_$path.log" 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats "$(find "$OUT/kt_$path" -name '*_results.db' | head -1)" > "$OUT/stats_${WL}_$path.txt"
  rm -rf "$OUT/kt_$path"
  head -12 "$OUT/stats_${WL}_$path.txt"
done
