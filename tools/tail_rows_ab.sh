#!/bin/bash
# k_tail tile shape A/B (experiment build: OMDS_TAIL_ROWS = 4 | 16 | 32 forces one): gpurun -- 'bash tools/tail_rows_ab.sh [workload ...]'
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tail_rows; mkdir -p $OUT
cd /tmp
for wl in "${@:-franka_shelf_1024x32}"; do
for rows in 32 4 16; do
  OMDS_TAIL_ROWS=$rows OMDS_LIB=$GRAFT_REPO_ROOT/optimalmodulationds_amd/csrc/libomds_hip_exp.so rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --path fp32 --no-secondary --no-cpu-baseline --steps 2 --warmup 1 --reps 1 --workload $wl > $OUT/log_$rows.txt 2>&1
  echo "$wl OMDS_TAIL_ROWS=$rows $(python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats "$(find $OUT/kt -name '*_results.db' | head -1)" | grep k_tail | head -1)"
  rm -rf $OUT/kt
done
done
