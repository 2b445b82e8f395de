#!/bin/bash
# Where k_tail's 72 us go: the kernel cut short behind its phases (experiment build: OMDS_TAIL_STOP = 1 top-k | 2 + forward and backward | 3 + modulation;
# pass2_body's own stops 10-15 through OMDS_TAIL_STOP too).  Wrong results by design; kernel times from rocprofv3.  gpurun -- 'bash tools/tail_phase_probe.sh'
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tail_probe; mkdir -p $OUT
cd /tmp
for st in 0 1 10 11 12 13 14 15 2 3; do
  OMDS_TAIL_STOP=$st OMDS_LIB=$GRAFT_REPO_ROOT/optimalmodulationds_amd/csrc/libomds_hip_exp.so rocprofv3 --kernel-trace --stats -d $OUT/kt_$st -- python3 $GRAFT_REPO_ROOT/bench.py --path fp32 --no-secondary --no-cpu-baseline --steps 3 --warmup 1 --reps 1 > $OUT/log_$st.txt 2>&1
  echo "OMDS_TAIL_STOP=$st $(python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats "$(find $OUT/kt_$st -name '*_results.db' | head -1)" | grep k_tail | head -1)"
  rm -rf $OUT/kt_$st
done
