#!/bin/bash
# Kernel-trace A/B of one k_tail_sel feature (experiment build, OMDS_TAIL_SEL_STOP=<code> switches it off; see tail_sel_ab.sh):
# average k_tail_sel / k_exact / k_screen launch time at 1024 x 32 and 4096 x 32, two rounds each, same box.
#   make -C optimalmodulationds_amd/csrc experiment ; bash tools/studies/tail_sel_trace_ab.sh 102 > gpurun_out/tail_sel_trace_ab.txt
export TMPDIR=/tmp
R=$(pwd)
export OMDS_LIB=$R/optimalmodulationds_amd/csrc/libomds_hip_exp.so
finddb() { find "$1" -name "*results.db" | head -1; }
OFF=${1:-102}
for wl in franka_shelf_1024x32 franka_shelf_4096x32; do
for rep in 1 2; do
for cfg in $OFF 0; do
  export OMDS_TAIL_SEL_STOP=$cfg
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python3 bench.py --workload $wl --steps 6 --warmup 2 --reps 2 --no-cpu-baseline --no-secondary > /tmp/prof_ab.log 2>&1
  echo "## $wl round $rep OMDS_TAIL_SEL_STOP=$cfg"
  python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_ab)" | grep -E "k_tail_sel|k_exact|k_screen" | head -3 | cut -c1-110
done
done
done
