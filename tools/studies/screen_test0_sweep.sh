#!/bin/bash
# k_screen with the zero test starting at k-chunk 4 / 8 (shipped) / 10 / 12 (variant builds: make variant V=t10 VFLAGS=-DOMDS_SC_TEST0=10),
# kernel trace of the 1024 x 32 and 4096 x 32 workloads on one box.
export TMPDIR=/tmp
R=$(pwd)
finddb() { find "$1" -name "*results.db" | head -1; }
for wl in franka_shelf_1024x32 franka_shelf_4096x32; do
for rep in 1 2; do
for v in ${VARIANTS:-t4 hip t10 t12}; do
  if [ $v = hip ]; then export OMDS_LIB=$R/optimalmodulationds_amd/csrc/libomds_hip.so; else export OMDS_LIB=$R/optimalmodulationds_amd/csrc/libomds_hip_$v.so; fi
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python3 bench.py --workload $wl --steps 6 --warmup 2 --reps 2 --no-cpu-baseline --no-secondary > /tmp/prof_ab.log 2>&1
  echo "## $wl round $rep $v: $(python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_ab)" | grep -E "k_screen" | head -1 | cut -c60-110)"
done
done
done
