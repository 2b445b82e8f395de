#!/bin/bash
# A/B of k_screen's dead-chunk skipping on one box (experiment build): OMDS_SCREEN_REORDER=0 keeps the natural unit order (hardly any
# 16-unit chunk of the shipped network is dead for a whole wave then: what is measured is the cost of the zero tests), 1 sorts the
# hidden units by how often they fire (screen_reorder); 2 = the order of the calibration batch only, without the refinement on the
# rollouts' own states (CFGS="2 1").  Kernel trace + bench, two rounds.
#   make -C optimalmodulationds_amd/csrc experiment ; bash tools/studies/screen_skip_ab.sh > gpurun_out/screen_skip_ab.txt
export TMPDIR=/tmp
R=$(pwd)
export OMDS_LIB=$R/optimalmodulationds_amd/csrc/libomds_hip_exp.so
finddb() { find "$1" -name "*results.db" | head -1; }
for wl in ${WORKLOADS:-franka_shelf_1024x32 franka_shelf_4096x32}; do
for rep in 1 2; do
for cfg in ${CFGS:-0 1}; do
  export OMDS_SCREEN_REORDER=$cfg
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python3 bench.py --workload $wl --steps 6 --warmup 2 --reps 2 --no-cpu-baseline --no-secondary > /tmp/prof_ab.log 2>&1
  echo "## $wl round $rep OMDS_SCREEN_REORDER=$cfg"
  python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_ab)" | grep -E "k_tail_sel|k_exact|k_screen" | head -3 | cut -c1-110
  python3 bench.py --workload $wl --steps 10 --warmup 3 --reps 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('value', round(d['value']), 'ms_per_iteration', round(d['ms_per_step'],3), 'k_screen ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'cand', round(d['screening']['candidates_per_rollout_step'],2), 'eps', d['screening']['eps'], 'fallbacks', d['screening']['fallbacks'])"
done
done
done
