#!/bin/bash
# Upper bound of "k_exact under k_screen" (EXPERIMENTS.md round 5), experiment build.  OMDS_EXACT_CORUN=1 launches
# a REDUNDANT k_exact (the previous step's candidate list, scratch outputs) on a second stream beside every k_screen; =2 launches
# the same redundant kernel in the main stream's order (control: what one more serial k_exact costs); 0 = the shipped step.
# If an overlapped design hid k_exact completely behind k_screen, the step would cost (shipped) - k_exact + [(=1) - (shipped)]:
# (=1) - (shipped) per step is what k_screen + the rest pay for the co-resident kernel.
#   make -C optimalmodulationds_amd/csrc experiment ; bash tools/studies/exact_corun_ab.sh > gpurun_out/exact_corun_ab.txt
export TMPDIR=/tmp
R=$(pwd)
export OMDS_LIB=$R/optimalmodulationds_amd/csrc/libomds_hip_exp.so
finddb() { find "$1" -name "*results.db" | head -1; }
for rep in 1 2; do
for cfg in 0 1 2; do
  export OMDS_EXACT_CORUN=$cfg
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python3 bench.py --path screened --steps 6 --warmup 2 --reps 2 --no-cpu-baseline --no-secondary > /tmp/prof_ab.log 2>&1
  echo "## round $rep OMDS_EXACT_CORUN=$cfg"
  python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_ab)" | grep -E "k_tail_sel|k_exact|k_screen" | head -3 | cut -c1-110
  python3 bench.py --path screened --steps 10 --warmup 3 --reps 6 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('value', round(d['value']), 'ms_per_iteration', round(d['ms_per_step'],3), 'per step us', round(1e3*d['ms_per_step']/32,1), 'k_screen ms', round(d['roofline']['avg_launch_ms'],4), 'cand', round(d['screening']['candidates_per_rollout_step'],2), 'fallbacks', d['screening']['fallbacks'])"
done
done
