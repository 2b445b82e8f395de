#!/bin/bash
# k_tail (fp32 step) on 16-row against 32-row tiles (experiment build, OMDS_TAIL_ROWS): kernel trace + bench at several rollout counts.
export TMPDIR=/tmp
export OMDS_LIB=$(pwd)/optimalmodulationds_amd/csrc/libomds_hip_exp.so
finddb() { find "$1" -name "*results.db" | head -1; }
for wl in ${WORKLOADS:-franka_shelf_1024x32 franka_shelf_4096x32}; do
for rows in 32 16; do
  export OMDS_TAIL_ROWS=$rows
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python3 bench.py --path fp32 --workload $wl --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary > /tmp/prof_ab.log 2>&1
  echo "## $wl OMDS_TAIL_ROWS=$rows"
  python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_ab)" | grep -E "k_tail|k_pass1" | head -2 | cut -c1-110
  python3 bench.py --path fp32 --workload $wl --steps 5 --warmup 1 --reps 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('value', round(d['value']), 'ms_per_iteration', round(d['ms_per_step'],3))"
done
done
