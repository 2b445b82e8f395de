#!/bin/bash
# What does the partial last tile of k_screen's per-CU chunk cost at 1024 x 32 (4 rollouts = 4 full tiles + 152 pairs)?
# Experiment build (make experiment VFLAGS=-DOMDS_SC_EXPERIMENT): OMDS_SCREEN_DBG=8 skips that tile (wrong results, guard off).
export OMDS_LIB=$PWD/optimalmodulationds_amd/csrc/libomds_hip_exp.so OMDS_SCREEN_NOGUARD=1
B="python bench.py --steps 10 --warmup 3 --reps 4 --no-cpu-baseline --no-secondary --prof-stride 1"
for r in 1 2; do for d in 0 8 16; do
  echo "DBG=$d: $(OMDS_SCREEN_DBG=$d $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_ms'], d['ms_per_step'])")"
done; done
