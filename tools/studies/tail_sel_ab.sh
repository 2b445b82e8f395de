#!/bin/bash
# A/B of k_tail_sel's mask request on one box (experiment build with -DOMDS_TAIL_TL):
#   OMDS_TAIL_SEL_STOP=101   masks of the k selected rows requested after the ranks are known (three dependent round trips)
#   OMDS_TAIL_SEL_STOP=0     masks of every candidate requested with its value (two)
# make -C optimalmodulationds_amd/csrc experiment VFLAGS=-DOMDS_TAIL_TL ; bash tools/studies/tail_sel_ab.sh > gpurun_out/tail_sel_ab.txt
export OMDS_LIB=optimalmodulationds_amd/csrc/libomds_hip_exp.so
for rep in 1 2; do
for cfg in 101 0; do
  echo "## round $rep: OMDS_TAIL_SEL_STOP=$cfg"
  OMDS_TAIL_SEL_STOP=$cfg OMDS_TAIL_TL_STEP=5 python bench.py --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary 2>&1 | python tools/tail_timeline.py
  OMDS_TAIL_SEL_STOP=$cfg python bench.py --steps 20 --warmup 3 --reps 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('ms_per_iteration', d['ms_per_step'], d['rep_ms_per_step'])"
done
done
