#!/bin/bash
# A/B of one k_tail_sel feature on one box (experiment build with -DOMDS_TAIL_TL): OMDS_TAIL_SEL_STOP=<code> switches it off
#   102   modulate_pre on the idle waves off (the modulation computes everything after the backward)
# make -C optimalmodulationds_amd/csrc experiment VFLAGS=-DOMDS_TAIL_TL ; bash tools/studies/tail_sel_ab.sh 102 > gpurun_out/tail_sel_ab.txt
export OMDS_LIB=optimalmodulationds_amd/csrc/libomds_hip_exp.so
OFF=${1:-102}
for rep in 1 2; do
for cfg in $OFF 0; do
  echo "## round $rep: OMDS_TAIL_SEL_STOP=$cfg"
  OMDS_TAIL_SEL_STOP=$cfg OMDS_TAIL_TL_STEP=5 python bench.py --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary 2>&1 | python tools/tail_timeline.py
  OMDS_TAIL_SEL_STOP=$cfg python bench.py --steps 20 --warmup 3 --reps 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('ms_per_iteration', d['ms_per_step'], d['rep_ms_per_step'])"
done
done
