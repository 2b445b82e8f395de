#!/bin/bash
# k_tail on 16- vs 32-row tiles at the headline shape (experiment build: OMDS_TAIL_ROWS).  gpurun -- 'bash tools/tail_rows_ab.sh'
for r in 0 16 32; do
  OMDS_TAIL_ROWS=$r OMDS_LIB=$PWD/optimalmodulationds_amd/csrc/libomds_hip_exp.so python bench.py --path fp32 --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --reps 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OMDS_TAIL_ROWS=$r', d['value'], d['ms_per_step'])"
done
