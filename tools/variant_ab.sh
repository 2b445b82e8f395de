#!/bin/bash
# same-box A/B of library variants: bash tools/variant_ab.sh <suffix> <suffix> ... (libomds_hip_<suffix>.so), three interleaved rounds each
for round in 1 2 3; do
  for v in "$@"; do
    OMDS_LIB=$PWD/optimalmodulationds_amd/csrc/libomds_hip_$v.so python bench.py --steps 10 --warmup 3 --reps 5 --no-cpu-baseline --no-secondary --path fp32 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],3), round(r['avg_launch_ms']*1e3,1))"
  done
done
