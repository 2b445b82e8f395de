export OMDS_LIB=$PWD/optimalmodulationds_amd/csrc/libomds_hip_exp.so
B="python bench.py --steps 20 --warmup 5 --reps 6 --no-cpu-baseline --no-secondary"
for round in 1 2; do
for cfg in "0 0" "3 0" "2 0" "0 1" "3 1"; do
  set -- $cfg
  echo "AB=$1 PRIO=$2: $(OMDS_AB=$1 OMDS_AB_PRIO=$2 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['rep_ms_per_step'], d['roofline']['avg_launch_ms'])")"
done; done
