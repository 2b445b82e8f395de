export TMPDIR=/tmp
for st in 1 2 3 0; do
  mkdir -p gpurun_out/kt
  OMDS_SCREEN_NOGUARD=1 OMDS_TAIL_SEL_STOP=$st rocprofv3 --kernel-trace --stats -d gpurun_out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/kt.log 2>&1
  echo "stop $st: $(python3 tools/rocprof_summary.py stats "$(find gpurun_out/kt -name "*_results.db" | head -1)" | grep k_tail_sel | head -1 | cut -c60-110)"
  rm -rf gpurun_out/kt
done
