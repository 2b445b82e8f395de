# k_exact launch time for candidate counts / tile schemes (GPU): bash tools/studies/exact_probe.sh
export TMPDIR=/tmp
for cfg in "0 0.0075" "0 0.015" "0 0.03"; do
  set -- $cfg
  mkdir -p gpurun_out/kt
  EPS=$2 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -- python3 tools/studies/exact_probe.py > gpurun_out/kt.log 2>&1
  echo "eps=$2: $(grep -o "candidates_per_rollout_step.: [0-9.]*" gpurun_out/kt.log)"
  python3 tools/rocprof_summary.py stats "$(find gpurun_out/kt -name "*_results.db" | head -1)" | grep -E "k_exact" | head -1 | cut -c1-110
  rm -rf gpurun_out/kt
done
