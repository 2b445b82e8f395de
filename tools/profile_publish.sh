#!/bin/bash
# Copies the summaries tools/profile_refresh.sh left under gpurun_out/prof_<round>/ (scratch, merged back by gpurun) into profiles/
# under their committed names.  Run here, after the gpurun call:  bash tools/profile_publish.sh r04
set -eu
R=${1:-r06}
S=gpurun_out/prof_$R
D=profiles
cp "$S/bench.json" "$D/${R}_bench_franka_shelf_1024x32.json"
cp "$S/stats_franka_shelf_1024x32.txt" "$D/${R}_kernel_trace_stats.txt"
cp "$S/stats_fp32.txt" "$D/${R}_kernel_trace_stats_fp32.txt"
for wl in franka_shelf_4096x32 planar7_1024x32 franka_tanh_4096x32 franka_dynamic_1024x32 franka_shelf_4096x64 franka_shelf_8192x32; do
  [ -s "$S/stats_$wl.txt" ] && cp "$S/stats_$wl.txt" "$D/${R}_kernel_trace_stats_$wl.txt"
done
cat "$S/pmc_FETCH_SIZE.txt" "$S/pmc_WRITE_SIZE.txt" > "$D/${R}_pmc_hbm.txt"
cat "$S/pmc_fp32_FETCH_SIZE.txt" "$S/pmc_fp32_WRITE_SIZE.txt" > "$D/${R}_pmc_hbm_fp32.txt"
cat "$S/pmc_p7_FETCH_SIZE.txt" "$S/pmc_p7_WRITE_SIZE.txt" > "$D/${R}_pmc_hbm_planar7_1024x32.txt"
cp "$S/pmc_sq.txt" "$D/${R}_pmc_sq.txt"
[ -s "$S/pmc_sq_fp32.txt" ] && cp "$S/pmc_sq_fp32.txt" "$D/${R}_pmc_sq_fp32.txt"
cp "$S/parity_fullsize.txt" "$D/${R}_parity_fullsize.txt"
cp "$S/pmc_traffic.json" "$D/pmc_traffic.json"
[ -s gpurun_out/${R}_soak.txt ] && cp gpurun_out/${R}_soak.txt "$D/${R}_screen_error_hist.txt"
[ -s gpurun_out/${R}_soak.json ] && cp gpurun_out/${R}_soak.json "$D/${R}_screen_error_hist.json"
ls -la $D | grep "${R}_"
