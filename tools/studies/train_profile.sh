#!/bin/bash
# Kernel trace of the SDF trainer at 1.03 M rows (profiles/rNN_kernel_trace_stats_train_sdf.txt): per-kernel table + per launch shape.
#   bash tools/studies/train_profile.sh > gpurun_out/train_profile.txt
export TMPDIR=/tmp
finddb() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -- python3 tools/train_sdf_hip.py --rows ${ROWS:-1048576} --epochs ${EPOCHS:-10} > /tmp/prof_tr.log 2>&1
tail -4 /tmp/prof_tr.log
python3 tools/rocprof_summary.py stats "$(finddb /tmp/prof_tr)" | head -12
echo
python3 tools/rocprof_summary.py shapes "$(finddb /tmp/prof_tr)" | head -24
