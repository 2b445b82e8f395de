#!/bin/bash
# Where the tall GEMM's time goes (experiment build): OMDS_TALL_DBG bit 1 = no A traffic, 2 = no C traffic, 4 = no mask traffic.
export TMPDIR=/tmp
export OMDS_LIB=$(pwd)/optimalmodulationds_amd/csrc/libomds_hip_exp.so
finddb() { find "$1" -name "*results.db" | head -1; }
for dbg in ${CFGS:-0 1 2 3 7}; do
  export OMDS_TALL_DBG=$dbg
  rm -rf /tmp/prof_tr
  rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -- python3 tools/train_sdf_hip.py --rows 1048576 --epochs 4 > /tmp/prof_tr.log 2>&1
  echo "## OMDS_TALL_DBG=$dbg"
  python3 tools/rocprof_summary.py shapes "$(finddb /tmp/prof_tr)" | grep -E "k_gemm_tall" | head -2
done
