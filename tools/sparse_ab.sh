#!/bin/bash
# A/B of k_pass1's exact zero-skip: kernel times and LDS / MFMA counters, dense (OMDS_FLAG_DENSE_PASS1) vs per-tile compaction.  gpurun -- 'bash tools/sparse_ab.sh'
set -u
OUT=$PWD/gpurun_out/sparse_ab
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary --path fp32"
for v in compacted dense; do
  F=""; [ $v = dense ] && F="--dense-pass1"
  rocprofv3 --kernel-trace --stats -d "$OUT/kt_$v" -- python3 $B $F > "$OUT/kt_$v.log" 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats "$(find "$OUT/kt_$v" -name '*_results.db' | head -1)" | head -4
  rm -rf "$OUT/kt_$v"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES -d "$OUT/pmc_$v" -- python3 $B $F > "$OUT/pmc_$v.log" 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py pmc "$(find "$OUT/pmc_$v" -name '*_results.db' | head -1)" | grep -i "pass1" | head -8
  rm -rf "$OUT/pmc_$v"
done
