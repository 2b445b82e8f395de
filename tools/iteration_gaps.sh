#!/bin/bash
# Where one planner iteration's wall time goes on the GPU timeline: busy time per kernel and the idle gaps between
# consecutive kernels (host synchronisations, launch ramps), from a rocprofv3 kernel trace of bench.py.
# usage (GPU box, repo root): bash tools/iteration_gaps.sh [workload]
set -u
WL=${1:-franka_shelf_1024x32}
export TMPDIR=/tmp
OUT=gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/kt -- python3 bench.py --workload $WL --steps 6 --warmup 2 --reps 1 --prof-stride 0 --no-cpu-baseline --no-secondary > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name '*_results.db' | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# iterations are delimited by k_sample launches
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_sample")]
its = [(idx[j], idx[j + 1]) for j in range(len(idx) - 1)][-3:]
for a, b in its:
    seg = rows[a:b]
    span = (rows[b][1] - seg[0][1]) / 1e3
    busy = sum(e - s for _, s, e in seg) / 1e3
    gaps = [((seg[i + 1][1] if i + 1 < len(seg) else rows[b][1]) - seg[i][2], seg[i][0][:28], (seg[i + 1][0] if i + 1 < len(seg) else rows[b][0])[:28]) for i in range(len(seg))]
    big = sorted(gaps, reverse=True)[:6]
    small = sum(g for g, _, _ in gaps if g < 5000) / 1e3
    print(f"iteration: span {span:.0f} us, kernels busy {busy:.0f} us, idle {span - busy:.0f} us (of which gaps < 5 us: {small:.0f} us over {len(seg)} launches)")
    for g, x, y in big:
        print(f"    gap {g / 1e3:7.1f} us  after {x:28s} before {y}")
PY
rm -rf $OUT/kt
