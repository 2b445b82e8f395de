#!/usr/bin/env python3
"""Phase timeline of k_tail_sel (and of k_tail on 4-row groups: --path fp32) from the diagnostic build (-DOMDS_TAIL_TL): reads the 'TL ...' lines the dump kernel printed
(stdin or a file) and reports, over the workgroups of one launch, when each phase boundary is reached after the first
workgroup's entry (median / max, microseconds at the measured shader clock) and the clock itself.

    make -C optimalmodulationds_amd/csrc experiment VFLAGS=-DOMDS_TAIL_TL        # -> libomds_hip_exp.so (reads OMDS_TAIL_TL_STEP)
    OMDS_LIB=optimalmodulationds_amd/csrc/libomds_hip_exp.so OMDS_TAIL_TL_STEP=5 python bench.py --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary 2>&1 | python tools/tail_timeline.py
"""
import statistics
import sys

NAMES = ["entry", "entry", "top-k done", "masks in MFMA layout", "seed done", "first GEMM: wave 0 done", "first GEMM: all waves",
         "first GEMM: epilogue done", "hidden layers done", "backward done", "modulation done", "next layer 1 done"]
rows = []
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if line.startswith("TL "):
        try:
            v = [int(x) for x in line.split()[1:]]
        except ValueError:   # the line the bench's JSON was appended to
            continue
        if len(v) >= 14 and v[1] and v[12]:
            rows.append(v)
if not rows:
    sys.exit("no TL lines")
# shader clock from the two real-time stamps (100 MHz) around the cycle stamps 1 and 11
ghz = statistics.median((r[12] - r[2]) / ((r[-1] - r[1]) * 10.0) for r in rows if r[-1] > r[1])
t0 = min(r[1] for r in rows)
print("%d workgroups, shader clock %.2f GHz, entries spread over %.2f us, last exit %.2f us after the first entry"
      % (len(rows), ghz, (max(r[1] for r in rows) - t0) / 100.0, (max(r[-1] for r in rows) - t0) / 100.0))
print("%-28s %10s %10s   (us after the workgroup's own entry)" % ("phase boundary", "median", "max"))
for i in range(3, 13):
    d = [(r[i] - r[2]) / (ghz * 1e3) for r in rows]
    print("%-28s %10.2f %10.2f" % (NAMES[i - 1], statistics.median(d), max(d)))
if len(rows[0]) >= 19:
    for i, nm in ((13, "mod: q + nominal DS"), (14, "mod: gradient blend"), (15, "mod: normal, sigmoids, act"), (16, "mod: policy"), (17, "mod: velocity")):
        d = [(r[i] - r[2]) / (ghz * 1e3) for r in rows if r[i]]
        if d:
            print("%-28s %10.2f %10.2f" % (nm, statistics.median(d), max(d)))
if len(rows[0]) >= 21 and rows[0][18]:   # k_tail on 4-row groups (pass2_body_g4 stamps 3, 17, 18): its own phase list
    print("-- k_tail<.., 4>: phase boundary, median / max us after the workgroup's own entry")
    for i, nm in ((3, "top-k done"), (4, "inputs gathered"), (18, "layer 1 done"), (19, "hidden forward done"), (5, "last layer + seed done"),
                  (9, "hidden backward done"), (10, "backward done"), (11, "modulation done"), (12, "next inputs written")):
        d = [(r[i] - r[2]) / (ghz * 1e3) for r in rows if r[i]]
        print("%-28s %10.2f %10.2f" % (nm, statistics.median(d), max(d)))
late = sorted(rows, key=lambda r: r[1])
print("entry of the 257th workgroup: %.2f us after the first" % ((late[256][1] - t0) / 100.0) if len(late) > 256 else "at most 256 workgroups")
