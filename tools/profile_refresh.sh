#!/bin/bash
# Regenerates the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun from the repo root).
# Every rocprofv3 call has python3 directly after "--"; counters are collected in their own passes.
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
# the library's default (screened) step unless a line says --path fp32; bench.py's own default primary is the all-fp32 step
B="bench.py --steps 5 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary --path screened"
P="bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-secondary --path screened"
finddb() { find "$1" -name '*_results.db' | head -1; }
# the all-fp32 step (omds_set_screening(0)): the precision-matched figure of the bench line (roofline.fp32_only), k_pass1 + k_tail
rocprofv3 --kernel-trace --stats -d "$OUT/kt_fp32" -- python3 $B --path fp32 > "$OUT/kt_fp32.log" 2>&1
python3 tools/rocprof_summary.py stats "$(finddb "$OUT/kt_fp32")" > "$OUT/stats_fp32.txt"
tail -1 "$OUT/kt_fp32.log" | grep '^{' > "$OUT/bench_profiled_fp32.json"
rm -rf "$OUT/kt_fp32"
for wl in franka_shelf_1024x32 franka_shelf_4096x32 planar7_1024x32 franka_tanh_4096x32 franka_dynamic_1024x32 franka_shelf_4096x64 franka_shelf_8192x32; do
  rocprofv3 --kernel-trace --stats -d "$OUT/kt_$wl" -- python3 $B --workload $wl > "$OUT/kt_$wl.log" 2>&1
  python3 tools/rocprof_summary.py stats "$(finddb "$OUT/kt_$wl")" > "$OUT/stats_$wl.txt"
  tail -1 "$OUT/kt_$wl.log" | grep '^{' > "$OUT/bench_profiled_$wl.json"
  rm -rf "$OUT/kt_$wl"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" -- python3 $P > "$OUT/pmc_$c.log" 2>&1
  python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_$c")" > "$OUT/pmc_$c.txt"
  rm -rf "$OUT/pmc_$c"
  rocprofv3 --pmc $c -d "$OUT/pmc_fp32_$c" -- python3 $P --path fp32 > "$OUT/pmc_fp32_$c.log" 2>&1
  python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_fp32_$c")" > "$OUT/pmc_fp32_$c.txt"
  rm -rf "$OUT/pmc_fp32_$c"
  rocprofv3 --pmc $c -d "$OUT/pmc_p7_$c" -- python3 $P --workload planar7_1024x32 > "$OUT/pmc_p7_$c.log" 2>&1
  python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_p7_$c")" > "$OUT/pmc_p7_$c.txt"
  rm -rf "$OUT/pmc_p7_$c"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/pmc_sq" -- python3 $P > "$OUT/pmc_sq.log" 2>&1
python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_sq")" > "$OUT/pmc_sq.txt"
rm -rf "$OUT/pmc_sq"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA -d "$OUT/pmc_sq_fp32" -- python3 $P --path fp32 > "$OUT/pmc_sq_fp32.log" 2>&1
python3 tools/rocprof_summary.py pmc "$(finddb "$OUT/pmc_sq_fp32")" > "$OUT/pmc_sq_fp32.txt"
rm -rf "$OUT/pmc_sq_fp32"
# HBM bytes per launch of each workload's kernels -> profiles/pmc_traffic.json (read by bench.py for roofline.traffic)
python3 - "$OUT" "$R" <<'PY'
import json, re, sys
out = sys.argv[1]
def table(path, counters=("FETCH_SIZE", "WRITE_SIZE")):
    t = {}
    try:
        for line in open(path):
            m = re.match(r"^(.{40}) (\S+)\s+(\d+)\s+([0-9.]+)", line)
            if m and m.group(2) in counters:
                t[m.group(1).strip()] = float(m.group(4))
    except OSError:
        pass
    return t
res = {"_round": sys.argv[2] if len(sys.argv) > 2 else "?", "_doc": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 from separate rocprofv3 --pmc passes of "
               "bench.py --workload <workload> (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md); mfma_insts = SQ_INSTS_MFMA per launch "
               "(wave-level MFMA instructions executed) from the SQ pass of the same command"}
for wl, tag in (("franka_shelf_1024x32", ""), ("franka_shelf_1024x32_fp32", "_fp32"), ("planar7_1024x32", "_p7")):
    f, w = table(out + "/pmc%s_FETCH_SIZE.txt" % tag), table(out + "/pmc%s_WRITE_SIZE.txt" % tag)
    mf = table(out + "/pmc_sq%s.txt" % tag, ("SQ_INSTS_MFMA",)) if tag != "_p7" else {}
    res[wl] = {}
    for name in f:
        short = "k_screen" if "k_screen" in name else "k_step_small" if "k_step_small" in name else "k_tail" if "k_tail" in name else \
                "k_exact" if "k_exact" in name else "k_select" if "k_select" in name else "k_audit" if "k_audit" in name else "k_pass1" if "k_pass1" in name else None
        if short and name in w:
            res[wl][short] = {"fetch_kb": f[name], "write_kb": w[name], "traffic_bytes": int((2 * f[name] + w[name]) * 1024)}
            if name in mf:
                res[wl][short]["mfma_insts"] = mf[name]
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
# the committed bench line once more, now beside THIS run's counters (roofline.traffic / SQ_INSTS_MFMA are read from profiles/pmc_traffic.json)
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 -m pytest tests/test_gpu_fullsize_parity.py -q -s > "$OUT/parity_fullsize.txt" 2>&1
ls -la "$OUT"
