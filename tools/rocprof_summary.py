#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (results.db) into the small text summaries kept under
profiles/: per-kernel stats of a --kernel-trace --stats run and per-kernel means of --pmc runs."""
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = [f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}"]
    for n, c, tot, avg, pct in rows:
        out.append(f"{n[:70]:70s} {c:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
    cols = "name, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x)"
    out.append("")
    out.append(f"{'kernel':40s} {'n':>5s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'grid':>9s} {'wg':>5s}")
    for r in cur.execute(f"select {cols} from kernels group by name order by sum(duration) desc"):
        out.append(f"{r[0][:40]:40s} {r[1]:5d} {r[2]:9.2f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:5d} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:7d} {r[10]:9d} {r[11]:5d}")
    return "\n".join(out)


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    out = [f"{'kernel':40s} {'counter':16s} {'n':>5s} {'mean':>16s} {'min':>16s} {'max':>16s}"]
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name")
    for r in cur.execute(q):
        out.append(f"{r[0][:40]:40s} {r[1]:16s} {r[2]:5d} {r[3]:16.1f} {r[4]:16.1f} {r[5]:16.1f}")
    return "\n".join(out)


def markers(db):
    """roctx ranges of an OMDS_ROCTX=1 run under --marker-trace: host-side enqueue time per range name (the kernels
    run asynchronously; the ranges annotate the timeline with the reference's record_function tags)."""
    import json
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for ext, dur in cur.execute("select extdata, duration from regions where category like 'MARKER%'"):
        try:
            name = json.loads(ext).get("message", "?")
        except Exception:
            name = "?"
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += dur / 1e3
    out = [f"{'roctx range':60s} {'n':>6s} {'host_us_total':>14s} {'host_us_avg':>12s}"]
    for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{name[:60]:60s} {n:6d} {tot:14.1f} {tot / n:12.2f}")
    return "\n".join(out)


def shapes(db):
    """Per (kernel, grid) launch shape: calls and mean / min duration -- one GEMM template serves several layer shapes, which the
    per-kernel table averages together."""
    cur = sqlite3.connect(db).cursor()
    out = [f"{'kernel':48s} {'grid':>22s} {'wg':>5s} {'n':>5s} {'avg_us':>10s} {'min_us':>10s}"]
    q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration)/1000.0, min(duration)/1000.0 from kernels "
         "group by name, grid_x, grid_y, grid_z order by sum(duration) desc")
    for r in cur.execute(q):
        out.append(f"{r[0][:48]:48s} {str(r[1]) + 'x' + str(r[2]) + 'x' + str(r[3]):>22s} {r[4]:5d} {r[5]:5d} {r[6]:10.2f} {r[7]:10.2f}")
    return "\n".join(out)


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    print({"stats": kernel_stats, "pmc": pmc, "markers": markers, "shapes": shapes}[mode](db))
