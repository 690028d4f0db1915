#!/usr/bin/env python3
"""Turn rocprofv3 result databases (gpurun_out/<dir>/*_results.db) into the small summaries committed under profiles/.

  python profiles/summarize.py stats  gpurun_out/r03a_stats/s_results.db  profiles/r03/a_kernel_stats.csv --skip-first 4 --bench-log gpurun_out/r03a_stats.log
  python profiles/summarize.py pmc    gpurun_out/pmc_fetch_r1_d/d_results.db gpurun_out/pmc_write_r1_d/d_results.db profiles/r01/d_pmc_traffic.json
  python profiles/summarize.py sq     gpurun_out/r02b_pmc3/p_results.db gpurun_out/r02b_pmc4/p_results.db profiles/r02/b_pmc_sq.json

`pmc` writes per-launch HBM traffic of the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests
at 64 B, so read bytes = 2 x FETCH_SIZE x 1024 (an upper bound for this kernel's narrow accesses, which the guide calls
uncalibrated); WRITE_SIZE x 1024 is taken as is.  bench.py reads the JSON for `roofline.traffic`.
"""
import json
import sqlite3
import sys


def stats(db, out, skip_first=0, bench_log=None):
    """per-kernel table (all launches) + for the dominant kernel the launches after the first `skip_first` (cold allocations):
    avg / median / min / max.  With a bench log (the JSON line of the same command) the steady-state average must not exceed the
    line's ms_per_step -- a kernel that takes longer than the step it is part of is not evidence of anything."""
    c = sqlite3.connect(db)
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), max(lds_size), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name order by sum(duration) desc")
    rows = list(c.execute(q)); tot = sum(r[2] for r in rows)
    dom = rows[0][0]
    d = [r[0] for r in c.execute("select duration from kernels where name=? order by start", (dom,))][skip_first:]
    d.sort(); n = len(d)
    avg = sum(d) / n; med = d[n // 2] if n % 2 else 0.5 * (d[n // 2 - 1] + d[n // 2])
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,Grid,Workgroup,LDS,VGPR,AGPR,SGPR,Scratch\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.4f,%d,%d,%d,%d,%d,%d,%d\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, *r[6:]))
        f.write('# steady state of "%s": launches after the first %d -> calls=%d avg_ns=%.1f median_ns=%.1f min_ns=%d max_ns=%d\n' % (dom, skip_first, n, avg, med, d[0], d[-1]))
        if bench_log:
            line = [ln for ln in open(bench_log) if ln.startswith('{"metric"')][-1]
            j = json.loads(line)
            f.write('# bench line of the same run: ms_per_step=%.3f roofline.kernel_ms=%.3f (HIP events) roofline.frac=%.5f\n' % (j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
            f.write('# source of the run: %s\n' % json.dumps(j.get("source")))
            if avg * 1e-6 > j["ms_per_step"]:
                print(open(out).read()); sys.exit("steady-state kernel average %.3f ms exceeds ms_per_step %.3f ms" % (avg * 1e-6, j["ms_per_step"]))
    print(open(out).read())


def source_of(bench_log):
    """the `source` block (git HEAD, kernel source / library hashes) of the bench line in a run's log, or None"""
    if not bench_log:
        return None
    try:
        line = [ln for ln in open(bench_log) if ln.startswith('{"metric"')][-1]
        return json.loads(line).get("source")
    except (OSError, IndexError, ValueError):
        return None


def per_launch(db, counter, kernel_prefix):
    c = sqlite3.connect(db)
    v = [r[0] for r in c.execute("select value from counters_collection where counter_name=? and kernel_name like ?", (counter, kernel_prefix + "%"))]
    return v


def pmc(fetch_db, write_db, out, kernel_prefix="ald_decomp_kernel", bench_log=None):
    f = per_launch(fetch_db, "FETCH_SIZE", kernel_prefix); w = per_launch(write_db, "WRITE_SIZE", kernel_prefix)
    fk = sum(f) / len(f); wk = sum(w) / len(w)
    d = {"kernel": kernel_prefix, "launches": [len(f), len(w)], "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
         "read_bytes_per_launch_corrected": 2 * fk * 1024, "write_bytes_per_launch": wk * 1024,
         "traffic_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
         "note": "gfx950: FETCH_SIZE doubled per the microarch guide (upper bound for narrow accesses); separate --pmc passes",
         "source": source_of(bench_log)}
    json.dump(d, open(out, "w"), indent=1); print(json.dumps(d, indent=1))


def main():
    if sys.argv[1] == "stats":
        # stats <db> <out.csv> [--skip-first N] [--bench-log <log with the JSON line>]
        a = sys.argv[4:]; skip = int(a[a.index("--skip-first") + 1]) if "--skip-first" in a else 0
        log = a[a.index("--bench-log") + 1] if "--bench-log" in a else None
        stats(sys.argv[2], sys.argv[3], skip, log)
    elif sys.argv[1] == "sq":
        # sq <db> [<db> ...] <out.json> [--bench-log <log>] [--kernel <prefix>]
        a = sys.argv[2:]; log = None; kern = "ald_decomp_kernel_c1"
        if "--bench-log" in a:
            i = a.index("--bench-log"); log = a[i + 1]; a = a[:i] + a[i + 2:]
        if "--kernel" in a:
            i = a.index("--kernel"); kern = a[i + 1]; a = a[:i] + a[i + 2:]
        sq(a[:-1], a[-1], kernel_prefix=kern, bench_log=log)
    else:
        # pmc <fetch db> <write db> <out.json> [--bench-log <log>]
        a = sys.argv[5:]
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], bench_log=(a[a.index("--bench-log") + 1] if "--bench-log" in a else None))


def sq(dbs, out, kernel_prefix="ald_decomp_kernel_c1", graphs=100000, bench_log=None):
    """SQ instruction / wait counters per launch and per graph from one or more --pmc result databases"""
    vals = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for n, v, k in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (kernel_prefix + "%",)):
            vals[n] = v
    d = {"kernel": kernel_prefix, "graphs_per_launch": graphs, "per_launch": vals, "per_graph": {k: v / graphs for k, v in vals.items()}, "source": source_of(bench_log)}
    json.dump(d, open(out, "w"), indent=1); print(json.dumps(d["per_graph"], indent=1))


if __name__ == "__main__":
    main()
