# When does each size-class kernel of the mixed batch (cfg3: 10k graphs, V in [8, 512]) start and end?
#   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/cfg3_timeline.py run     (one warm-up run + one traced run)
#   python tools/cfg3_timeline.py parse gpurun_out/tl                                                      (timeline of the LAST run)
import glob, os, sys, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import aletsch_amd as A
    pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload()
        for rep in range(2):
            b.run(); b.download()
        print("kernel_ms", b.kernel_ms())
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ald_decomp_kernel" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0) // 64))
    rows.sort()
    names = [r[2] for r in rows]
    half = len(rows) // 2                                    # two runs: keep the second
    rows = rows[half:]
    t0 = rows[0][0]
    for s, e, n, wg in rows: print(f"  {n:28s} workgroups {wg:6d}  start {(s - t0) / 1e6:8.2f} ms  end {(e - t0) / 1e6:8.2f} ms  ({(e - s) / 1e6:7.2f} ms)")
    print(f"  whole batch: {(max(r[1] for r in rows) - t0) / 1e6:.2f} ms")
