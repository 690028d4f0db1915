# catch-all class (hot state in HBM) and class 4 at scale: GPU vs oracle
import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A, common
thr = max(1, min(16, len(os.sched_getaffinity(0))))
for kw in (dict(seed=301, n_graphs=200, v_min=600, v_max=1500, edges_per_vertex=3),
           dict(seed=302, n_graphs=120, v_min=700, v_max=1200, edges_per_vertex=4, phasing_per_graph=30, n_samples=3, weight_mode=1),
           dict(seed=303, n_graphs=60, v_min=1500, v_max=2000, edges_per_vertex=3, strand_mode=1, layout_mode=1),
           dict(seed=304, n_graphs=3000, v_min=300, v_max=512, edges_per_vertex=4, phasing_per_graph=10, weight_mode=2)):
    pg = A.synth(**kw)
    t0 = time.time(); want = common.oracle_run(pg, threads=thr)[0]; t1 = time.time()
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); b.run(); b.download(); got = b.result(); ms = b.kernel_ms()
        cls = [(c, b.class_info(c)["n_graphs"]) for c in range(13) if b.class_info(c)["n_graphs"]]
    bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
    print(kw, "classes", cls, "oracle %.1f s, kernel %.0f ms" % (t1 - t0, ms), "status!=0", int((want.status != 0).sum()), "MISMATCH " + str(bad[:3]) if bad else "ok", flush=True)
