# cfg3 (10 000 graphs, V in [8, 512]) against the oracle, graph by graph, with the library's own choice of LDS form / slab twins
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A, common
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
t0 = time.time(); want = common.oracle_run(pg, threads=16)[0]; t1 = time.time()
with A.DecompBatch(0) as b:
    b.add(pg); b.upload(); b.run(); b.download(); got = b.result()
    used = {c: b.class_info(c)["n_graphs"] for c in range(13) if b.class_info(c)["n_graphs"]}
    ms = b.kernel_ms()
bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
print("cfg3: classes", used, "kernel %.1f ms, oracle %.1f s on 16 threads, mismatching graphs: %d of %d" % (ms, t1 - t0, len(bad), pg.n), flush=True)
