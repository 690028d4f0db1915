import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
for name, kw in (("c5 (LDS, V 700..1000)", dict(seed=311, n_graphs=256, v_min=700, v_max=1000, edges_per_vertex=4)),
                 ("c6 (HBM, V 1100..1500)", dict(seed=312, n_graphs=256, v_min=1100, v_max=1500, edges_per_vertex=4))):
    pg = A.synth(**kw)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); ms = []
        for rep in range(2):
            b.run(); b.download(); ms.append(b.kernel_ms())
        print(name, "graphs", pg.n, "kernel_ms %.0f" % min(ms), [(c, b.class_info(c)["n_graphs"]) for c in range(13) if b.class_info(c)["n_graphs"]], flush=True)
