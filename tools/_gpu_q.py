# quick A/B: kernel ms for the bench workload (100k x 64v/256e) and cfg3 (10k, V in [8,512])
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
for name, kw in (("cfg2", dict(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)), ("cfg2flow", dict(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256, weight_mode=2)), ("cfg3", dict(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4))):
    pg = A.synth(**kw)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload()
        ms = []
        for rep in range(3):
            b.run(); b.download(); ms.append(b.kernel_ms())
        r = b.result()
        print(name, "kernel_ms", ["%.2f" % x for x in ms], "graphs/s %.0f" % (pg.n / (min(ms) / 1e3)), "bad", int((r.status != 0).sum()), "grids", [b.class_info(c)["blocks_last_run"] for c in range(13)], "paths", len(r.weight), [b.class_info(c)["blocks_per_cu"] for c in range(13)], flush=True)
