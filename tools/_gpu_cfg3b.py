# cfg3 broken down by size class: each class's graphs alone, then all together
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
V = pg.g_nv
def run(sel, name):
    sub = pg.select(np.nonzero(sel)[0])
    with A.DecompBatch(0) as b:
        b.add(sub); b.upload(); ms = []
        for rep in range(2):
            b.run(); b.download(); ms.append(b.kernel_ms())
        info = [(c, b.class_info(c)["n_graphs"], b.class_info(c)["blocks_per_cu"]) for c in range(13) if b.class_info(c)["n_graphs"]]
        print(name, "graphs", sub.n, "kernel_ms %.1f" % min(ms), "classes(n, wg/cu)", info, flush=True)
run(V <= 32, "V<=32")
run((V > 32) & (V <= 64), "33..64")
run((V > 64) & (V <= 128), "65..128")
run((V > 128) & (V <= 256), "129..256")
run((V > 256) & (V <= 384), "257..384")
run((V > 384), "385..512")
run(V > 0, "all")
