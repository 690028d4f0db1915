import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
for k in (1, 2, 3):
    os.environ["ALD_WG_PER_CU"] = str(k)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); ms = []
        for rep in range(2):
            b.run(); b.download(); ms.append(b.kernel_ms())
        print("cap", k, "kernel_ms %.1f" % min(ms), flush=True)
