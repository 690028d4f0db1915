import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
pg = A.synth(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)
for k in (8, 12, 16, 18, 20):
    os.environ["ALD_WG_PER_CU"] = str(k)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); ms = []
        for rep in range(2):
            b.run(); b.download(); ms.append(b.kernel_ms())
        print(k, "kernel_ms %.2f" % min(ms), flush=True)
