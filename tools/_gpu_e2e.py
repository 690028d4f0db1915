# host-buffer-to-host-results rate (PCIe inclusive): add -> upload -> run -> download, buffers warm on the second round
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
n = 100000
pg = A.synth(seed=1002, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)
with A.DecompBatch(0) as b:
    for rnd in range(3):
        t0 = time.time(); b.clear(); b.add(pg); t1 = time.time(); b.upload(); t2 = time.time(); b.run(); b.sync(); t3 = time.time(); b.download(); t4 = time.time()
        print("round %d: add %.3f upload %.3f kernel %.3f download %.3f total %.3f s -> %.0f graphs/s" % (rnd, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, n / (t4 - t0)), flush=True)
