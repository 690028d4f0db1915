# cfg3 with the size-class kernels launched in another order (ALD_LDS_FIRST: 0 = by cost, the product; 1 = every LDS class before the slab-resident
# ones; 2 = the large LDS classes 5..9 first):   python tools/cfg3_launch_order.py
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
for rnd in range(2):
    for mode in ("0", "1", "2"):
        os.environ["ALD_LDS_FIRST"] = mode
        with A.DecompBatch(0) as b:
            b.add(pg); b.upload(); ms = []
            for rep in range(3): b.run(); b.download(); ms.append(b.kernel_ms())
            r = b.result()
        print("ALD_LDS_FIRST=%s kernel_ms %s bad %d" % (mode, ["%.1f" % x for x in ms], int((r.status != 0).sum())), flush=True)
