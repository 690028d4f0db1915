# slab-resident twins of classes 7 / 8 versus their LDS form: batches of only large graphs (V 390..512), several counts
import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
for n in (400, 800, 1200, 1600, 2500, 5000):
    pg = A.synth(seed=1003, n_graphs=n, v_min=390, v_max=512, edges_per_vertex=4)
    out = []
    for force in ("0", "1", None):
        if force is None: os.environ.pop("ALD_DEBUG_TWIN", None)
        else: os.environ["ALD_DEBUG_TWIN"] = force
        with A.DecompBatch(0) as b:
            b.add(pg); b.upload(); ms = []
            for rep in range(2):
                b.run(); b.download(); ms.append(b.kernel_ms())
            out.append(min(ms))
    print("graphs %5d  LDS form %.1f ms  slab twins %.1f ms  chosen by the library %.1f ms" % (n, out[0], out[1], out[2]), flush=True)
# the realistic shape: a bench-sized batch of small graphs plus a few hundred large ones
from aletsch_amd.packed import PackedGraphs
small = A.synth(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)
for nb in (50, 300, 1500):
    big = A.synth(seed=1003, n_graphs=nb, v_min=390, v_max=512, edges_per_vertex=4)
    pg = PackedGraphs.concat([small, big]); out = []
    for force in ("0", "1", None):
        if force is None: os.environ.pop("ALD_DEBUG_TWIN", None)
        else: os.environ["ALD_DEBUG_TWIN"] = force
        with A.DecompBatch(0) as b:
            b.add(pg); b.upload(); ms = []
            for rep in range(2):
                b.run(); b.download(); ms.append(b.kernel_ms())
            out.append(min(ms))
    print("100000 small + %4d large  LDS form %.1f ms  slab twins %.1f ms  chosen by the library %.1f ms" % (nb, out[0], out[1], out[2]), flush=True)
