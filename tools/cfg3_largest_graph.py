# run time of the LARGEST graphs of cfg3 alone (one wave each, nothing else on the GPU), LDS form of their class against the slab twin:
#   python tools/cfg3_largest_graph.py        -> the bound a batch cannot go below however its waves are scheduled
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
order = np.argsort(-(pg.g_ne.astype(np.int64) * 1024 + pg.g_nv))
for top in (1, 16, 256):
    sub = pg.select(np.sort(order[:top]))
    with A.DecompBatch(0) as b:
        b.add(sub); b.upload(); ms = []
        for rep in range(3): b.run(); b.download(); ms.append(b.kernel_ms())
        r = b.result(); it = b.iterations() if hasattr(b, "iterations") else None
        info = [(c, b.class_info(c)["n_graphs"]) for c in range(14) if b.class_info(c)["n_graphs"]]
    print("   the %%3d largest graphs (V %%d..%%d, E %%d..%%d) twin=%%s: kernel ms %%s classes %%s bad %%d" %% (top, sub.g_nv.min(), sub.g_nv.max(), sub.g_ne.min(), sub.g_ne.max(), os.environ.get("ALD_DEBUG_TWIN"), ["%%.1f" %% x for x in ms], info, int((r.status != 0).sum())), flush=True)
''' % ROOT
for tw in ("0", "1"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, ALD_DEBUG_TWIN=tw), check=False)
