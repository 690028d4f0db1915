# cfg3 broken down by size band (each band's graphs alone, LDS form and slab twins), then the whole batch:
#   python tools/cfg3_bands.py [lib.so ...]
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, "ROOTDIR")
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
V = pg.g_nv
def run(sel, name, twin=None):
    sub = pg.select(np.nonzero(sel)[0])
    if twin is not None: os.environ["ALD_DEBUG_TWIN"] = twin
    else: os.environ.pop("ALD_DEBUG_TWIN", None)
    with A.DecompBatch(0) as b:
        b.add(sub); b.upload(); ms = []
        for rep in range(2):
            b.run(); b.download(); ms.append(b.kernel_ms())
        info = [(c, b.class_info(c)["n_graphs"], b.class_info(c)["blocks_per_cu"]) for c in range(14) if b.class_info(c)["n_graphs"]]
        print("   ", name, "twin=%s" % twin, "graphs", sub.n, "kernel_ms %.1f" % min(ms), "classes(n, wg/cu)", info, flush=True)
run(V <= 64, "V<=64")
run((V > 64) & (V <= 128), "65..128")
run((V > 128) & (V <= 256), "129..256")
run((V > 256) & (V <= 384), "257..384")
run((V > 384), "385..512", "0")
run((V > 384), "385..512", "1")
run(V > 0, "all")
'''.replace("ROOTDIR", ROOT)
for lib in (sys.argv[1:] or [os.path.join(ROOT, "aletsch_amd/lib/libaletsch_decomp.so")]):
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, ALETSCH_DECOMP_LIB=os.path.abspath(lib)), check=False)
