# quick A/B of library builds: kernel ms for 100k graphs of cfg1's shape (32v/96e: class 0), the bench workload (100k x 64v/256e), its flow-weight variant and cfg3 (10k, V in [8,512]);
#   python tools/kernel_ab.py [lib.so ...]      (each library in a fresh child process: ALETSCH_DECOMP_LIB selects the build)
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import aletsch_amd as A
for name, kw in (("cfg1x1000", dict(seed=1001, n_graphs=100000, v_min=32, v_max=32, fixed_edges=96)), ("cfg2", dict(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)), ("cfg2flow", dict(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256, weight_mode=2)), ("cfg3", dict(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4))):
    pg = A.synth(**kw)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload()
        ms = []
        for rep in range(3):
            b.run(); b.download(); ms.append(b.kernel_ms())
        r = b.result()
        print("   ", name, "kernel_ms", ["%%.2f" %% x for x in ms], "bad", int((r.status != 0).sum()), "paths", len(r.weight), "wg/CU", [b.class_info(c)["blocks_per_cu"] for c in (0, 1, 2, 4, 8, 12)], flush=True)
''' % ROOT
for lib in (sys.argv[1:] or [os.path.join(ROOT, "aletsch_amd/lib/libaletsch_decomp.so")]):
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, ALETSCH_DECOMP_LIB=os.path.abspath(lib)), check=False)
