# cfg3 with n6 / n5 graphs of LDS classes 6 / 5 routed to the slab twins (ALD_TWIN_SPILL=n6,n5; "default" = the library's own rule):
#   python tools/cfg3_twin_spill.py 0,0 200,0 400,200 default
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import aletsch_amd as A
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
with A.DecompBatch(0) as b:
    b.add(pg); b.upload(); ms = []
    for rep in range(3):
        b.run(); b.download(); ms.append(b.kernel_ms())
    r = b.result()
    print("  kernel_ms", ["%%.1f" %% x for x in ms], "bad", int((r.status != 0).sum()), "classes", {c: b.class_info(c)["n_graphs"] for c in range(14) if b.class_info(c)["n_graphs"]}, flush=True)
''' % ROOT
for sp in sys.argv[1:]:
    print("ALD_TWIN_SPILL=%s" % sp, flush=True)
    env = dict(os.environ)
    if sp != "default": env["ALD_TWIN_SPILL"] = sp
    subprocess.run([sys.executable, "-c", CHILD], env=env)
