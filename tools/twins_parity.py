# the slab-resident twins (classes 11 / 12) on their own, GPU vs oracle: graphs of 385..512 vertices with ALD_DEBUG_TWIN=1 (every graph of
# classes 7 / 8 goes to its twin), one library per child process:   python tools/twins_parity.py [lib.so ...]
# (tests/test_gpu_parity.py::test_register_form_of_the_small_fans_on_the_twins runs it on the STARREG=1 build and on the product)
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import aletsch_amd as A, common
tot = 0; nbad = 0; ntwin = 0
for seed, epv in ((11, 3), (12, 3), (13, 4), (14, 2)):
    pg = A.synth(seed=seed, n_graphs=50, v_min=385, v_max=512, edges_per_vertex=epv)
    want = common.oracle_run(pg, threads=max(1, min(16, len(os.sched_getaffinity(0)))))[0]
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); b.run(); b.download(); got = b.result()
        info = {c: b.class_info(c)["n_graphs"] for c in range(14) if b.class_info(c)["n_graphs"]}
    bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
    tot += pg.n; nbad += len(bad); ntwin += info.get(11, 0) + info.get(12, 0)
    print("   seed", seed, "classes", info, "mismatch", len(bad), "status", dict(zip(*[a.tolist() for a in np.unique(got.status, return_counts=True)])), flush=True)
print("   TOTAL graphs", tot, "on the twins", ntwin, "mismatches", nbad, flush=True)
sys.exit(0 if nbad == 0 and ntwin == tot else 1)
''' % (ROOT, ROOT)
rc = 0
for lib in (sys.argv[1:] or [os.path.join(ROOT, "aletsch_amd/lib/libaletsch_decomp.so")]):
    print("library", os.path.basename(lib), flush=True)
    r = subprocess.run(["timeout", "-k", "10", "240", sys.executable, "-c", CHILD], env=dict(os.environ, ALETSCH_DECOMP_LIB=os.path.abspath(lib), ALD_DEBUG_TWIN="1"), check=False)
    print("   rc", r.returncode, flush=True)
    if r.returncode != 0: rc = 1; break        # (after a GPU fault nothing else is started)
sys.exit(rc)
