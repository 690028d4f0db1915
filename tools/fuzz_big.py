# randomized parity sweep over LARGE graphs (the one-workgroup-per-CU class, the catch-all and the class beyond it: hot state in the slab,
# 32-bit creation ids above 2 048 vertices) GPU vs oracle:   FUZZ_SEED=1 FUZZ_SECONDS=300 python tools/fuzz_big.py
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aletsch_amd as A, common
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", "300"))
thr = max(1, min(16, len(os.sched_getaffinity(0))))
ntot = 0; nbad = 0; k = 0; by_class = {}
while time.time() < t_end:
    k += 1
    lo = int(rng.choice([520, 700, 1030, 1500, 2050, 2600]))
    if os.environ.get("FUZZ_VMIN"): lo = int(os.environ["FUZZ_VMIN"]) + int(rng.integers(0, 64))      # e.g. FUZZ_VMIN=385 ALD_DEBUG_TWIN=1: the slab twins' band
    kw = dict(seed=int(rng.integers(1, 1 << 30)), v_min=lo, v_max=int(lo * rng.choice([1.0, 1.2, 1.5])), edges_per_vertex=int(rng.choice([2, 3, 4])),
              weight_mode=int(rng.choice([0, 1, 2])), n_samples=int(rng.choice([1, 1, 2, 4])), phasing_per_graph=int(rng.choice([0, 0, 5, 40])),
              strand_mode=int(rng.choice([0, 0, 1])), layout_mode=int(rng.choice([0, 1])), n_graphs=int(rng.choice([2, 4, 8])) * (8 if os.environ.get("FUZZ_VMIN") else 1))
    p = A.default_params()
    if rng.random() < 0.25: p.max_decompose_error_ratio[7] = float(rng.choice([1.2, 1.5]))
    if rng.random() < 0.2: p.max_num_exons = int(rng.choice([600, 1200, 2100]))      # some graphs leave the rule loop at once, or after they grew
    pg = A.synth(**kw)
    want = common.oracle_run(pg, params=p, threads=thr)[0]
    with A.DecompBatch(0, params=p) as b:
        b.add(pg); b.upload(); b.run(); b.download(); got = b.result()
        for c in range(14):
            n = b.class_info(c)["n_graphs"]
            if n: by_class[c] = by_class.get(c, 0) + n
    bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
    ntot += pg.n; nbad += len(bad)
    print(k, "graphs", pg.n, kw, "status!=0", int((got.status != 0).sum()), "MISMATCH " + str(bad[:3]) if bad else "ok", flush=True)
print("TOTAL large graphs", ntot, "by first class", by_class, "mismatches", nbad)
