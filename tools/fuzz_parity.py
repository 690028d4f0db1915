import os
# randomized parity sweep GPU vs oracle: many configurations x seeds at sizes where rare interleavings occur
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aletsch_amd as A, common
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
thr = max(1, min(16, len(os.sched_getaffinity(0))))
budget = float(os.environ.get("FUZZ_SECONDS", "240")); t_end = time.time() + budget
nbad = 0; ntot = 0; k = 0
while time.time() < t_end:
    k += 1
    vmin = int(rng.choice([4, 8, 16, 32, 64, 100])); vmax = int(min(520, vmin * rng.choice([1, 2, 4])))
    kw = dict(seed=int(rng.integers(1, 1 << 30)), v_min=vmin, v_max=vmax, edges_per_vertex=int(rng.choice([2, 3, 4, 5])),
              weight_mode=int(rng.choice([0, 1, 2])), n_samples=int(rng.choice([1, 1, 2, 4])), phasing_per_graph=int(rng.choice([0, 0, 3, 10, 25])),
              strand_mode=int(rng.choice([0, 0, 1])), layout_mode=int(rng.choice([0, 1])))
    work = (vmin + vmax) / 2 * kw["edges_per_vertex"]
    kw["n_graphs"] = int(max(50, min(20000, 2.5e6 / work)))
    if os.environ.get("FUZZ_EMU"): kw["n_graphs"] = max(20, kw["n_graphs"] // 25)
    p = A.default_params()
    if rng.random() < 0.3: p.max_decompose_error_ratio[7] = float(rng.choice([1.2, 1.5, 3.0]))
    if rng.random() < 0.3: p.max_decompose_error_ratio[0] = float(rng.choice([0.1, 0.2, 0.5]))
    if rng.random() < 0.2: p.min_transcript_coverage = float(rng.choice([0.5, 5.0]))
    if rng.random() < 0.2: p.max_num_exons = int(rng.choice([12, 40, 150]))       # cascade cut short -> the greedy phase does the work
    pg = A.synth(**kw)
    if rng.random() < 0.3:
        cnt = pg.sample_counts() + rng.integers(0, 3, pg.edge_target.size).astype(np.int32); pg.edge_count = cnt.astype(np.int32)
    want = common.oracle_run(pg, params=p, threads=thr)[0]
    got = common.emu_run(pg, params=p)[0] if os.environ.get("FUZZ_EMU") else A.decompose(pg, device=0, params=p)      # FUZZ_EMU=1: the single-lane emulation on the CPU (ALD_EMU_LIB: a build with -DALD_EMU_CHECK verifies its caches / rows as it goes)
    bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
    ntot += pg.n; nbad += len(bad)
    print(k, "graphs", pg.n, {a: b for a, b in kw.items() if a != "n_graphs"}, "params", [round(x, 2) for x in p.max_decompose_error_ratio], p.min_transcript_coverage, p.max_num_exons,
          "status!=0", int((want.status != 0).sum()), "MISMATCH " + str(bad[:3]) if bad else "ok", flush=True)
print("TOTAL graphs", ntot, "mismatches", nbad, flush=True)
