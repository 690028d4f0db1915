# randomized parity sweep for RAW graphs (row f1: the pre-steps of assembler::assemble on the device) GPU vs oracle:
#   FUZZ_SEED=1 FUZZ_SECONDS=240 python tools/fuzz_raw.py
# Every round: a few hundred gene-like graphs as assemble(gx, px, sid) receives them (runs of touching partial exons, junctions, several
# close start / end boundaries, junctions over one partial exon, multi-sample supports, counts, a random creation order) with phase sets
# in exon coordinates (some invalid), a random max_group_boundary_distance; the oracle runs its own pre-steps + decomposition graph by
# graph, the GPU takes the graphs raw through the bulk entry point.  Graphs on which the reference would assert must end >= 100.
import os, sys, time, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aletsch_amd as A, common
from aletsch_amd.packed import PackedGraphs
import test_pre_steps_cpu as T
O = common.oracle_lib()
O.ora_pre_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
O.ora_staged_view.argtypes = [C.c_void_p, C.c_void_p]; O.ora_staged_free.argtypes = [C.c_void_p]; O.ora_staged_boundary_maps.argtypes = [C.c_void_p] * 5
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", "240"))
ntot = 0; nbad = 0; nassert = 0; k = 0
while time.time() < t_end:
    k += 1
    dist = int(rng.choice([0, 50, 500, 10000, 10000, 1000000]))
    big = rng.random() < 0.3
    n = 150 if big else 400
    want_parts = []; asserted = []; raws = []; all_phases = []
    for t in range(n):
        g, phases = common.gene_like_raw(rng, n_runs=int(rng.integers(3, 40 if big else 12)), strand="+-."[int(rng.integers(0, 3))])
        if rng.random() < 0.15: phases = phases + phases[:2]
        if rng.random() < 0.1: phases = []
        pg = PackedGraphs.from_graphs([g])
        pg.edge_rank = np.array(sorted(range(len(g["edges"])), key=lambda q: (g["edges"][q][0], g["edges"][q][1])), np.int32)
        if rng.random() < 0.5: pg.edge_count = (pg.sample_counts() + rng.integers(0, 3, pg.edge_target.size)).astype(np.int32)
        want, _, _, rc_o = A.pre_assemble(pg, phases, dist, _lib=O, _prefix="ora")
        asserted.append(rc_o != 0); raws.append(pg); all_phases.append(phases)
        if rc_o == 0: want_parts.append(want)
    asserted = np.array(asserted)
    if os.environ.get("FUZZ_EMU"):           # the same device code in the single-lane emulation (CPU): a dry run of this script
        got = common.emu_run_raw([(pg, ph, dist) for pg, ph in zip(raws, all_phases)])[0]
    else:
        with A.DecompBatch(0) as b:
            b.add_packed_raw(PackedGraphs.concat(raws), dist, all_phases); b.upload(); b.run(); b.download()
            got = b.result()
    bad = []
    if not (got.status[asserted] >= 100).all(): bad.append(("asserting graphs must end with an invariant status", got.status[asserted].tolist()[:10]))
    if want_parts:
        batch = PackedGraphs.concat(want_parts)
        want = common.oracle_run(batch, threads=8)[0]
        sub = T.common_select_results(got, np.nonzero(~asserted)[0])
        bad += common.compare_results(want, sub, batch.n, conf_tol=1e-9)
    ntot += n; nbad += len(bad); nassert += int(asserted.sum())
    print(k, "graphs", n, "dist", dist, "big" if big else "small", "asserted", int(asserted.sum()), "MISMATCH " + str(bad[:3]) if bad else "ok", flush=True)
print("TOTAL raw graphs", ntot, "of which asserting in the pre-steps", nassert, "mismatches", nbad)
