import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import common, aletsch_amd as A, numpy as np
for variant in ("all","jump","small","cov"):
    p = A.default_params()
    if variant in ("all","jump"): p.max_decompose_error_ratio[7] = 1.5
    if variant in ("all","small"): p.max_decompose_error_ratio[0] = 0.2
    if variant in ("all","cov"): p.min_transcript_coverage = 5.0
    pg = A.synth(seed=31, n_graphs=200, v_min=10, v_max=70, edges_per_vertex=3, phasing_per_graph=8, weight_mode=1)
    want, st, _, tr = common.oracle_run(pg, params=p, trace=True)
    with A.DecompBatch(0, params=p, trace_events=4096) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        got = b.result()
        bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
        print(variant, 'mismatch', bad[:3])
        nb=0
        for g in range(pg.n):
            mine = b.trace(g)
            if mine != tr[g]:
                k = next((i for i,(x,y) in enumerate(zip(mine,tr[g])) if x!=y), min(len(mine),len(tr[g])))
                print('  graph',g,'V',pg.g_nv[g],'first divergence at event',k,'gpu',mine[k-1:k+2],'oracle',tr[g][k-1:k+2])
                nb+=1
                if nb>=3: break
