import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import common, aletsch_amd as A, numpy as np
pg = A.synth(**common.PARITY_CONFIGS["phasing"])
want, st, _, tr = common.oracle_run(pg, trace=True)
with A.DecompBatch(0, trace_events=4096) as b:
    b.add(pg); b.upload(); b.run(); b.download()
    got = b.result()
    print('mismatch', common.compare_results(want, got, pg.n, conf_tol=1e-9)[:3])
    for g in (65,):
        mine = b.trace(g)
        k = next((i for i,(x,y) in enumerate(zip(mine,tr[g])) if x!=y), min(len(mine),len(tr[g])))
        print('graph',g,'V',pg.g_nv[g],'E',pg.g_ne[g],'len gpu',len(mine),'oracle',len(tr[g]),'first divergence',k)
        print(' gpu', mine[max(0,k-3):k+3]); print(' ora', tr[g][max(0,k-3):k+3])
    r, it, cl = common.emu_run(pg, trace_cap=0)
    print('emu status 65', r.status[65], 'class', cl[65])
    for fc in (1,2):
        r, it, cl = common.emu_run(pg, force_class=fc); print('emu forced', fc, r.status[65], cl[65])
