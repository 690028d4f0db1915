import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A, common
pg = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
with A.DecompBatch(0) as b:
    t0=time.time(); b.add(pg); b.upload(); t1=time.time()
    for rep in range(3):
        b.run(); b.download()
        print('cfg3 run: kernel_ms %.1f  -> %.0f graphs/s' % (b.kernel_ms(), pg.n/(b.kernel_ms()/1e3)), [ (c, b.class_info(c)['n_graphs'], b.class_info(c)['blocks_last_run']) for c in range(5)], flush=True)
    r=b.result()
    print('status counts', np.unique(r.status, return_counts=True))
sub = pg.select(np.arange(0, 200))
want = common.oracle_run(sub)[0]; got = A.decompose(sub, 0)
print('parity on 200:', common.compare_results(want, got, 200, conf_tol=1e-9))
