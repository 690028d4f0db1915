import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
for n in (2000, 20000, 100000):
    pg = A.synth(seed=1002, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)
    with A.DecompBatch(0) as b:
        t0=time.time(); b.add(pg); t1=time.time(); b.upload(); t2=time.time()
        for rep in range(3):
            b.run(); b.sync(); 
        t3=time.time(); b.download(); t4=time.time()
        r=b.result()
        print(n, 'add %.2fs upload %.2fs run(3x) %.3fs download %.2fs kernel_ms(last) %.2f'%(t1-t0,t2-t1,t3-t2,t4-t3,b.kernel_ms()), 'graphs/s(kernel)=%.0f'%(n/(b.kernel_ms()/1e3)), 'status!=0', int((r.status!=0).sum()), 'paths', len(r.weight), b.class_info(1), flush=True)
