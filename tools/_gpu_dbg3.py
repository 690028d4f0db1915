import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import common, aletsch_amd as A, numpy as np
pg = A.synth(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)
with A.DecompBatch(0) as b:
    b.add(pg); b.upload(); b.run(); b.download(); got = b.result()
npaths = np.diff(got.path_offset)
np.save('/root/repo/gpurun_out/npaths_gpu.npy', npaths)
print('total', npaths.sum(), 'bad status', int((got.status != 0).sum()))
