# host sink throughput: all transcripts of a 100k-graph batch through ald_tset_add_batch
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import aletsch_amd as A
n = 100000
pg = A.synth(seed=1002, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)
with A.DecompBatch(0) as b:
    b.add(pg); b.upload(); b.run(); b.download()
    t0 = time.time(); r = b.result(); t1 = time.time()
    s = A.TranscriptSink(0.8); sid = (np.arange(n) % 8).astype(np.int32)
    t2 = time.time(); s.add_batch(b, sid); t3 = time.time()
    print("index/export %.2f s; sink add_batch %.2f s for %d transcripts -> %.0f transcripts/s, %.0f graphs/s" % (t1 - t0, t3 - t2, len(r.weight), len(r.weight) / (t3 - t2), n / (t3 - t2)), flush=True)
    t4 = time.time(); it = s._lib  # noqa
    import ctypes as C
    a = C.c_int64(); e = C.c_int64(); sm = C.c_int64(); s._lib.ald_tset_size(s._h, C.byref(a), C.byref(e), C.byref(sm)); print("items", a.value, "exons", e.value)
