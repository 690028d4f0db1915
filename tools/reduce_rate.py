# transcript-set reduction of the bench batch (100k x 64v/256e, ~1.88 M transcripts): device path (ald_batch_reduce_transcripts) vs the
# host sink (ald_tset_add_batch); prints milliseconds per batch
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aletsch_amd as A
n = int(os.environ.get("N", "100000"))
pg = A.synth(seed=1002, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)
sid = (np.arange(n) % 8).astype(np.int32)
import ctypes as C
with A.DecompBatch(0) as b:
    b.add(pg); b.upload(); b.run(); b.download()
    lib = b._lib
    for rep in range(3):
        t0 = time.time(); s = A.TranscriptSink(0.8); s.add_batch(b, sid); t1 = time.time()
        h = C.c_void_p(); sp = C.c_void_p(sid.ctypes.data)
        rc = lib.ald_batch_reduce_transcripts(b._h, sp, C.c_int64(0), C.c_int32(0), C.c_double(0.8), C.byref(h)); t2 = time.time()
        assert rc == 0
        st = [C.c_double(), C.c_double(), C.c_int64(), C.c_int64()]; lib.ald_tset_flat_stats(h, *[C.byref(x) for x in st])
        ni = C.c_int64(); lib.ald_tset_flat_size(h, C.byref(ni), None, None); lib.ald_tset_flat_free(h)
        print("rep %d: host sink %.1f ms | device reduction: call %.1f ms (device part %.1f ms), %d groups on the device, %d items on the host, %d items"
              % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st[0].value, st[2].value, st[3].value, ni.value), flush=True)
        s.close()
