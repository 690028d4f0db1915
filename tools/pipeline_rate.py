import os
# Sustained host-arrays-in -> merged-transcript-set-out rate with the stages overlapped on host threads (ctypes drops the GIL):
#   stage (add + upload)  |  kernel (run + sync)  |  download (D2H + decode)  |  merge      -- four batches in flight
#   PIPE_MERGE=host (default): the 16-thread host sink (ald_tset_add_batch);  PIPE_MERGE=reduce: the batch reduced on the GPU
#   (ald_batch_reduce_transcripts) and the reduced set zipped into the persistent one (ald_tset_add_flat)
import sys, time, threading, queue, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aletsch_amd as A
n = 100000; rounds = 8
pgs = [A.synth(seed=1002 + k, n_graphs=n, v_min=64, v_max=64, fixed_edges=256) for k in range(2)]
sid = (np.arange(n) % 8).astype(np.int32)
batches = [A.DecompBatch(0) for _ in range(4)]
free = queue.Queue(); [free.put(b) for b in batches]
staged = queue.Queue(maxsize=1); ran = queue.Queue(maxsize=1); done = queue.Queue(maxsize=1)
sink = A.TranscriptSink(0.8)
tstage = tkern = tdl = tsink = 0.0
def stage():
    global tstage
    for r in range(rounds):
        b = free.get(); t = time.time(); b.clear(); b.add(pgs[r % 2]); b.upload(); tstage += time.time() - t; staged.put(b)
    staged.put(None)
def kern():
    global tkern
    while True:
        b = staged.get()
        if b is None: ran.put(None); return
        t = time.time(); b.run(); b.sync(); tkern += time.time() - t; ran.put(b)
def fetch():
    global tdl
    while True:
        b = ran.get()
        if b is None: done.put(None); return
        t = time.time(); b.download(); tdl += time.time() - t; done.put(b)
def merge():
    global tsink
    r = 0
    while True:
        b = done.get()
        if b is None: return
        t = time.time()
        if os.environ.get("PIPE_MERGE", "host") == "reduce": b.reduce_into(sink, sid, tid_base=r << 44)
        else: sink.add_batch(b, sid, tid_base=r << 44)
        tsink += time.time() - t; r += 1; free.put(b)
# warm-up round (allocations), then the timed pipeline
for b in batches: b.add(pgs[0]); b.upload(); b.run(); b.download(); b.clear()
th = [threading.Thread(target=f) for f in (stage, kern, fetch, merge)]
t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; el = time.time() - t0
print(os.environ.get("PIPE_MERGE", "host"), "merge; pipeline: %d batches x %d graphs in %.2f s -> %.0f graphs/s end to end (stage %.2f s, kernel %.2f s, download %.2f s, merge %.2f s busy)" % (rounds, n, el, rounds * n / el, tstage, tkern, tdl, tsink), flush=True)
