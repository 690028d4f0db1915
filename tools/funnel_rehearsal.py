#!/usr/bin/env python3
"""Rank 0's funnel at N > 1, rehearsed on ONE GPU: W ranks as threads of this process, RCCL replaced by tests/host_adapter/mock_rccl.cc
(ALD_RCCL_LIB; device-to-device copies behind the nccl* entry points), every rank holding the finished-transcript stream of ITS OWN
cfg4 shard (BASELINE.json configs[3]: 125 000 graphs of 64v / 256e, seed 1004 + rank) in device memory.

Per step every rank calls ald_comm_gather_begin; rank 0 then waits for stream r to land in pinned host memory (ald_comm_gather_wait(r))
and merges it (ald_tset_add_stream) while stream r + 1 is still being copied.  Printed: receive (mock: D2D) + D2H until everything has
landed, the merge per stream, the whole step, and the same with the next step's gather begun BEFORE the merge (the buffers exist twice).
What the rehearsal cannot show is the xGMI time of the real receive: 8 x ~180 MB over 7 links of ~50 GB/s is a few milliseconds.

    ALD_RCCL_LIB=tests/_build/libmock_rccl.so python tools/funnel_rehearsal.py [W=8] [graphs per rank=125000] [steps=3]
"""
import ctypes as C, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mock = os.environ.get("ALD_RCCL_LIB")
if not mock:
    mock = os.path.join(ROOT, "tests", "_build", "libmock_rccl.so"); os.makedirs(os.path.dirname(mock), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", mock, os.path.join(ROOT, "tests", "host_adapter", "mock_rccl.cc")], check=True)
    os.environ["ALD_RCCL_LIB"] = mock
import numpy as np
import torch
import aletsch_amd as A
from aletsch_amd.distributed import _device_words
lib = A.load_library()
lib.ald_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
lib.ald_comm_gather_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
lib.ald_comm_gather_wait.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int32))]
lib.ald_comm_destroy.argtypes = [C.c_void_p]
lib.ald_tset_add_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64]

# every rank's stream, built by its own decomposition run, parked in a device tensor of its own
dev = torch.device("cuda", 0); streams = []; t0 = time.perf_counter()
with A.DecompBatch(0) as b:
    for r in range(W):
        pg = A.synth(seed=1004 + r, n_graphs=NG, v_min=64, v_max=64, fixed_edges=256)
        b.clear(); b.add(pg); b.upload(); b.run(); b.download()
        ptr, n = b.device_transcript_stream(None, True)                     # single-exon transcripts left out, as the reference's default
        streams.append(_device_words(ptr, n, dev).clone())
torch.cuda.synchronize()
print(f"{W} shards of {NG} graphs decomposed in {time.perf_counter() - t0:.1f} s; stream sizes (MB): {[round(4 * s.numel() / 1e6, 1) for s in streams]}", flush=True)

uid = (C.c_uint8 * 128)(); assert lib.ald_comm_unique_id(uid) == 0, lib.ald_last_error()
sink = A.TranscriptSink(0.8)
bar = threading.Barrier(W); rep = {}; err = []

def rank(r):
    try:
        comm = C.c_void_p(); assert lib.ald_comm_create(uid, W, r, 0, C.byref(comm)) == 0, lib.ald_last_error()
        src = streams[r]
        for mode in ("serial", "overlapped"):
            for k in range(STEPS):
                bar.wait(); t_begin = time.perf_counter()
                assert lib.ald_comm_gather_begin(comm, C.c_void_p(src.data_ptr()), C.c_int64(src.numel()), C.c_int32(r * NG)) == 0, lib.ald_last_error()
                t_enq = time.perf_counter()
                allw = C.POINTER(C.c_uint32)(); offs = C.POINTER(C.c_int64)(); goffs = C.POINTER(C.c_int32)()
                if r != 0:
                    assert lib.ald_comm_gather_wait(comm, -1, C.byref(allw), C.byref(offs), C.byref(goffs)) == 0
                    continue
                if mode == "serial":
                    assert lib.ald_comm_gather_wait(comm, -1, C.byref(allw), C.byref(offs), C.byref(goffs)) == 0
                    t_land = time.perf_counter(); tm = []
                    for q in range(W):
                        t_ = time.perf_counter()
                        assert lib.ald_tset_add_stream(sink._h, C.cast(C.addressof(allw.contents) + 4 * offs[q], C.c_void_p), C.c_int64(offs[q + 1] - offs[q]), C.c_int32(goffs[q]), C.c_int64((k + 1) << 44)) == 0, lib.ald_last_error()
                        tm.append(time.perf_counter() - t_)
                    t_end = time.perf_counter()
                    rep.setdefault(mode, []).append((t_enq - t_begin, t_land - t_begin, sum(tm), t_end - t_begin, tm))
                else:
                    tm = []; tl = []
                    for q in range(W):                                   # stream q is merged as soon as it has landed, q + 1 .. still on their way
                        assert lib.ald_comm_gather_wait(comm, q, C.byref(allw), C.byref(offs), C.byref(goffs)) == 0
                        tl.append(time.perf_counter() - t_begin); t_ = time.perf_counter()
                        assert lib.ald_tset_add_stream(sink._h, C.cast(C.addressof(allw.contents) + 4 * offs[q], C.c_void_p), C.c_int64(offs[q + 1] - offs[q]), C.c_int32(goffs[q]), C.c_int64((k + 9) << 44)) == 0, lib.ald_last_error()
                        tm.append(time.perf_counter() - t_)
                    t_end = time.perf_counter()
                    rep.setdefault(mode, []).append((t_enq - t_begin, tl[-1], sum(tm), t_end - t_begin, tm))
        # the buffers exist twice: gather k + 1 is begun BEFORE gather k is merged -- a step then costs max(gather, merge), not their sum
        bar.wait(); t_all = time.perf_counter()
        assert lib.ald_comm_gather_begin(comm, C.c_void_p(src.data_ptr()), C.c_int64(src.numel()), C.c_int32(r * NG)) == 0, lib.ald_last_error()
        for k in range(STEPS):
            allw = C.POINTER(C.c_uint32)(); offs = C.POINTER(C.c_int64)(); goffs = C.POINTER(C.c_int32)()
            assert lib.ald_comm_gather_wait(comm, -1, C.byref(allw), C.byref(offs), C.byref(goffs)) == 0
            mine = [(C.addressof(allw.contents) + 4 * offs[q], offs[q + 1] - offs[q], goffs[q]) for q in range(W)] if r == 0 else []
            if k + 1 < STEPS:
                assert lib.ald_comm_gather_begin(comm, C.c_void_p(src.data_ptr()), C.c_int64(src.numel()), C.c_int32(r * NG)) == 0, lib.ald_last_error()
            for a, n, g0 in mine:
                assert lib.ald_tset_add_stream(sink._h, C.c_void_p(a), C.c_int64(n), C.c_int32(g0), C.c_int64((k + 17) << 44)) == 0, lib.ald_last_error()
        if r == 0: rep["double"] = (time.perf_counter() - t_all) / STEPS
        bar.wait()
        assert lib.ald_comm_destroy(comm) == 0
    except BaseException as e:
        err.append(e)
        try: bar.abort()
        except Exception: pass

ths = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
for t in ths: t.start()
for t in ths: t.join()
if err: raise err[0]
tot_mb = sum(4 * s.numel() for s in streams) / 1e6
for mode in ("serial", "overlapped"):
    for k, (enq, land, merge, whole, tm) in enumerate(rep[mode]):
        print(f"{mode:10s} step {k}: begin returned after {1e3 * enq:7.1f} ms | all {W} streams ({tot_mb:.0f} MB) in pinned host memory after {1e3 * land:7.1f} ms | merge of the {W} streams {1e3 * merge:7.1f} ms ({', '.join('%.0f' % (1e3 * x) for x in tm)}) | step {1e3 * whole:7.1f} ms = {W * NG / whole / 1e6:.2f} M bundles/s through rank 0", flush=True)
print(f"double-buffered (gather k + 1 begun before gather k is merged): {1e3 * rep['double']:.1f} ms per step = {W * NG / rep['double'] / 1e6:.2f} M bundles/s through rank 0")
print(f"items in rank 0's set: {len(sink.items())}")
print("FUNNEL_OK")
