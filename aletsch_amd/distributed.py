"""Multi-GPU plumbing: bundles shard embarrassingly (one process per GPU, no data-path collective); the only exchange
is the final gather of every rank's FINISHED transcripts to rank 0 -- the analogue of the reference's per-graph
``tm.add(ts, ...)`` under ``mylock`` (meta/assembler.cc:1127-1132), done once per batch over RCCL (backend "nccl")
or gloo (CPU tests).  What travels is the self-contained transcript stream of ``ald_batch_transcript_stream``
(records of abandoned capacity attempts and of failed graphs already dropped, exons already joined), so rank 0 needs
nothing else from the sender to merge it: ``ald_tset_add_stream`` in rank order == ascending global graph id,
independent of the number of ranks.  The raw record pool can be gathered too (same gatherer: any
int32 stream), but it is only meaningful together with the sender's status / attempt arrays and vertex coordinates."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REC_HDR_WORDS = 16          # decomp_common.h: record header words ([14] = number of exon words behind the vertices)


def shard_range(n_graphs: int, rank: int, world: int):
    """Contiguous block partition of graph ids [lo, hi) for this rank."""
    base, rem = divmod(n_graphs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device_words(ptr: int, n_words: int, device: torch.device) -> torch.Tensor:
    """Zero-copy int32 view of `n_words` record words at device address `ptr` (what ald_batch_device_records returns)."""
    class _Span:
        pass
    sp = _Span()
    sp.__cuda_array_interface__ = {"shape": (max(int(n_words), 1),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(sp, device=device)[: int(n_words)]


class StreamGatherer:
    """The one exchange step of the multi-GPU path: every rank's word stream (finished transcripts) -> rank 0.

    Every rank sends a fixed-capacity buffer [n_words, graph_offset, 0, 0, records...] through a `gather` (RCCL when the tensors
    are on the GPU, gloo on CPU).  The capacity is agreed once (all_gather of the first step's sizes, plus head-room), so a step
    needs no size exchange and NOTHING in it blocks the host: the gather is only enqueued, the next batch's kernel is launched
    right behind it and the two share the GPU, and on rank 0 the gathered streams go to pinned host memory by async copies that
    the copy engine performs meanwhile.  Graph ids stay local in the stream; each rank's `graph_offset` rides in the header, so
    rank 0 makes them global when it consumes a stream (`aletsch_amd.records_add_graph_offset`, a C loop).  The agreement is a
    collective, so it is never entered by one rank alone: a stream that outgrows the capacity raises, and a workload whose batch
    shape changes calls `renegotiate()` on every rank at that point (the capacity carries 3 % head-room over the largest stream).
    """
    HDR = 4

    def __init__(self, device: torch.device, headroom: float = 1.0 / 32):
        self.device = device
        self.world = dist.get_world_size(); self.rank = dist.get_rank()
        self.headroom = headroom
        self.cap = 0
        self._pad = None; self._out = None; self._host = None; self._done = None

    def _largest(self, n_words: int) -> int:
        dev = self.device
        mine = torch.tensor([int(n_words)], dtype=torch.int64, device=dev)
        allv = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(allv, mine)
        return max(int(v.item()) for v in allv)

    def renegotiate(self, n_words: int):
        """Blocking: agree on a capacity that holds every rank's stream of this shape."""
        self._set_capacity(self._largest(n_words))

    def gather_any(self, words: torch.Tensor, graph_offset: int = 0):
        """`gather` for callers whose batch shape varies from step to step: the sizes are exchanged first (one 8-byte all_gather,
        blocking), and a stream that would not fit grows the capacity ON EVERY RANK in the same step -- all ranks see the same sizes,
        so all take the same decision and nothing is ever raised or entered alone.  Costs one small collective per step; the
        fixed-capacity `gather` stays the form for steady shapes."""
        mx = self._largest(int(words.numel()))
        if self.cap == 0 or self.HDR + mx > self.cap:
            self._set_capacity(mx)
        return self.gather(words, graph_offset)

    def _set_capacity(self, mx: int):
        dev = self.device
        cap = self.HDR + mx + int(mx * self.headroom) + 1024
        cap += (-cap) % 64
        if self._done is not None:
            self._done.synchronize()
        self.cap = cap
        self._pad = torch.zeros(cap, dtype=torch.int32, device=dev)
        if self.rank == 0:
            self._out = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(self.world)]
            self._host = torch.empty((self.world, cap), dtype=torch.int32, pin_memory=(dev.type == "cuda"))

    def gather(self, words: torch.Tensor, graph_offset: int = 0):
        """words: 1-D int32 tensor on self.device (this rank's record stream).  Enqueues the exchange and returns an event (None on
        CPU) that is reached once `words` has been read -- the caller waits for it before it lets anything overwrite `words`."""
        dev = self.device
        n = int(words.numel())
        if self.cap == 0:
            self.renegotiate(n)                                           # collective: on first use every rank gets here together
        elif self.HDR + n > self.cap:
            # never enter a collective alone: the other ranks are about to enqueue the gather of this step
            raise ValueError(f"record stream of {n} words exceeds the capacity agreed by all ranks ({self.cap - self.HDR}): "
                             "call renegotiate() on EVERY rank when the batch shape changes")
        if self._done is not None:
            self._done.synchronize()                                      # the previous step's host copies must be out of the buffers
        pad = self._pad
        pad[: self.HDR] = torch.tensor([n, int(graph_offset), 0, 0], dtype=torch.int32).to(dev, non_blocking=True)
        pad[self.HDR: self.HDR + n] = words
        read = None
        if dev.type == "cuda":
            read = torch.cuda.Event(); read.record()
        dist.gather(pad, self._out if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            for i, o in enumerate(self._out):
                self._host[i].copy_(o, non_blocking=True)                 # pinned target: the copy engine works behind the next kernel
            if dev.type == "cuda":
                self._done = torch.cuda.Event(); self._done.record()
        return read

    def streams(self):
        """Rank 0: [(uint32 record words of rank i, graph_offset of rank i)] in rank order == ascending global graph id."""
        if self.rank != 0:
            return None
        if self._done is not None:
            self._done.synchronize()
        out = []
        for i in range(self.world):
            row = self._host[i]; n = int(row[0]); off = int(row[1])
            out.append((row[self.HDR: self.HDR + n].numpy().view(np.uint32), off))
        return out


RecordGatherer = StreamGatherer          # round-1 name


def merge_streams(sink, streams, tid_base: int = 0):
    """Rank 0: merge gathered transcript streams [(words, graph_offset)] in rank order (== ascending global graph id) into a
    ``TranscriptSink``; the result equals the single-rank merge of the unsharded batch."""
    for words, off in streams:
        sink.add_stream(np.ascontiguousarray(words, dtype=np.uint32), graph_offset=int(off), tid_base=tid_base)
    return sink


def gather_records(rec, device: torch.device, graph_offset: int = 0):
    """One-shot form: `rec` is a uint32 numpy stream (or an int32 tensor already on `device`).  Returns on rank 0 the list of
    per-rank streams with GLOBAL graph ids (copies), None elsewhere."""
    if isinstance(rec, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.uint32).view(np.int32)).to(device)
    else:
        t = rec
    g = StreamGatherer(device)
    g.gather(t, graph_offset)
    st = g.streams()
    if st is None:
        return None
    from .native import records_add_graph_offset
    out = []
    for words, off in st:
        w = words.copy()
        if off:
            records_add_graph_offset(w, off)
        out.append(w)
    return out


def parse_records(words: np.ndarray):
    """Decode a record stream into a list of dicts sorted by (graph, path index)."""
    out = []; o = 0
    while o + REC_HDR_WORDS <= words.size:
        nv = int(words[o + 2]); nx = int(words[o + 14])
        f = words[o + 6:o + 14].view(np.float64)
        out.append(dict(graph=int(words[o]), index=int(words[o + 1]), length=int(words[o + 3]), count=int(words[o + 4]),
                        strand=chr(int(words[o + 5]) & 0xFF), attempt=(int(words[o + 5]) >> 8) & 0xFF,
                        weight=float(f[0]), abd=float(f[1]), conf=float(f[2]), reads=float(f[3]),
                        v=words[o + REC_HDR_WORDS:o + REC_HDR_WORDS + nv].astype(np.int32).tolist(),
                        exons=words[o + REC_HDR_WORDS + nv:o + REC_HDR_WORDS + nv + nx].astype(np.int32).reshape(-1, 2).tolist()))
        w = REC_HDR_WORDS + nv + nx; o += w + (w & 1)
    out.sort(key=lambda r: (r["graph"], r["index"]))
    return out
