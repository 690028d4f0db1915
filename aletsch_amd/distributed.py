"""Multi-GPU plumbing: bundles shard embarrassingly (one process per GPU, no data-path collective); the only exchange
is the final gather of the packed path records to rank 0 -- the analogue of the reference's per-graph
``tm.add(ts, ...)`` under ``mylock`` (meta/assembler.cc:1127-1132), done once per batch over RCCL (backend "nccl")
or gloo (CPU tests).  Merge order on rank 0 is ascending global graph id, independent of the number of ranks."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REC_HDR_WORDS = 14


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


class RecordGatherer:
    """The one exchange step of the multi-GPU path: every rank's packed path records -> rank 0.

    Sizes travel by all_gather, payloads by a padded `gather` (RCCL when the tensors are on the GPU, gloo on CPU).  Buffers are
    kept across steps.  Nothing on the data path is touched by Python per record: graph ids stay local in the stream and each
    rank's `graph_offset` travels with the sizes, so rank 0 can make them global when it consumes a stream
    (`aletsch_amd.records_add_graph_offset`, a C loop).  On a GPU rank 0 the gathered streams are copied to pinned host memory
    with an async copy on the current stream (the copy engine works while the next batch's kernel runs); `streams()` waits for it.
    """

    def __init__(self, device: torch.device):
        self.device = device
        self.world = dist.get_world_size(); self.rank = dist.get_rank()
        self._pad = None; self._out = None; self._host = None; self._meta = None; self._done = None

    def gather(self, words: torch.Tensor, graph_offset: int = 0):
        """words: 1-D int32 tensor on self.device (this rank's record stream).  Starts the exchange; rank 0 reads `streams()`."""
        dev = self.device
        mine = torch.tensor([int(words.numel()), int(graph_offset)], dtype=torch.int64, device=dev)
        meta = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(meta, mine)
        self._meta = [(int(m[0].item()), int(m[1].item())) for m in meta]
        mx = max(max(n for n, _ in self._meta), 1)
        if self._pad is None or self._pad.numel() < mx:
            cap = mx + mx // 8 + 64                                       # head-room: the stream length varies a little from batch to batch
            self._pad = torch.zeros(cap, dtype=torch.int32, device=dev)
            self._out = None
        if self._done is not None:
            self._done.synchronize()                                      # the previous step's host copy must be out of the buffers
        pad = self._pad[:mx]
        pad[: words.numel()] = words
        if self.rank == 0:
            if self._out is None or self._out[0].numel() < mx:
                cap = self._pad.numel()
                self._out = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(self.world)]
                self._host = torch.empty((self.world, cap), dtype=torch.int32, pin_memory=(dev.type == "cuda"))
            outs = [o[:mx] for o in self._out]
            dist.gather(pad, outs, dst=0)
        else:
            outs = None
            dist.gather(pad, None, dst=0)
        if dev.type == "cuda":
            # the exchange has read `words` (the batch's record pool, which its next run overwrites) once this event is reached
            ev = torch.cuda.Event(); ev.record(); ev.synchronize()
        if self.rank == 0:
            for i, o in enumerate(outs):
                self._host[i, :mx].copy_(o, non_blocking=True)            # pinned target: the copy engine works behind the next kernel
            if dev.type == "cuda":
                self._done = torch.cuda.Event(); self._done.record()

    def streams(self):
        """Rank 0: [(uint32 record words of rank i, graph_offset of rank i)] in rank order == ascending global graph id."""
        if self.rank != 0:
            return None
        if self._done is not None:
            self._done.synchronize()
        return [(self._host[i, :n].numpy().view(np.uint32), off) for i, (n, off) in enumerate(self._meta)]


def gather_records(rec, device: torch.device, graph_offset: int = 0):
    """One-shot form: `rec` is a uint32 numpy stream (or an int32 tensor already on `device`).  Returns on rank 0 the list of
    per-rank streams with GLOBAL graph ids (copies), None elsewhere."""
    if isinstance(rec, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.uint32).view(np.int32)).to(device)
    else:
        t = rec
    g = RecordGatherer(device)
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
        nv = int(words[o + 2])
        f = words[o + 6:o + 14].view(np.float64)
        out.append(dict(graph=int(words[o]), index=int(words[o + 1]), length=int(words[o + 3]), count=int(words[o + 4]),
                        strand=chr(int(words[o + 5]) & 0xFF), attempt=(int(words[o + 5]) >> 8) & 0xFF,
                        weight=float(f[0]), abd=float(f[1]), conf=float(f[2]), reads=float(f[3]),
                        v=words[o + REC_HDR_WORDS:o + REC_HDR_WORDS + nv].astype(np.int32).tolist()))
        w = REC_HDR_WORDS + nv; o += w + (w & 1)
    out.sort(key=lambda r: (r["graph"], r["index"]))
    return out
