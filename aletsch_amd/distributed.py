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


def gather_records(rec: np.ndarray, device: torch.device, graph_offset: int = 0):
    """Variable-length gather of uint32 record streams to rank 0: all_gather the sizes, pad to the max, gather.
    `graph_offset` is added to the graph-id word of every local record so ids are global.  Returns the list of
    per-rank streams on rank 0, None elsewhere."""
    world = dist.get_world_size(); rank = dist.get_rank()
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    if graph_offset:
        rec = rec.copy()
        o = 0
        while o + REC_HDR_WORDS <= rec.size:
            rec[o] += np.uint32(graph_offset)
            w = REC_HDR_WORDS + int(rec[o + 2]); o += w + (w & 1)
    t = torch.from_numpy(rec.view(np.int32)).to(device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.numel()], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.int32, device=device); pad[: t.numel()] = t
    out = [torch.empty(mx, dtype=torch.int32, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, out, dst=0)
    if rank != 0:
        return None
    return [o[:s].cpu().numpy().view(np.uint32) for o, s in zip(out, sizes)]


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
