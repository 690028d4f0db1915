"""Packed splice-graph batches: the host-side (numpy) view of the wire format that
``ald_batch_add_packed`` (include/aletsch_decomp.h) takes.

A batch is the concatenation over graphs of the per-vertex / per-edge / per-sample arrays of
``ald_graph_view``; ``vertex_offset``, ``edge_sample_offset`` and ``phasing_offset`` are per-graph
local CSR offsets (each restarts at 0).  This mirrors, as flat arrays, what the reference keeps in
``splice_graph`` (rnacore/splice_graph.h:25-143), ``edge_info`` (rnacore/edge_info.h:14-35),
``vertex_info`` (rnacore/vertex_info.h:12-43) and ``hyper_set::nodes`` (scallop/hyper_set.h:34).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np

_I32 = np.int32
_F64 = np.float64


@dataclass
class PackedGraphs:
    g_nv: np.ndarray            # [n] int32  vertices per graph (incl. source 0 / sink V-1)
    g_ne: np.ndarray            # [n] int32
    g_np: np.ndarray            # [n] int32  phasing paths per graph
    vertex_offset: np.ndarray   # [sum(V+1)] int32 local out-CSR
    edge_target: np.ndarray     # [sum E] int32
    edge_weight: np.ndarray     # [sum E] float64
    edge_strand: np.ndarray     # [sum E] uint8
    edge_abd: np.ndarray        # [sum E] float64
    edge_sample_offset: np.ndarray  # [sum(E+1)] int32 local
    sample_id: np.ndarray       # [sum S] int32
    sample_abd: np.ndarray      # [sum S] float64
    vertex_weight: np.ndarray   # [sum V] float64
    vertex_lpos: np.ndarray     # [sum V] int32
    vertex_rpos: np.ndarray     # [sum V] int32
    vertex_type: np.ndarray     # [sum V] int32
    phasing_offset: np.ndarray  # [sum(P+1)] int32 local
    phasing_vertex: np.ndarray  # [sum len] int32
    phasing_count: np.ndarray   # [sum P] int32
    graph_strand: np.ndarray    # [n] int8 ('+', '-', '.')
    edge_count: Optional[np.ndarray] = None   # [sum E] int32 edge_info.count at hand-over; None => the number of supporting samples
    edge_rank: Optional[np.ndarray] = None    # [sum E] int32 creation rank (scallop edge index) per graph: a permutation of 0..E-1; None => CSR position

    @property
    def n(self) -> int:
        return int(self.g_nv.shape[0])

    def sample_counts(self) -> np.ndarray:
        """[sum E] number of supporting samples per edge (what edge_count defaults to)."""
        E = self.g_ne.astype(np.int64); eo = np.concatenate([[0], np.cumsum(E + 1)])
        d = np.diff(self.edge_sample_offset.astype(np.int64))
        keep = np.ones(d.shape[0], bool); keep[eo[1:-1] - 1] = False            # drop the differences across graph borders
        return d[keep].astype(np.int32)

    def identity_rank(self) -> np.ndarray:
        """[sum E] the default creation rank: every edge's CSR position inside its graph."""
        E = self.g_ne.astype(np.int64)
        return (np.arange(int(E.sum()), dtype=np.int64) - np.repeat(np.concatenate([[0], np.cumsum(E)[:-1]]), E)).astype(np.int32)

    def c_args(self):
        """Pointers in the argument order shared by ald_batch_add_packed / ora_run_packed."""
        def p(a, t):
            return a.ctypes.data_as(C.POINTER(t))
        return (
            C.c_int32(self.n),
            p(self.g_nv, C.c_int32), p(self.g_ne, C.c_int32), p(self.g_np, C.c_int32),
            p(self.vertex_offset, C.c_int32), p(self.edge_target, C.c_int32),
            p(self.edge_weight, C.c_double), p(self.edge_strand, C.c_uint8), p(self.edge_abd, C.c_double),
            p(self.edge_sample_offset, C.c_int32), p(self.sample_id, C.c_int32), p(self.sample_abd, C.c_double),
            p(self.vertex_weight, C.c_double), p(self.vertex_lpos, C.c_int32), p(self.vertex_rpos, C.c_int32),
            p(self.vertex_type, C.c_int32),
            p(self.phasing_offset, C.c_int32), p(self.phasing_vertex, C.c_int32), p(self.phasing_count, C.c_int32),
            p(self.graph_strand, C.c_char),
            p(np.ascontiguousarray(self.edge_count, np.int32), C.c_int32) if self.edge_count is not None else None,
            p(np.ascontiguousarray(self.edge_rank, np.int32), C.c_int32) if self.edge_rank is not None else None,
        )

    def graph_slices(self):
        """Per-graph start offsets into every concatenated array (host convenience)."""
        n = self.n
        V = self.g_nv.astype(np.int64); E = self.g_ne.astype(np.int64); P = self.g_np.astype(np.int64)
        ov = np.concatenate([[0], np.cumsum(V)]); ovo = np.concatenate([[0], np.cumsum(V + 1)])
        oe = np.concatenate([[0], np.cumsum(E)]); oeo = np.concatenate([[0], np.cumsum(E + 1)])
        op = np.concatenate([[0], np.cumsum(P)]); opo = np.concatenate([[0], np.cumsum(P + 1)])
        ns = self.edge_sample_offset[(oeo[1:] - 1)].astype(np.int64) if n else np.zeros(0, np.int64)
        os_ = np.concatenate([[0], np.cumsum(ns)])
        npv = self.phasing_offset[(opo[1:] - 1)].astype(np.int64) if n else np.zeros(0, np.int64)
        opv = np.concatenate([[0], np.cumsum(npv)])
        return dict(v=ov, vo=ovo, e=oe, eo=oeo, s=os_, p=op, po=opo, pv=opv)

    def select(self, idx) -> "PackedGraphs":
        """Sub-batch of the given graph indices (used to shard a batch across ranks)."""
        idx = np.asarray(idx, dtype=np.int64)
        o = self.graph_slices()

        def cat(arr, starts, lens):
            if len(idx) == 0:
                return arr[:0].copy()
            return np.concatenate([arr[starts[g]:starts[g] + lens[g]] for g in idx])
        V = self.g_nv.astype(np.int64); E = self.g_ne.astype(np.int64); P = self.g_np.astype(np.int64)
        ns = np.diff(o["s"]); npv = np.diff(o["pv"])
        return PackedGraphs(
            g_nv=self.g_nv[idx].copy(), g_ne=self.g_ne[idx].copy(), g_np=self.g_np[idx].copy(),
            vertex_offset=cat(self.vertex_offset, o["vo"], V + 1), edge_target=cat(self.edge_target, o["e"], E),
            edge_weight=cat(self.edge_weight, o["e"], E), edge_strand=cat(self.edge_strand, o["e"], E),
            edge_abd=cat(self.edge_abd, o["e"], E), edge_sample_offset=cat(self.edge_sample_offset, o["eo"], E + 1),
            sample_id=cat(self.sample_id, o["s"], ns), sample_abd=cat(self.sample_abd, o["s"], ns),
            vertex_weight=cat(self.vertex_weight, o["v"], V), vertex_lpos=cat(self.vertex_lpos, o["v"], V),
            vertex_rpos=cat(self.vertex_rpos, o["v"], V), vertex_type=cat(self.vertex_type, o["v"], V),
            phasing_offset=cat(self.phasing_offset, o["po"], P + 1), phasing_vertex=cat(self.phasing_vertex, o["pv"], npv),
            phasing_count=cat(self.phasing_count, o["p"], P), graph_strand=self.graph_strand[idx].copy(),
            edge_count=None if self.edge_count is None else cat(self.edge_count, o["e"], E),
            edge_rank=None if self.edge_rank is None else cat(self.edge_rank, o["e"], E),
        )

    @staticmethod
    def concat(parts) -> "PackedGraphs":
        kw = {f.name: np.concatenate([getattr(p, f.name) for p in parts]) for f in fields(PackedGraphs) if f.name not in ("edge_count", "edge_rank")}
        if any(p.edge_count is not None for p in parts):
            kw["edge_count"] = np.concatenate([p.edge_count if p.edge_count is not None else p.sample_counts() for p in parts]).astype(np.int32)
        if any(p.edge_rank is not None for p in parts):
            kw["edge_rank"] = np.concatenate([p.edge_rank if p.edge_rank is not None else p.identity_rank() for p in parts]).astype(np.int32)
        return PackedGraphs(**kw)

    @staticmethod
    def from_graphs(graphs) -> "PackedGraphs":
        """Build from a list of dicts: V, edges=[(s,t,w[,strand[,{sid:abd}]])], vw, lpos, rpos,
        optional vtype, phasing=[([v...], count)], strand.  Edges are sorted into CSR order."""
        acc = {f.name: [] for f in fields(PackedGraphs) if f.name not in ("edge_count", "edge_rank")}
        for g in graphs:
            V = int(g["V"]); edges = sorted(g["edges"], key=lambda e: (e[0], e[1]))
            voff = np.zeros(V + 1, _I32)
            for e in edges:
                voff[e[0] + 1] += 1
            voff = np.cumsum(voff).astype(_I32)
            esoff = [0]; sid = []; sabd = []; eabd = []
            for e in edges:
                sp = e[4] if len(e) > 4 and e[4] is not None else {0: float(e[2])}
                for k in sorted(sp):
                    sid.append(k); sabd.append(float(sp[k]))
                esoff.append(len(sid)); eabd.append(float(sum(sp.values())))
            ph = sorted(g.get("phasing", []), key=lambda p: list(p[0]))
            poff = [0]; pv = []
            for v, c in ph:
                pv.extend(v); poff.append(len(pv))
            acc["g_nv"].append(np.array([V], _I32)); acc["g_ne"].append(np.array([len(edges)], _I32))
            acc["g_np"].append(np.array([len(ph)], _I32)); acc["vertex_offset"].append(voff)
            acc["edge_target"].append(np.array([e[1] for e in edges], _I32))
            acc["edge_weight"].append(np.array([e[2] for e in edges], _F64))
            acc["edge_strand"].append(np.array([(e[3] if len(e) > 3 else 0) for e in edges], np.uint8))
            acc["edge_abd"].append(np.array(eabd, _F64)); acc["edge_sample_offset"].append(np.array(esoff, _I32))
            acc["sample_id"].append(np.array(sid, _I32)); acc["sample_abd"].append(np.array(sabd, _F64))
            acc["vertex_weight"].append(np.asarray(g["vw"], _F64)); acc["vertex_lpos"].append(np.asarray(g["lpos"], _I32))
            acc["vertex_rpos"].append(np.asarray(g["rpos"], _I32))
            acc["vertex_type"].append(np.asarray(g.get("vtype", [-1] * V), _I32))
            acc["phasing_offset"].append(np.array(poff, _I32)); acc["phasing_vertex"].append(np.array(pv, _I32))
            acc["phasing_count"].append(np.array([c for _, c in ph], _I32))
            acc["graph_strand"].append(np.frombuffer(g.get("strand", ".").encode(), np.int8).copy())
        kw = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in acc.items()}
        return PackedGraphs(**kw)


@dataclass
class DecompResult:
    """Bulk export of a decomposed batch (what ``scallop::paths`` holds, rnacore/path.h:14-35)."""
    status: np.ndarray        # [n] int32
    path_offset: np.ndarray   # [n+1] int32
    weight: np.ndarray
    abd: np.ndarray
    conf: np.ndarray
    reads: np.ndarray
    length: np.ndarray
    count: np.ndarray
    strand: np.ndarray        # int8
    pv_offset: np.ndarray     # [paths+1] int64
    path_vertices: np.ndarray  # int32

    def paths_of(self, g: int):
        out = []
        for p in range(int(self.path_offset[g]), int(self.path_offset[g + 1])):
            v = self.path_vertices[int(self.pv_offset[p]):int(self.pv_offset[p + 1])]
            out.append(dict(v=v.tolist(), weight=float(self.weight[p]), abd=float(self.abd[p]), conf=float(self.conf[p]),
                            reads=float(self.reads[p]), length=int(self.length[p]), count=int(self.count[p]),
                            strand=chr(int(self.strand[p]))))
        return out


def export_via(fn, handle, n: int, check=None) -> DecompResult:
    """Drive an ``ald_batch_export``-shaped C function (two-pass: sizes, then fill); ``check(rc)`` may raise the caller's own error."""
    tp = C.c_int64(0); tv = C.c_int64(0)
    nul = None
    rc = fn(handle, C.byref(tp), C.byref(tv), nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul)
    if rc != 0 and check is not None:
        check(rc)
    if rc != 0:
        raise RuntimeError(f"export(sizes) failed rc={rc}")
    P, TV = tp.value, tv.value
    r = DecompResult(
        status=np.zeros(n, _I32), path_offset=np.zeros(n + 1, _I32),
        weight=np.zeros(P, _F64), abd=np.zeros(P, _F64), conf=np.zeros(P, _F64), reads=np.zeros(P, _F64),
        length=np.zeros(P, _I32), count=np.zeros(P, _I32), strand=np.zeros(P, np.int8),
        pv_offset=np.zeros(P + 1, np.int64), path_vertices=np.zeros(TV, _I32))

    def p(a, t):
        return a.ctypes.data_as(C.POINTER(t))
    rc = fn(handle, C.byref(tp), C.byref(tv), p(r.status, C.c_int32), p(r.path_offset, C.c_int32),
            p(r.weight, C.c_double), p(r.abd, C.c_double), p(r.conf, C.c_double), p(r.reads, C.c_double),
            p(r.length, C.c_int32), p(r.count, C.c_int32), p(r.strand, C.c_char),
            p(r.pv_offset, C.c_int64), p(r.path_vertices, C.c_int32))
    if rc != 0:
        raise RuntimeError(f"export(fill) failed rc={rc}")
    return r
