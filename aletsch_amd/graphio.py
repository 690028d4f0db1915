"""Text formats of the reference for splice graphs and phasing sets <-> PackedGraphs (tooling / replay).

Two formats exist in the reference tree:

A. the *graph file* read by ``splice_graph::build(const string &file)`` (rnacore/splice_graph.cc:329-376):
       <n>                                   number of vertices
       <name> <weight> <length>              n lines, vertex i
       <x> <y> <weight> <length>             one line per edge x -> y, in creation order
   It carries no coordinates and no edge_info beyond ``length``: vertices get lpos = rpos = 0, edges count 0 and no supporting
   sample -- exactly what ``build`` leaves in the graph (the reference's merges assert ``count > 0`` on such edges, and so does the
   kernel: ALD_INV_COUNT).  ``read_graph_file(..., one_sample=True)`` attaches the one sample {0: weight} / count 1 instead, so
   that such a graph can be decomposed.

B. the *bundle dump* written by ``splice_graph::write(ostream&)`` (rnacore/splice_graph.cc:422-477) followed by
   ``hyper_set::write(ostream&)`` (scallop/hyper_set.cc:1109-1128):
       # <gid> <chrm> <strand>
       region <lpos> <rpos> <weight>         internal vertices with lpos < rpos, ascending
       sbound <lpos of target> <weight> 1    edges source -> v, in out_edges(0) order (target, creation)
       tbound <rpos of source> <weight> 1    edges v -> sink, in in_edges(n) order (source, creation)
       junction <rpos of s> <lpos of t> <weight> 1     every other edge with rpos(s) < lpos(t), in creation order
       path <n> <v0> ... <vn-1> <count> 1    phasing vertex lists with more than two vertices
   all numbers ``fixed`` with two decimals.  The dump is lossy by design: edges between touching regions are not written, nor is
   edge_info (samples, abundances, strand).  Nothing in the live reference reads it back; the dead meta-scallop loader
   (meta/combined_graph.cc:355-498) shows the intended reading, which ``read_bundle_dump`` follows: vertices = source + regions +
   sink, edges created in file order (sbounds, tbounds, junctions -- that order is their creation rank), then one edge between
   every pair of touching consecutive regions, weight = the weight of the region with fewer edges on the touching side
   (ties: the right one), at least ``min_guaranteed_edge_weight``.  Every edge gets count = the trailing field and the single
   sample {0: weight}.

``write_bundle_dump`` emits byte for byte what the reference's two writers would print for the same graph (same line order, same
number formatting), so a dump made here and a dump made by an Aletsch build elsewhere are interchangeable.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .packed import PackedGraphs


def _graph_dicts(pg: PackedGraphs, g: int):
    sl = pg.graph_slices()
    V, E, P = int(pg.g_nv[g]), int(pg.g_ne[g]), int(pg.g_np[g])
    ov, ovo, oe = int(sl["v"][g]), int(sl["vo"][g]), int(sl["e"][g])
    vo = pg.vertex_offset[ovo:ovo + V + 1]
    src = np.repeat(np.arange(V), np.diff(vo))
    tgt = pg.edge_target[oe:oe + E]
    w = pg.edge_weight[oe:oe + E]
    rank = pg.edge_rank[oe:oe + E] if pg.edge_rank is not None else np.arange(E)
    return V, E, P, ov, src, tgt, w, rank, sl


# ------------------------------------------------------------------------------------------------ format A
def write_graph_file(pg: PackedGraphs, g: int = 0) -> str:
    """Graph `g` in the format splice_graph::build(file) reads; edges in creation order."""
    V, E, P, ov, src, tgt, w, rank, _ = _graph_dicts(pg, g)
    out = ["%d" % V]
    for i in range(V):
        out.append("%d %r %d" % (i, float(pg.vertex_weight[ov + i]), int(pg.vertex_rpos[ov + i]) - int(pg.vertex_lpos[ov + i])))
    for k in np.argsort(rank, kind="stable"):
        out.append("%d %d %r %d" % (int(src[k]), int(tgt[k]), float(w[k]), 0))
    return "\n".join(out) + "\n"


def read_graph_file(text: str, one_sample: bool = False) -> PackedGraphs:
    """splice_graph::build(file): one graph.  Vertex intervals are [0, 0) (the file has no coordinates); see the module docstring
    for `one_sample`."""
    lines = text.splitlines()
    n = int(lines[0].split()[0])                                    # atoi(line)
    vw = []
    for i in range(n):
        f = lines[1 + i].split()
        vw.append(float(f[1]))
    edges = []
    for ln in lines[1 + n:]:
        f = ln.split()
        if len(f) < 3:
            continue
        x, y, wt = int(f[0]), int(f[1]), float(f[2])
        if not (x != y and 0 <= x < n and 0 <= y < n):
            raise ValueError("edge %d -> %d out of range (the reference asserts here)" % (x, y))
        edges.append((x, y, wt, 0, ({0: wt} if one_sample else {})))
    pg = PackedGraphs.from_graphs([dict(V=n, edges=edges, vw=vw, lpos=[0] * n, rpos=[0] * n)])
    _set_rank_from_listing(pg, [edges])
    if not one_sample:
        pg.edge_count = np.zeros(len(edges), np.int32)
    return pg


def _set_rank_from_listing(pg: PackedGraphs, per_graph_edges):
    """from_graphs sorts edges into CSR order (stable by (source, target)); the creation rank is the position in the listing."""
    ranks = []
    for edges in per_graph_edges:
        order = sorted(range(len(edges)), key=lambda k: (edges[k][0], edges[k][1]))        # stable: what from_graphs did
        ranks.append(np.array(order, np.int32))
    pg.edge_rank = np.concatenate(ranks) if ranks else np.zeros(0, np.int32)


# ------------------------------------------------------------------------------------------------ format B
def write_bundle_dump(pg: PackedGraphs, gids: Optional[List[str]] = None, chrm: str = "1") -> str:
    """splice_graph::write(os) + hyper_set::write(os) for every graph of the batch, concatenated."""
    out = []
    sl = pg.graph_slices()
    for g in range(pg.n):
        V, E, P, ov, src, tgt, w, rank, _ = _graph_dicts(pg, g)
        n = V - 1
        lp = pg.vertex_lpos[ov:ov + V]; rp = pg.vertex_rpos[ov:ov + V]
        out.append("# %s %s %s" % (gids[g] if gids else "gene.%d" % g, chrm, chr(int(pg.graph_strand[g]))))
        for i in range(1, n):
            if lp[i] >= rp[i]:
                continue
            out.append("region %d %d %.2f" % (lp[i], rp[i], pg.vertex_weight[ov + i]))
        ks = [k for k in range(E) if src[k] == 0 and tgt[k] != n]
        for k in sorted(ks, key=lambda k: (tgt[k], rank[k])):                      # out_edges(0): by (target, creation)
            out.append("sbound %d %.2f 1" % (lp[tgt[k]], w[k]))
        ks = [k for k in range(E) if tgt[k] == n and src[k] != 0]
        for k in sorted(ks, key=lambda k: (src[k], rank[k])):                      # in_edges(n): by (source, creation)
            out.append("tbound %d %.2f 1" % (rp[src[k]], w[k]))
        for k in np.argsort(rank, kind="stable"):                                  # edges(): creation order
            s, t = int(src[k]), int(tgt[k])
            if s == 0 or t == n:
                continue
            p1, p2 = int(rp[s]), int(lp[t])
            if p1 >= p2:
                continue
            out.append("junction %d %d %.2f 1" % (p1, p2, w[k]))
        opo, opv, op = int(sl["po"][g]), int(sl["pv"][g]), int(sl["p"][g])
        po = pg.phasing_offset[opo:opo + P + 1]
        nodes = {}                                                                 # hyper_set::nodes is a std::map<vector<int>, int>: add_node_list
        for p in range(P):                                                         # sorts every list and adds the counts of equal ones (hyper_set.cc:40-48)
            v = tuple(sorted(int(x) for x in pg.phasing_vertex[opv + po[p]:opv + po[p + 1]]))
            nodes[v] = nodes.get(v, 0) + int(pg.phasing_count[op + p])
        for v in sorted(nodes):                                                    # ... and hyper_set::write walks it in key order: lexicographic
            if len(v) <= 2:
                continue
            out.append("path %d %s %d 1" % (len(v), " ".join(str(x) for x in v), nodes[v]))
    return "\n".join(out) + "\n"


def read_bundle_dump(text: str, min_guaranteed_edge_weight: float = 0.01) -> Tuple[PackedGraphs, List[dict]]:
    """-> (PackedGraphs, [dict(gid, chrm, strand)]): every '# gid chrm strand' block of the dump as one graph (module docstring)."""
    blocks = []; cur = None
    for ln in text.splitlines():
        f = ln.split()
        if not f:
            continue
        if f[0] == "#":
            cur = dict(gid=f[1] if len(f) > 1 else "", chrm=f[2] if len(f) > 2 else "", strand=f[3] if len(f) > 3 else ".",
                       regions=[], sb=[], tb=[], jn=[], paths=[])
            blocks.append(cur)
        elif cur is None:
            raise ValueError("dump does not start with a '# gid chrm strand' line")
        elif f[0] == "region":
            cur["regions"].append((int(f[1]), int(f[2]), float(f[3])))
        elif f[0] == "sbound":
            cur["sb"].append((int(f[1]), float(f[2]), int(f[3]) if len(f) > 3 else 1))
        elif f[0] == "tbound":
            cur["tb"].append((int(f[1]), float(f[2]), int(f[3]) if len(f) > 3 else 1))
        elif f[0] == "junction":
            cur["jn"].append((int(f[1]), int(f[2]), float(f[3]), int(f[4]) if len(f) > 4 else 1))
        elif f[0] == "path":
            n = int(f[1]); cur["paths"].append(([int(x) for x in f[2:2 + n]], int(f[2 + n])))
    graphs = []; listings = []; counts = []; meta = []
    for b in blocks:
        R = b["regions"]; nr = len(R); V = nr + 2
        for i in range(nr):
            if not R[i][0] < R[i][1] or (i and R[i - 1][1] > R[i][0]):
                raise ValueError("regions of %s are not ascending, disjoint, non-empty intervals" % b["gid"])
        lindex = {R[i][0]: i + 1 for i in range(nr)}; rindex = {R[i][1]: i + 1 for i in range(nr)}
        left = min([p for p, _, _ in b["sb"]] + [R[0][0]] if nr else [0]); right = max([p for p, _, _ in b["tb"]] + [R[-1][1]] if nr else [0])
        edges = []; cnt = []
        for p, wt, c in b["sb"]:
            edges.append((0, lindex[p], wt)); cnt.append(c)
        for p, wt, c in b["tb"]:
            edges.append((rindex[p], V - 1, wt)); cnt.append(c)
        for p1, p2, wt, c in b["jn"]:
            if p1 not in rindex or p2 not in lindex:
                continue                                                           # combined_graph.cc:452-453
            edges.append((rindex[p1], lindex[p2], wt)); cnt.append(c)
        outd = [0] * V; ind = [0] * V
        for s, t, _ in edges:
            outd[s] += 1; ind[t] += 1
        for i in range(1, nr):                                                     # connect touching regions (combined_graph.cc:467-497)
            if R[i - 1][1] != R[i][0]:
                continue
            wt = R[i - 1][2] if outd[i] < ind[i + 1] else R[i][2]
            wt = max(wt, min_guaranteed_edge_weight)
            edges.append((i, i + 1, wt)); cnt.append(1); outd[i] += 1; ind[i + 1] += 1
        graphs.append(dict(V=V, edges=[(s, t, wt, 0, {0: wt}) for s, t, wt in edges], vw=[0.0] + [r[2] for r in R] + [0.0],
                           lpos=[left] + [r[0] for r in R] + [right], rpos=[left] + [r[1] for r in R] + [right],
                           phasing=b["paths"], strand=b["strand"] or "."))
        listings.append(edges); counts.append(cnt)
        meta.append(dict(gid=b["gid"], chrm=b["chrm"], strand=b["strand"]))
    pg = PackedGraphs.from_graphs(graphs)
    _set_rank_from_listing(pg, listings)
    # edge_count in CSR order
    ec = []
    for edges, cnt in zip(listings, counts):
        order = sorted(range(len(edges)), key=lambda k: (edges[k][0], edges[k][1]))
        ec.append(np.array([cnt[k] for k in order], np.int32))
    pg.edge_count = np.concatenate(ec) if ec else np.zeros(0, np.int32)
    return pg, meta
