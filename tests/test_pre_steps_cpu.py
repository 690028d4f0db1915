"""CPU tier: the pre-steps of assembler::assemble(gx, px, sid) (meta/assembler.cc:1075-1086) behind the ABI -- extend_strands,
group_start/end_boundaries, project_boundaries, hyper_set ctor, filter_nodes (aletsch_amd/csrc/pre_steps.cpp: host code on flat
arrays) -- against the oracle's container-based restatement (unpinned: their reference files need config.h / Boost / htslib to
build), array by array, then through the engine."""
import ctypes as C

import numpy as np

import aletsch_amd as A
from aletsch_amd.packed import PackedGraphs
import common


def both(pg, phases, dist=10000):
    O = common.oracle_lib()
    O.ora_pre_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    O.ora_staged_view.argtypes = [C.c_void_p, C.c_void_p]; O.ora_staged_free.argtypes = [C.c_void_p]
    O.ora_staged_boundary_maps.argtypes = [C.c_void_p] * 5
    a = A.pre_assemble(pg, phases, dist)
    b = A.pre_assemble(pg, phases, dist, _lib=O, _prefix="ora")
    return a, b


def same_graph(x: PackedGraphs, y: PackedGraphs):
    for f in ("g_nv", "g_ne", "g_np", "vertex_offset", "edge_target", "edge_weight", "edge_strand", "edge_abd", "edge_sample_offset", "sample_id", "sample_abd",
              "vertex_weight", "vertex_lpos", "vertex_rpos", "vertex_type", "phasing_offset", "phasing_vertex", "phasing_count", "graph_strand", "edge_count", "edge_rank"):
        a, b = getattr(x, f), getattr(y, f)
        assert np.array_equal(a, b), (f, a, b)


def test_pre_steps_match_the_oracle_and_feed_the_engine():
    rng = np.random.default_rng(1075)
    n_grouped = n_strand = n_phase_kept = n_phase_dropped = n_assert = 0
    staged = []
    for t in range(150):
        g, phases = common.gene_like_raw(rng, n_runs=int(rng.integers(3, 9)), strand="+-."[t % 3])
        pg = PackedGraphs.from_graphs([g])
        # from_graphs lays the edges out as CSR; the caller's creation order is the order of the listing
        order = sorted(range(len(g["edges"])), key=lambda k: (g["edges"][k][0], g["edges"][k][1]))
        pg.edge_rank = np.array(order, np.int32)
        pg.edge_count = (pg.sample_counts() + rng.integers(0, 3, pg.edge_target.size)).astype(np.int32)
        dist = int(rng.choice([10000, 10000, 150, 0]))
        (mine, sm, tm, rc), (want, sm_o, tm_o, rc_o) = both(pg, phases, dist)
        assert rc == rc_o, (t, rc, rc_o)
        if rc:                                           # a phase whose exons run backwards: the reference asserts (essential.cc:364), both say so
            n_assert += 1; continue
        same_graph(want, mine)
        assert sm == sm_o and tm == tm_o
        n_grouped += len(sm) + len(tm)
        n_phase_kept += int(mine.g_np[0]); n_phase_dropped += len(phases) - int(mine.g_np[0])
        assert int(mine.g_ne[0]) == int(pg.g_ne[0]) - len(sm) - len(tm)            # every grouped boundary loses its source / sink edge
        staged.append(mine)
    assert n_grouped > 50 and n_phase_kept > 100 and n_phase_dropped > 20 and n_assert < 30, (n_grouped, n_phase_kept, n_phase_dropped, n_assert)
    # what comes out is an ordinary batch: the engine and the oracle decompose it alike
    batch = PackedGraphs.concat(staged)
    res_o, st, _, _ = common.oracle_run(batch); res_e, it, _ = common.emu_run(batch)
    assert not common.compare_results(res_o, res_e, batch.n)
    ok = res_o.status == 0
    assert np.array_equal(it[ok], st[ok, 3]) and ok.sum() > 90


def test_pre_steps_in_the_load_phase_of_the_engine():
    """SURVEY 8f row f1 where it belongs: the graph goes to the device AS assemble(gx, px, sid) receives it, its phase set in exon
    coordinates, and the wave that loads it runs extend_strands / boundary grouping / phase projection / hyper_set ctor / filter_nodes
    (decomp_device.h: pre_assemble_device; here through the single-lane emulation).  The decomposition must equal the oracle's
    pre-steps + decomposition graph for graph -- and a graph on which the reference would have asserted ends with an invariant status."""
    rng = np.random.default_rng(1077)
    items = []; staged = []; asserted = []
    for t in range(260):
        g, phases = common.gene_like_raw(rng, n_runs=int(rng.integers(3, 10)), strand="+-."[t % 3])
        if t % 7 == 0:
            phases = phases + phases[:2]                                            # duplicate phases: equal lists fold their counts
        pg = PackedGraphs.from_graphs([g])
        pg.edge_rank = np.array(sorted(range(len(g["edges"])), key=lambda k: (g["edges"][k][0], g["edges"][k][1])), np.int32)
        pg.edge_count = (pg.sample_counts() + rng.integers(0, 3, pg.edge_target.size)).astype(np.int32)
        dist = int(rng.choice([10000, 10000, 150, 0]))
        O = common.oracle_lib()
        O.ora_pre_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        O.ora_staged_view.argtypes = [C.c_void_p, C.c_void_p]; O.ora_staged_free.argtypes = [C.c_void_p]; O.ora_staged_boundary_maps.argtypes = [C.c_void_p] * 5
        want, sm, tm, rc_o = A.pre_assemble(pg, phases, dist, _lib=O, _prefix="ora")
        items.append((pg, phases, dist)); asserted.append(rc_o != 0)
        if rc_o == 0:
            staged.append(want)
    got, it = common.emu_run_raw(items)
    asserted = np.array(asserted)
    assert (got.status[asserted] >= 100).all() and 0 < asserted.sum() < 60
    batch = PackedGraphs.concat(staged)
    want, st, _, _ = common.oracle_run(batch)
    ok = ~asserted
    # compare graph by graph: the emulated batch holds the asserted graphs too
    sub = common_select_results(got, np.nonzero(ok)[0])
    assert not common.compare_results(want, sub, batch.n)
    good = want.status == 0
    assert np.array_equal(it[ok][good], st[good, 3]) and good.sum() > 150


def common_select_results(r, keep):
    """the results of graphs `keep` (ascending) as a DecompResult of their own"""
    from aletsch_amd.packed import DecompResult
    po = r.path_offset; pv = r.pv_offset
    paths = np.concatenate([np.arange(po[g], po[g + 1]) for g in keep]) if len(keep) else np.zeros(0, np.int64)
    cnt = np.array([po[g + 1] - po[g] for g in keep], np.int64)
    new_po = np.concatenate([[0], np.cumsum(cnt)]).astype(r.path_offset.dtype)
    lens = (pv[paths + 1] - pv[paths]) if len(paths) else np.zeros(0, np.int64)
    new_pv = np.concatenate([[0], np.cumsum(lens)]).astype(r.pv_offset.dtype)
    verts = np.concatenate([r.path_vertices[pv[p]:pv[p + 1]] for p in paths]) if len(paths) else np.zeros(0, r.path_vertices.dtype)
    return DecompResult(status=r.status[keep], path_offset=new_po, weight=r.weight[paths], abd=r.abd[paths], conf=r.conf[paths], reads=r.reads[paths],
                        length=r.length[paths], count=r.count[paths], strand=r.strand[paths], pv_offset=new_pv, path_vertices=verts)


def test_pre_steps_by_hand():
    """one small case worked out by hand: three start boundaries on a run of touching partial exons, the third too far away"""
    #            0      1          2          3          4          5      6
    lpos = [100, 100, 200, 300, 5000, 9000, 9500]; rpos = [100, 200, 300, 400, 5200, 9200, 9500]
    e = [(0, 1, 10.0), (0, 2, 4.0), (0, 4, 3.0), (1, 2, 9.0), (2, 3, 12.0), (3, 4, 11.0), (3, 5, 2.0), (4, 5, 13.0), (5, 6, 15.0), (4, 6, 1.0)]
    g = dict(V=7, edges=[(s, t, w, 1, {0: w, 3: 1.0}) for s, t, w in e], vw=[0, 10, 14, 13, 14, 15, 0], lpos=lpos, rpos=rpos, strand="+")
    pg = PackedGraphs.from_graphs([g])
    phases = [([200, 400, 5000, 5200], 4), ([100, 400], 2), ([200, 300], 5), ([200, 400, 5000, 5200], 1), ([5000, 5200, 9000, 9200], 3), ([300, 401], 9),
              ([100, 200, 9000, 9200], 7)]          # vertices [1, 5]: no edge 1 -> 5, so filter_nodes / check_valid_path (essential.cc:448-459) drops it
    out, sm, tm, rc = A.pre_assemble(pg, phases, 500)
    assert rc == 0 and sm == [(200, 100)] and tm == []          # 0->2 folds into 0->1 (touching, 100 apart); 0->4 is 4800 away; 4->6 / 5->6: 4,5 do not touch
    W = {}
    vo = out.vertex_offset
    for s in range(7):
        for k in range(vo[s], vo[s + 1]):
            W[(s, int(out.edge_target[k]))] = (float(out.edge_weight[k]), int(out.edge_count[k]))
    assert (0, 2) not in W and W[(0, 1)] == (14.0, 4) and W[(1, 2)] == (13.0, 4)      # 10 + 4; count 2 + 2; the edge on the way 9 + 4
    assert list(out.vertex_weight) == [0, 14, 14, 13, 14, 15, 0]                       # vertex 1 (on the way) + 4
    # phases: (200,400,..) starts on the folded boundary -> 100; merges with nothing else; ([100,400]) stays; ([200,300]) -> (100,300);
    # ([300,401]) has an unknown right coordinate -> dropped; ([100,200,9000,9200]) is not a path of the graph -> dropped; vertex lists: [1,2,3,4] x (4+1), [1,2,3] x 2, [1,2] x 5, [4,5] x 3
    got = {tuple(out.phasing_vertex[out.phasing_offset[p]:out.phasing_offset[p + 1]]): int(out.phasing_count[p]) for p in range(int(out.g_np[0]))}
    assert got == {(1, 2, 3, 4): 5, (1, 2, 3): 2, (1, 2): 5, (4, 5): 3}
