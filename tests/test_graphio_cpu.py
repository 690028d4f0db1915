"""CPU tier: the reference's text formats for splice graphs / phasing sets (aletsch_amd/graphio.py) -- the graph
file of splice_graph::build(file) (splice_graph.cc:329-376) and the bundle dump of splice_graph::write + hyper_set::write
(splice_graph.cc:422-477, hyper_set.cc:1109-1128).  Replayed graphs go through the engine (emulated here, the HIP kernels in the GPU
tier) and the oracle like any other batch."""
import numpy as np

import aletsch_amd as A
from aletsch_amd import graphio
import common


def test_graph_file_round_trip_and_replay():
    pg = A.synth(seed=81, n_graphs=6, v_min=8, v_max=30, edges_per_vertex=3, weight_mode=0)
    for g in range(pg.n):
        one = pg.select(np.array([g]))
        text = graphio.write_graph_file(one, 0)
        back = graphio.read_graph_file(text, one_sample=True)
        assert np.array_equal(back.vertex_offset, one.vertex_offset) and np.array_equal(back.edge_target, one.edge_target)
        assert np.array_equal(back.edge_weight, one.edge_weight) and np.array_equal(back.vertex_weight, one.vertex_weight)       # %r round-trips doubles
        assert np.array_equal(back.edge_rank, one.identity_rank())
        text2 = graphio.write_graph_file(back, 0)              # (the length column is vertex_info.length, which only the file carries)
        assert graphio.write_graph_file(graphio.read_graph_file(text2, one_sample=True), 0) == text2
        want = common.oracle_run(back)[0]; got = common.emu_run(back)[0]
        assert not common.compare_results(want, got, 1)
    # as build() leaves it: no supporting samples, count 0 -> the reference's merges assert (count > 0), and so does the engine
    bare = graphio.read_graph_file(graphio.write_graph_file(pg.select(np.array([0])), 0))
    assert (bare.edge_count == 0).all() and bare.sample_id.size == 0
    want = common.oracle_run(bare)[0]; got = common.emu_run(bare)[0]
    assert want.status[0] == got.status[0] == 100 + 3                                       # ALD_ST_INVARIANT + ALD_INV_COUNT


def test_bundle_dump_round_trip_and_replay():
    # integer weights survive the two-decimal format; layout_mode=0 keeps every edge a junction, so nothing is lost
    pg = A.synth(seed=82, n_graphs=25, v_min=6, v_max=60, edges_per_vertex=3, weight_mode=1, phasing_per_graph=8, strand_mode=1)
    pg.edge_strand[:] = 0; pg.vertex_weight[:] = np.round(pg.vertex_weight)                # the dump carries no edge strands
    text = graphio.write_bundle_dump(pg, gids=["gene.7.%d.0" % g for g in range(pg.n)], chrm="12")
    back, meta = graphio.read_bundle_dump(text)
    assert [m["gid"] for m in meta] == ["gene.7.%d.0" % g for g in range(pg.n)] and all(m["chrm"] == "12" for m in meta)
    assert np.array_equal(back.g_nv, pg.g_nv) and np.array_equal(back.g_ne, pg.g_ne)
    assert np.array_equal(back.vertex_offset, pg.vertex_offset) and np.array_equal(back.edge_target, pg.edge_target) and np.array_equal(back.edge_weight, pg.edge_weight)
    assert np.array_equal(back.vertex_lpos[1:-1], pg.vertex_lpos[1:-1]) and np.array_equal(back.graph_strand, pg.graph_strand)
    assert graphio.write_bundle_dump(back, gids=[m["gid"] for m in meta], chrm="12") == text          # a second trip changes nothing
    # creation order of a replayed graph is the order of the dump's lines: sbounds, tbounds, junctions
    assert back.edge_rank is not None and not np.array_equal(back.edge_rank, back.identity_rank())
    want, st, _, _ = common.oracle_run(back); got, it, _ = common.emu_run(back)
    assert not common.compare_results(want, got, back.n) and np.array_equal(it, st[:, 3])
    assert (want.status == 0).all() and len(want.weight) > 50
    # phasing lists of up to two vertices are not written (hyper_set.cc:1119)
    assert back.g_np.sum() <= pg.g_np.sum() and back.g_np.sum() > 0


HAND_DUMP = """# gene.3.1.0 7 +
region 1000 1200 12.00
region 1200 1350 9.50
region 2000 2100 14.00
region 2100 2300 6.00
region 3000 3400 11.00
sbound 1000 12.00 1
sbound 2000 3.00 1
tbound 2300 5.00 1
tbound 3400 11.00 1
junction 1200 2000 4.00 1
junction 1350 2000 8.00 1
junction 1350 3000 1.50 1
junction 2100 3000 7.00 1
junction 2300 3000 2.50 1
path 3 1 2 3 4 1
path 4 1 2 3 5 2 1
# gene.3.2.0 7 -
region 500 600 2.00
sbound 500 2.00 1
tbound 600 2.00 1
"""


def test_a_dump_as_an_aletsch_build_prints_it():
    pg, meta = graphio.read_bundle_dump(HAND_DUMP)
    assert [m["strand"] for m in meta] == ["+", "-"] and list(pg.g_nv) == [7, 3]
    # graph 0: 9 written edges + the two touching pairs (1200: regions 1|2, 2100: regions 3|4), which the dump leaves implicit
    assert int(pg.g_ne[0]) == 11
    edges = {}
    vo = pg.vertex_offset[:8]
    for s in range(7):
        for k in range(vo[s], vo[s + 1]):
            edges[(s, int(pg.edge_target[k]))] = float(pg.edge_weight[k])
    # region 1 has 2 out-edges (junction 1200->2000 and the new one is not counted yet: 1) vs region 2 with 0 in-edges: the right
    # region's weight is taken when the left side has no fewer edges (combined_graph.cc:487-489)
    assert edges[(1, 2)] == 9.5 and edges[(3, 4)] == 6.0
    assert edges[(0, 1)] == 12.0 and edges[(2, 5)] == 1.5 and edges[(4, 6)] == 5.0
    assert graphio.write_bundle_dump(pg, gids=[m["gid"] for m in meta], chrm="7") == HAND_DUMP
    want, st, _, _ = common.oracle_run(pg); got, it, _ = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n) and np.array_equal(it, st[:, 3])
    assert (want.status == 0).all() and want.path_offset[1] >= 2 and want.path_offset[2] - want.path_offset[1] == 1
