"""CPU tier: the reference's text formats for splice graphs / phasing sets (aletsch_amd/graphio.py) -- the graph
file of splice_graph::build(file) (splice_graph.cc:329-376) and the bundle dump of splice_graph::write + hyper_set::write
(splice_graph.cc:422-477, hyper_set.cc:1109-1128).  Replayed graphs go through the engine (emulated here, the HIP kernels in the GPU
tier) and the oracle like any other batch."""
import os

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


# ------------------------------------------------------------------------------------------------ the C++ reader / writer (aletsch_amd/host/graph_io.hpp)
def _replay_tool():
    import subprocess
    exe = os.path.join(common.ROOT, "tests", "_build", "replay_dump"); os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(common.ROOT, "aletsch_amd", "lib")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(common.ROOT, "include"), os.path.join(common.ROOT, "tools", "replay_dump.cc"),
                    "-o", exe, "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib, "-pthread"], check=True)
    return exe


def test_cxx_reader_and_writer_round_trip_a_dump():
    """aletsch::read_bundle_dump + write_bundle_dump (C++11, beside gpu_scallop.hpp): the hand-typed dump and a large synthetic one come
    back byte for byte, and equal what the Python twin writes"""
    import subprocess
    exe = _replay_tool()
    for text in (HAND_DUMP, graphio.write_bundle_dump(A.synth(seed=5150, n_graphs=60, v_min=6, v_max=70, edges_per_vertex=3, layout_mode=1, weight_mode=1, phasing_per_graph=6, strand_mode=1), chrm="3")):
        back = subprocess.run([exe, "--echo"], input=text, capture_output=True, text=True, check=True).stdout
        pg, meta = graphio.read_bundle_dump(text)
        assert back == graphio.write_bundle_dump(pg, gids=[m["gid"] for m in meta], chrm=meta[0]["chrm"])
        if text is HAND_DUMP:
            assert back == text


def _listing(gid, chrm, strand, V, lpos, rpos, vw, edges, paths=()):
    out = ["%s %s %s %d %d %d" % (gid, chrm, strand, V, len(edges), len(paths))]
    out += ["%r %d %d" % (float(vw[i]), lpos[i], rpos[i]) for i in range(V)]
    out += ["%d %d %r" % (s, t, float(w)) for s, t, w in edges]
    out += ["%d %d %s" % (len(v), c, " ".join(map(str, v))) for v, c in paths]
    return "\n".join(out) + "\n"


def test_dump_line_order_against_the_reference_containers():
    """The ORDER of a dump's lines is the iteration order of the reference's containers: sbound = out_edges(0), tbound = in_edges(n),
    junction = edges() (rnacore/splice_graph.cc:440-476).  tests/golden/ref_graph.json holds those orders as printed by the reference's own
    graph/*.cc (oracle/_ref/ref_graph) for graphs built by scripts of add / remove / move in a known creation order -- parallel edges
    included.  Both writers (C++ and Python) must list the edges in exactly those orders; every edge carries its handle as its weight."""
    import json
    import subprocess
    exe = _replay_tool()
    cases = json.load(open(os.path.join(common.ROOT, "tests", "golden", "ref_graph.json")))
    checked = 0
    for c in cases:
        lines = c["script"].split("\n")
        if not lines[0].startswith("D"):
            continue
        n_v = int(lines[0].split()[1]); edges = {}; h = 0
        for ln in lines[1:]:                                     # replay the script: final (source, target) of every live handle
            f = ln.split()
            if not f:
                continue
            if f[0] == "a":
                edges[h] = (int(f[1]), int(f[2])); h += 1
            elif f[0] == "r":
                edges.pop(int(f[1]), None)
            elif f[0] == "m":
                edges[int(f[1])] = (int(f[2]), int(f[3]))
            elif f[0] == "c":
                for k in [k for k, (s, t) in edges.items() if int(f[1]) in (s, t)]:
                    edges.pop(k)
        last = c["dump"].strip().split("end\n")[-1] if c["dump"].strip().endswith("end") else c["dump"]
        blocks = [b for b in c["dump"].split("end\n") if b.strip()]
        d = {ln.split(":")[0]: [int(x) for x in ln.split(":")[1].split()] for ln in blocks[-1].splitlines() if ":" in ln and not ln.startswith("edge ")}
        if any(not (s < t) for s, t in edges.values()) or not edges:
            continue
        n = n_v - 1
        live = d["edges"]                                         # creation order of the live handles, as the reference iterates them
        assert sorted(live) == sorted(edges)
        lpos = [0] + [1000 * i for i in range(1, n)] + [1000 * n]; rpos = [0] + [1000 * i + 200 for i in range(1, n)] + [1000 * n]
        listing = "1\n" + _listing("g", "1", "+", n_v, lpos, rpos, [0] + [5] * (n - 1) + [0], [(edges[k][0], edges[k][1], k + 0.25) for k in live])
        text = subprocess.run([exe, "--from-listing"], input=listing, capture_output=True, text=True, check=True).stdout
        hid = lambda ln: int(float(ln.split()[-2]) - 0.25 + 0.5)
        sb = [hid(ln) for ln in text.splitlines() if ln.startswith("sbound")]
        tb = [hid(ln) for ln in text.splitlines() if ln.startswith("tbound")]
        jn = [hid(ln) for ln in text.splitlines() if ln.startswith("junction")]
        assert sb == [k for k in d["out 0"] if edges[k][1] != n], (c["script"], sb, d["out 0"])
        assert tb == [k for k in d["in %d" % n] if edges[k][0] != 0]
        assert jn == [k for k in d["edges"] if edges[k][0] != 0 and edges[k][1] != n]
        # the Python twin prints the same text for the same graph
        g = dict(V=n_v, edges=[(edges[k][0], edges[k][1], k + 0.25, 0, {0: 1.0}) for k in live], vw=[0] + [5] * (n - 1) + [0], lpos=lpos, rpos=rpos, strand="+")
        from aletsch_amd.packed import PackedGraphs
        pg = PackedGraphs.from_graphs([g]); graphio._set_rank_from_listing(pg, [g["edges"]])
        assert graphio.write_bundle_dump(pg, gids=["g"], chrm="1") == text
        checked += 1
    assert checked >= 8
