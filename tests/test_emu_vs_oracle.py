"""CPU tier: the array-based engine that the HIP kernels are compiled from (aletsch_amd/csrc/decomp_device.h), run as a
single-lane emulation, against the container-based oracle -- two independent statements of the reference algorithm.
Bit-exact on every field.  (The GPU tier repeats this through the C ABI on a real MI355X.)"""
import numpy as np
import pytest

import aletsch_amd as A
import common


@pytest.mark.parametrize("name", list(common.PARITY_CONFIGS))
def test_engine_matches_oracle(name):
    pg = common.make_batch(name)                 # the full configuration, exactly what the GPU tier runs
    want, st, _, _ = common.oracle_run(pg)
    got, it, cl = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])


@pytest.mark.parametrize("name", ["cfg2_64v256e", "stranded", "everything", "real_shaped", "minimal_3v"])
def test_engine_records_carry_the_transcripts(name):
    """The engine joins the exons of every path into its record (build_transcript, essential.cc:719-748: touching intervals of
    consecutive vertices fuse, empty ones vanish); coverage = log(1 + weight) is taken by the host decode.  Both must equal the oracle's
    transcripts exactly -- no host-side join exists any more."""
    pg = common.make_batch(name)
    _, wcov, weo, wlr = common.oracle_transcripts(pg)
    _, gcov, geo, glr = common.emu_transcripts(pg)
    assert np.array_equal(weo, geo) and np.array_equal(wlr, glr)
    assert np.array_equal(wcov, gcov)


@pytest.mark.parametrize("name", ["cfg1_32v96e", "cfg2_64v256e", "cfg3_mixed", "flow_weights", "multi_sample", "stranded", "phasing", "everything", "real_shaped"])
def test_row_form_of_the_engine_matches_oracle(name):
    """The A/B build of round 4 (aletsch_amd/csrc/decomp_device_rows.h, make ROWS=1): per-vertex adjacency ROWS in a segment pool instead
    of linked lists.  Same results bit for bit, same iteration counts; its emulation is built with the row checker (rows sorted by
    (endpoint, creation id), every live edge in exactly its two rows, segments disjoint -- verified after every rule firing; a
    violation aborts the process)."""
    pg = common.make_batch(name)
    want, st, _, _ = common.oracle_run(pg)
    got, it, cl = common.emu_run(pg, rows=True)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])


@pytest.mark.parametrize("name", ["cfg2_64v256e", "cfg3_mixed", "flow_weights", "multi_sample", "stranded", "phasing", "everything", "real_shaped"])
def test_kept_sweep_records_are_never_stale(name):
    """Round 4: the slab-resident classes keep the per-vertex records of the two evaluation sweeps (smallest-edge evaluation; trivial class +
    balance ratio) between sweeps and bring only MARKED vertices up to date -- the ones a rule edited and, when a degree crossed the 1 | 2
    line, the vertices at the far end of that list.  This build keeps them in EVERY class and checks, at every sweep, every record that is
    not marked against a fresh evaluation (a stale one aborts); results and iteration counts must equal the oracle's as ever."""
    pg = common.make_batch(name)
    want, st, _, _ = common.oracle_run(pg)
    got, it, cl = common.emu_run(pg, keep=True)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])


def test_engine_joins_touching_and_drops_empty_intervals():
    """hand-made: vertices 1|2 touch (one exon), 3 is an empty interval (vanishes), 4 stands alone"""
    from aletsch_amd.packed import PackedGraphs
    g = dict(V=6, edges=[(0, 1, 9.0), (1, 2, 9.0), (2, 3, 9.0), (3, 4, 9.0), (4, 5, 9.0)], vw=[0, 10, 10, 0, 10, 0],
             lpos=[0, 100, 200, 400, 500, 600], rpos=[0, 200, 300, 400, 600, 600])
    pg = PackedGraphs.from_graphs([g])
    r, cov, eo, lr = common.emu_transcripts(pg)
    assert len(cov) == 1 and lr.tolist() == [[100, 300], [500, 600]]
    _, wcov, weo, wlr = common.oracle_transcripts(pg)
    assert np.array_equal(wlr, lr) and np.array_equal(wcov, cov)


def test_engine_matches_oracle_at_scale():
    """10 000 graphs of the bench shape (64v / 256e): rare interleavings of the cascade (a removal that flips a degree guard of a
    vertex evaluated earlier, a fan above the fast-path limit, ...) need thousands of graphs to occur at all"""
    pg = A.synth(seed=1002, n_graphs=10000, v_min=64, v_max=64, fixed_edges=256)
    want, st, _, _ = common.oracle_run(pg, threads=4)
    got, it, cl = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])


def test_engine_op_trace_matches_oracle():
    """rule id / vertex or edge id / ratio of every firing, in order -- the sharpest parity check available"""
    pg = A.synth(**dict(common.PARITY_CONFIGS["everything"], n_graphs=25))
    _, _, _, traces = common.oracle_run(pg, trace=True)
    import ctypes as C
    E = common.emu_lib(); h = C.c_void_p()
    assert E.emu_run_packed(*pg.c_args(), None, C.c_int32(4096), C.c_int32(0), C.byref(h)) == 0
    for g in range(pg.n):
        n = C.c_int32(); E.emu_result_trace(h, g, C.byref(n), None, None, 0)
        codes = np.zeros(3 * max(1, n.value), np.int32); vals = np.zeros(max(1, n.value))
        E.emu_result_trace(h, g, C.byref(n), codes.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)), n.value)
        mine = [(int(codes[3 * i]), int(codes[3 * i + 1]), int(codes[3 * i + 2]), float(vals[i])) for i in range(n.value)]
        assert mine == traces[g], f"graph {g}: first divergence at {next((i for i, (a, b) in enumerate(zip(mine, traces[g])) if a != b), min(len(mine), len(traces[g])))}"
    E.emu_result_free(h)


def test_capacity_retry_moves_up_a_class():
    """ALD_DEBUG_UNDERCLASS starts every graph two classes too low: the engine must report ALD_ST_CAPACITY for it, the host re-queues
    it one class up until it fits, and the final answer is the same (records of abandoned attempts are dropped)"""
    import os
    pg = A.synth(seed=77, n_graphs=40, v_min=40, v_max=200, edges_per_vertex=4, phasing_per_graph=3)
    want = common.oracle_run(pg)[0]
    _, _, cl_plain = common.emu_run(pg)
    os.environ["ALD_DEBUG_UNDERCLASS"] = "2"
    try:
        got, _, cl = common.emu_run(pg)
    finally:
        del os.environ["ALD_DEBUG_UNDERCLASS"]
    assert not common.compare_results(want, got, pg.n)
    assert (cl >= np.maximum(cl_plain - 2, 0)).all() and (cl > np.maximum(cl_plain - 2, 0)).any()      # at least one graph had to climb


def test_nondefault_parameters():
    """jump_ratio > 1.02 exercises resolve_trivial_vertex_fast and the early `break` of resolve_trivial_vertex"""
    p = A.default_params()
    p.max_decompose_error_ratio[7] = 1.5; p.max_decompose_error_ratio[0] = 0.2; p.min_transcript_coverage = 5.0
    pg = A.synth(seed=31, n_graphs=60, v_min=10, v_max=70, edges_per_vertex=3, phasing_per_graph=8, weight_mode=1)
    want, st, _, _ = common.oracle_run(pg, params=p)
    got, it, _ = common.emu_run(pg, params=p)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])


def test_edge_cases_and_invariant_classes():
    from aletsch_amd.packed import PackedGraphs
    graphs = [
        # empty graph: source + sink only
        dict(V=2, edges=[], vw=[0, 0], lpos=[0, 0], rpos=[0, 0]),
        # a single chain
        dict(V=4, edges=[(0, 1, 5.0), (1, 2, 5.0), (2, 3, 5.0)], vw=[0, 10, 10, 0], lpos=[0, 100, 300, 400], rpos=[0, 200, 400, 400]),
        # isolated internal vertex + below-coverage path (nothing reported)
        dict(V=4, edges=[(0, 1, 1.0), (1, 3, 1.0)], vw=[0, 1, 1, 0], lpos=[0, 100, 300, 400], rpos=[0, 200, 400, 400]),
        # broken vertex (no out-edge) -> resolve_broken_vertex
        dict(V=5, edges=[(0, 1, 9.0), (1, 2, 4.0), (1, 3, 5.0), (3, 4, 5.0)], vw=[0, 3, 3, 3, 0], lpos=[0, 10, 30, 50, 60], rpos=[0, 20, 40, 60, 60]),
        # edge without sample support: the reference asserts in merge_adjacent_equal_edges (count > 0)
        dict(V=4, edges=[(0, 1, 5.0, 0, {}), (1, 2, 5.0), (2, 3, 5.0)], vw=[0, 10, 10, 0], lpos=[0, 100, 300, 400], rpos=[0, 200, 400, 400]),
        # disjoint sample sets on the two sides of a vertex: empty intersection (count becomes 0 downstream)
        dict(V=5, edges=[(0, 1, 6.0, 0, {1: 6.0}), (1, 2, 6.0, 0, {2: 6.0}), (2, 3, 6.0, 0, {1: 6.0}), (3, 4, 6.0, 0, {1: 6.0})], vw=[0, 1, 1, 1, 0], lpos=[0, 10, 30, 50, 60], rpos=[0, 20, 40, 60, 60]),
        # EMPTY_VERTEX (-9) on a path: collected but not reported (scallop.cc:2788-2794)
        dict(V=4, edges=[(0, 1, 5.0), (1, 2, 5.0), (2, 3, 5.0)], vw=[0, 10, 10, 0], lpos=[0, 100, 300, 400], rpos=[0, 200, 400, 400], vtype=[-1, -9, -1, -1]),
        # mixed strands meeting at one vertex
        dict(V=6, edges=[(0, 1, 8.0, 1), (0, 2, 6.0, 2), (1, 3, 8.0, 1), (2, 3, 6.0, 2), (3, 4, 7.0, 1), (3, 5, 7.0, 2), (4, 5, 7.0, 1)], vw=[0, 1, 1, 1, 1, 0],
             lpos=[0, 10, 30, 50, 70, 80], rpos=[0, 20, 40, 60, 80, 80]),
    ]
    pg = PackedGraphs.from_graphs(graphs)
    want = common.oracle_run(pg)[0]
    got, _, _ = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n)
    assert want.status[0] == 0 and want.path_offset[1] == 0           # empty graph: no paths, no failure
    assert want.path_offset[2] - want.path_offset[1] == 1              # chain: one path
    assert want.status[4] >= 100                                       # assert class reported, not a crash
    assert want.path_offset[7] - want.path_offset[6] == 0              # EMPTY_VERTEX path dropped


def test_staging_normalises_unsorted_input():
    """rows not sorted by target, sample lists not ascending, duplicate / unsorted phasing lists: staging must put them into
    the canonical order (edge id = CSR position after a stable sort by target; hyper_set::add_node_list semantics)"""
    import copy
    from aletsch_amd.packed import PackedGraphs
    a = A.synth(seed=5, n_graphs=10, v_min=12, v_max=20, edges_per_vertex=3, n_samples=3, phasing_per_graph=6, weight_mode=1)
    b = copy.deepcopy(a)
    o = b.graph_slices()
    for g in range(b.n):
        V = int(b.g_nv[g]); vo = b.vertex_offset[o["vo"][g]:o["vo"][g] + V + 1]
        for s in range(V):
            lo, hi = int(o["e"][g] + vo[s]), int(o["e"][g] + vo[s + 1])
            if hi - lo >= 2:                  # reverse the row: distinct targets, so the stable sort restores it
                eso = b.edge_sample_offset[o["eo"][g]:o["eo"][g] + int(b.g_ne[g]) + 1]
                if len(set(np.diff(eso[vo[s]:vo[s + 1] + 1]).tolist())) == 1:      # equal support sizes: the sample slices can stay where they are
                    for arr in (b.edge_target, b.edge_weight, b.edge_strand, b.edge_abd):
                        arr[lo:hi] = arr[lo:hi][::-1].copy()
                    k = int(np.diff(eso[vo[s]:vo[s + 1] + 1])[0]); so = int(o["s"][g]) + int(eso[vo[s]])
                    for arr in (b.sample_id, b.sample_abd):
                        blk = arr[so:so + k * (hi - lo)].reshape(hi - lo, k)[::-1].copy(); arr[so:so + k * (hi - lo)] = blk.reshape(-1)
    ra, _, _ = common.emu_run(a)
    rb, _, _ = common.emu_run(b)
    assert not common.compare_results(ra, rb, a.n)


def test_large_classes():
    """graphs beyond 512 vertices: the one-workgroup-per-CU class up to 1024 vertices, then the catch-all class whose hot state lives
    in the wave's slab"""
    for kw, want_cls in ((dict(seed=99, n_graphs=2, v_min=700, v_max=900, edges_per_vertex=4), 9), (dict(seed=96, n_graphs=2, v_min=1100, v_max=1300, edges_per_vertex=3), 10)):
        pg = A.synth(**kw)
        want, st, _, _ = common.oracle_run(pg)
        got, it, cl = common.emu_run(pg)
        assert (cl == want_cls).all(), cl
        assert not common.compare_results(want, got, pg.n)
        assert np.array_equal(it, st[:, 3])


def test_class_beyond_the_catch_all():
    """graphs of more than 2 048 vertices: the largest class (hot state in the slab, 32-bit creation ids); small graphs forced through it
    as well, so that its id width sees parallel edges, phasing lists and multi-sample supports; a graph beyond it is refused with its
    own status instead of ALD_ST_CAPACITY (the reference leaves its rule loop above max_num_exons = 10 000 vertices anyway)"""
    pg = A.synth(seed=78, n_graphs=2, v_min=2500, v_max=2500, edges_per_vertex=3)
    want = common.oracle_run(pg, threads=2)[0]
    got, it, cl = common.emu_run(pg)
    assert (cl == 13).all(), cl
    assert not common.compare_results(want, got, pg.n)
    pg = A.synth(seed=77, n_graphs=60, v_min=8, v_max=120, edges_per_vertex=4, phasing_per_graph=5, n_samples=2)
    want = common.oracle_run(pg, threads=4)[0]
    got, it, cl = common.emu_run(pg, force_class=13)
    assert (cl == 13).all()
    assert not common.compare_results(want, got, pg.n)
    pg = A.synth(seed=5, n_graphs=1, v_min=10300, v_max=10300, edges_per_vertex=2)
    got, it, cl = common.emu_run(pg)
    assert got.status[0] == 4 and cl[0] == -1 and np.diff(got.path_offset)[0] == 0          # ALD_ST_TOO_LARGE


def test_explicit_edge_counts():
    """edge_info.count handed over separately from the sample sets (group_start_boundaries adds counts along grouped boundaries,
    graph_reviser.cc:965-975): larger than |samples| on some edges, zero on a few (router.cc:269 treats those as absent; a merge of
    such an edge is the reference's assert(ei1.count > 0 && ei2.count > 0))"""
    pg = A.synth(seed=91, n_graphs=120, v_min=8, v_max=50, edges_per_vertex=3, n_samples=3, phasing_per_graph=3, weight_mode=1)
    rng = np.random.default_rng(7)
    cnt = pg.sample_counts() + rng.integers(0, 4, pg.edge_target.size).astype(np.int32)
    cnt[rng.random(cnt.size) < 0.002] = 0
    pg.edge_count = cnt.astype(np.int32)
    want, st, _, _ = common.oracle_run(pg)
    got, it, cl = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n)
    plain = common.oracle_run(A.synth(seed=91, n_graphs=120, v_min=8, v_max=50, edges_per_vertex=3, n_samples=3, phasing_per_graph=3, weight_mode=1))[0]
    assert (want.status != 0).any() or not np.array_equal(want.count, plain.count)      # the counts do reach the output


def test_max_num_exons_skips_the_cascade():
    """|V| > max_num_exons: the rule cascade is skipped and the greedy phase decomposes the untouched graph (scallop.cc:49); status
    ALD_ST_SKIPPED_LARGE = 1 for those graphs, paths still reported"""
    p = A.default_params(); p.max_num_exons = 30
    pg = A.synth(seed=55, n_graphs=80, v_min=10, v_max=60, edges_per_vertex=3, weight_mode=2, phasing_per_graph=2)
    want, st, _, _ = common.oracle_run(pg, params=p)
    got, it, cl = common.emu_run(pg, params=p)
    assert not common.compare_results(want, got, pg.n)
    big = pg.g_nv > 30
    assert big.any() and (~big).any()
    assert (want.status[big] == 1).all() and (want.status[~big] == 0).any() and (want.status[~big] == 1).any()     # some grow past the limit mid-run
    assert (np.diff(want.path_offset)[big] > 0).any()


def permuted_rank(pg, seed):
    """A random creation rank per graph (a permutation of 0..E-1), as a reference heap layout other than creation order would give."""
    rng = np.random.default_rng(seed)
    out = np.empty(int(pg.g_ne.sum()), np.int32); o = 0
    for e in pg.g_ne:
        out[o:o + e] = rng.permutation(int(e)).astype(np.int32); o += int(e)
    return out


def test_edge_creation_rank_crosses_the_boundary():
    """ald_graph_view.edge_creation_rank: the scallop edge index (position in the reference's gr.edges(), graph_base.cc:139-153) of
    every input edge.  Ids are behaviour (pe2w / route order, parallel edges, thread_leaf's scan): a permuted rank changes some
    decompositions, identically in the engine and the oracle; the identity rank reproduces the default (CSR position) bit for bit."""
    import copy
    pg = A.synth(seed=21, n_graphs=300, v_min=10, v_max=70, edges_per_vertex=3, phasing_per_graph=10, weight_mode=1, n_samples=3)
    base_o = common.oracle_run(pg)[0]
    ident = copy.copy(pg); ident.edge_rank = pg.identity_rank()
    assert not common.compare_results(base_o, common.oracle_run(ident)[0], pg.n)
    assert not common.compare_results(base_o, common.emu_run(ident)[0], pg.n)
    perm = copy.copy(pg); perm.edge_rank = permuted_rank(pg, 5)
    want, st, _, _ = common.oracle_run(perm)
    got, it, _ = common.emu_run(perm)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(it, st[:, 3])
    changed = [g for g in range(pg.n) if want.paths_of(g) != base_o.paths_of(g)]
    assert changed, "a permuted creation rank should change at least one decomposition (integer weights give ties)"
    # sub-batches and concatenations carry the column along
    sub = perm.select(np.arange(50, 120))
    assert not common.compare_results(common.oracle_run(sub)[0], common.emu_run(sub)[0], sub.n)
    from aletsch_amd.packed import PackedGraphs
    mix = PackedGraphs.concat([pg.select(np.arange(10)), sub])          # graphs without a rank in front of graphs with one
    assert not common.compare_results(common.oracle_run(mix)[0], common.emu_run(mix)[0], mix.n)


def test_full_record_pool_grows_and_the_batch_runs_again(monkeypatch):
    """ALD_ST_POOL_FULL is its own status (not the class-overflow code): the host grows the pool and decomposes the batch again
    instead of walking the graph up through every larger class"""
    pg = A.synth(seed=33, n_graphs=60, v_min=20, v_max=90, edges_per_vertex=4)
    want = common.oracle_run(pg)[0]
    _, _, cl_plain = common.emu_run(pg)
    monkeypatch.setenv("ALD_DEBUG_POOL_WORDS", "900")
    got, _, cl = common.emu_run(pg)
    assert not common.compare_results(want, got, pg.n)
    assert np.array_equal(cl, cl_plain)                  # nobody changed class because of the pool


def test_fixed_size_star_declines_where_the_reference_asserts():
    """The fixed-size form of the trivial decomposition (star_fixed) checks everything the sequential form would trip over -- a weight
    below min_guaranteed_edge_weight, a zero edge count, an exhausted id space, a centre edge that keeps a remainder -- BEFORE its first
    write and hands such a star to the whole-wave form: graphs with zero counts and with a minimum weight far above the balanced weights
    must end exactly as the oracle's (status words of the reference's asserts included)."""
    rng = np.random.default_rng(77)
    for k in range(6):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), v_min=int(rng.choice([8, 32, 64])), v_max=int(rng.choice([64, 100])), edges_per_vertex=int(rng.choice([3, 4])),
                  weight_mode=int(rng.choice([0, 1, 2])), n_samples=int(rng.choice([1, 2])), phasing_per_graph=int(rng.choice([0, 5])), n_graphs=120)
        p = A.default_params()
        if k % 2 == 0: p.min_guaranteed_edge_weight = float(rng.choice([5.0, 30.0]))
        pg = A.synth(**kw)
        if k % 3 != 0:
            cnt = pg.sample_counts() - rng.integers(0, 2, pg.edge_target.size).astype(np.int32); pg.edge_count = np.maximum(cnt, 0).astype(np.int32)
        want = common.oracle_run(pg, params=p)[0]
        got = common.emu_run(pg, params=p)[0]
        bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
        assert not bad, f"round {k}: {len(bad)} mismatches, first {bad[:3]}"
        if k % 3 != 0: assert int((want.status >= 100).sum()) > 0          # the case is really exercised: some graphs end on an assert class
