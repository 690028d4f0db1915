"""GPU tier: the reference-shaped C++ surface (aletsch_amd/host/gpu_scallop.hpp) over the C ABI.

tests/host_adapter/adapter_test.cc instantiates aletsch::gpu_scallop with mock types that carry the member names of the
reference's splice_graph / hyper_set / parameters / path (scallop/scallop.h:31-51), is compiled here with g++ -std=c++11
(what the reference's build uses) and linked against the in-tree library; its output must equal what the ctypes path gives
for the same graph, and the oracle's.
"""
import os
import subprocess

import numpy as np
import pytest

import aletsch_amd as A
import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "adapter_test")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "aletsch_amd", "lib")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_adapter", "adapter_test.cc"),
                    "-o", BIN, "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib], check=True)


def shuffled_rank(one, shuffle_seed):
    """creation rank of every CSR edge of a single-graph batch when the caller's gr.edges() lists the edges in the scrambled order of
    graph_text(one, shuffle_seed): the adapter hands the position in gr.edges() over as edge_creation_rank"""
    V, E = int(one.g_nv[0]), int(one.g_ne[0])
    keys = [(s, int(one.edge_target[k])) for s in range(V) for k in range(one.vertex_offset[s], one.vertex_offset[s + 1])]
    perm = np.random.default_rng(shuffle_seed).permutation(E)
    slots = {}
    for pos, k in enumerate(perm):
        slots.setdefault(keys[k], []).append(pos)
    rank = np.zeros(E, np.int32); nxt = {}
    for k in range(E):
        j = nxt.get(keys[k], 0); nxt[keys[k]] = j + 1
        rank[k] = slots[keys[k]][j]
    return rank


def graph_text(one, shuffle_seed=None):
    """a single-graph batch in adapter_test's stdin format; edges optionally in scrambled order: the adapter lays them out as CSR by
    (source, target, position in gr.edges()) and hands the position in gr.edges() over as the creation rank (= scallop's edge index)"""
    V, E, P = int(one.g_nv[0]), int(one.g_ne[0]), int(one.g_np[0])
    lines = ["%d %d %d" % (V, E, P)]
    for i in range(V):
        lines.append("%r %d %d" % (float(one.vertex_weight[i]), int(one.vertex_lpos[i]), int(one.vertex_rpos[i])))
    edges = [(s, int(one.edge_target[k]), float(one.edge_weight[k])) for s in range(V) for k in range(one.vertex_offset[s], one.vertex_offset[s + 1])]
    if shuffle_seed is not None:
        perm = np.random.default_rng(shuffle_seed).permutation(E)
        slots = {}                                   # positions each (s, t) group occupies after the shuffle, ascending
        for pos, k in enumerate(perm):
            slots.setdefault(edges[k][:2], []).append(pos)
        out = [None] * E; nxt = {}
        for k in range(E):                           # parallel edges keep their relative order: that IS the tie-break
            key = edges[k][:2]; j = nxt.get(key, 0); nxt[key] = j + 1
            out[slots[key][j]] = edges[k]
        edges = out
    for s, t, w in edges:
        lines.append("%d %d %r" % (s, t, w))
    for p in range(P):
        vs = one.phasing_vertex[one.phasing_offset[p]:one.phasing_offset[p + 1]]
        lines.append("%d %d %s" % (len(vs), int(one.phasing_count[p]), " ".join(str(int(x)) for x in vs)))
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(seed=61, n_graphs=6, v_min=8, v_max=40, edges_per_vertex=3),
                                 dict(seed=62, n_graphs=6, v_min=10, v_max=50, edges_per_vertex=3, phasing_per_graph=6, weight_mode=2)])
def test_adapter_matches_abi_and_oracle(cfg):
    build()
    pg = A.synth(**cfg)
    # the mock edge_info of adapter_test carries one sample (id 0) with abundance = weight: make the packed batch say the same
    pg.sample_id[:] = 0; pg.sample_abd[:] = pg.edge_weight; pg.edge_abd[:] = pg.edge_weight
    want = common.oracle_run(pg)[0]
    got = A.decompose(pg, device=0)
    assert not common.compare_results(want, got, pg.n, conf_tol=1e-9)
    whole = want
    for g in range(pg.n):
        for seed in (None, 5):
            one = pg.select(np.array([g]))
            out = subprocess.run([BIN], input=graph_text(one, seed), capture_output=True, text=True, check=True).stdout.splitlines()
            if seed is None:
                want = whole; gg = g
            else:                                   # a caller whose gr.edges() iterates in another order: the ids differ, and so may the paths
                one.edge_rank = shuffled_rank(one, seed); want = common.oracle_run(one)[0]; gg = 0
            head = out[0].split(); assert head[0] == "status" and int(head[1]) == int(want.status[gg])
            a, b = int(want.path_offset[gg]), int(want.path_offset[gg + 1])
            assert int(head[3]) == b - a
            for k, line in enumerate(out[1:]):
                f, vs = line.split(" :"); f = f.split(); i = a + k
                assert float(f[0]) == want.weight[i] and float(f[1]) == want.abd[i] and float(f[2]) == want.reads[i]
                assert int(f[3]) == want.length[i] and int(f[4]) == want.count[i] and f[5] == chr(want.strand[i])
                pv = want.path_vertices[want.pv_offset[i]:want.pv_offset[i + 1]]
                assert [int(x) for x in vs.split()] == [int(x) for x in pv]
                # junctions: consecutive internal vertices that do not touch (scallop.cc:2797-2822)
                ov = int(pg.g_nv[:g].sum()); lp = pg.vertex_lpos[ov:]; rp = pg.vertex_rpos[ov:]
                nj = sum(1 for q in range(2, len(pv) - 1) if lp[pv[q]] != rp[pv[q - 1]])
                assert int(f[6]) == nj


DBIN = os.path.join(ROOT, "tests", "_build", "dispatch_test")


def build_dispatch():
    os.makedirs(os.path.dirname(DBIN), exist_ok=True)
    lib = os.path.join(ROOT, "aletsch_amd", "lib")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_adapter", "dispatch_test.cc"),
                    "-o", DBIN, "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib], check=True)


def _parse_sink(lines):
    out = []
    for line in lines:
        head, ex, sm = line.split(" :")
        f = head.split()
        out.append(dict(hash=int(f[0]), count=int(f[1]), strand=f[2], coverage=float(f[3]), cov2=float(f[4]), conf=float(f[5]), abd=float(f[6]),
                        count1=int(f[7]), count2=int(f[8]), tid=int(f[9]), exons=[tuple(int(x) for x in e.split("-")) for e in ex.split()],
                        samples=[dict(sid=int(q[0]), cov2=float(q[1]), conf=float(q[2]), abd=float(q[3]), count1=int(q[4])) for q in (s.split(",") for s in sm.split())]))
    return out


@pytest.mark.gpu
def test_dispatch_queue_matches_the_serial_merge():
    """aletsch::gpu_assembly_queue (aletsch_amd/host/gpu_dispatch.hpp: queue + flusher in place of the inline calls of
    meta/incubator.cc:553-577 / assembler.cc:1075-1136): graphs submitted one by one, batched across submitters, three batches
    rotating through staging / kernel / merge.  One submitting thread -> the merged set equals, bit for bit, one whole-batch
    run through the ctypes path; four threads -> the same set up to the order of the floating-point additions."""
    build_dispatch()
    pg = A.synth(seed=63, n_graphs=300, v_min=8, v_max=40, edges_per_vertex=3, phasing_per_graph=4, weight_mode=2, layout_mode=1)
    pg.sample_id[:] = 0; pg.sample_abd[:] = pg.edge_weight; pg.edge_abd[:] = pg.edge_weight      # what the mock edge_info carries
    sid = np.arange(pg.n, dtype=np.int32) % 3
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        n_failed = int((b.result().status != 0).sum())
        sink = A.TranscriptSink(0.8)
        sink.add_batch(b, sid=sid)
        want = sink.items()
    assert len(want) > 100
    text = "".join(graph_text(pg.select(np.array([g]))) for g in range(pg.n))
    # one submitter, batches of 64 graphs, 3 slots: five batches, the last one partial
    out = subprocess.run([DBIN], input="%d 1 64 3 1\n%s" % (pg.n, text), capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "submitted %d failed %d batches 5" % (pg.n, n_failed), out[0]
    assert _parse_sink(out[1:]) == want
    # several devices in one process (here: three GPU threads with two batch objects each, all on device 0): batches finish in any
    # order, the merge takes them in the order they were cut -- same set, bit for bit
    out = subprocess.run([DBIN], input="%d 1 32 2 3\n%s" % (pg.n, text), capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "submitted %d failed %d batches 10" % (pg.n, n_failed), out[0]
    assert _parse_sink(out[1:]) == want
    # four submitters, batches of 32, 2 slots (submitters wait for a free slot): ticket order is up to the scheduler
    out = subprocess.run([DBIN], input="%d 4 32 2 2\n%s" % (pg.n, text), capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].startswith("submitted %d failed %d batches" % (pg.n, n_failed)), out[0]
    got = _parse_sink(out[1:])
    key = lambda x: (x["hash"], x["strand"], tuple(tuple(e) for e in x["exons"]))
    multi = [x for x in want if len(x["exons"]) > 1]                     # single-exon clusters depend on arrival order (bounds widen as they merge)
    gm = {key(x): x for x in got if len(x["exons"]) > 1}
    assert len(gm) == len(multi)
    for w in multi:
        g = gm[key(w)]
        assert g["count"] == w["count"] and g["count2"] == w["count2"]
        assert np.isclose(g["coverage"], w["coverage"], rtol=1e-12) and np.isclose(g["cov2"], w["cov2"], rtol=1e-12)
        assert sorted(s["sid"] for s in g["samples"]) == sorted(s["sid"] for s in w["samples"])


@pytest.mark.gpu
def test_batched_adapter_enqueue_flush_paths():
    """aletsch::gpu_scallop_batch: many graphs enqueued, ONE flush, `paths(ticket)` per graph; the batch
    object is cleared and reused for a second round, whose output is the one compared -- graph by graph against the oracle."""
    build()
    pg = A.synth(seed=64, n_graphs=40, v_min=8, v_max=60, edges_per_vertex=3, phasing_per_graph=5, weight_mode=1)
    pg.sample_id[:] = 0; pg.sample_abd[:] = pg.edge_weight; pg.edge_abd[:] = pg.edge_weight
    from aletsch_amd.packed import PackedGraphs
    ones = []
    for g in range(pg.n):
        one = pg.select(np.array([g])); one.edge_rank = shuffled_rank(one, 9) if g % 2 else one.identity_rank(); ones.append(one)
    want = common.oracle_run(PackedGraphs.concat(ones))[0]
    text = "".join(graph_text(pg.select(np.array([g])), 9 if g % 2 else None) for g in range(pg.n))
    out = subprocess.run([BIN, "batch", str(pg.n)], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
    pos = 0
    for g in range(pg.n):
        head = out[pos].split(); pos += 1
        a, b = int(want.path_offset[g]), int(want.path_offset[g + 1])
        assert head[0] == "status" and int(head[1]) == int(want.status[g]) and int(head[3]) == b - a, (g, head)
        for i in range(a, b):
            f, vs = out[pos].split(" :"); f = f.split(); pos += 1
            assert float(f[0]) == want.weight[i] and float(f[1]) == want.abd[i] and float(f[2]) == want.reads[i]
            assert int(f[3]) == want.length[i] and int(f[4]) == want.count[i] and f[5] == chr(want.strand[i])
            assert [int(x) for x in vs.split()] == [int(x) for x in want.path_vertices[want.pv_offset[i]:want.pv_offset[i + 1]]]
    assert pos == len(out)


@pytest.mark.gpu
def test_adapter_takes_graphs_as_assemble_receives_them():
    """aletsch::gpu_scallop_batch::enqueue_raw: the call shape of assembler::assemble(gx, px, sid) (meta/assembler.cc:1075) -- the graph
    BEFORE extend_strands / boundary grouping and the phase set in exon coordinates; the library runs the pre-steps (ald_pre_assemble).
    Output per graph == the oracle's pre-steps followed by the oracle's decomposition."""
    build()
    from aletsch_amd.packed import PackedGraphs
    rng = np.random.default_rng(12)
    texts = []; staged = []
    for t in range(30):
        g, phases = common.gene_like_raw(rng, n_runs=int(rng.integers(3, 8)), strand=".")
        g["edges"] = [(s, tt, w, 0, {0: w}) for s, tt, w, _st, _sp in g["edges"]]          # what the mock edge_info of adapter_test carries
        pg = PackedGraphs.from_graphs([g])
        order = sorted(range(len(g["edges"])), key=lambda k: (g["edges"][k][0], g["edges"][k][1]))
        pg.edge_rank = np.array(order, np.int32)
        lines = ["%d %d %d" % (g["V"], len(g["edges"]), len(phases))]
        lines += ["%r %d %d" % (float(g["vw"][i]), g["lpos"][i], g["rpos"][i]) for i in range(g["V"])]
        lines += ["%d %d %r" % (s, tt, w) for s, tt, w, _a, _b in g["edges"]]                # listing order == creation order == gr.edges()
        lines += ["%d %d %s" % (len(co), c, " ".join(str(x) for x in co)) for co, c in phases]
        texts.append("\n".join(lines) + "\n")
        import ctypes as C
        O = common.oracle_lib()
        O.ora_pre_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        O.ora_staged_view.argtypes = [C.c_void_p, C.c_void_p]; O.ora_staged_free.argtypes = [C.c_void_p]; O.ora_staged_boundary_maps.argtypes = [C.c_void_p] * 5
        want, _, _, rc = A.pre_assemble(pg, phases, 10000, _lib=O, _prefix="ora")
        staged.append((want, rc))
    out = subprocess.run([BIN, "raw", str(len(texts))], input="".join(texts), capture_output=True, text=True, check=True).stdout.splitlines()
    pos = 0; n_paths = 0
    for want_pg, rc in staged:
        head = out[pos].split(); pos += 1
        if rc:
            assert int(head[1]) == rc and int(head[3]) == 0
            continue
        want = common.oracle_run(want_pg)[0]
        assert head[0] == "status" and int(head[1]) == int(want.status[0]) and int(head[3]) == int(want.path_offset[1]), head
        for i in range(int(want.path_offset[1])):
            f, vs = out[pos].split(" :"); f = f.split(); pos += 1; n_paths += 1
            assert float(f[0]) == want.weight[i] and float(f[1]) == want.abd[i] and float(f[2]) == want.reads[i]
            assert [int(x) for x in vs.split()] == [int(x) for x in want.path_vertices[want.pv_offset[i]:want.pv_offset[i + 1]]]
    assert pos == len(out) and n_paths > 50


@pytest.mark.gpu
def test_comm_gather_with_several_ranks_on_one_gpu():
    """ald_comm_gather_streams with W = 2, 3 and 8 ranks (the multi-rank offsets / receive sizing had never executed).  RCCL
    refuses two ranks on one device, so the ranks are threads of one process and librccl.so is replaced, through ALD_RCCL_LIB, by
    tests/host_adapter/mock_rccl.cc (device-to-device copies behind the nine nccl* entry points the library binds); then a refused
    ncclSend: the failing rank must report the error with its thread out of group mode, and nobody may hang."""
    bld = os.path.join(ROOT, "tests", "_build"); os.makedirs(bld, exist_ok=True)
    lib = os.path.join(ROOT, "aletsch_amd", "lib"); mock = os.path.join(bld, "libmock_rccl.so"); exe = os.path.join(bld, "comm_ranks_test")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", mock, os.path.join(ROOT, "tests", "host_adapter", "mock_rccl.cc")], check=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_adapter", "comm_ranks_test.cc"), "-o", exe, "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib,
                    "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-pthread"], check=True)
    env = dict(os.environ, ALD_RCCL_LIB=mock)
    for world in ("2", "3", "8"):                               # 8: BASELINE.json configs[3]'s world size (one step goes through ald_comm_gather_begin / _wait)
        r = subprocess.run([exe, world], capture_output=True, text=True, timeout=90, env=env)
        assert r.returncode == 0 and "COMM_RANKS_OK world=" + world in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    for fail in ("1", "0"):
        r = subprocess.run([exe, "2", fail], capture_output=True, text=True, timeout=90, env=dict(env, ALD_MOCK_RCCL_FAIL_SEND=fail))
        assert r.returncode == 0 and "injected send failure handled" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_replay_dump_program_on_gpu():
    """tools/replay_dump.cc (in C++ as the host side is): a bundle dump in the reference's own text form
    (splice_graph::write + hyper_set::write) -> aletsch::read_bundle_dump -> gpu_assembly_queue -> merged transcript set -> GTF through
    ald_gtf_format_transcript.  The GTF must equal, byte for byte, what the Python path makes of the same dump (graphio reader ->
    DecompBatch -> TranscriptSink -> format_transcript), whose pieces are each checked against the oracle / the reference's writers."""
    from aletsch_amd import graphio
    bld = os.path.join(ROOT, "tests", "_build"); os.makedirs(bld, exist_ok=True)
    lib = os.path.join(ROOT, "aletsch_amd", "lib"); exe = os.path.join(bld, "replay_dump")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "replay_dump.cc"),
                    "-o", exe, "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib, "-pthread"], check=True)
    src = A.synth(seed=8086, n_graphs=700, v_min=6, v_max=90, edges_per_vertex=3, layout_mode=1, weight_mode=1, phasing_per_graph=5, strand_mode=1)
    text = graphio.write_bundle_dump(src, chrm="17")
    r = subprocess.run([exe, "-b", "256"], input=text, capture_output=True, text=True, timeout=300)      # three batches through two slots
    assert r.returncode == 0, r.stderr[-2000:]
    pg, meta = graphio.read_bundle_dump(text)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        assert (b.result().status == 0).all()
        sink = A.TranscriptSink(0.8); sink.add_batch(b, np.zeros(pg.n, np.int32), tid_base=0, skip_single_exon=True)
    want = []
    for it in sink.items():
        g, p = it["tid"] >> 20, it["tid"] & ((1 << 20) - 1)
        want.append(A.format_transcript("17", "aletsch", meta[g]["gid"], A.transcript_id("17", meta[g]["gid"], p), it["strand"], it["coverage"], it["exons"], cov2=-1.0, count=it["count2"]))
    assert len(want) > 300 and r.stdout == "".join(want)
    assert ("%d bundles, 0 not decomposed, %d transcripts" % (pg.n, len(want))) in r.stderr
