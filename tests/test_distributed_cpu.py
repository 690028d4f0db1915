"""N > 1 path on CPU: world_size-2 gloo run of the record gather + the deterministic merge order."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from aletsch_amd.distributed import gather_records, parse_records, shard_range, REC_HDR_WORDS
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_total = 7
    lo, hi = shard_range(n_total, rank, world)
    # fabricate this rank's record stream: graph g (local id) has g+1 paths of 3+k vertices
    words = []
    for g in range(hi - lo):
        for k in range(lo + g + 1):
            nv = 3 + (k %% 3)
            hdr = np.zeros(REC_HDR_WORDS, np.uint32)
            hdr[0] = g; hdr[1] = k; hdr[2] = nv; hdr[3] = 100 * nv; hdr[4] = 1; hdr[5] = ord("+")
            hdr[6:14] = np.array([1.5 + k, 2.0, 1.0, 3.25 * (lo + g)], np.float64).view(np.uint32)
            v = np.arange(nv, dtype=np.uint32)
            rec = np.concatenate([hdr, v, np.zeros((REC_HDR_WORDS + nv) & 1, np.uint32)])
            words.append(rec)
    rec = np.concatenate(words) if words else np.zeros(0, np.uint32)
    got = gather_records(rec, torch.device("cpu"), graph_offset=lo)
    if rank == 0:
        allrec = parse_records(np.concatenate(got))
        graphs = [r["graph"] for r in allrec]
        assert graphs == sorted(graphs), graphs
        assert sorted(set(graphs)) == list(range(n_total))
        for g in range(n_total):
            mine = [r for r in allrec if r["graph"] == g]
            assert [r["index"] for r in mine] == list(range(g + 1))
            assert all(abs(r["reads"] - 3.25 * g) < 1e-12 for r in mine)
        print("GATHER_OK", len(allrec))
    dist.barrier()
    # the persistent form bench.py uses: one gatherer, several steps of differently sized streams inside the agreed capacity, a
    # renegotiation entered by both ranks when the shape grows, and a loud error instead of a lone collective on overflow
    from aletsch_amd.distributed import RecordGatherer
    def stream(n_rec, tag):
        out = []
        for k in range(n_rec):
            hdr = np.zeros(REC_HDR_WORDS, np.uint32); hdr[0] = k; hdr[1] = 0; hdr[2] = 2; hdr[4] = tag; hdr[5] = ord(".")
            out.append(np.concatenate([hdr, np.array([0, 1], np.uint32), np.zeros(REC_HDR_WORDS & 1, np.uint32)]))
        return np.concatenate(out) if out else np.zeros(0, np.uint32)
    G = RecordGatherer(torch.device("cpu"))
    for step, n_rec in enumerate((40 + 3 * rank, 41, 0, 43 - rank)):
        w = stream(n_rec, 10 * step + rank)
        G.gather(torch.from_numpy(w.view(np.int32).copy()), graph_offset=1000 * rank)
        if rank == 0:
            st = G.streams()
            assert [off for _, off in st] == [0, 1000]
            want = [(40, 41, 0, 43)[step], (43, 41, 0, 42)[step]]
            for r, (words, off) in enumerate(st):
                recs = parse_records(words.copy())
                assert len(recs) == want[r] and all(x["count"] == 10 * step + r for x in recs), (step, r, len(recs))
    big = stream(400, 7)
    try:
        G.gather(torch.from_numpy(big.view(np.int32).copy()))
        raise SystemExit("overflow was not refused")
    except ValueError:
        pass
    G.renegotiate(big.size)                      # both ranks
    G.gather(torch.from_numpy(big.view(np.int32).copy()), graph_offset=5 * rank)
    if rank == 0:
        assert [len(parse_records(wd.copy())) for wd, _ in G.streams()] == [400, 400]
        print("STEPS_OK")
    dist.barrier()
    # shapes that change from step to step, rank by rank: gather_any agrees on the sizes first and grows the capacity everywhere at once
    G2 = RecordGatherer(torch.device("cpu"))
    for step, sizes in enumerate(((5, 9), (300, 2), (0, 0), (2, 2500), (2600, 1))):
        w = stream(sizes[rank], 50 + step)
        G2.gather_any(torch.from_numpy(w.view(np.int32).copy()), graph_offset=7 * rank)
        if rank == 0:
            assert [len(parse_records(wd.copy())) for wd, _ in G2.streams()] == list(sizes), step
            assert [off for _, off in G2.streams()] == [0, 7]
    if rank == 0:
        print("ANY_OK")
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_gloo_world2_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK 28" in r.stdout and "STEPS_OK" in r.stdout and "ANY_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from aletsch_amd.distributed import shard_range
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


SHARD_WORKER = textwrap.dedent('''
    import os, sys, json, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import aletsch_amd as A, common
    from aletsch_amd.distributed import StreamGatherer, merge_streams, shard_range
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    spec = json.loads(os.environ["ALD_TEST_SPEC"])
    pg = A.synth(**spec)                                   # every rank knows the whole batch and takes its block of it
    sid = (np.arange(pg.n) %% 4).astype(np.int32)
    lo, hi = shard_range(pg.n, rank, world)
    mine = pg.select(np.arange(lo, hi))
    res, _, _ = common.emu_run(mine)                       # the engine (single-lane emulation of the HIP kernels) on this rank's shard
    words = common.transcript_stream_from_result(mine, res, sid[lo:hi])
    G = StreamGatherer(torch.device("cpu"))
    G.gather(torch.from_numpy(words.view(np.int32).copy()), graph_offset=lo)
    if rank == 0:
        sink = merge_streams(A.TranscriptSink(0.8), G.streams())
        json.dump(sink.items(), open(os.environ["ALD_TEST_OUT"], "w"))
        print("MERGED", len(sink.items()))
    dist.barrier()
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def test_sharded_batch_merges_to_the_single_rank_result(tmp_path):
    """SURVEY.md 8e's determinism rule, tested: a real batch is sharded over two gloo ranks, every rank decomposes its block with the
    engine (emulated), the finished transcript streams are gathered and merged on rank 0 in ascending global graph id -- and the
    merged transcript set (every field of every item, tids included) equals the one-rank merge of the unsharded batch, which in turn
    equals the same transcripts fed through the reference-pinned ald_tset_add."""
    import json
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import aletsch_amd as A
    import common
    spec = dict(seed=71, n_graphs=90, v_min=8, v_max=60, edges_per_vertex=3, layout_mode=1, weight_mode=2, phasing_per_graph=3, strand_mode=1)
    pg = A.synth(**spec)
    sid = (np.arange(pg.n) % 4).astype(np.int32)
    res, _, _ = common.emu_run(pg)
    single = A.TranscriptSink(0.8); single.add_stream(common.transcript_stream_from_result(pg, res, sid))
    want = single.items()
    assert len(want) > 50
    # the stream merge against the per-group entry point whose rules are pinned to the reference's transcript_set.cc
    groups = []
    import math
    sl = pg.graph_slices()
    for g in range(pg.n):
        ts = []
        for k, p in enumerate(range(int(res.path_offset[g]), int(res.path_offset[g + 1]))):
            v = res.path_vertices[int(res.pv_offset[p]):int(res.pv_offset[p + 1])]; ex = []
            for x in v[1:-1]:
                l, r = int(pg.vertex_lpos[sl["v"][g] + x]), int(pg.vertex_rpos[sl["v"][g] + x])
                if l >= r: continue
                if ex and ex[-1][1] == l: ex[-1] = (ex[-1][0], r)
                else: ex.append((l, r))
            ts.append((chr(int(res.strand[p])), math.log(1.0 + float(res.weight[p])), float(res.conf[p]), float(res.abd[p]), int(res.count[p]), (g << 20) | k, ex))
        groups.append((int(sid[g]), ts))
    pinned = A.TranscriptSink(0.8); pinned.add_groups(groups)
    assert pinned.items() == want
    out = tmp_path / "merged.json"
    script = tmp_path / "shard_worker.py"; script.write_text(SHARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ALD_TEST_SPEC=json.dumps(spec), ALD_TEST_OUT=str(out))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = json.load(open(out))
    for x in got:
        x["exons"] = [tuple(e) for e in x["exons"]]
    assert got == want


CFG4_WORKER = textwrap.dedent('''
    import os, sys, json, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import aletsch_amd as A, common
    from aletsch_amd.distributed import StreamGatherer, merge_streams
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(os.environ["ALD_TEST_N"])
    pg = A.synth(seed=1004 + rank, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)      # SURVEY 8d: cfg4 = seed 1004 + rank, 64v/256e
    res, _, _ = common.emu_run(pg)
    words = common.transcript_stream_from_result(pg, res, None, skip_single_exon=True)
    G = StreamGatherer(torch.device("cpu"))
    G.gather(torch.from_numpy(words.view(np.int32).copy()), graph_offset=rank * n)           # bench.py's offsets: rank * graphs per rank
    if rank == 0:
        sink = merge_streams(A.TranscriptSink(0.8), G.streams())
        json.dump(sink.items(), open(os.environ["ALD_TEST_OUT"], "w"))
        print("MERGED", len(sink.items()))
    dist.barrier()
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def test_cfg4_shards_merge_in_rank_order(tmp_path):
    """BASELINE configs[3] with two of its eight ranks: every rank synthesises ITS shard (64v/256e, seed 1004 + rank), decomposes it
    with the engine (emulated), the finished transcripts are gathered and merged on rank 0 in rank order = ascending global graph id.
    The merged set equals what one process gets from the two shards' streams merged in that order."""
    import json
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import aletsch_amd as A
    import common
    n = 300
    single = A.TranscriptSink(0.8)
    for rank in range(2):
        pg = A.synth(seed=1004 + rank, n_graphs=n, v_min=64, v_max=64, fixed_edges=256)
        res, _, _ = common.emu_run(pg)
        single.add_stream(common.transcript_stream_from_result(pg, res, None, skip_single_exon=True), graph_offset=rank * n)
    want = single.items()
    assert len(want) > 1000
    out = tmp_path / "merged.json"
    script = tmp_path / "cfg4_worker.py"; script.write_text(CFG4_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ALD_TEST_N=str(n), ALD_TEST_OUT=str(out))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29521", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = json.load(open(out))
    for x in got:
        x["exons"] = [tuple(e) for e in x["exons"]]
    assert got == want


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (fresh child processes, before any GPU call), relays rank
    0's JSON line and reports the world the collective actually saw; here as a CPU / gloo dry run of that plumbing"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--graphs", "300"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_gather"] == [0, 1] and d["graph_offsets"] == [0, 300] and d["graphs_staged"] == [300, 300]
    # a launcher world that contradicts --gpus is refused
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"], capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode == 2 and "WORLD_SIZE=1" in r2.stderr
