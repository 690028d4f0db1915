"""Test-side helpers.  This module (with bench.py's cpu_baseline leg and smoke()) is the only place that loads
anything under oracle/ or the single-lane emulation; the product package never does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from aletsch_amd.packed import PackedGraphs, DecompResult, export_via

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = None
_EMU = None


def _ensure(path: str, make_dir: str):
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, make_dir), "-j8"], check=True, stdout=subprocess.DEVNULL)
    return path


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = C.CDLL(_ensure(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle"))
        _ORACLE.ora_result_seconds.restype = C.c_double
        _ORACLE.ora_result_seconds.argtypes = [C.c_void_p]
        _ORACLE.ora_result_free.argtypes = [C.c_void_p]
    return _ORACLE


_EMU_ROWS = None


def emu_rows_lib():
    """the single-lane emulation of the ADJACENCY-ROW form of the engine (aletsch_amd/csrc/decomp_device_rows.h, make ROWS=1), built with
    the row checker: every row sorted, every live edge in exactly the rows it belongs to, verified after every rule"""
    global _EMU_ROWS
    if _EMU_ROWS is None:
        path = os.path.join(ROOT, "tests", "_build", "libkernel_emu_rows.so")
        src = [os.path.join(ROOT, "aletsch_amd", "csrc", f) for f in ("decomp_device_rows.h", "decomp_common.h")]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in src):
            subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "kernel_emu"), "-j8", "ROWS=1"], check=True, stdout=subprocess.DEVNULL)
        _EMU_ROWS = C.CDLL(path)
        _EMU_ROWS.emu_result_free.argtypes = [C.c_void_p]
    return _EMU_ROWS


_EMU_KEEP = None


def emu_keep_lib():
    """the emulation with the sweep records kept between sweeps in EVERY size class (make KEEP=1), built with the checker: at every sweep
    every kept record that is not marked must equal a fresh evaluation (a stale one aborts the process)"""
    global _EMU_KEEP
    if _EMU_KEEP is None:
        path = os.path.join(ROOT, "tests", "_build", "libkernel_emu_keep.so")
        src = [os.path.join(ROOT, "aletsch_amd", "csrc", f) for f in ("decomp_device.h", "decomp_common.h")]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in src):
            subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "kernel_emu"), "-j8", "KEEP=1"], check=True, stdout=subprocess.DEVNULL)
        _EMU_KEEP = C.CDLL(path)
        _EMU_KEEP.emu_result_free.argtypes = [C.c_void_p]
    return _EMU_KEEP


def emu_lib():
    global _EMU
    if _EMU is None:
        # ALD_EMU_LIB: another build of the emulation (e.g. one compiled with -DALD_EMU_CHECK, which verifies the adjacency rows after every rule)
        _EMU = C.CDLL(os.environ.get("ALD_EMU_LIB") or _ensure(os.path.join(ROOT, "tests", "_build", "libkernel_emu.so"), "tests/kernel_emu"))
        _EMU.emu_result_free.argtypes = [C.c_void_p]
    return _EMU


def oracle_run(pg: PackedGraphs, threads: int = 1, trace: bool = False, params=None):
    """-> (DecompResult, stats[n,6], seconds, traces or None)"""
    O = oracle_lib()
    h = C.c_void_p()
    rc = O.ora_run_packed(*pg.c_args(), C.byref(params) if params is not None else None, C.c_int32(threads), C.c_int32(1 if trace else 0), C.byref(h))
    assert rc == 0
    r = export_via(O.ora_result_export, h, pg.n)
    st = np.zeros((pg.n, 6), np.int32)
    O.ora_result_stats(h, st.ctypes.data_as(C.POINTER(C.c_int32)))
    sec = O.ora_result_seconds(h)
    traces = None
    if trace:
        traces = []
        for g in range(pg.n):
            n = C.c_int32()
            O.ora_result_trace(h, g, C.byref(n), None, None, 0)
            codes = np.zeros(3 * max(n.value, 1), np.int32); vals = np.zeros(max(n.value, 1))
            O.ora_result_trace(h, g, C.byref(n), codes.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)), n.value)
            traces.append([(int(codes[3 * i]), int(codes[3 * i + 1]), int(codes[3 * i + 2]), float(vals[i])) for i in range(n.value)])
    O.ora_result_free(h)
    return r, st, sec, traces


def oracle_transcripts(pg: PackedGraphs):
    """coverage + exon lists per path (scallop.cc:3250-3266, essential.cc:719-748) from the oracle."""
    O = oracle_lib()
    h = C.c_void_p()
    assert O.ora_run_packed(*pg.c_args(), None, C.c_int32(1), C.c_int32(0), C.byref(h)) == 0
    r = export_via(O.ora_result_export, h, pg.n)
    te = C.c_int64()
    O.ora_result_export_transcripts(h, C.byref(te), None, None, None)
    P = len(r.weight)
    cov = np.zeros(P); eo = np.zeros(P + 1, np.int64); lr = np.zeros(2 * max(te.value, 1), np.int32)
    O.ora_result_export_transcripts(h, C.byref(te), cov.ctypes.data_as(C.POINTER(C.c_double)), eo.ctypes.data_as(C.POINTER(C.c_int64)), lr.ctypes.data_as(C.POINTER(C.c_int32)))
    O.ora_result_free(h)
    return r, cov, eo, lr[:2 * te.value].reshape(-1, 2)


def emu_run(pg: PackedGraphs, trace_cap: int = 0, force_class: int = 0, params=None, rows: bool = False, keep: bool = False):
    """single-lane emulation of the HIP engine -> (DecompResult, iterations[n], class[n]); rows: its adjacency-row form (make ROWS=1);
    keep: sweep records kept in every class + stale-record checker (make KEEP=1)"""
    E = emu_rows_lib() if rows else (emu_keep_lib() if keep else emu_lib())
    h = C.c_void_p()
    rc = E.emu_run_packed(*pg.c_args(), C.byref(params) if params is not None else None, C.c_int32(trace_cap), C.c_int32(force_class), C.byref(h))
    assert rc == 0, rc
    r = export_via(E.emu_result_export, h, pg.n)
    it = np.zeros(pg.n, np.int32); cl = np.zeros(pg.n, np.int32)
    E.emu_result_iters(h, it.ctypes.data_as(C.POINTER(C.c_int32)), cl.ctypes.data_as(C.POINTER(C.c_int32)))
    E.emu_result_free(h)
    return r, it, cl


def emu_run_raw(items, params=None):
    """items: [(single-graph PackedGraphs, phases, max_group_boundary_distance), ...] as assembler::assemble(gx, px, sid) receives them;
    staged raw (HostBatch::add_graph_raw) and run through the single-lane emulation of the engine, whose load phase does the pre-steps
    -> (DecompResult, iterations[n])"""
    from aletsch_amd.native import GraphView, PhaseView
    E = emu_lib()
    E.emu_batch_new.restype = C.c_void_p
    E.emu_batch_free.argtypes = [C.c_void_p]; E.emu_batch_add_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    E.emu_batch_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    B = C.c_void_p(E.emu_batch_new())
    try:
        for pg, phases, dist in items:
            gv = GraphView.from_packed(pg, 0); pv = PhaseView.from_lists(phases)
            rc = E.emu_batch_add_raw(B, C.byref(gv), C.byref(pv), C.c_int32(dist))
            assert rc == 0, rc
        h = C.c_void_p()
        rc = E.emu_batch_run(B, C.byref(params) if params is not None else None, C.c_int32(0), C.c_int32(0), C.byref(h))
        assert rc == 0, rc
        r = export_via(E.emu_result_export, h, len(items))
        it = np.zeros(len(items), np.int32); cl = np.zeros(len(items), np.int32)
        E.emu_result_iters(h, it.ctypes.data_as(C.POINTER(C.c_int32)), cl.ctypes.data_as(C.POINTER(C.c_int32)))
        E.emu_result_free(h)
    finally:
        E.emu_batch_free(B)
    return r, it


def emu_transcripts(pg: PackedGraphs):
    """coverage + exon lists per path as the engine's records carry them (single-lane emulation) -> (DecompResult, cov, exon_offset, exons[.,2])"""
    E = emu_lib()
    h = C.c_void_p()
    assert E.emu_run_packed(*pg.c_args(), None, C.c_int32(0), C.c_int32(0), C.byref(h)) == 0
    r = export_via(E.emu_result_export, h, pg.n)
    te = C.c_int64()
    E.emu_result_export_transcripts(h, C.byref(te), None, None, None)
    P = len(r.weight)
    cov = np.zeros(P); eo = np.zeros(P + 1, np.int64); lr = np.zeros(2 * max(te.value, 1), np.int32)
    E.emu_result_export_transcripts(h, C.byref(te), cov.ctypes.data_as(C.POINTER(C.c_double)), eo.ctypes.data_as(C.POINTER(C.c_int64)), lr.ctypes.data_as(C.POINTER(C.c_int32)))
    E.emu_result_free(h)
    return r, cov, eo, lr[:2 * te.value].reshape(-1, 2)


def compare_results(want: DecompResult, got: DecompResult, n: int, tol: float = 0.0, conf_tol: float = 0.0):
    """Per graph: same status, same number of paths, identical vertex lists in identical order, exact length /
    count / strand; weight / abd / conf / reads bit-identical when tol == 0, else |d| <= tol * max(1, |x|)
    (north_star allows 1e-4).  conf = exp(sum of log(...)) goes through libm log/exp, which differ by an ulp between
    glibc and the device math library: conf_tol is its own relative tolerance (GPU tests use 1e-9).
    Returns the list of mismatches."""
    bad = []
    if not np.array_equal(want.status[:n], got.status[:n]):
        for g in np.nonzero(want.status[:n] != got.status[:n])[0][:5]:
            bad.append((int(g), "status", int(want.status[g]), int(got.status[g])))
        return bad
    if not np.array_equal(want.path_offset, got.path_offset):
        d = np.nonzero(np.diff(want.path_offset) != np.diff(got.path_offset))[0]
        for g in d[:5]:
            bad.append((int(g), "npaths", int(want.path_offset[g + 1] - want.path_offset[g]), int(got.path_offset[g + 1] - got.path_offset[g])))
        return bad
    if not np.array_equal(want.pv_offset, got.pv_offset) or not np.array_equal(want.path_vertices, got.path_vertices):
        for g in range(n):
            if want.paths_of(g) != got.paths_of(g) and [p["v"] for p in want.paths_of(g)] != [p["v"] for p in got.paths_of(g)]:
                bad.append((g, "vertices")); break
        return bad
    for name in ("length", "count", "strand"):
        if not np.array_equal(getattr(want, name), getattr(got, name)):
            bad.append((-1, name))
    for name in ("weight", "abd", "conf", "reads"):
        a, b = getattr(want, name), getattr(got, name)
        t = max(tol, conf_tol) if name == "conf" else tol
        if t == 0.0:
            ok = np.array_equal(a, b)
        else:
            ok = bool(np.all(np.abs(a - b) <= t * np.maximum(1.0, np.abs(a))))
        if not ok:
            i = int(np.argmax(np.abs(a - b)))
            g = int(np.searchsorted(want.path_offset, i, side="right") - 1)
            bad.append((g, name, float(a[i]), float(b[i])))
    return bad


# the synthetic configurations used across the parity tests (name -> ald_synth_spec fields)
PARITY_CONFIGS = {
    "cfg1_32v96e": dict(seed=1001, n_graphs=100, v_min=32, v_max=32, fixed_edges=96),
    "cfg2_64v256e": dict(seed=1002, n_graphs=120, v_min=64, v_max=64, fixed_edges=256),
    "tiny_8v16e": dict(seed=5, n_graphs=200, v_min=8, v_max=8, fixed_edges=16),
    "minimal_3v": dict(seed=6, n_graphs=50, v_min=3, v_max=5, edges_per_vertex=2),
    "cfg3_mixed": dict(seed=1003, n_graphs=40, v_min=8, v_max=300, edges_per_vertex=4),
    "int_weights": dict(seed=11, n_graphs=150, v_min=10, v_max=60, edges_per_vertex=3, weight_mode=1),
    "flow_weights": dict(seed=12, n_graphs=150, v_min=10, v_max=60, edges_per_vertex=3, weight_mode=2),
    "multi_sample": dict(seed=13, n_graphs=150, v_min=10, v_max=60, edges_per_vertex=3, n_samples=5),
    "stranded": dict(seed=14, n_graphs=150, v_min=10, v_max=60, edges_per_vertex=3, strand_mode=1, layout_mode=1),
    "phasing": dict(seed=15, n_graphs=150, v_min=10, v_max=60, edges_per_vertex=3, phasing_per_graph=12, weight_mode=2),
    "everything": dict(seed=16, n_graphs=150, v_min=6, v_max=100, edges_per_vertex=3, phasing_per_graph=20, weight_mode=1, n_samples=4, strand_mode=1, layout_mode=1),
    # the shape of real multi-sample data (tens of samples, phasing paths in the hundreds, edge counts that were ADDED along
    # grouped boundaries and so differ from the number of supporting samples, graph_reviser.cc:965-975)
    "real_shaped": dict(seed=17, n_graphs=60, v_min=20, v_max=120, edges_per_vertex=3, n_samples=30, phasing_per_graph=150, weight_mode=2, strand_mode=1, layout_mode=1,
                        _extra_counts=3),
}


def make_batch(name: str) -> PackedGraphs:
    """The synthetic batch of a PARITY_CONFIGS entry (keys starting with '_' are post-processing steps, not generator fields)."""
    import aletsch_amd as A
    kw = dict(PARITY_CONFIGS[name]); extra = kw.pop("_extra_counts", 0)
    pg = A.synth(**kw)
    if extra:
        rng = np.random.default_rng(kw["seed"])
        pg.edge_count = (pg.sample_counts() + rng.integers(0, extra + 1, pg.edge_target.size)).astype(np.int32)
    return pg


def transcript_stream_from_result(pg: PackedGraphs, res: DecompResult, sid=None, skip_single_exon: bool = False) -> np.ndarray:
    """Test-side restatement of ald_batch_transcript_stream (include/aletsch_decomp.h) over any DecompResult -- e.g. the single-lane
    emulation's, so that the CPU tier can drive the multi-rank exchange and the stream merge without a GPU.  Exons: the path's
    internal vertices' [lpos, rpos) with touching intervals joined (essential.cc:719-748)."""
    sl = pg.graph_slices(); out = []
    for g in range(pg.n):
        if int(res.status[g]) not in (0, 1):
            continue
        ov = int(sl["v"][g])
        for k, p in enumerate(range(int(res.path_offset[g]), int(res.path_offset[g + 1]))):
            v = res.path_vertices[int(res.pv_offset[p]):int(res.pv_offset[p + 1])]
            ex = []
            for x in v[1:-1]:
                l, r = int(pg.vertex_lpos[ov + x]), int(pg.vertex_rpos[ov + x])
                if l >= r:
                    continue
                if ex and ex[-1] == l:
                    ex[-1] = r
                else:
                    ex += [l, r]
            if len(ex) <= 2 and skip_single_exon:
                continue
            hdr = np.zeros(12, np.uint32)
            hdr[0] = g; hdr[1] = k; hdr[2] = np.array([-1 if sid is None else int(sid[g])], np.int32).view(np.uint32)[0]
            hdr[3] = int(res.strand[p]) & 0xFF; hdr[4] = int(res.count[p]); hdr[5] = len(ex) // 2
            hdr[6:12] = np.array([res.weight[p], res.conf[p], res.abd[p]], np.float64).view(np.uint32)
            out.append(hdr); out.append(np.array(ex, np.int32).view(np.uint32))
    return np.concatenate(out) if out else np.zeros(0, np.uint32)


def oracle_features(pg: PackedGraphs, extras=None):
    """scallop::update_trst_features restated in the oracle (unpinned: scallop.cc needs Boost/htslib to build) ->
    per graph (list of feature dicts, complete flags, asserted?); extras: list of aletsch_amd.GraphExtras, one per graph, or None"""
    from aletsch_amd.native import TrstFeatures, GraphExtras
    O = oracle_lib()
    arr = None
    if extras is not None:
        arr = (GraphExtras * pg.n)()
        for g, x in enumerate(extras):
            C.memmove(C.byref(arr[g]), C.byref(x), C.sizeof(GraphExtras))
        O.ora_set_extras(arr)
    h = C.c_void_p()
    assert O.ora_run_packed(*pg.c_args(), None, C.c_int32(1), C.c_int32(0), C.byref(h)) == 0
    r = export_via(O.ora_result_export, h, pg.n)
    out = []
    for g in range(pg.n):
        k = int(r.path_offset[g + 1] - r.path_offset[g])
        f = (TrstFeatures * max(k, 1))(); comp = np.zeros(max(k, 1), np.int32)
        bad = O.ora_result_features(h, g, C.byref(f), C.c_void_p(comp.ctypes.data))
        out.append(([f[i].as_dict() for i in range(k)], comp[:k].copy(), bool(bad)))
    O.ora_result_free(h)
    return r, out


def gene_like_raw(rng, n_runs=6, strand="+"):
    """A splice graph as assembler::assemble(gx, px, sid) receives it (before its pre-steps), plus a phase set in exon coordinates:
    exons cut into runs of TOUCHING partial exons (edges j -> j+1 inside a run), junctions between runs, several start / end boundaries
    close to each other (so that group_start/end_boundaries fold some), junctions that jump exactly over one partial exon (so that
    extend_strands lends a strand), multi-sample edge_info with counts; phases walk real paths, some start / end on boundaries that
    get grouped away, some are invalid (unknown coordinate, broken continuity).  -> (graph dict for PackedGraphs.from_graphs, phases)"""
    runs = []; pos = 1000; v = 1; lpos = [0]; rpos = [0]
    for r in range(n_runs):
        k = int(rng.integers(1, 5)); run = []
        for _ in range(k):
            ln = int(rng.integers(30, 400)); lpos.append(pos); rpos.append(pos + ln); pos += ln; run.append(v); v += 1
        runs.append(run); pos += int(rng.integers(50, 3000))
    V = v + 1; lpos.append(pos); rpos.append(pos); lpos[0] = rpos[0] = 1000
    ns = int(rng.integers(1, 6))

    def info(w):
        ids = sorted(set([0] + [int(x) for x in rng.integers(0, 8, ns)]))      # one sample supports every edge: intersections never go empty
        return {i: float(rng.integers(1, 30)) for i in ids}
    edges = []
    have = set()

    def add(s, t, w, st=0):
        if (s, t) in have or not (0 <= s < t < V):
            return
        have.add((s, t)); edges.append((s, t, float(w), st, info(w)))
    sd = {"+": 1, "-": 2, ".": 0}[strand]
    for run in runs:
        for a, b in zip(run, run[1:]):
            add(a, b, rng.integers(2, 60))
    for a, b in zip(runs, runs[1:]):                       # junctions between consecutive runs (+ a few skips)
        add(a[-1], b[0], rng.integers(2, 80), sd)
        if len(a) > 1 and rng.random() < 0.5:
            add(a[int(rng.integers(0, len(a)))], b[int(rng.integers(0, len(b)))], rng.integers(1, 30), sd)
    for i in range(len(runs) - 2):
        if rng.random() < 0.4:
            add(runs[i][-1], runs[i + 2][0], rng.integers(1, 25), sd)
    for run in runs:                                       # a junction over exactly one partial exon: s -> s+2 with s+1 filling the gap
        if len(run) >= 3 and rng.random() < 0.7:
            add(run[0], run[2], rng.integers(40, 90), sd)
    starts = set([runs[0][0]]) | set(int(x) for x in rng.choice(runs[0] + runs[1], size=min(3, len(runs[0] + runs[1])), replace=False))
    ends = set([runs[-1][-1]]) | set(int(x) for x in rng.choice(runs[-1] + runs[-2], size=min(3, len(runs[-1] + runs[-2])), replace=False))
    for s in sorted(starts):
        add(0, s, rng.integers(1, 40))
    for t in sorted(ends):
        add(t, V - 1, rng.integers(1, 40))
    order = rng.permutation(len(edges))                    # creation order of the caller's graph (pointer order in the reference)
    edges = [edges[i] for i in order]
    vw = [0.0] + [float(rng.integers(1, 50)) for _ in range(V - 2)] + [0.0]
    g = dict(V=V, edges=edges, vw=vw, lpos=lpos, rpos=rpos, strand=strand)
    # phases: exon-coordinate lists along forward walks of the graph
    adj = {}
    for s, t, *_ in edges:
        adj.setdefault(s, []).append(t)
    phases = []
    for _ in range(int(rng.integers(3, 14))):
        x = int(rng.choice(sorted(starts) + [r[0] for r in runs])); walk = [x]
        while x in adj and len(walk) < 12 and rng.random() < 0.85:
            nxt = [t for t in adj[x] if t != V - 1]
            if not nxt:
                break
            x = int(rng.choice(nxt)); walk.append(x)
        co = []
        for a in walk:                                     # merge touching consecutive vertices into exons
            if co and co[-1] == lpos[a]:
                co[-1] = rpos[a]
            else:
                co += [lpos[a], rpos[a]]
        kind = rng.random()
        if kind < 0.1:
            co[int(rng.integers(0, len(co)))] += 7          # a coordinate no vertex has
        elif kind < 0.15 and len(co) >= 4:
            co[1], co[2] = co[2], co[1]
        phases.append((co, int(rng.integers(1, 6))))
    return g, phases
