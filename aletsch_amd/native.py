"""ctypes bindings of the C ABI (include/aletsch_decomp.h).

``DecompBatch`` mirrors the call shape the reference uses for one graph --
``scallop sx(gx, hx, cfg); sx.assemble(); sx.paths`` (meta/assembler.cc:1110-1121, scallop/scallop.h:31-51) --
lifted to a batch: ``add`` graphs, ``upload``, ``run``, ``download``, read ``paths``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .packed import PackedGraphs, DecompResult, export_via

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DecompError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"aletsch_decomp error {code}: {msg}")
        self.code = code


class AldParams(C.Structure):
    _fields_ = [("max_decompose_error_ratio", C.c_double * 8), ("min_guaranteed_edge_weight", C.c_double),
                ("min_transcript_coverage", C.c_double), ("max_num_exons", C.c_int32), ("reserved", C.c_int32)]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_graphs", C.c_int32), ("v_min", C.c_int32), ("v_max", C.c_int32),
                ("edges_per_vertex", C.c_int32), ("fixed_edges", C.c_int32), ("weight_mode", C.c_int32),
                ("n_samples", C.c_int32), ("phasing_per_graph", C.c_int32), ("strand_mode", C.c_int32),
                ("layout_mode", C.c_int32)]


_I, _D = C.c_int32, C.c_double
FEATURE_FIELDS = [("gr_vertices", _I), ("gr_edges", _I), ("gr_reads", _I), ("gr_subgraph", _I), ("num_vertices", _I), ("num_edges", _I), ("junc_ratio", _D),
                  ("max_mid_exon_len", _I), ("start_loss1", _D), ("start_loss2", _D), ("start_loss3", _D), ("end_loss1", _D), ("end_loss2", _D), ("end_loss3", _D),
                  ("start_merged_loss", _D), ("end_merged_loss", _D), ("introns", _I), ("start_introns", _I), ("end_introns", _I), ("intron_ratio", _D),
                  ("start_intron_ratio", _D), ("end_intron_ratio", _D), ("uni_junc", _I), ("seq_min_wt", _D), ("seq_min_cnt", _I), ("seq_min_abd", _D),
                  ("seq_min_ratio", _D), ("seq_max_wt", _D), ("seq_max_cnt", _I), ("seq_max_abd", _D), ("seq_max_ratio", _D),
                  ("unbridge_start_coming_count", _I), ("unbridge_start_coming_ratio", _D), ("unbridge_end_leaving_count", _I), ("unbridge_end_leaving_ratio", _D),
                  ("start_cnt", _I), ("start_weight", _D), ("start_abd", _D), ("end_cnt", _I), ("end_weight", _D), ("end_abd", _D)]


class TrstFeatures(C.Structure):
    """ald_trst_features == transcript::TrstFeatures (gtf/transcript.h:60-104)"""
    _fields_ = FEATURE_FIELDS

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in FEATURE_FIELDS}


class GraphExtras(C.Structure):
    """ald_graph_extras: what only the feature block reads (vertex_info's boundary / unbridged fields, splice_graph::reads / subgraph)"""
    _fields_ = [("boundary_loss1", C.POINTER(_D)), ("boundary_loss2", C.POINTER(_D)), ("boundary_loss3", C.POINTER(_D)), ("boundary_merged_loss", C.POINTER(_D)),
                ("unbridge_leaving_count", C.POINTER(_I)), ("unbridge_leaving_ratio", C.POINTER(_D)), ("unbridge_coming_count", C.POINTER(_I)),
                ("unbridge_coming_ratio", C.POINTER(_D)), ("gr_reads", _I), ("gr_subgraph", _I)]

    @classmethod
    def from_arrays(cls, gr_reads=0, gr_subgraph=0, **arrays):
        x = cls(); x.gr_reads = gr_reads; x.gr_subgraph = gr_subgraph; x._keep = []
        for name, a in arrays.items():
            t = dict(cls._fields_)[name]._type_
            a = np.ascontiguousarray(a, np.int32 if t is _I else np.float64); x._keep.append(a)
            setattr(x, name, a.ctypes.data_as(C.POINTER(t)))
        return x


class GraphView(C.Structure):
    """ald_graph_view (include/aletsch_decomp.h): one splice graph as caller-owned arrays"""
    _fields_ = [("num_vertices", _I), ("num_edges", _I), ("vertex_offset", C.POINTER(_I)), ("edge_target", C.POINTER(_I)), ("edge_weight", C.POINTER(_D)),
                ("edge_strand", C.POINTER(C.c_uint8)), ("edge_abd", C.POINTER(_D)), ("edge_sample_offset", C.POINTER(_I)), ("sample_id", C.POINTER(_I)),
                ("sample_abd", C.POINTER(_D)), ("vertex_weight", C.POINTER(_D)), ("vertex_lpos", C.POINTER(_I)), ("vertex_rpos", C.POINTER(_I)),
                ("vertex_type", C.POINTER(_I)), ("num_phasing", _I), ("phasing_offset", C.POINTER(_I)), ("phasing_vertex", C.POINTER(_I)),
                ("phasing_count", C.POINTER(_I)), ("strand", C.c_char), ("edge_count", C.POINTER(_I)), ("edge_creation_rank", C.POINTER(_I))]

    @classmethod
    def from_packed(cls, pg: PackedGraphs, g: int = 0):
        """view of graph g of a packed batch (the arrays stay alive with the returned object)"""
        sl = pg.graph_slices(); V, E, P = int(pg.g_nv[g]), int(pg.g_ne[g]), int(pg.g_np[g])
        v = cls(); v._keep = []

        def put(name, arr, dt, t):
            a = np.ascontiguousarray(arr, dt)
            if a.size == 0:
                a = np.zeros(1, dt)
            v._keep.append(a); setattr(v, name, a.ctypes.data_as(C.POINTER(t)))
        ov, ovo, oe, oeo, os_, op, opo, opv = (int(sl[k][g]) for k in ("v", "vo", "e", "eo", "s", "p", "po", "pv"))
        ns = int(sl["s"][g + 1] - sl["s"][g]); npv = int(sl["pv"][g + 1] - sl["pv"][g])
        v.num_vertices = V; v.num_edges = E; v.num_phasing = P; v.strand = bytes([int(pg.graph_strand[g]) & 0xFF])
        put("vertex_offset", pg.vertex_offset[ovo:ovo + V + 1], np.int32, _I); put("edge_target", pg.edge_target[oe:oe + E], np.int32, _I)
        put("edge_weight", pg.edge_weight[oe:oe + E], np.float64, _D); put("edge_strand", pg.edge_strand[oe:oe + E], np.uint8, C.c_uint8)
        put("edge_abd", pg.edge_abd[oe:oe + E], np.float64, _D); put("edge_sample_offset", pg.edge_sample_offset[oeo:oeo + E + 1], np.int32, _I)
        put("sample_id", pg.sample_id[os_:os_ + ns], np.int32, _I); put("sample_abd", pg.sample_abd[os_:os_ + ns], np.float64, _D)
        put("vertex_weight", pg.vertex_weight[ov:ov + V], np.float64, _D); put("vertex_lpos", pg.vertex_lpos[ov:ov + V], np.int32, _I)
        put("vertex_rpos", pg.vertex_rpos[ov:ov + V], np.int32, _I); put("vertex_type", pg.vertex_type[ov:ov + V], np.int32, _I)
        put("phasing_offset", pg.phasing_offset[opo:opo + P + 1], np.int32, _I); put("phasing_vertex", pg.phasing_vertex[opv:opv + npv], np.int32, _I)
        put("phasing_count", pg.phasing_count[op:op + P], np.int32, _I)
        if pg.edge_count is not None:
            put("edge_count", pg.edge_count[oe:oe + E], np.int32, _I)
        if pg.edge_rank is not None:
            put("edge_creation_rank", pg.edge_rank[oe:oe + E], np.int32, _I)
        return v

    def to_packed(self) -> PackedGraphs:
        """copy out as a single-graph batch"""
        V, E, P = self.num_vertices, self.num_edges, self.num_phasing

        def get(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt).copy() if n > 0 else np.zeros(0, dt)
        eso = get(self.edge_sample_offset, E + 1, np.int32); po = get(self.phasing_offset, P + 1, np.int32)
        ns = int(eso[E]) if E > 0 else 0; npv = int(po[P]) if P > 0 else 0
        return PackedGraphs(
            g_nv=np.array([V], np.int32), g_ne=np.array([E], np.int32), g_np=np.array([P], np.int32),
            vertex_offset=get(self.vertex_offset, V + 1, np.int32), edge_target=get(self.edge_target, E, np.int32), edge_weight=get(self.edge_weight, E, np.float64),
            edge_strand=get(self.edge_strand, E, np.uint8), edge_abd=get(self.edge_abd, E, np.float64), edge_sample_offset=eso,
            sample_id=get(self.sample_id, ns, np.int32), sample_abd=get(self.sample_abd, ns, np.float64), vertex_weight=get(self.vertex_weight, V, np.float64),
            vertex_lpos=get(self.vertex_lpos, V, np.int32), vertex_rpos=get(self.vertex_rpos, V, np.int32), vertex_type=get(self.vertex_type, V, np.int32),
            phasing_offset=po, phasing_vertex=get(self.phasing_vertex, npv, np.int32), phasing_count=get(self.phasing_count, P, np.int32),
            graph_strand=np.frombuffer(self.strand, np.int8).copy(), edge_count=get(self.edge_count, E, np.int32) if self.edge_count else None,
            edge_rank=get(self.edge_creation_rank, E, np.int32) if self.edge_creation_rank else None)


class PhaseView(C.Structure):
    """ald_phase_view: phase_set::pmap as flat arrays (exon-coordinate lists with counts)"""
    _fields_ = [("num_phases", _I), ("phase_offset", C.POINTER(_I)), ("phase_coord", C.POINTER(_I)), ("phase_count", C.POINTER(_I))]

    @classmethod
    def from_lists(cls, phases):
        """phases: [([l0, r0, l1, r1, ...], count), ...]"""
        v = cls(); off = [0]; co = []; cnt = []
        for coords, c in phases:
            co += list(coords); off.append(len(co)); cnt.append(c)
        v._keep = [np.array(off, np.int32), np.array(co if co else [0], np.int32), np.array(cnt if cnt else [0], np.int32)]
        v.num_phases = len(phases)
        v.phase_offset, v.phase_coord, v.phase_count = (a.ctypes.data_as(C.POINTER(_I)) for a in v._keep)
        return v


def pre_assemble(pg: PackedGraphs, phases, max_group_boundary_distance: int = 10000, g: int = 0, _lib=None, _prefix="ald"):
    """The pre-steps of assembler::assemble(gx, px, sid) (meta/assembler.cc:1075-1086) on graph g of `pg` and the phase set `phases`
    -> (single-graph PackedGraphs as scallop would receive it, smap pairs, tmap pairs, rc); rc > 0: the reference would have asserted."""
    lib = _lib or load_library()
    gv = GraphView.from_packed(pg, g); pv = PhaseView.from_lists(phases)
    h = C.c_void_p()
    rc = getattr(lib, _prefix + "_pre_assemble")(C.byref(gv), C.byref(pv), C.c_int32(max_group_boundary_distance), C.byref(h))
    if rc < 0:
        _check(rc)
    if rc > 0:
        return None, None, None, rc
    out = GraphView(); getattr(lib, _prefix + "_staged_view")(h, C.byref(out))
    back = out.to_packed()
    ns, nt = _I(), _I(); sp, tp = C.POINTER(_I)(), C.POINTER(_I)()
    getattr(lib, _prefix + "_staged_boundary_maps")(h, C.byref(ns), C.byref(sp), C.byref(nt), C.byref(tp))
    smap = [(sp[2 * i], sp[2 * i + 1]) for i in range(ns.value)]; tmap = [(tp[2 * i], tp[2 * i + 1]) for i in range(nt.value)]
    getattr(lib, _prefix + "_staged_free")(h)
    return back, smap, tmap, 0


class _ResultView(C.Structure):
    _fields_ = [("status", C.c_int32), ("num_paths", C.c_int32), ("num_iterations", C.c_int32), ("reserved", C.c_int32)]


def library_path() -> str:
    # ALETSCH_DECOMP_LIB selects a diagnostic build of the same HIP library (e.g. the -DALD_PROF variant); never a fallback
    return os.environ.get("ALETSCH_DECOMP_LIB") or os.path.join(_HERE, "lib", "libaletsch_decomp.so")


def load_library():
    """Load the in-tree HIP library; fail loudly if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise DecompError(-100, f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback for the decomposition path")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # only takes effect if the HIP runtime has not initialised yet
    lib = C.CDLL(path)
    lib.ald_last_error.restype = C.c_char_p
    lib.ald_version.restype = C.c_char_p
    lib.ald_batch_last_kernel_ms.restype = C.c_double
    lib.ald_batch_last_kernel_ms.argtypes = [C.c_void_p]
    for name in ("ald_batch_destroy", "ald_batch_clear", "ald_batch_upload", "ald_batch_run", "ald_batch_sync",
                 "ald_batch_download", "ald_batch_num_graphs"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.ald_batch_result_index.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int64))]
    lib.ald_batch_last_download_ms.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int64)]
    lib.ald_batch_enable_trace.argtypes = [C.c_void_p, C.c_int32]
    lib.ald_batch_device_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.ald_batch_device_transcript_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.ald_records_add_graph_offset.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    lib.ald_tset_destroy.argtypes = [C.c_void_p]
    lib.ald_tset_create.argtypes = [C.c_double, C.POINTER(C.c_void_p)]
    lib.ald_tset_add.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 10 + [C.c_int32]
    lib.ald_tset_add_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.ald_tset_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
    lib.ald_batch_transcript_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int64)]
    lib.ald_tset_add_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64]
    lib.ald_batch_export_iterations.argtypes = [C.c_void_p, C.c_void_p]
    lib.ald_batch_features.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ald_pre_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.ald_staged_view.argtypes = [C.c_void_p, C.c_void_p]
    lib.ald_staged_boundary_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ald_staged_free.argtypes = [C.c_void_p]
    lib.ald_batch_add_graph_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.ald_gtf_format_transcript.restype = C.c_int64
    lib.ald_gtf_format_transcript.argtypes = [C.c_char_p, C.c_int64] + [C.c_char_p] * 6 + [C.c_char, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p]
    lib.ald_gtf_format_features.restype = C.c_int64
    lib.ald_gtf_format_features.argtypes = [C.c_char_p, C.c_int64, C.c_int32] + [C.c_char_p] * 3 + [C.c_double] * 4 + [C.c_int32] * 3 + [C.c_void_p]
    lib.ald_transcript_id.restype = C.c_int64
    lib.ald_transcript_id.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int32]
    lib.ald_tset_export.argtypes = [C.c_void_p] * 19
    lib.ald_batch_reduce_transcripts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.POINTER(C.c_void_p)]
    lib.ald_tset_reduce_stream.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.POINTER(C.c_void_p)]
    lib.ald_tset_flat_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
    lib.ald_tset_flat_export.argtypes = [C.c_void_p] * 19
    lib.ald_tset_flat_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ald_tset_add_flat.argtypes = [C.c_void_p, C.c_void_p]
    lib.ald_tset_merge.argtypes = [C.c_void_p, C.c_void_p]
    lib.ald_tset_flat_free.argtypes = [C.c_void_p]
    _LIB = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise DecompError(rc, (load_library().ald_last_error() or b"").decode())


def default_params() -> AldParams:
    p = AldParams()
    _check(load_library().ald_default_params(C.byref(p)))
    return p


def synth(**kw) -> PackedGraphs:
    """Deterministic synthetic batch (SURVEY.md 8d generator; csrc/synth.cpp)."""
    lib = load_library()
    spec = SynthSpec(**kw)
    t = [C.c_int64() for _ in range(5)]
    _check(lib.ald_synth_sizes(C.byref(spec), *[C.byref(x) for x in t]))
    tv, te, ts, tp, tpv = [x.value for x in t]
    n = spec.n_graphs
    pg = PackedGraphs(
        g_nv=np.zeros(n, np.int32), g_ne=np.zeros(n, np.int32), g_np=np.zeros(n, np.int32),
        vertex_offset=np.zeros(tv + n, np.int32), edge_target=np.zeros(te, np.int32), edge_weight=np.zeros(te),
        edge_strand=np.zeros(te, np.uint8), edge_abd=np.zeros(te), edge_sample_offset=np.zeros(te + n, np.int32),
        sample_id=np.zeros(ts, np.int32), sample_abd=np.zeros(ts), vertex_weight=np.zeros(tv),
        vertex_lpos=np.zeros(tv, np.int32), vertex_rpos=np.zeros(tv, np.int32), vertex_type=np.zeros(tv, np.int32),
        phasing_offset=np.zeros(tp + n, np.int32), phasing_vertex=np.zeros(tpv, np.int32),
        phasing_count=np.zeros(tp, np.int32), graph_strand=np.zeros(n, np.int8))
    a = pg.c_args()
    _check(lib.ald_synth_fill(C.byref(spec), *a[1:]))
    return pg


class DecompBatch:
    """One batch of splice graphs on one MI355X (one HIP stream)."""

    def __init__(self, device: int = 0, params: Optional[AldParams] = None, trace_events: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.ald_batch_create(C.byref(params) if params is not None else None, C.c_int(device), C.byref(self._h)))
        if trace_events:
            _check(self._lib.ald_batch_enable_trace(self._h, trace_events))
        self._keep = []

    def close(self):
        if self._h:
            self._lib.ald_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def n(self) -> int:
        return int(self._lib.ald_batch_num_graphs(self._h))

    def clear(self):
        _check(self._lib.ald_batch_clear(self._h))

    def add(self, pg: PackedGraphs):
        _check(self._lib.ald_batch_add_packed(self._h, *pg.c_args()))

    def add_packed_raw(self, pg: PackedGraphs, max_group_boundary_distance: int = 10000, phases=None):
        """every graph of `pg` as a RAW graph (pre-steps on the device): ald_batch_add_packed_raw; phases: per graph a list of
        ([l0, r0, l1, r1, ...], count) or None"""
        n = pg.n
        dist = np.full(n, max_group_boundary_distance, np.int32)
        nph = np.zeros(n, np.int32); off = []; co = []; cnt = []
        for g in range(n):
            ph = phases[g] if phases is not None else []
            nph[g] = len(ph); off.append(0); base = len(co)
            for coords, c in ph:
                co += list(coords); off.append(len(co) - base); cnt.append(c)
        off = np.array(off, np.int32); co = np.array(co if co else [0], np.int32); cnt = np.array(cnt if cnt else [0], np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        _check(self._lib.ald_batch_add_packed_raw(self._h, *pg.c_args(), p(dist), p(nph), p(off), p(co), p(cnt)))

    def add_raw(self, pg: PackedGraphs, phases, max_group_boundary_distance: int = 10000, g: int = 0) -> int:
        """assemble(gx, px, sid) as received: ald_batch_add_graph_raw stages graph g of `pg` raw, with its phases; the pre-steps run on the
        device.  Returns 0 (a graph on which the reference would have asserted in them ends with an invariant status)."""
        gv = GraphView.from_packed(pg, g); pv = PhaseView.from_lists(phases)
        rc = self._lib.ald_batch_add_graph_raw(self._h, C.byref(gv), C.byref(pv), C.c_int32(max_group_boundary_distance))
        if rc < 0:
            _check(rc)
        return rc

    def upload(self):
        _check(self._lib.ald_batch_upload(self._h))

    def run(self):
        _check(self._lib.ald_batch_run(self._h))

    def sync(self):
        _check(self._lib.ald_batch_sync(self._h))

    def download(self):
        _check(self._lib.ald_batch_download(self._h))

    def kernel_ms(self) -> float:
        return float(self._lib.ald_batch_last_kernel_ms(self._h))

    def algorithmic_bytes(self):
        a = C.c_int64(); b = C.c_int64()
        _check(self._lib.ald_batch_algorithmic_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def raw_records(self) -> np.ndarray:
        """Packed path-record stream of the last download (uint32 words; copy)."""
        w = C.POINTER(C.c_uint32)(); n = C.c_int64()
        _check(self._lib.ald_batch_raw_records(self._h, C.byref(w), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(w, shape=(n.value,)).copy()

    def result_index(self):
        """(index[n_paths_total] uint64, graph_first[n] int64): the kernel-written index into raw_records() (copies)"""
        ip = C.POINTER(C.c_uint64)(); n = C.c_int64(); gp = C.POINTER(C.c_int64)()
        _check(self._lib.ald_batch_result_index(self._h, C.byref(ip), C.byref(n), C.byref(gp)))
        idx = np.ctypeslib.as_array(ip, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)
        gf = np.ctypeslib.as_array(gp, shape=(self.n,)).copy() if self.n else np.zeros(0, np.int64)
        return idx, gf

    def download_ms(self):
        """stages of the last download in host milliseconds + bytes moved to the host (diagnostic)"""
        v = [C.c_double() for _ in range(4)]; nb = C.c_int64()
        _check(self._lib.ald_batch_last_download_ms(self._h, *[C.byref(x) for x in v], C.byref(nb)))
        return dict(wait_kernel=v[0].value, status_retries=v[1].value, copy=v[2].value, decode=v[3].value, bytes_to_host=nb.value)

    def device_records(self):
        """(device pointer, word count) of the record stream in HBM -- for a zero-copy hand-over to RCCL (distributed.py)."""
        p = C.c_void_p(); n = C.c_int64()
        _check(self._lib.ald_batch_device_records(self._h, C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def device_transcript_stream(self, sid=None, skip_single_exon: bool = False):
        """(device pointer, word count) of the finished-transcript stream built ON the device (== transcript_stream(), word for word):
        for a zero-copy hand-over to the exchange step (distributed.StreamGatherer / ald_comm_gather_streams)."""
        p = C.c_void_p(); n = C.c_int64()
        sp = None
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int32); assert len(sid) == self.n; sp = C.c_void_p(sid.ctypes.data)
        _check(self._lib.ald_batch_device_transcript_stream(self._h, sp, C.c_int32(int(skip_single_exon)), C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def result(self) -> DecompResult:
        return export_via(self._lib.ald_batch_export, self._h, self.n, check=_check)

    def transcripts(self):
        """(coverage[paths], exon_offset[paths+1], exons[n,2]): scallop::build_transcripts without the ML features."""
        te = C.c_int64()
        _check(self._lib.ald_batch_export_transcripts(self._h, C.byref(te), None, None, None))
        P = len(self.result().weight)
        cov = np.zeros(P); eo = np.zeros(P + 1, np.int64); lr = np.zeros(2 * max(te.value, 1), np.int32)
        _check(self._lib.ald_batch_export_transcripts(self._h, C.byref(te), cov.ctypes.data_as(C.POINTER(C.c_double)),
                                                      eo.ctypes.data_as(C.POINTER(C.c_int64)), lr.ctypes.data_as(C.POINTER(C.c_int32))))
        return cov, eo, lr[:2 * te.value].reshape(-1, 2)

    def iterations(self) -> np.ndarray:
        """main-loop rule firings per graph (the `steps` of SURVEY.md 8d's steps/s/CU)"""
        out = np.zeros(max(self.n, 1), np.int32)
        _check(self._lib.ald_batch_export_iterations(self._h, C.c_void_p(out.ctypes.data)))
        return out[:self.n]

    def features(self, graph: int, extras: Optional["GraphExtras"] = None):
        """-> (features: list of TrstFeatures, complete: int32 array, rc): scallop::update_trst_features for every path of `graph`;
        rc > 0 where the reference would have asserted."""
        rv = _ResultView(); _check(self._lib.ald_batch_get_result(self._h, graph, C.byref(rv)))
        k = max(rv.num_paths, 1)
        f = (TrstFeatures * k)(); comp = np.zeros(k, np.int32)
        rc = self._lib.ald_batch_features(self._h, graph, C.byref(extras) if extras is not None else None, C.byref(f), C.c_void_p(comp.ctypes.data))
        if rc < 0:
            _check(rc)
        return [f[i] for i in range(rv.num_paths)], comp[:rv.num_paths], rc

    def reduce_transcripts(self, sid=None, tid_base: int = 0, skip_single_exon: bool = False, single_exon_overlap: float = 0.8, into: Optional["TranscriptSink"] = None):
        """The batch's transcripts merged into an EMPTY set on the GPU (ald_batch_reduce_transcripts) -> (items in TranscriptSink.items()
        form, stats dict); `into`: also merge the reduced set into that persistent sink (ald_tset_add_flat)."""
        sp = None
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int32); assert len(sid) == self.n; sp = C.c_void_p(sid.ctypes.data)
        h = C.c_void_p()
        _check(self._lib.ald_batch_reduce_transcripts(self._h, sp, C.c_int64(tid_base), C.c_int32(int(skip_single_exon)), C.c_double(single_exon_overlap), C.byref(h)))
        try:
            if into is not None:
                _check(self._lib.ald_tset_add_flat(into._h, h))
            st = [C.c_double(), C.c_double(), C.c_int64(), C.c_int64()]
            _check(self._lib.ald_tset_flat_stats(h, *[C.byref(x) for x in st]))
            items = _export_items(lambda *a: self._lib.ald_tset_flat_size(h, *a), lambda *a: self._lib.ald_tset_flat_export(h, *a))
        finally:
            self._lib.ald_tset_flat_free(h)
        return items, dict(device_ms=st[0].value, total_ms=st[1].value, device_groups=st[2].value, host_items=st[3].value)

    def reduce_into(self, sink: "TranscriptSink", sid=None, tid_base: int = 0, skip_single_exon: bool = False, single_exon_overlap: float = 0.8):
        """reduce_transcripts + ald_tset_add_flat without exporting the items to Python: the merge step of a pipelined caller -> stats dict"""
        sp = None
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int32); assert len(sid) == self.n; sp = C.c_void_p(sid.ctypes.data)
        h = C.c_void_p()
        _check(self._lib.ald_batch_reduce_transcripts(self._h, sp, C.c_int64(tid_base), C.c_int32(int(skip_single_exon)), C.c_double(single_exon_overlap), C.byref(h)))
        try:
            _check(self._lib.ald_tset_add_flat(sink._h, h))
            st = [C.c_double(), C.c_double(), C.c_int64(), C.c_int64()]
            _check(self._lib.ald_tset_flat_stats(h, *[C.byref(x) for x in st]))
        finally:
            self._lib.ald_tset_flat_free(h)
        return dict(device_ms=st[0].value, total_ms=st[1].value, device_groups=st[2].value, host_items=st[3].value)

    def transcript_stream(self, sid=None, skip_single_exon: bool = False) -> np.ndarray:
        """Finished transcripts of the downloaded batch as one self-contained uint32 stream (copy): what ranks exchange in the
        multi-GPU gather and what TranscriptSink.add_stream merges."""
        sp = None
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int32); assert len(sid) == self.n; sp = C.c_void_p(sid.ctypes.data)
        w = C.POINTER(C.c_uint32)(); n = C.c_int64()
        _check(self._lib.ald_batch_transcript_stream(self._h, sp, C.c_int32(int(skip_single_exon)), C.byref(w), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(w, shape=(n.value,)).copy()

    def trace(self, g: int):
        n = C.c_int32(); codes = C.POINTER(C.c_int32)(); vals = C.POINTER(C.c_double)()
        _check(self._lib.ald_batch_get_trace(self._h, g, C.byref(n), C.byref(codes), C.byref(vals)))
        return [(codes[3 * i], codes[3 * i + 1], codes[3 * i + 2], vals[i]) for i in range(n.value)]

    def class_info(self, cls: int):
        v = [C.c_int32() for _ in range(4)]; sb = C.c_int64(); ng = C.c_int32()
        _check(self._lib.ald_batch_class_info(self._h, cls, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), C.byref(sb), C.byref(ng)))
        return dict(maxv=v[0].value, maxe=v[1].value, blocks_per_cu=v[2].value, blocks_last_run=v[3].value, slab_bytes=sb.value, n_graphs=ng.value)


def _export_items(size_fn, export_fn):
    """drive an (ald_tset_size, ald_tset_export)-shaped pair -> list of dicts"""
    n = C.c_int64(); ne = C.c_int64(); ns = C.c_int64()
    _check(size_fn(C.byref(n), C.byref(ne), C.byref(ns)))
    n, ne, ns = n.value, ne.value, ns.value
    z = lambda k, dt: np.zeros(max(k, 1), dt)
    h = z(n, np.uint64); cnt = z(n, np.int32); st = z(n, np.int8); cov = z(n, np.float64); cov2 = z(n, np.float64); conf = z(n, np.float64); abd = z(n, np.float64)
    c1 = z(n, np.int32); c2 = z(n, np.int32); tid = z(n, np.int64); eo = z(n + 1, np.int64); lr = z(2 * ne, np.int32)
    so = z(n + 1, np.int64); ssid = z(ns, np.int32); scov2 = z(ns, np.float64); sconf = z(ns, np.float64); sabd = z(ns, np.float64); sc1 = z(ns, np.int32)
    _check(export_fn(*[C.c_void_p(x.ctypes.data) for x in (h, cnt, st, cov, cov2, conf, abd, c1, c2, tid, eo, lr, so, ssid, scov2, sconf, sabd, sc1)]))
    out = []
    for i in range(n):
        out.append(dict(hash=int(h[i]), count=int(cnt[i]), strand=chr(st[i]), coverage=float(cov[i]), cov2=float(cov2[i]), conf=float(conf[i]), abd=float(abd[i]),
                        count1=int(c1[i]), count2=int(c2[i]), tid=int(tid[i]), exons=[(int(lr[2 * k]), int(lr[2 * k + 1])) for k in range(eo[i], eo[i + 1])],
                        samples=[dict(sid=int(ssid[k]), cov2=float(scov2[k]), conf=float(sconf[k]), abd=float(sabd[k]), count1=int(sc1[k])) for k in range(so[i], so[i + 1])]))
    return out


class TranscriptSink:
    """The result sink (transcript_set::add / trans_item::merge, rnacore/transcript_set.cc:38-175) behind ald_tset_*."""

    def __init__(self, single_exon_overlap: float = 0.8):
        self._lib = load_library(); self._h = C.c_void_p()
        _check(self._lib.ald_tset_create(C.c_double(single_exon_overlap), C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.ald_tset_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_groups(self, groups, skip_single_exon: bool = False):
        """groups: list of (sid, [ (strand, coverage, conf, abd, count1, tid, [(l, r), ...]), ... ]) -- one group per graph."""
        go = [0]; sid = []; st = []; cov = []; conf = []; abd = []; c1 = []; tid = []; eo = [0]; lr = []
        for s, ts in groups:
            sid.append(s)
            for t in ts:
                st.append(ord(t[0])); cov.append(t[1]); conf.append(t[2]); abd.append(t[3]); c1.append(t[4]); tid.append(t[5])
                for l, r in t[6]:
                    lr += [l, r]
                eo.append(len(lr) // 2)
            go.append(len(st))
        a = lambda x, dt: np.ascontiguousarray(np.array(x, dt) if len(x) else np.zeros(1, dt))
        go, sid, st, cov, conf, abd, c1, tid, eo, lr = (a(go, np.int64), a(sid, np.int32), a(st, np.int8), a(cov, np.float64), a(conf, np.float64),
                                                        a(abd, np.float64), a(c1, np.int32), a(tid, np.int64), a(eo, np.int64), a(lr, np.int32))
        _check(self._lib.ald_tset_add(self._h, C.c_int32(len(groups)), C.c_void_p(go.ctypes.data), C.c_void_p(sid.ctypes.data), C.c_void_p(st.ctypes.data),
                                      C.c_void_p(cov.ctypes.data), C.c_void_p(conf.ctypes.data), C.c_void_p(abd.ctypes.data), C.c_void_p(c1.ctypes.data),
                                      C.c_void_p(tid.ctypes.data), C.c_void_p(eo.ctypes.data), C.c_void_p(lr.ctypes.data), C.c_int32(int(skip_single_exon))))

    def add_batch(self, batch: "DecompBatch", sid=None, tid_base: int = 0, skip_single_exon: bool = False):
        """Every transcript of a downloaded batch, graph by graph (meta/assembler.cc:1105-1133)."""
        sp = None
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int32); assert len(sid) == batch.n; sp = C.c_void_p(sid.ctypes.data)
        _check(self._lib.ald_tset_add_batch(self._h, batch._h, sp, C.c_int64(tid_base), C.c_int32(int(skip_single_exon))))

    def merge(self, other: "TranscriptSink"):
        """transcript_set::add(transcript_set&): every bucket of `other` zipped into this set; `other` is left empty."""
        _check(self._lib.ald_tset_merge(self._h, other._h))

    def add_stream(self, words: np.ndarray, graph_offset: int = 0, tid_base: int = 0):
        """Merge a transcript stream (DecompBatch.transcript_stream, possibly gathered from another rank) graph by graph."""
        words = np.ascontiguousarray(words, np.uint32)
        _check(self._lib.ald_tset_add_stream(self._h, C.c_void_p(words.ctypes.data), C.c_int64(words.size), C.c_int32(int(graph_offset)), C.c_int64(int(tid_base))))

    def items(self):
        """List of dicts in the reference's iteration order (bucket hash ascending, then bucket order)."""
        return _export_items(lambda *a: self._lib.ald_tset_size(self._h, *a), lambda *a: self._lib.ald_tset_export(self._h, *a))


def reduce_stream(words: np.ndarray, coverage=None, tid=None, tid_base: int = 0, skip_single_exon: bool = False, single_exon_overlap: float = 0.8, device: int = 0):
    """A transcript stream merged into an EMPTY set by the GPU reduction (ald_tset_reduce_stream) -> (items, stats)"""
    lib = load_library()
    words = np.ascontiguousarray(words, np.uint32)
    cp = tp = None
    if coverage is not None:
        coverage = np.ascontiguousarray(coverage, np.float64); cp = C.c_void_p(coverage.ctypes.data)
    if tid is not None:
        tid = np.ascontiguousarray(tid, np.int64); tp = C.c_void_p(tid.ctypes.data)
    h = C.c_void_p()
    _check(lib.ald_tset_reduce_stream(C.c_int32(device), C.c_void_p(words.ctypes.data), C.c_int64(words.size), cp, tp, C.c_int64(tid_base), C.c_int32(int(skip_single_exon)),
                                      C.c_double(single_exon_overlap), C.byref(h)))
    try:
        st = [C.c_double(), C.c_double(), C.c_int64(), C.c_int64()]
        _check(lib.ald_tset_flat_stats(h, *[C.byref(x) for x in st]))
        items = _export_items(lambda *a: lib.ald_tset_flat_size(h, *a), lambda *a: lib.ald_tset_flat_export(h, *a))
    finally:
        lib.ald_tset_flat_free(h)
    return items, dict(device_ms=st[0].value, total_ms=st[1].value, device_groups=st[2].value, host_items=st[3].value)


def _two_pass(fn, *args) -> str:
    n = fn(None, 0, *args)
    buf = C.create_string_buffer(int(n) + 1)
    fn(buf, int(n) + 1, *args)
    return buf.value.decode()


def format_transcript(seqname, source, gene_id, transcript_id, strand, coverage, exons, cov2=-1.0, count=-1, gene_type="", transcript_type="") -> str:
    """transcript::write (gtf/transcript.cc:318-360): the GTF records of one transcript; exons = [(l, r), ...]"""
    lr = np.ascontiguousarray(np.array(exons, np.int32).reshape(-1))
    return _two_pass(load_library().ald_gtf_format_transcript, seqname.encode(), source.encode(), gene_id.encode(), transcript_id.encode(), gene_type.encode(),
                     transcript_type.encode(), strand.encode(), float(coverage), float(cov2), int(count), len(exons), C.c_void_p(lr.ctypes.data) if len(exons) else None)


def format_features(transcript_id, meta_tid, seqname, coverage, cov2, abd, conf, count1, count2, n_exons, features: "TrstFeatures", fixed2=False) -> str:
    """transcript::write_features (gtf/transcript.cc:362-494): one row of *.trstFeature.csv; fixed2 = the file form (2 decimals)"""
    return _two_pass(load_library().ald_gtf_format_features, int(bool(fixed2)), transcript_id.encode(), meta_tid.encode(), seqname.encode(), float(coverage), float(cov2),
                     float(abd), float(conf), int(count1), int(count2), int(n_exons), C.byref(features))


def transcript_id(chrm: str, gid: str, path_index: int) -> str:
    return _two_pass(load_library().ald_transcript_id, chrm.encode(), gid.encode(), int(path_index))


def records_add_graph_offset(words: np.ndarray, graph_offset: int):
    """In place: make the graph ids of a record stream global (C loop over the variable-length records)."""
    assert words.dtype == np.uint32 and words.flags.c_contiguous and words.flags.writeable
    _check(load_library().ald_records_add_graph_offset(C.c_void_p(words.ctypes.data), C.c_int64(words.size), C.c_int32(int(graph_offset))))
    return words


def decompose(pg: PackedGraphs, device: int = 0, params: Optional[AldParams] = None) -> DecompResult:
    """Convenience: stage, upload, run, download one batch."""
    with DecompBatch(device, params) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        return b.result()


def subsetsum_batch(instances, device: int = 0):
    """instances: list of (source [(value,label)...], target [(value,label)...]).  Returns list of
    (error, S labels, T labels) or None for an instance the reference would assert on (scallop/subsetsum.cc)."""
    lib = load_library()
    n = len(instances)
    ns = np.array([len(s) for s, _ in instances], np.int32); nt = np.array([len(t) for _, t in instances], np.int32)
    sv = np.array([v for s, _ in instances for v, _l in s], np.int32); sl = np.array([l for s, _ in instances for _v, l in s], np.int32)
    tv = np.array([v for _, t in instances for v, _l in t], np.int32); tl = np.array([l for _, t in instances for _v, l in t], np.int32)
    err = np.zeros(n); ons = np.zeros(n, np.int32); ont = np.zeros(n, np.int32); os_ = np.zeros(64 * n, np.int32); ot = np.zeros(64 * n, np.int32)

    def p(a, t):
        return a.ctypes.data_as(C.POINTER(t))
    _check(lib.ald_subsetsum_batch(C.c_int(device), C.c_int32(n), p(ns, C.c_int32), p(nt, C.c_int32), p(sv, C.c_int32), p(sl, C.c_int32),
                                   p(tv, C.c_int32), p(tl, C.c_int32), p(err, C.c_double), p(ons, C.c_int32), p(ont, C.c_int32),
                                   p(os_, C.c_int32), p(ot, C.c_int32)))
    out = []
    for i in range(n):
        if ons[i] < 0:
            out.append(None)
        else:
            out.append((float(err[i]), os_[64 * i:64 * i + ons[i]].tolist(), ot[64 * i:64 * i + ont[i]].tolist()))
    return out
