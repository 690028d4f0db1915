"""aletsch_amd -- MI355X-native per-bundle splice-graph decomposition (the "Scallop core" hot path of
Shao-Group/aletsch: reference meta/assembler.cc:1110-1111, scallop/scallop.cc:38-188).

The compute path is hand-written HIP for gfx950 behind a C ABI (include/aletsch_decomp.h,
aletsch_amd/lib/libaletsch_decomp.so).  This Python package is host plumbing for tests and the benchmark:
ctypes bindings, numpy views of the wire format, and the sharding helper used by bench.py.
There is no CPU compute fallback: without a HIP device every compute call raises.
"""
from .packed import PackedGraphs, DecompResult
from .native import (DecompBatch, DecompError, default_params, load_library, library_path, synth,
                     subsetsum_batch, decompose, SynthSpec, AldParams, TranscriptSink, records_add_graph_offset,
                     TrstFeatures, GraphExtras, FEATURE_FIELDS, format_transcript, format_features, transcript_id,
                     GraphView, PhaseView, pre_assemble, reduce_stream)

__all__ = ["PackedGraphs", "DecompResult", "DecompBatch", "DecompError", "default_params", "load_library",
           "library_path", "synth", "subsetsum_batch", "decompose", "SynthSpec", "AldParams", "TranscriptSink", "records_add_graph_offset",
           "TrstFeatures", "GraphExtras", "FEATURE_FIELDS", "format_transcript", "format_features", "transcript_id",
           "GraphView", "PhaseView", "pre_assemble", "reduce_stream"]
