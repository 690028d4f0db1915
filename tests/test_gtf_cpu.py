"""CPU tier: the GTF / feature-table writers behind the C ABI (ald_gtf_format_transcript, ald_gtf_format_features) against the bytes
the REFERENCE's own writers emit (gtf/transcript.cc:318-494 compiled unmodified into oracle/_ref/ref_gtf; tests/golden/ref_gtf.json
made by tests/golden/make_golden.py): pinned parity, byte for byte.  Host-only code: no GPU involved."""
import json
import os

import aletsch_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_gtf.json")))


def _features(c):
    f = A.TrstFeatures()
    for k, v in c["features"].items():
        setattr(f, k, v)
    return f


def test_transcript_records_match_the_reference_writer():
    assert len(CASES) >= 100
    for c in CASES:
        got = A.format_transcript(c["seqname"], c["source"], c["gene_id"], c["transcript_id"], c["strand"], c["coverage"], [tuple(e) for e in c["exons"]],
                                  cov2=c["w_cov2"], count=c["w_count"], gene_type=c["gene_type"], transcript_type=c["transcript_type"])
        assert got == c["T"], (c["transcript_id"], got, c["T"])
    # optional attributes really were exercised both ways
    assert any('cov2 "' in c["T"] for c in CASES) and any('cov2 "' not in c["T"] for c in CASES)
    assert any('gene_type "' in c["T"] for c in CASES) and any('count "' not in c["T"] for c in CASES)


def test_feature_rows_match_the_reference_writer_in_both_forms():
    for c in CASES:
        f = _features(c)
        args = (c["transcript_id"], c["meta_tid"], c["seqname"], c["coverage"], c["cov2"], c["abd"], c["conf"], c["count1"], c["count2"], len(c["exons"]), f)
        assert A.format_features(*args) == c["F"], c["transcript_id"]                    # stream form: default ostream state (incubator.cc:781)
        assert A.format_features(*args, fixed2=True) == c["G"], c["transcript_id"]       # file form: fixed, 2 decimals (transcript.cc:430-434)


def test_snprintf_contract_and_ids():
    import ctypes as C
    lib = A.load_library()
    c = CASES[0]; want = c["T"].encode()
    import numpy as np
    lr = np.array(c["exons"], np.int32).reshape(-1)
    args = (c["seqname"].encode(), c["source"].encode(), c["gene_id"].encode(), c["transcript_id"].encode(), c["gene_type"].encode(), c["transcript_type"].encode(),
            c["strand"].encode(), c["coverage"], c["w_cov2"], c["w_count"], len(c["exons"]), C.c_void_p(lr.ctypes.data))
    assert lib.ald_gtf_format_transcript(None, 0, *args) == len(want)                   # size query
    small = C.create_string_buffer(b"\xff" * 40, 40)
    assert lib.ald_gtf_format_transcript(small, 17, *args) == len(want)                 # truncated, NUL-terminated, nothing past cap
    assert small.raw[:16] == want[:16] and small.raw[16] == 0 and small.raw[17:] == b"\xff" * 23
    assert A.format_transcript("1", "aletsch", "g", "t", "+", 1.0, []) == ""            # no exons: nothing is written (transcript.cc:323)
    assert A.transcript_id("1", "gene.4.0", 12) == "chr1.gene.4.0.12"                   # scallop.cc:3258
