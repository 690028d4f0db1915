"""Result sink: ald_tset_* against the REFERENCE's transcript_set.cc.

tests/golden/ref_tset.json was produced by oracle/_ref/ref_tset (the reference's rnacore/transcript_set.cc +
gtf/transcript.cc built from source, oracle/Makefile) replaying meta/assembler.cc:1105-1133 over random transcript groups.
The sink is host code behind the C ABI, so it runs in the CPU tier; every field is compared bit for bit.
"""
import json
import os
import subprocess

import pytest

import aletsch_amd as A

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ref_tset.json")))


def as_groups(case):
    return [(sid, [(st, cov, conf, abd, c1, tid, [tuple(e) for e in ex]) for st, cov, conf, abd, c1, tid, ex in ts]) for sid, ts in case["groups"]]


def check(items, want):
    assert len(items) == len(want)
    for a, b in zip(items, want):
        for k in ("hash", "count", "coverage", "cov2", "conf", "abd", "count1", "count2", "tid"):
            assert a[k] == b[k], (k, a, b)
        assert [list(e) for e in a["exons"]] == b["exons"]
        assert len(a["samples"]) == len(b["samples"]) == a["count2"]
        for x, y in zip(a["samples"], b["samples"]):
            for k in ("sid", "cov2", "conf", "abd", "count1"):
                assert x[k] == y[k], (k, x, y)
            # the two derived per-sample fields of the reference (transcript_set.cc:70-74)
            assert y["coverage"] == a["coverage"] and y["count2"] == a["count2"]


@pytest.mark.parametrize("i", range(len(CASES)))
def test_sink_matches_reference_golden(i):
    s = A.TranscriptSink(0.8)
    s.add_groups(as_groups(CASES[i]))
    check(s.items(), CASES[i]["items"])


def test_sink_incremental_equals_one_call():
    """Feeding group by group (what the per-graph loop does) is the same as one call."""
    c = CASES[3]; g = as_groups(c)
    s = A.TranscriptSink(0.8)
    for x in g:
        s.add_groups([x])
    check(s.items(), c["items"])


def test_sink_skip_single_exon():
    c = CASES[2]; g = as_groups(c)
    s = A.TranscriptSink(0.8); s.add_groups(g, skip_single_exon=True)
    t = A.TranscriptSink(0.8); t.add_groups([(sid, [x for x in ts if len(x[6]) > 1]) for sid, ts in g])
    assert s.items() == t.items() and all(len(x["exons"]) > 1 for x in s.items())


def test_sink_merge_semantics_small():
    """Hand-checked against transcript_set.cc:38-75: multi-exon coverage adds and bounds widen; single exon keeps the max."""
    s = A.TranscriptSink(0.8)
    s.add_groups([(0, [("+", 3.5, 1.0, 10.0, 1, 0, [(100, 200), (300, 400), (500, 600)]), ("+", 2.0, 1.0, 5.0, 1, 1, [(1000, 1500)])]),
                  (1, [("+", 1.5, 0.5, 7.0, 1, 2, [(90, 200), (300, 400), (500, 650)]), ("+", 9.0, 1.0, 5.0, 1, 3, [(1000, 1600)])])])
    it = {x["tid"]: x for x in s.items()}
    assert set(it) == {0, 1}
    assert it[0]["coverage"] == 5.0 and it[0]["exons"] == [(90, 200), (300, 400), (500, 650)] and it[0]["count"] == 2 and it[0]["count2"] == 2
    assert it[1]["coverage"] == 9.0 and it[1]["exons"] == [(1000, 1600)] and it[1]["cov2"] == 9.0


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_tset")), reason="oracle/_ref/ref_tset not built")
def test_sink_matches_live_reference_build():
    import random
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    rng = random.Random(77)
    for ng, ns, nc in ((25, 2, 5), (200, 6, 30)):
        groups = mg.tset_case(rng, ng, ns, nc)
        out = subprocess.run([os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_tset")], input=mg.tset_text(groups), capture_output=True, text=True, check=True).stdout
        s = A.TranscriptSink(0.8); s.add_groups(groups)
        check(s.items(), mg.tset_parse(out))


def test_sink_containers_under_sanitizers():
    """the sink's own containers (small_vec: first elements inside their owner; chain_table: index over entries that never move) against
    standard-container models under AddressSanitizer + UBSan, object lifetimes counted; then the sink against a plain std::map / std::vector
    restatement of transcript_set::add / merge_sorted_trans_items on random transcripts incl. chains beyond the inline eight exons and
    items with many samples (tests/host_adapter/sink_containers_test.cc; C++11 as the reference builds)"""
    import subprocess
    out = os.path.join(HERE, "_build"); os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "sink_containers_test")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        os.path.join(HERE, "host_adapter", "sink_containers_test.cc"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sink containers ok" in r.stdout, r.stdout + r.stderr
