#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

  ref_graph.json      scripts + dumps produced by oracle/_ref/ref_graph  (the REFERENCE's graph/*.cc, built from
                      /root/reference by oracle/Makefile)            -> pins the oracle's container orders
  ref_subsetsum.json  instances + answers produced by oracle/_ref/ref_subsetsum (the REFERENCE's subsetsum.cc)
                      -> pins the subset-sum restatement and the HIP subset-sum kernel
  oracle_paths.npz    small synthetic batches + the oracle's own decomposition (regression fixture: scallop / router /
                      hyper_set have no reference-authored vectors; "parity unpinned")

  ref_tset.json       transcript groups + the merged set produced by oracle/_ref/ref_tset (the REFERENCE's
                      rnacore/transcript_set.cc + gtf/transcript.cc)  -> pins the result sink (ald_tset_*)

  ref_router.json     one-vertex cases (edges in creation order, supporting samples, phasing routes) and what the REFERENCE's
                      scallop/router.cc makes of them (oracle/_ref/ref_router: type, degree, ratio, pe2w, edge confidences)
                      -> pins the oracle's Router (classify + thread + isolate attachment)

  ref_gtf.json        transcripts (+ feature blocks) and the bytes the REFERENCE's writers emit for them: transcript::write,
                      write_features(ostream), write_features(int) (gtf/transcript.cc:318-494, oracle/_ref/ref_gtf)
                      -> pins ald_gtf_format_transcript / ald_gtf_format_features

Needs /root/reference (for the _ref binaries); run from the repo root:  python tests/golden/make_golden.py
(`python tests/golden/make_golden.py gtf` / `router` regenerate ref_gtf.json / ref_router.json only)
"""
import json
import os
import random
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))


def graph_script(rng, directed, n, ops):
    s = [("D" if directed else "U") + " %d" % n]; live = []; h = 0
    for _ in range(ops):
        r = rng.random()
        if r < 0.55 or not live:
            a, b = rng.randrange(n), rng.randrange(n)
            if a == b:
                continue
            if a > b:
                a, b = b, a                       # forward edges: the splice graphs are DAGs
            s.append("a %d %d" % (a, b)); live.append(h); h += 1
        elif r < 0.70:
            k = rng.choice(live); live.remove(k); s.append("r %d" % k)
        elif r < 0.85 and directed:
            k = rng.choice(live); a, b = rng.randrange(n), rng.randrange(n)
            if a < b:
                s.append("m %d %d %d" % (k, a, b))
        else:
            s.append("q")
    s.append("q")
    return "\n".join(s) + "\n"


def tset_case(rng, n_groups, n_samples, n_chains):
    """Transcript groups with plenty of shared intron chains, single-exon overlaps and both strands."""
    chains = []
    for _ in range(n_chains):
        ne = rng.choice([1, 1, 2, 2, 3, 4, 6, 9]); x = rng.randrange(1000, 200000); ex = []
        for _k in range(ne):
            l = x + rng.randrange(20, 400); r = l + rng.randrange(30, 900); ex.append((l, r)); x = r
        chains.append((rng.choice("+-."), ex))
    groups = []; tid = 0
    for g in range(n_groups):
        sid = rng.randrange(n_samples); ts = []
        for _ in range(rng.randint(0, 6)):
            st, ex = chains[rng.randrange(n_chains)]; ex = list(ex)
            if len(ex) == 1:
                d = rng.choice([0, 0, 5, 40, 300, 2000]); ex[0] = (ex[0][0] + d, ex[0][1] + rng.choice([d, d + 3, 2 * d]))
            else:
                ex[0] = (ex[0][0] - rng.choice([0, 0, 7, 90]), ex[0][1]); ex[-1] = (ex[-1][0], ex[-1][1] + rng.choice([0, 0, 11, 250]))
            if rng.random() < 0.1:
                st = rng.choice("+-.")
            ts.append((st, round(rng.uniform(0.1, 30), 3), round(rng.random(), 4), round(rng.uniform(0, 50), 2), rng.randint(1, 9), tid, ex)); tid += 1
        groups.append((sid, ts))
    return groups


def tset_text(groups):
    s = ["%d" % len(groups)]
    for sid, ts in groups:
        s.append("%d %d" % (sid, len(ts)))
        for st, cov, conf, abd, c1, tid, ex in ts:
            s.append("%s %r %r %r %d %d %s %d" % (st, cov, conf, abd, c1, len(ex), " ".join("%d %d" % e for e in ex), tid))
    return "\n".join(s) + "\n"


def tset_parse(out):
    items = []
    for line in out.splitlines():
        a, b = line.split(" | "); f = a.split(); ne = int(f[9]); g = b.replace("{", "").replace("}", "").split(); ns = int(g[0])
        items.append({"hash": int(f[0]), "count": int(f[1]), "coverage": float(f[2]), "cov2": float(f[3]), "conf": float(f[4]), "abd": float(f[5]),
                      "count1": int(f[6]), "count2": int(f[7]), "tid": int(f[8]), "exons": [[int(f[10 + 2 * k]), int(f[11 + 2 * k])] for k in range(ne)],
                      "samples": [{"sid": int(g[1 + 7 * k]), "coverage": float(g[2 + 7 * k]), "cov2": float(g[3 + 7 * k]), "conf": float(g[4 + 7 * k]),
                                   "abd": float(g[5 + 7 * k]), "count1": int(g[6 + 7 * k]), "count2": int(g[7 + 7 * k])} for k in range(ns)]})
    return items


FEATURE_ORDER = ["gr_vertices", "gr_edges", "gr_reads", "gr_subgraph", "num_vertices", "num_edges", "junc_ratio", "max_mid_exon_len", "start_loss1", "start_loss2",
                 "start_loss3", "end_loss1", "end_loss2", "end_loss3", "start_merged_loss", "end_merged_loss", "introns", "start_introns", "end_introns", "intron_ratio",
                 "start_intron_ratio", "end_intron_ratio", "uni_junc", "seq_min_wt", "seq_min_cnt", "seq_min_abd", "seq_min_ratio", "seq_max_wt", "seq_max_cnt",
                 "seq_max_abd", "seq_max_ratio", "unbridge_start_coming_count", "unbridge_start_coming_ratio", "unbridge_end_leaving_count",
                 "unbridge_end_leaving_ratio", "start_cnt", "start_weight", "start_abd", "end_cnt", "end_weight", "end_abd"]    # transcript.h:60-104
INT_FEATURES = {"gr_vertices", "gr_edges", "gr_reads", "gr_subgraph", "num_vertices", "num_edges", "max_mid_exon_len", "introns", "start_introns", "end_introns", "uni_junc",
                "seq_min_cnt", "seq_max_cnt", "unbridge_start_coming_count", "unbridge_end_leaving_count", "start_cnt", "end_cnt"}


def gtf_cases(rng, n):
    """transcripts with every kind of number the writers meet: rounding ties at 4 / 2 decimals, six-significant-digit switches to
    exponent form, huge and tiny magnitudes, DBL_MAX / INT_MAX sentinels, zeros, omitted cov2 / count, optional gene / transcript types"""
    def num():
        k = rng.randrange(10)
        if k == 0: return 0.0
        if k == 1: return float(rng.randrange(1, 2000))
        if k == 2: return rng.choice([0.00005, 0.00015, 0.12345, 0.123449999, 2.675, 1.005, 0.125, 0.375, 999999.5, 1234567.0, 0.000123456, 1e-7, 123456.5])
        if k == 3: return 1.7976931348623157e308
        if k == 4: return rng.uniform(0, 1)
        if k == 5: return rng.uniform(0, 1e6)
        if k == 6: return 10.0 ** rng.uniform(-9, 12)
        return round(rng.uniform(0, 500), rng.randrange(0, 6))
    cases = []
    for t in range(n):
        ne = rng.choice([1, 2, 2, 3, 5, 12]); x = rng.randrange(0, 3000000); ex = []
        for _ in range(ne):
            l = x + rng.randrange(1, 5000); r = l + rng.randrange(1, 3000); ex.append([l, r]); x = r
        f = {k: (rng.choice([0, 1, 7, 2147483647, rng.randrange(0, 100000)]) if k in INT_FEATURES else num()) for k in FEATURE_ORDER}
        c = dict(seqname=rng.choice(["1", "chr1", "X", "GL000194.1"]), source="aletsch", gene_id=rng.choice(["gene.12.0", "g", "bundle.7.3.0"]),
                 transcript_id="chr1.gene.%d.%d" % (t, rng.randrange(40)), meta_tid=rng.choice(["chr1.m.%d" % t, "x"]),
                 gene_type=rng.choice(["", "", "protein_coding"]), transcript_type=rng.choice(["", "", "lncRNA"]), strand=rng.choice("+-."),
                 coverage=num(), cov2=num(), conf=num(), abd=num(), count1=rng.randrange(0, 50), count2=rng.randrange(0, 50),
                 w_cov2=rng.choice([-1.0, -1.0, num()]), w_count=rng.choice([-1, -1, 0, rng.randrange(1, 60)]), exons=ex, features=f)
        cases.append(c)
    return cases


def gtf_text(cases):
    out = ["%d" % len(cases)]
    e = lambda s: s if s else "-"
    for c in cases:
        out.append(" ".join([e(c["seqname"]), e(c["source"]), e(c["gene_id"]), e(c["transcript_id"]), e(c["meta_tid"]), e(c["gene_type"]), e(c["transcript_type"]), c["strand"],
                             repr(c["coverage"]), repr(c["cov2"]), repr(c["conf"]), repr(c["abd"]), str(c["count1"]), str(c["count2"]), repr(c["w_cov2"]), str(c["w_count"]),
                             str(len(c["exons"]))] + ["%d %d" % tuple(x) for x in c["exons"]]))
        out.append(" ".join(repr(c["features"][k]) for k in FEATURE_ORDER))
    return "\n".join(out) + "\n"


def make_gtf():
    import tempfile
    rng = random.Random(318360)
    cases = gtf_cases(rng, 120)
    with tempfile.TemporaryDirectory() as tmp:
        out = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_gtf"), tmp], input=gtf_text(cases), capture_output=True, text=True, check=True).stdout
    blocks = out.split("@T\n")[1:]
    assert len(blocks) == len(cases)
    for c, b in zip(cases, blocks):
        t, rest = b.split("@F\n"); f, g = rest.split("@G\n")
        c["T"], c["F"], c["G"] = t, f, g
    json.dump(cases, open(os.path.join(HERE, "ref_gtf.json"), "w"))
    return len(cases)


def router_case(rng):
    """One vertex with its in- and out-edges as scallop hands it to the router: 1..6 edges a side created in a random order (parallel
    edges included), weights with deliberate ties, 1..4 supporting samples per edge with sample 0 everywhere (so that every isolated
    node finds a partner: the reference asserts otherwise), a random set of phasing routes.  -> script text of oracle/ref_drivers/ref_router_main.cc"""
    nin = rng.choice([1, 2, 2, 2, 3, 3, 4, 5, 6]); nout = rng.choice([1, 2, 2, 2, 3, 3, 4, 5, 6])
    nfar_in = rng.randint(1, nin); nfar_out = rng.randint(1, nout)
    root = nfar_in; nv = nfar_in + 1 + nfar_out
    strand = rng.choice([0, 0, 0, 1, 2])
    wpool = [float(rng.randint(1, 40)) for _ in range(3)] + [rng.uniform(0.5, 90.0) for _ in range(3)]
    edges = []
    for i in range(nin):
        edges.append((rng.randrange(nfar_in) if i >= nfar_in else i, root))
    for j in range(nout):
        edges.append((root, root + 1 + (rng.randrange(nfar_out) if j >= nfar_out else j)))
    rng.shuffle(edges)                                           # creation order
    lines = []
    for s, t in edges:
        w = rng.choice(wpool) if rng.random() < 0.5 else rng.uniform(0.5, 120.0)
        smp = sorted(set([0] + [rng.randint(1, 4) for _ in range(rng.randint(0, 3))]))
        ab = [rng.choice([3.0, 7.5, 12.0]) if rng.random() < 0.4 else rng.uniform(0.2, 60.0) for _ in smp]
        lines.append("%d %d %r %d %d %d %s" % (s, t, w, strand if rng.random() < 0.7 else 0, rng.randint(1, 3), len(smp), " ".join("%d %r" % (a, b) for a, b in zip(smp, ab))))
    ins = [k for k, (s, t) in enumerate(edges) if t == root]; outs = [k for k, (s, t) in enumerate(edges) if s == root]
    routes = []
    dens = rng.choice([0.0, 0.0, 0.15, 0.4, 0.8])
    for a in ins:
        for b in outs:
            if rng.random() < dens:
                routes.append((a, b, rng.randint(1, 9)))
    head = "R %d %d %d %d %r" % (nv, root, len(edges), len(routes), rng.choice([0.01, 0.01, 0.5]))
    return "\n".join([head] + lines + ["%d %d %d" % r for r in routes]) + "\n"


def make_router(n_cases=240):
    rng = random.Random(73811)
    cases = []
    while len(cases) < n_cases:
        sc = router_case(rng)
        r = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_router")], input=sc, capture_output=True, text=True)      # one process per case: an assert aborts it
        if r.returncode != 0:
            continue
        out = "".join(ln + "\n" for ln in r.stdout.splitlines() if ln.startswith("@"))
        cases.append({"script": sc, "out": out})
    json.dump(cases, open(os.path.join(HERE, "ref_router.json"), "w"), indent=0)
    return len(cases)


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    if len(sys.argv) > 1 and sys.argv[1] == "gtf":
        print("ref_gtf.json:", make_gtf(), "transcripts")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "router":
        print("ref_router.json:", make_router(), "cases")
        return
    rng = random.Random(20250211)
    # ---- graph layer ----
    cases = []
    for t in range(24):
        sc = graph_script(rng, t % 3 != 2, 5 + t % 4, 40 + 5 * t)
        out = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_graph")], input=sc, capture_output=True, text=True, check=True).stdout
        cases.append({"script": sc, "dump": out})
    json.dump(cases, open(os.path.join(HERE, "ref_graph.json"), "w"), indent=0)
    # ---- subset sum: the reference's own KAT first (subsetsum.cc:263-282), then random instances ----
    inst = [([(10, 1), (20, 2), (39, 3)], [(29, 1), (54, 2)])]
    for t in range(200):
        ns, nt = rng.randint(1, 12), rng.randint(1, 12)
        hi = rng.choice([20, 100, 1000, 5000])
        inst.append(([(rng.randint(1, hi), i) for i in range(ns)], [(rng.randint(1, hi), 100 + i) for i in range(nt)]))
    ans = []
    for s, t in inst:               # one process per instance: the reference asserts (aborts) when no cross pair exists
        txt = "1\n%d %d\n" % (len(s), len(t)) + " ".join("%d %d" % p for p in s) + "\n" + " ".join("%d %d" % p for p in t) + "\n"
        r = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_subsetsum")], input=txt, capture_output=True, text=True)
        if r.returncode != 0 or not r.stdout.strip():
            ans.append(None); continue
        f = r.stdout.split(); e = float(f[0]); k = int(f[1]); ss = [int(x) for x in f[2:2 + k]]; m = int(f[2 + k]); tt = [int(x) for x in f[3 + k:3 + k + m]]
        ans.append({"e": e, "s": ss, "t": tt})
    json.dump({"instances": [{"s": s, "t": t} for s, t in inst], "answers": ans}, open(os.path.join(HERE, "ref_subsetsum.json"), "w"))
    # ---- result sink: the reference's transcript_set over random transcript groups ----
    tcases = []
    for t, (ng, nsmp, nch) in enumerate([(1, 1, 3), (6, 2, 4), (40, 3, 10), (120, 8, 25), (300, 20, 40), (60, 1, 6)]):
        groups = tset_case(rng, ng, nsmp, nch)
        out = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_tset")], input=tset_text(groups), capture_output=True, text=True, check=True).stdout
        tcases.append({"groups": groups, "items": tset_parse(out)})
    json.dump(tcases, open(os.path.join(HERE, "ref_tset.json"), "w"))
    make_gtf()
    make_router()
    # ---- oracle regression fixture ----
    import aletsch_amd as A
    import common
    from dataclasses import fields
    from aletsch_amd.packed import PackedGraphs
    blob = {}
    for name in ("cfg1_32v96e", "tiny_8v16e", "everything"):
        kw = dict(common.PARITY_CONFIGS[name]); kw["n_graphs"] = 12
        pg = A.synth(**kw)
        r = common.oracle_run(pg)[0]
        for f in fields(PackedGraphs):
            if getattr(pg, f.name) is not None:
                blob[f"{name}/in/{f.name}"] = getattr(pg, f.name)
        for k in ("status", "path_offset", "weight", "abd", "conf", "reads", "length", "count", "strand", "pv_offset", "path_vertices"):
            blob[f"{name}/out/{k}"] = getattr(r, k)
    np.savez_compressed(os.path.join(HERE, "oracle_paths.npz"), **blob)
    print("golden fixtures written:", len(cases), "graph scripts,", len(ans), "subset-sum instances")


if __name__ == "__main__":
    main()
