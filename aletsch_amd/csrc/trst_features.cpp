// trst_features.cpp -- host side of the ABI: the per-transcript feature block and the GTF / feature-table writers.
//
//   ald_batch_features          scallop::update_trst_features (scallop/scallop.cc:3268-3451) + unique_junc (:3472-3497)
//   ald_gtf_format_transcript   transcript::write           (gtf/transcript.cc:318-360)
//   ald_gtf_format_features     transcript::write_features  (gtf/transcript.cc:362-494, both forms)
//
// The features are read from the graph exactly as it was staged -- the reference reads them from `gr_ori`, the copy it takes before
// the decomposition starts (scallop.cc:42,179) -- through the batch's host-side CSR: edge(s, t) is the NEWEST parallel edge
// (directed_graph.cc:60-76), get_out_weights / get_in_weights add in adjacency order (splice_graph.cc:174-198).
#include "ald_internal.h"
#include <cfloat>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <utility>

namespace {

struct GraphRO {                       // read-only view of one staged graph
    const HostBatch &hb; int64_t ov, ovo, oe; int V, E;
    GraphRO(const HostBatch &h, int g) : hb(h), ov(h.off_v[g]), ovo(h.off_v[g] + g), oe(h.off_e[g]), V(h.g_nv[g]), E(h.g_ne[g]) {}
    int lpos(int v) const { return hb.vertex_lpos[ov + v]; }
    int rpos(int v) const { return hb.vertex_rpos[ov + v]; }
    int edge(int s, int t) const {     // newest parallel edge s -> t, or -1: rows are sorted by (target, creation rank)
        if(s < 0 || s >= V) return -1;
        int best = -1;
        for(int k = hb.vertex_offset[ovo + s]; k < hb.vertex_offset[ovo + s + 1]; k++) { const int tt = hb.edge_target[oe + k]; if(tt == t) best = k; else if(tt > t) break; }
        return best;
    }
    double w(int e) const { return hb.edge_weight[oe + e]; }
    int cnt(int e) const { return hb.edge_count[oe + e]; }
    double abd(int e) const { return hb.edge_abd[oe + e]; }
    double out_weights(int v) const { double s = 0; for(int k = hb.vertex_offset[ovo + v]; k < hb.vertex_offset[ovo + v + 1]; k++) s += hb.edge_weight[oe + k]; return s; }
    double in_weights(int v) const { double s = 0; for(int k = hb.in_offset[ovo + v]; k < hb.in_offset[ovo + v + 1]; k++) s += hb.edge_weight[oe + hb.in_edge[oe + k]]; return s; }
};

typedef std::pair<int, int> Junc;
// path::junc (scallop.cc:2812-2820): consecutive INTERNAL vertices that do not touch
void junctions(const GraphRO &G, const uint32_t *v, int n, std::vector<Junc> &out)
{
    out.clear();
    for(int i = 2; i + 1 < n; i++) if(G.lpos((int)v[i]) != G.rpos((int)v[i - 1])) out.push_back(Junc((int)v[i - 1], (int)v[i]));
}

} // namespace

extern "C" {

int ald_batch_features(const ald_batch *b, int32_t graph, const ald_graph_extras *X, ald_trst_features *features, int32_t *complete)
{
    if(!b || !features || graph < 0 || graph >= b->hb.n()) return ALD_ERR_INVALID;
    if(!b->downloaded) return ald_set_err(ALD_ERR_STATE, "ald_batch_features before ald_batch_download");
    { int rc = ald_ensure_index(b); if(rc != ALD_OK) return rc; }
    // gr_ori is the graph as scallop received it, i.e. AFTER the pre-steps of assemble(): a raw graph was grouped on the device, so
    // its staged form is made here, on demand, by the host restatement of the same steps (pre_steps.cpp)
    HostBatch staged_copy; int gi = graph;
    if(b->hb.g_rawdist[(size_t)graph] >= 0) {
        const HostBatch &h = b->hb; const int64_t ov = h.off_v[graph], ovo = ov + graph, oe = h.off_e[graph], oeo = oe + graph, os = h.off_s[graph], orp = h.off_rp[graph], orc = h.off_rc[graph];
        ald_graph_view gv; memset(&gv, 0, sizeof(gv));
        gv.num_vertices = h.g_nv[graph]; gv.num_edges = h.g_ne[graph]; gv.vertex_offset = &h.vertex_offset[ovo]; gv.edge_target = h.edge_target.data() + oe; gv.edge_weight = h.edge_weight.data() + oe;
        gv.edge_strand = h.edge_strand.data() + oe; gv.edge_abd = h.edge_abd.data() + oe; gv.edge_sample_offset = &h.edge_sample_offset[oeo]; gv.sample_id = h.sample_id.data() + os; gv.sample_abd = h.sample_abd.data() + os;
        gv.vertex_weight = h.vertex_weight.data() + ov; gv.vertex_lpos = h.vertex_lpos.data() + ov; gv.vertex_rpos = h.vertex_rpos.data() + ov; gv.vertex_type = h.vertex_type.data() + ov;
        gv.strand = h.graph_strand[(size_t)graph]; gv.edge_count = h.edge_count.data() + oe; gv.edge_creation_rank = h.has_rank ? h.edge_rank.data() + oe : nullptr;
        ald_phase_view pv; pv.num_phases = (int32_t)(h.off_rp[graph + 1] - orp); pv.phase_offset = &h.rphase_offset[(size_t)(orp + graph)]; pv.phase_coord = h.rphase_coord.data() + orc; pv.phase_count = h.rphase_count.data() + orp;
        ald_staged *S = nullptr;
        const int rc = ald_pre_assemble(&gv, &pv, h.g_rawdist[(size_t)graph], &S);
        if(rc != ALD_OK) return rc;
        ald_graph_view sv; ald_staged_view(S, &sv);
        const int rc2 = staged_copy.add_graph(sv);
        ald_staged_free(S);
        if(rc2 != ALD_OK) return ald_set_err(rc2, staged_copy.err);
        gi = 0;
    }
    const GraphRO G(gi == graph && b->hb.g_rawdist[(size_t)graph] < 0 ? b->hb : staged_copy, gi);
    const int64_t p0 = b->res.path_begin[graph]; const int np = (int)(b->res.path_begin[graph + 1] - p0);
    std::vector<std::vector<Junc>> junc((size_t)np);
    for(int k = 0; k < np; k++) { const PathRec p = b->res.path((int64_t)((size_t)(p0 + k))); junctions(G, b->res.vertices(p), p.nv, junc[(size_t)k]); }
    // unique_junc (scallop.cc:3472-3497): owner of every junction over the whole path set, -1 once two paths share it
    std::map<Junc, int> owner;
    for(int k = 0; k < np; k++) for(const Junc &j : junc[(size_t)k]) { auto it = owner.find(j); if(it == owner.end()) owner[j] = k; else if(it->second != k && it->second != -1) it->second = -1; }
    int status = ALD_OK;
    auto need = [&](int e) { if(e < 0) status = ALD_ST_INVARIANT + ALD_INV_OTHER; return e >= 0; };      // assert(gr.edge(..).second)
    auto dx = [&](const double *a, int v) { return a ? a[v] : 0.0; };
    auto ix = [&](const int32_t *a, int v) { return a ? a[v] : 0; };
    static const ald_graph_extras none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if(!X) X = &none;
    for(int pid = 0; pid < np; pid++) {
        const PathRec P = b->res.path((int64_t)((size_t)(p0 + pid))); const uint32_t *pv = b->res.vertices(P); const int n = P.nv;
        ald_trst_features &F = features[pid]; memset(&F, 0, sizeof(F));
        if(complete) complete[pid] = 0;
        if(n < 3) { status = ALD_ST_INVARIANT + ALD_INV_OTHER; continue; }                                     // assert(n >= 3)
        F.num_vertices = n - 2; F.num_edges = n - 3; F.gr_vertices = G.V; F.gr_edges = G.E; F.gr_reads = X->gr_reads; F.gr_subgraph = X->gr_subgraph;
        F.max_mid_exon_len = 0;
        const std::vector<Junc> &J = junc[(size_t)pid]; const int nj = (int)J.size();
        if(nj == 0) continue;                                                                                  // single exon: nothing else is set
        {   // junctions over the span between the first and the last spliced vertex
            int is = -1, it = -1;
            for(int i = 0; i < n; i++) { if((int)pv[i] == J.front().first && is < 0) is = i; if((int)pv[i] == J.back().second) it = i; }
            F.junc_ratio = 1.0 * nj / (it - is);
        }
        for(int i = 1; i < nj; i++) { const int len = G.rpos(J[(size_t)i].first) - G.lpos(J[(size_t)i - 1].second); if(len > F.max_mid_exon_len) F.max_mid_exon_len = len; }
        const int sv = (int)pv[1], ev = (int)pv[n - 2];
        F.start_loss1 = dx(X->boundary_loss1, sv); F.start_loss2 = dx(X->boundary_loss2, sv); F.start_loss3 = dx(X->boundary_loss3, sv);
        F.end_loss1 = dx(X->boundary_loss1, ev); F.end_loss2 = dx(X->boundary_loss2, ev); F.end_loss3 = dx(X->boundary_loss3, ev);
        F.start_merged_loss = dx(X->boundary_merged_loss, sv); F.end_merged_loss = dx(X->boundary_merged_loss, ev);
        for(const Junc &j : J) { auto it = owner.find(j); if(it != owner.end() && it->second == pid) F.uni_junc++; }
        // introns of OTHER paths that fall inside one exon of this path (scallop.cc:3326-3396): the exon before the first junction
        // (start), between two junctions (middle), behind the last one (end); each with the ratio junction weight / the smaller of
        // the two flanking within-exon edges
        auto ratio_of = [&](const Junc &q, double &dst) {
            const int e = G.edge(q.first, q.second), e1 = G.edge(q.first, q.first + 1), e2 = G.edge(q.second - 1, q.second);
            if(!need(e) || !need(e1) || !need(e2)) return;
            const double r = G.w(e) / std::min(G.w(e1), G.w(e2));
            if(dst < r) dst = r;
        };
        for(int o = 0; o < np && nj >= 2; o++) {
            if(o == pid) continue;
            const std::vector<Junc> &K = junc[(size_t)o];
            if(K.empty()) continue;
            int mid = 0, head = 0, tail = 0;
            for(int i = 0; i < nj; i++) for(const Junc &q : K) {
                if(i == 0) { if(q.first >= sv && q.second <= J[0].first) { head++; ratio_of(q, F.start_intron_ratio); } }
                else if(q.second <= J[(size_t)i].first && q.first >= J[(size_t)i - 1].second) { mid++; ratio_of(q, F.intron_ratio); }
                if(i == nj - 1) { if(q.first >= J[(size_t)i].second && q.second <= ev) { tail++; ratio_of(q, F.end_intron_ratio); } }
            }
            if(F.introns < mid) F.introns = mid;
            if(F.start_introns < head) F.start_introns = head;
            if(F.end_introns < tail) F.end_introns = tail;
        }
        // along the path's own edges (scallop.cc:3399-3448)
        F.seq_min_wt = DBL_MAX; F.seq_min_cnt = INT_MAX; F.seq_min_abd = DBL_MAX; F.seq_min_ratio = 1.0;
        for(int i = 1; i < n; i++) {
            const int v1 = (int)pv[i - 1], v2 = (int)pv[i];
            const int e = G.edge(v1, v2);
            if(!need(e)) continue;
            const double w = G.w(e), r = w / std::max(G.in_weights(v2), G.out_weights(v1));
            F.seq_min_wt = std::min(F.seq_min_wt, w); F.seq_min_cnt = std::min(F.seq_min_cnt, G.cnt(e)); F.seq_min_abd = std::min(F.seq_min_abd, G.abd(e)); F.seq_min_ratio = std::min(F.seq_min_ratio, r);
            F.seq_max_wt = std::max(F.seq_max_wt, w); F.seq_max_cnt = std::max(F.seq_max_cnt, G.cnt(e)); F.seq_max_abd = std::max(F.seq_max_abd, G.abd(e)); F.seq_max_ratio = std::max(F.seq_max_ratio, r);
            if(i == 1) { F.unbridge_start_coming_count = ix(X->unbridge_coming_count, v2); F.unbridge_start_coming_ratio = dx(X->unbridge_coming_ratio, v2); F.start_cnt = G.cnt(e); F.start_weight = w; F.start_abd = G.abd(e); }
            else if(i == n - 2) { F.unbridge_end_leaving_count = ix(X->unbridge_leaving_count, v2); F.unbridge_end_leaving_ratio = dx(X->unbridge_leaving_ratio, v2); }
            else if(i == n - 1) { F.end_cnt = G.cnt(e); F.end_weight = w; F.end_abd = G.abd(e); }
        }
        if(complete) complete[pid] = 1;
    }
    return status;
}

} // extern "C"

namespace {
// append-only text sink with snprintf's contract: never writes past cap, always counts what the full text needs
struct Text {
    char *buf; int64_t cap, len = 0;
    Text(char *b, int64_t c) : buf(b), cap(c < 0 ? 0 : c) {}
    void put(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        va_list ap; va_start(ap, fmt);
        const int64_t room = cap > len ? cap - len : 0;
        const int k = vsnprintf(room > 0 ? buf + len : nullptr, (size_t)room, fmt, ap);
        va_end(ap);
        if(k > 0) len += k;
    }
    int64_t done() { if(cap > 0) buf[len < cap ? len : cap - 1] = 0; return len; }
};
const char *str(const char *s) { return s ? s : ""; }
} // namespace

extern "C" {

// ostream << fixed << setprecision(4) for the coverages, plain integers elsewhere; positions are written 1-based / closed on the left
int64_t ald_gtf_format_transcript(char *buf, int64_t cap, const char *seqname, const char *source, const char *gene_id, const char *transcript_id,
                                  const char *gene_type, const char *transcript_type, char strand, double coverage, double cov2, int32_t count,
                                  int32_t n_exons, const int32_t *exon_lr)
{
    Text T(buf, cap);
    if(n_exons <= 0 || !exon_lr) return T.done();              // transcript.cc:323: nothing is written for a transcript without exons
    T.put("%s\t%s\ttranscript\t%d\t%d\t1000\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\"; ", str(seqname), str(source), exon_lr[0] + 1, exon_lr[2 * n_exons - 1], strand, str(gene_id), str(transcript_id));
    if(gene_type && *gene_type) T.put("gene_type \"%s\"; ", gene_type);
    if(transcript_type && *transcript_type) T.put("transcript_type \"%s\"; ", transcript_type);
    T.put("cov \"%.4f\"; ", coverage);
    if(cov2 >= -0.5) T.put("cov2 \"%.4f\"; ", cov2);
    if(count >= -0.5) T.put("count \"%d\"; ", count);
    T.put("\n");
    for(int k = 0; k < n_exons; k++)
        T.put("%s\t%s\texon\t%d\t%d\t1000\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\"; exon \"%d\"; \n", str(seqname), str(source), exon_lr[2 * k] + 1, exon_lr[2 * k + 1], strand, str(gene_id), str(transcript_id), k + 1);
    return T.done();
}

int64_t ald_gtf_format_features(char *buf, int64_t cap, int32_t fixed2, const char *transcript_id, const char *meta_tid, const char *seqname,
                                double coverage, double cov2, double abd, double conf, int32_t count1, int32_t count2, int32_t n_exons, const ald_trst_features *f)
{
    Text T(buf, cap);
    if(!f) return T.done();
    const char *D = fixed2 ? "%.2f\t" : "%g\t";               // ostream default: %g with 6 significant digits; the file form: fixed, precision 2
    auto d = [&](double x) { T.put(D, x); };
    auto i = [&](int x) { T.put("%d\t", x); };
    T.put("%s\t%s\t%s\t", str(transcript_id), str(meta_tid), str(seqname));
    d(coverage); d(cov2); d(abd); d(conf); i(count1); i(count2); i(n_exons);
    i(f->gr_vertices); i(f->gr_edges); i(f->gr_reads); i(f->gr_subgraph); i(f->num_vertices); i(f->num_edges); d(f->junc_ratio); i(f->max_mid_exon_len);
    d(f->start_loss1); d(f->start_loss2); d(f->start_loss3); d(f->end_loss1); d(f->end_loss2); d(f->end_loss3); d(f->start_merged_loss); d(f->end_merged_loss);
    i(f->introns); d(f->intron_ratio); i(f->start_introns); d(f->start_intron_ratio); i(f->end_introns); d(f->end_intron_ratio); i(f->uni_junc);
    d(f->seq_min_wt); i(f->seq_min_cnt); d(f->seq_min_abd); d(f->seq_min_ratio); d(f->seq_max_wt); i(f->seq_max_cnt); d(f->seq_max_abd); d(f->seq_max_ratio);
    i(f->start_cnt); d(f->start_weight); d(f->start_abd); i(f->end_cnt); d(f->end_weight); d(f->end_abd);
    i(f->unbridge_start_coming_count); d(f->unbridge_start_coming_ratio); i(f->unbridge_end_leaving_count);
    T.put(fixed2 ? "%.2f\n" : "%g\n", f->unbridge_end_leaving_ratio);
    return T.done();
}

int64_t ald_transcript_id(char *buf, int64_t cap, const char *chrm, const char *gid, int32_t path_index)
{
    Text T(buf, cap);
    T.put("chr%s.%s.%d", str(chrm), str(gid), path_index);
    return T.done();
}

} // extern "C"
