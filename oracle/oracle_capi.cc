/*
 * oracle_capi.cc -- TEST INFRASTRUCTURE ONLY.  C entry points over scallop_oracle.hpp so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the CPU restatement
 * with exactly the packed arrays the GPU ABI (include/aletsch_decomp.h, ald_batch_add_packed)
 * takes.  The product library never links or loads this file.
 */
#include "scallop_oracle.hpp"
#include "subsetsum_oracle.hpp"
#include "../include/aletsch_decomp.h"
#include <thread>
#include <atomic>
#include <chrono>
#include <cstring>

namespace {

struct GraphResult {
    int status = 0;
    int iterations = 0;
    std::vector<ora::Path> paths;
    std::vector<ora::Transcript> trsts;
    std::vector<ora::TraceEvent> trace;
    ora::Stats st;
    int feature_assert = 0;
};

struct Packed {
    int32_t n;
    const int32_t *g_nv, *g_ne, *g_np;
    const int32_t *vertex_offset, *edge_target; const double *edge_weight; const uint8_t *edge_strand; const double *edge_abd;
    const int32_t *edge_sample_offset, *sample_id; const double *sample_abd;
    const double *vertex_weight; const int32_t *vertex_lpos, *vertex_rpos, *vertex_type;
    const int32_t *phasing_offset, *phasing_vertex, *phasing_count; const char *graph_strand; const int32_t *edge_count, *edge_rank;
    const ald_graph_extras *extras;            // [n] or null: what only the feature block reads
};

struct Offsets { int64_t v, vo, e, eo, s, p, po, pv; };

void run_one(const Packed &P, const Offsets &o, int g, const ora::Params &cfg, bool want_trace, GraphResult &out)
{
    int V = P.g_nv[g], E = P.g_ne[g], NP = P.g_np ? P.g_np[g] : 0;
    ora::Graph gr;
    for(int i = 0; i < V; i++) gr.add_vertex();
    gr.strand = P.graph_strand ? P.graph_strand[g] : '.';
    for(int i = 0; i < V; i++) {
        gr.vwrt[i] = P.vertex_weight[o.v + i];
        gr.vinf[i].lpos = P.vertex_lpos[o.v + i]; gr.vinf[i].rpos = P.vertex_rpos[o.v + i];
        gr.vinf[i].type = P.vertex_type ? P.vertex_type[o.v + i] : -1;
    }
    if(P.extras) {
        const ald_graph_extras &X = P.extras[g]; gr.reads = X.gr_reads; gr.subgraph = X.gr_subgraph;
        for(int i = 0; i < V; i++) {
            ora::VertexInfo &vi = gr.vinf[i];
            if(X.boundary_loss1) vi.boundary_loss1 = X.boundary_loss1[i]; if(X.boundary_loss2) vi.boundary_loss2 = X.boundary_loss2[i]; if(X.boundary_loss3) vi.boundary_loss3 = X.boundary_loss3[i];
            if(X.boundary_merged_loss) vi.boundary_merged_loss = X.boundary_merged_loss[i];
            if(X.unbridge_leaving_count) vi.unbridge_leaving_count = X.unbridge_leaving_count[i]; if(X.unbridge_leaving_ratio) vi.unbridge_leaving_ratio = X.unbridge_leaving_ratio[i];
            if(X.unbridge_coming_count) vi.unbridge_coming_count = X.unbridge_coming_count[i]; if(X.unbridge_coming_ratio) vi.unbridge_coming_ratio = X.unbridge_coming_ratio[i];
        }
    }
    const int32_t *vo = P.vertex_offset + o.vo; const int32_t *so = P.edge_sample_offset + o.eo;
    // edges are created in the order of their creation rank (the reference's gr.edges() order: graph_base.cc:139-153 assigns the
    // scallop edge indices by walking `se`); without a rank that order is the CSR position
    std::vector<int> src_of(E), order(E);
    for(int s = 0; s < V; s++) for(int k = vo[s]; k < vo[s + 1]; k++) src_of[k] = s;
    for(int k = 0; k < E; k++) order[P.edge_rank ? P.edge_rank[o.e + k] : k] = k;
    for(int q = 0; q < E; q++) {
        const int k = order[q], s = src_of[k];
        int e = gr.add_edge(s, P.edge_target[o.e + k]);
        gr.ewrt[e] = P.edge_weight[o.e + k];
        ora::EdgeInfo &ei = gr.einf[e];
        ei.strand = P.edge_strand ? P.edge_strand[o.e + k] : 0;
        double sum = 0;
        for(int j = so[k]; j < so[k + 1]; j++) { int sid = P.sample_id[o.s + j]; double a = P.sample_abd[o.s + j]; ei.samples.insert(sid); ei.spAbd[sid] = a; sum += a; }
        ei.count = P.edge_count ? P.edge_count[o.e + k] : so[k + 1] - so[k];       // the hand-over count is not always |samples| (graph_reviser.cc:965-975)
        ei.abd = P.edge_abd ? P.edge_abd[o.e + k] : sum;
    }
    ora::HyperSet hs;
    if(NP > 0) {
        const int32_t *po = P.phasing_offset + o.po;
        for(int p = 0; p < NP; p++) {
            std::vector<int> v(P.phasing_vertex + o.pv + po[p], P.phasing_vertex + o.pv + po[p + 1]);
            int c = P.phasing_count[o.p + p];
            if(hs.nodes.count(v)) hs.nodes[v] += c; else hs.nodes[v] = c;     // hyper_set.cc:40-48 add_node_list
        }
    }
    try {
        ora::Scallop sc(gr, hs, cfg);
        if(want_trace) sc.trace = &out.trace;
        sc.assemble();
        out.paths = sc.paths; out.trsts = sc.trsts; out.st = sc.st; out.iterations = sc.st.iterations; out.feature_assert = sc.feature_assert;
        if(sc.st.cut_short) out.status = ALD_ST_SKIPPED_LARGE;        // the loop left through `num_vertices() > max_num_exons` (scallop.cc:49), at the start or after growing
    } catch(const ora::AssertFail &a) {
        out.status = ALD_ST_INVARIANT + a.cls;
        out.paths.clear(); out.trsts.clear();
        if(getenv("ORA_VERBOSE")) fprintf(stderr, "oracle: graph %d assert class %d line %d: %s\n", g, a.cls, a.line, a.what);
    }
}

} // namespace

struct ora_result {
    std::vector<GraphResult> gr;
    double seconds = 0;
};

extern "C" {

static const ald_graph_extras *g_next_extras = nullptr;      // consumed by the next ora_run_packed (keeps the argument list shared with ald_batch_add_packed)
void ora_set_extras(const ald_graph_extras *per_graph) { g_next_extras = per_graph; }

int ora_run_packed(int32_t n,
                   const int32_t *g_nv, const int32_t *g_ne, const int32_t *g_np,
                   const int32_t *vertex_offset, const int32_t *edge_target,
                   const double *edge_weight, const uint8_t *edge_strand, const double *edge_abd,
                   const int32_t *edge_sample_offset, const int32_t *sample_id, const double *sample_abd,
                   const double *vertex_weight, const int32_t *vertex_lpos, const int32_t *vertex_rpos,
                   const int32_t *vertex_type,
                   const int32_t *phasing_offset, const int32_t *phasing_vertex, const int32_t *phasing_count,
                   const char *graph_strand, const int32_t *edge_count, const int32_t *edge_rank,
                   const ald_params *prm, int32_t n_threads, int32_t want_trace, ora_result **out)
{
    Packed P{n, g_nv, g_ne, g_np, vertex_offset, edge_target, edge_weight, edge_strand, edge_abd, edge_sample_offset, sample_id, sample_abd,
             vertex_weight, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, graph_strand, edge_count, edge_rank, g_next_extras};
    g_next_extras = nullptr;
    ora::Params cfg;
    if(prm) {
        for(int i = 0; i < 8; i++) cfg.max_decompose_error_ratio[i] = prm->max_decompose_error_ratio[i];
        cfg.min_guaranteed_edge_weight = prm->min_guaranteed_edge_weight;
        cfg.min_transcript_coverage = prm->min_transcript_coverage;
        cfg.max_num_exons = prm->max_num_exons;
    }
    std::vector<Offsets> off(n + 1);
    Offsets o{0, 0, 0, 0, 0, 0, 0, 0};
    for(int g = 0; g < n; g++) {
        off[g] = o;
        int V = g_nv[g], E = g_ne[g], NP = g_np ? g_np[g] : 0;
        int64_t ns = (edge_sample_offset + o.eo)[E];
        int64_t npv = NP > 0 ? (phasing_offset + o.po)[NP] : 0;
        o.v += V; o.vo += V + 1; o.e += E; o.eo += E + 1; o.s += ns; o.p += NP; o.po += NP + 1; o.pv += npv;
    }
    ora_result *R = new ora_result();
    R->gr.resize(n);
    auto t0 = std::chrono::steady_clock::now();
    if(n_threads <= 1) {
        for(int g = 0; g < n; g++) run_one(P, off[g], g, cfg, want_trace != 0, R->gr[g]);
    } else {
        std::atomic<int> next(0);
        std::vector<std::thread> th;
        for(int t = 0; t < n_threads; t++) th.emplace_back([&]() {
            while(true) { int g = next.fetch_add(1); if(g >= n) break; run_one(P, off[g], g, cfg, want_trace != 0, R->gr[g]); }
        });
        for(auto &t : th) t.join();
    }
    R->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = R;
    return 0;
}

double ora_result_seconds(const ora_result *R) { return R->seconds; }
void ora_result_free(ora_result *R) { delete R; }

int ora_result_export(const ora_result *R, int64_t *total_paths, int64_t *total_path_vertices,
                      int32_t *status, int32_t *path_offset,
                      double *weight, double *abd, double *conf, double *reads,
                      int32_t *length, int32_t *count, char *strand,
                      int64_t *pv_offset, int32_t *path_vertices)
{
    int64_t tp = 0, tv = 0;
    for(auto &g : R->gr) { tp += (int64_t)g.paths.size(); for(auto &p : g.paths) tv += (int64_t)p.v.size(); }
    if(total_paths) *total_paths = tp;
    if(total_path_vertices) *total_path_vertices = tv;
    if(!status) return 0;
    int64_t ip = 0, iv = 0;
    for(size_t g = 0; g < R->gr.size(); g++) {
        status[g] = R->gr[g].status; path_offset[g] = (int32_t)ip;
        for(auto &p : R->gr[g].paths) {
            weight[ip] = p.weight; abd[ip] = p.abd; conf[ip] = p.conf; reads[ip] = p.reads; length[ip] = p.length; count[ip] = p.count; strand[ip] = p.strand;
            pv_offset[ip] = iv;
            for(int x : p.v) path_vertices[iv++] = x;
            ip++;
        }
    }
    path_offset[R->gr.size()] = (int32_t)ip; pv_offset[ip] = iv;
    return 0;
}

/* transcripts: exon lists + coverage (scallop.cc:3250-3266, essential.cc:719-748) */
int ora_result_export_transcripts(const ora_result *R, int64_t *total_exons, double *coverage, int64_t *exon_offset, int32_t *exon_lr)
{
    int64_t te = 0; for(auto &g : R->gr) for(auto &t : g.trsts) te += (int64_t)t.exons.size();
    if(total_exons) *total_exons = te;
    if(!coverage) return 0;
    int64_t it = 0, ie = 0;
    for(auto &g : R->gr) for(auto &t : g.trsts) {
        coverage[it] = t.coverage; exon_offset[it] = ie;
        for(auto &x : t.exons) { exon_lr[2 * ie] = x.first; exon_lr[2 * ie + 1] = x.second; ie++; }
        it++;
    }
    exon_offset[it] = ie;
    return 0;
}

/* feature block of every transcript of one graph (scallop.cc:3268-3451); returns 0, or 1 when an assert of update_trst_features fired */
int ora_result_features(const ora_result *R, int32_t graph, ald_trst_features *out, int32_t *complete)
{
    const GraphResult &G = R->gr[graph];
    for(size_t k = 0; k < G.trsts.size(); k++) {
        const ora::Features &f = G.trsts[k].features; ald_trst_features &o = out[k];
        o.gr_vertices = f.gr_vertices; o.gr_edges = f.gr_edges; o.gr_reads = f.gr_reads; o.gr_subgraph = f.gr_subgraph; o.num_vertices = f.num_vertices; o.num_edges = f.num_edges;
        o.junc_ratio = f.junc_ratio; o.max_mid_exon_len = f.max_mid_exon_len;
        o.start_loss1 = f.start_loss1; o.start_loss2 = f.start_loss2; o.start_loss3 = f.start_loss3; o.end_loss1 = f.end_loss1; o.end_loss2 = f.end_loss2; o.end_loss3 = f.end_loss3;
        o.start_merged_loss = f.start_merged_loss; o.end_merged_loss = f.end_merged_loss;
        o.introns = f.introns; o.start_introns = f.start_introns; o.end_introns = f.end_introns; o.intron_ratio = f.intron_ratio; o.start_intron_ratio = f.start_intron_ratio; o.end_intron_ratio = f.end_intron_ratio;
        o.uni_junc = f.uni_junc;
        o.seq_min_wt = f.seq_min_wt; o.seq_min_cnt = f.seq_min_cnt; o.seq_min_abd = f.seq_min_abd; o.seq_min_ratio = f.seq_min_ratio;
        o.seq_max_wt = f.seq_max_wt; o.seq_max_cnt = f.seq_max_cnt; o.seq_max_abd = f.seq_max_abd; o.seq_max_ratio = f.seq_max_ratio;
        o.unbridge_start_coming_count = f.unbridge_start_coming_count; o.unbridge_start_coming_ratio = f.unbridge_start_coming_ratio;
        o.unbridge_end_leaving_count = f.unbridge_end_leaving_count; o.unbridge_end_leaving_ratio = f.unbridge_end_leaving_ratio;
        o.start_cnt = f.start_cnt; o.start_weight = f.start_weight; o.start_abd = f.start_abd; o.end_cnt = f.end_cnt; o.end_weight = f.end_weight; o.end_abd = f.end_abd;
        if(complete) complete[k] = f.complete ? 1 : 0;
    }
    return G.feature_assert ? 1 : 0;
}

/* per-graph diagnostics: [max_live_edges, max_vertices, total_edge_ids, iterations, router_builds, max_mev] */
int ora_result_stats(const ora_result *R, int32_t *stats6)
{
    for(size_t g = 0; g < R->gr.size(); g++) {
        const ora::Stats &s = R->gr[g].st;
        int32_t *o = stats6 + 6 * g;
        o[0] = s.max_live_edges; o[1] = s.max_vertices; o[2] = s.total_edge_ids; o[3] = s.iterations; o[4] = s.router_builds; o[5] = s.max_mev;
    }
    return 0;
}

int ora_result_trace(const ora_result *R, int32_t graph, int32_t *n_events, int32_t *codes3, double *values, int32_t cap)
{
    const auto &t = R->gr[graph].trace;
    *n_events = (int32_t)t.size();
    if(!codes3) return 0;
    for(int i = 0; i < (int)t.size() && i < cap; i++) { codes3[3 * i] = t[i].code; codes3[3 * i + 1] = t[i].a; codes3[3 * i + 2] = t[i].b; values[i] = t[i].val; }
    return 0;
}

/* the pre-steps of assembler::assemble(gx, px, sid) (meta/assembler.cc:1075-1086) restated on the oracle's containers; the result in the
 * layout of ald_staged_view (CSR by (source, target, creation), compacted creation ranks) so that it can be compared array by array */
struct ora_staged {
    std::vector<int32_t> vertex_offset, edge_target, edge_sample_offset, sample_id, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, edge_count, edge_rank;
    std::vector<double> edge_weight, edge_abd, sample_abd, vertex_weight; std::vector<uint8_t> edge_strand; char strand = '.';
    std::vector<int32_t> smap, tmap;
};
int ora_pre_assemble(const ald_graph_view *g, const ald_phase_view *ph, int32_t dist, ora_staged **out)
{
    const int V = g->num_vertices, NE = g->num_edges;
    ora::Graph gr; for(int i = 0; i < V; i++) gr.add_vertex();
    gr.strand = g->strand ? g->strand : '.';
    for(int i = 0; i < V; i++) { gr.vwrt[i] = g->vertex_weight[i]; gr.vinf[i].lpos = g->vertex_lpos[i]; gr.vinf[i].rpos = g->vertex_rpos[i]; gr.vinf[i].type = g->vertex_type ? g->vertex_type[i] : -1; }
    std::vector<int> src_of(NE), order(NE);
    for(int s = 0; s < V; s++) for(int k = g->vertex_offset[s]; k < g->vertex_offset[s + 1]; k++) src_of[k] = s;
    for(int k = 0; k < NE; k++) order[g->edge_creation_rank ? g->edge_creation_rank[k] : k] = k;
    for(int q = 0; q < NE; q++) {
        const int k = order[q]; int e = gr.add_edge(src_of[k], g->edge_target[k]);
        gr.ewrt[e] = g->edge_weight[k]; ora::EdgeInfo &ei = gr.einf[e]; ei.strand = g->edge_strand ? g->edge_strand[k] : 0;
        double sum = 0;
        for(int j = g->edge_sample_offset[k]; j < g->edge_sample_offset[k + 1]; j++) { ei.samples.insert(g->sample_id[j]); ei.spAbd[g->sample_id[j]] = g->sample_abd[j]; sum += g->sample_abd[j]; }
        ei.count = g->edge_count ? g->edge_count[k] : g->edge_sample_offset[k + 1] - g->edge_sample_offset[k]; ei.abd = g->edge_abd ? g->edge_abd[k] : sum;
    }
    ora::PhaseSet px;
    ora_staged *S = new ora_staged();
    try {
        if(ph) for(int p = 0; p < ph->num_phases; p++) px.add(std::vector<int32_t>(ph->phase_coord + ph->phase_offset[p], ph->phase_coord + ph->phase_offset[p + 1]), ph->phase_count[p]);
        ora::extend_strands(gr);
        std::map<int32_t, int32_t> smap, tmap;
        ora::group_start_boundaries(gr, smap, dist);
        ora::group_end_boundaries(gr, tmap, dist);
        px.project_boundaries(smap, tmap);
        ora::HyperSet hx(gr, px);
        hx.filter_nodes(gr);
        for(auto &x : smap) { S->smap.push_back(x.first); S->smap.push_back(x.second); }
        for(auto &x : tmap) { S->tmap.push_back(x.first); S->tmap.push_back(x.second); }
        S->vertex_offset.assign(V + 1, 0); S->edge_sample_offset.push_back(0); S->phasing_offset.push_back(0);
        std::map<int, int> newrank; { int r = 0; for(int e : gr.se) newrank[e] = r++; }
        for(int s = 0; s < V; s++) {
            for(int e : gr.out_edges(s)) {                                      // (target, creation) order
                S->edge_target.push_back(gr.et[e]); S->edge_weight.push_back(gr.ewrt[e]); S->edge_strand.push_back((uint8_t)gr.einf[e].strand); S->edge_abd.push_back(gr.einf[e].abd);
                S->edge_count.push_back(gr.einf[e].count); S->edge_rank.push_back(newrank[e]);
                for(int sid : gr.einf[e].samples) { S->sample_id.push_back(sid); S->sample_abd.push_back(gr.einf[e].spAbd[sid]); }
                S->edge_sample_offset.push_back((int32_t)S->sample_id.size());
            }
            S->vertex_offset[s + 1] = (int32_t)S->edge_target.size();
        }
        for(int i = 0; i < V; i++) { S->vertex_weight.push_back(gr.vwrt[i]); S->vertex_lpos.push_back(gr.vinf[i].lpos); S->vertex_rpos.push_back(gr.vinf[i].rpos); S->vertex_type.push_back(gr.vinf[i].type); }
        S->strand = gr.strand;
        for(auto &x : hx.nodes) { for(int q : x.first) S->phasing_vertex.push_back(q); S->phasing_offset.push_back((int32_t)S->phasing_vertex.size()); S->phasing_count.push_back(x.second); }
    } catch(const ora::AssertFail &a) { delete S; return ALD_ST_INVARIANT + a.cls; }
    *out = S;
    return 0;
}
int ora_staged_view(const ora_staged *S, ald_graph_view *g)
{
    static const int32_t zi = 0; static const double zd = 0; static const uint8_t zb = 0;
    memset(g, 0, sizeof(*g));
    g->num_vertices = (int32_t)S->vertex_weight.size(); g->num_edges = (int32_t)S->edge_target.size();
    g->vertex_offset = S->vertex_offset.data(); g->edge_target = S->edge_target.empty() ? &zi : S->edge_target.data(); g->edge_weight = S->edge_weight.empty() ? &zd : S->edge_weight.data();
    g->edge_strand = S->edge_strand.empty() ? &zb : S->edge_strand.data(); g->edge_abd = S->edge_abd.empty() ? &zd : S->edge_abd.data();
    g->edge_sample_offset = S->edge_sample_offset.data(); g->sample_id = S->sample_id.empty() ? &zi : S->sample_id.data(); g->sample_abd = S->sample_abd.empty() ? &zd : S->sample_abd.data();
    g->vertex_weight = S->vertex_weight.data(); g->vertex_lpos = S->vertex_lpos.data(); g->vertex_rpos = S->vertex_rpos.data(); g->vertex_type = S->vertex_type.data();
    g->num_phasing = (int32_t)S->phasing_count.size(); g->phasing_offset = S->phasing_offset.data();
    g->phasing_vertex = S->phasing_vertex.empty() ? &zi : S->phasing_vertex.data(); g->phasing_count = S->phasing_count.empty() ? &zi : S->phasing_count.data();
    g->strand = S->strand; g->edge_count = S->edge_count.empty() ? &zi : S->edge_count.data(); g->edge_creation_rank = S->edge_rank.empty() ? nullptr : S->edge_rank.data();
    return 0;
}
int ora_staged_boundary_maps(const ora_staged *S, int32_t *n_smap, const int32_t **smap_pairs, int32_t *n_tmap, const int32_t **tmap_pairs)
{
    *n_smap = (int32_t)(S->smap.size() / 2); *smap_pairs = S->smap.data(); *n_tmap = (int32_t)(S->tmap.size() / 2); *tmap_pairs = S->tmap.data();
    return 0;
}
void ora_staged_free(ora_staged *S) { delete S; }

/* Graph-layer script runner: the same edit/dump language as oracle/ref_drivers/ref_graph_main.cc, executed on the oracle's
 * creation-ordered containers, so that the dump can be diffed against the reference-built oracle/_ref/ref_graph. */
int ora_graph_script(const char *script, char *out, int32_t cap)
{
    std::string o; char buf[64];
    ora::Graph dg; ora::UGraph ug; bool directed = true;
    std::vector<int> h;   // handle -> edge id (or -1)
    const char *p = script;
    auto next_int = [&](int &v) { while(*p == ' ' || *p == '\n') p++; char *e; v = (int)strtol(p, &e, 10); p = e; };
    while(*p) {
        while(*p == ' ' || *p == '\n') p++;
        if(!*p) break;
        char op = *p++;
        if(op == 'D') { int n; next_int(n); dg = ora::Graph(); directed = true; h.clear(); for(int i = 0; i < n; i++) dg.add_vertex(); }
        else if(op == 'U') { int n; next_int(n); ug = ora::UGraph(); directed = false; h.clear(); for(int i = 0; i < n; i++) ug.add_vertex(); }
        else if(op == 'a') { int s, t; next_int(s); next_int(t); h.push_back(directed ? dg.add_edge(s, t) : ug.add_edge(s, t)); }
        else if(op == 'r') { int k; next_int(k); if(directed) dg.remove_edge(h[k]); else ug.remove_edge(h[k]); h[k] = -1; }
        else if(op == 'm') { int k, x, y; next_int(k); next_int(x); next_int(y); dg.move_edge(h[k], x, y); }
        else if(op == 'c') { int v; next_int(v);
            if(directed) { std::vector<int> es = dg.in_edges(v); for(int e : dg.out_edges(v)) es.push_back(e); for(int e : es) { dg.remove_edge(e); } for(auto &x : h) if(x >= 0 && !dg.alive(x)) x = -1; }
            else { std::vector<int> es; for(auto &k : ug.so[v]) es.push_back(std::get<2>(k)); ug.clear_vertex(v); for(auto &x : h) for(int e : es) if(x == e) x = -1; } }
        else if(op == 'q') {
            // handles are creation-numbered, and so are the oracle's edge ids: handle == id
            int n = directed ? dg.num_vertices() : ug.num_vertices();
            for(int v = 0; v < n; v++) {
                if(directed) { snprintf(buf, sizeof buf, "in %d:", v); o += buf; for(int e : dg.in_edges(v)) { snprintf(buf, sizeof buf, " %d", e); o += buf; } o += "\n"; }
                snprintf(buf, sizeof buf, "out %d:", v); o += buf;
                if(directed) { for(int e : dg.out_edges(v)) { snprintf(buf, sizeof buf, " %d", e); o += buf; } }
                else { for(auto &k : ug.so[v]) { snprintf(buf, sizeof buf, " %d", std::get<2>(k)); o += buf; } }
                o += "\n";
            }
            o += "edges:"; for(int e : (directed ? dg.se : ug.se)) { snprintf(buf, sizeof buf, " %d", e); o += buf; } o += "\n";
            if(directed) {
                o += "topo:"; for(int x : dg.topological_sort()) { snprintf(buf, sizeof buf, " %d", x); o += buf; } o += "\n";
                for(int s = 0; s < n; s++) for(int t = 0; t < n; t++) { if(s == t) continue; int e = dg.edge(s, t); if(e >= 0) { snprintf(buf, sizeof buf, "edge %d %d: %d\n", s, t, e); o += buf; } }
            } else {
                for(auto &cc : ug.compute_connected_components()) { o += "cc:"; for(int x : cc) { snprintf(buf, sizeof buf, " %d", x); o += buf; } o += "\n"; }
            }
            o += "end\n";
        }
    }
    if((int)o.size() + 1 > cap) return -1;
    memcpy(out, o.c_str(), o.size() + 1);
    return (int)o.size();
}

/* Router script runner: the same case language and the same '@' output lines as oracle/ref_drivers/ref_router_main.cc (the REFERENCE's
 * scallop/router.cc built from source), executed by the oracle's Router -- so that classify + thread, the isolate attachments, the
 * confidence side effect and the clamp can be diffed against the reference's own object code (tests/golden/ref_router.json). */
int ora_router_script(const char *script, char *out, int32_t cap)
{
    std::string o; char buf[160];
    const char *p = script;
    auto skip = [&]() { while(*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') p++; };
    auto next_int = [&](int &v) { skip(); char *e; v = (int)strtol(p, &e, 10); p = e; };
    auto next_dbl = [&](double &v) { skip(); char *e; v = strtod(p, &e); p = e; };
    for(;;) {
        skip();
        if(*p != 'R') break;
        p++;
        int nv, root, ne, nr; double minw;
        next_int(nv); next_int(root); next_int(ne); next_int(nr); next_dbl(minw);
        ora::Graph gr; for(int i = 0; i < nv; i++) gr.add_vertex();
        for(int k = 0; k < ne; k++) {
            int s, t, strand, count, ns; double w;
            next_int(s); next_int(t); next_dbl(w); next_int(strand); next_int(count); next_int(ns);
            const int e = gr.add_edge(s, t);                      // ids are creation-numbered: e == k
            ora::EdgeInfo &ei = gr.einf[e]; ei.strand = strand; ei.count = count; ei.confidence = 0; ei.abd = 0;
            for(int j = 0; j < ns; j++) { int sp; double a; next_int(sp); next_dbl(a); ei.samples.insert(sp); ei.spAbd[sp] = a; ei.abd += a; }
            gr.ewrt[e] = w;
        }
        ora::MPII mpi;
        for(int k = 0; k < nr; k++) { int a, b, c; next_int(a); next_int(b); next_int(c); mpi[{a, b}] += c; }
        ora::Params cfg; cfg.min_guaranteed_edge_weight = minw;
        try {
            ora::Router rt(root, gr, mpi, cfg);
            rt.classify();
            snprintf(buf, sizeof buf, "@case type %d degree %d\n", rt.type, rt.degree); o += buf;
            if(rt.type == ora::UNSPLITTABLE_SINGLE || rt.type == ora::SPLITTABLE_PURE) {
                rt.build();
                snprintf(buf, sizeof buf, "@ratio %.17g\n", rt.ratio); o += buf;
                for(auto &kv : rt.pe2w) { snprintf(buf, sizeof buf, "@pair %d %d %.17g\n", kv.first.first, kv.first.second, kv.second); o += buf; }
                for(int k = 0; k < ne; k++) { snprintf(buf, sizeof buf, "@conf %d %.17g\n", k, gr.einf[k].confidence); o += buf; }
            }
        } catch(const ora::AssertFail &f) { snprintf(buf, sizeof buf, "@assert %d\n", f.cls); o += buf; }
        o += "@end\n";
    }
    if((int)o.size() + 1 > cap) return -1;
    memcpy(out, o.c_str(), o.size() + 1);
    return (int)o.size();
}

/* subset-sum restatement (oracle/subsetsum_oracle.hpp), one instance */
int ora_subsetsum(int32_t ns, int32_t nt, const int32_t *src_val, const int32_t *src_lab, const int32_t *tgt_val, const int32_t *tgt_lab,
                  double *err, int32_t *out_ns, int32_t *out_nt, int32_t *out_s, int32_t *out_t)
{
    std::vector<std::pair<int, int>> s, t;
    for(int i = 0; i < ns; i++) s.push_back({src_val[i], src_lab[i]});
    for(int i = 0; i < nt; i++) t.push_back({tgt_val[i], tgt_lab[i]});
    ora::SubsetSum ss(s, t);
    if(ss.solve() != 0) return -1;
    *err = ss.e; *out_ns = (int)ss.eqn_s.size(); *out_nt = (int)ss.eqn_t.size();
    for(size_t i = 0; i < ss.eqn_s.size(); i++) out_s[i] = ss.eqn_s[i];
    for(size_t i = 0; i < ss.eqn_t.size(); i++) out_t[i] = ss.eqn_t[i];
    return 0;
}

} // extern "C"
