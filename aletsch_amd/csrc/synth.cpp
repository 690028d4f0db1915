// synth.cpp -- deterministic synthetic splice-graph batches (SURVEY.md section 8d generator).
//
// The reference has a random generator of its own (rnacore/splice_graph.cc:479-570 simulate) but it
// leaves lpos = rpos = 0 and edge_info.count = 0, which trips the reference's own assert at
// scallop.cc:2300; so the bench/test inputs come from this generator instead: forward DAG, every
// internal vertex has >= 1 in- and >= 1 out-edge, coordinates and sample support filled in.
//
// Host-only C++; exported through the C ABI (ald_synth_sizes / ald_synth_fill).
#include "../../include/aletsch_decomp.h"
#include <random>
#include <vector>
#include <set>
#include <map>
#include <algorithm>
#include <cstring>

namespace {

struct OneGraph {
    int V = 0, E = 0;
    std::vector<int> voff, etgt;
    std::vector<double> ew, eabd;
    std::vector<uint8_t> estrand;
    std::vector<int> esoff, sid; std::vector<double> sabd;
    std::vector<double> vw; std::vector<int> lpos, rpos, vtype;
    std::vector<int> poff, pv, pc;
    char strand = '.';
};

typedef std::mt19937_64 Rng;

static int uni(Rng &r, int lo, int hi) { return lo + (int)(r() % (uint64_t)(hi - lo + 1)); }   // inclusive, platform independent
static double unif(Rng &r) { return (double)(r() >> 11) * (1.0 / 9007199254740992.0); }        // [0,1)

static void gen_one(Rng &rng, const ald_synth_spec &sp, OneGraph &g)
{
    int V = uni(rng, sp.v_min, sp.v_max);
    if(V < 3) V = 3;
    long maxE = (long)V * (V - 1) / 2 - 1;
    long wantE = sp.fixed_edges > 0 ? sp.fixed_edges : (long)V * sp.edges_per_vertex;
    if(wantE > maxE) wantE = maxE;
    std::set<std::pair<int, int>> es;
    for(int i = 1; i <= V - 2; i++) es.insert({uni(rng, 0, i - 1), i});
    std::vector<char> has_out(V, 0);
    for(auto &e : es) has_out[e.first] = 1;
    for(int i = 1; i <= V - 2; i++) if(!has_out[i]) { es.insert({i, uni(rng, i + 1, V - 1)}); has_out[i] = 1; }
    while((long)es.size() < wantE) {
        int s = uni(rng, 0, V - 2), t = uni(rng, s + 1, V - 1);
        if(s == 0 && t == V - 1) continue;
        es.insert({s, t});
    }
    g.V = V; g.E = (int)es.size();
    g.voff.assign(V + 1, 0); g.etgt.clear();
    for(auto &e : es) { g.voff[e.first + 1]++; g.etgt.push_back(e.second); }
    for(int i = 0; i < V; i++) g.voff[i + 1] += g.voff[i];
    std::vector<int> esrc(g.E);
    for(int s = 0; s < V; s++) for(int k = g.voff[s]; k < g.voff[s + 1]; k++) esrc[k] = s;

    // vertices
    g.vw.assign(V, 10.0); g.lpos.assign(V, 0); g.rpos.assign(V, 0); g.vtype.assign(V, -1);
    int pos = 1000;
    for(int i = 1; i <= V - 2; i++) {
        bool touch = (sp.layout_mode == 1 && i > 1 && uni(rng, 0, 9) < 3);
        int l = touch ? g.rpos[i - 1] : pos;
        int len = (sp.layout_mode == 1) ? uni(rng, 50, 400) : 200;
        g.lpos[i] = l; g.rpos[i] = l + len;
        pos = (sp.layout_mode == 1) ? g.rpos[i] + uni(rng, 100, 900) : 1000 * (i + 1);
        if(sp.layout_mode == 1) g.vw[i] = 1.0 + 20.0 * unif(rng);
    }
    g.vw[0] = 0; g.vw[V - 1] = 0;
    g.lpos[0] = g.rpos[0] = (V > 2 ? g.lpos[1] : 0);
    g.lpos[V - 1] = g.rpos[V - 1] = (V > 2 ? g.rpos[V - 2] : 0);

    // weights
    g.ew.assign(g.E, 0.0);
    if(sp.weight_mode == 0) { for(int k = 0; k < g.E; k++) g.ew[k] = 1.0 + 99.0 * unif(rng); }
    else if(sp.weight_mode == 1) { for(int k = 0; k < g.E; k++) g.ew[k] = (double)uni(rng, 1, 100); }
    else {
        // flow conserving: every edge lies on >= 1 random s-t path with an integer abundance
        std::vector<std::vector<int>> inl(V);
        for(int k = 0; k < g.E; k++) inl[g.etgt[k]].push_back(k);
        for(int k = 0; k < g.E; k++) {
            double a = (double)uni(rng, 1, 20);
            g.ew[k] += a;
            int x = esrc[k];
            while(x != 0) { int e = inl[x][uni(rng, 0, (int)inl[x].size() - 1)]; g.ew[e] += a; x = esrc[e]; }
            x = g.etgt[k];
            while(x != V - 1) { int e = uni(rng, g.voff[x], g.voff[x + 1] - 1); g.ew[e] += a; x = g.etgt[e]; }
        }
    }

    // strands
    g.strand = '.'; g.estrand.assign(g.E, 0);
    if(sp.strand_mode == 1) {
        int code = uni(rng, 1, 2);
        g.strand = code == 1 ? '+' : '-';
        for(int k = 0; k < g.E; k++) if(uni(rng, 0, 9) < 7) g.estrand[k] = (uint8_t)code;
    }

    // sample support: sample 0 is always present so that any two edges share a sample (router precondition, router.cc:1028)
    g.esoff.assign(g.E + 1, 0); g.sid.clear(); g.sabd.clear(); g.eabd.assign(g.E, 0.0);
    int NS = sp.n_samples < 1 ? 1 : sp.n_samples;
    for(int k = 0; k < g.E; k++) {
        std::vector<int> ids{0};
        for(int s = 1; s < NS; s++) if(uni(rng, 0, 1)) ids.push_back(s);
        std::vector<double> fr(ids.size()); double tot = 0;
        for(auto &f : fr) { f = 0.1 + unif(rng); tot += f; }
        double sum = 0;
        for(size_t j = 0; j < ids.size(); j++) {
            double a = NS == 1 ? g.ew[k] : g.ew[k] * fr[j] / tot;
            g.sid.push_back(ids[j]); g.sabd.push_back(a); sum += a;
        }
        g.eabd[k] = NS == 1 ? g.ew[k] : sum;
        g.esoff[k + 1] = (int)g.sid.size();
    }

    // phasing paths: forward walks over internal vertices, unique + lexicographically sorted (hyper_set::nodes is a map)
    g.poff.assign(1, 0); g.pv.clear(); g.pc.clear();
    if(sp.phasing_per_graph > 0 && V >= 5) {
        std::map<std::vector<int>, int> nodes;
        for(int p = 0; p < sp.phasing_per_graph; p++) {
            int x = uni(rng, 1, V - 2), len = uni(rng, 2, 6);
            std::vector<int> v{x};
            while((int)v.size() < len) {
                std::vector<int> cand;
                for(int k = g.voff[x]; k < g.voff[x + 1]; k++) if(g.etgt[k] != V - 1) cand.push_back(g.etgt[k]);
                if(cand.empty()) break;
                x = cand[uni(rng, 0, (int)cand.size() - 1)]; v.push_back(x);
            }
            if(v.size() < 2) continue;
            int c = uni(rng, 0, 9) == 0 ? 1 : uni(rng, 2, 10);
            nodes[v] += c;
        }
        for(auto &kv : nodes) { for(int x : kv.first) g.pv.push_back(x); g.poff.push_back((int)g.pv.size()); g.pc.push_back(kv.second); }
    }
}

} // namespace

extern "C" {

int ald_synth_sizes(const ald_synth_spec *s, int64_t *tot_v, int64_t *tot_e, int64_t *tot_s, int64_t *tot_p, int64_t *tot_pv)
{
    if(!s || s->n_graphs < 0 || s->v_min > s->v_max) return ALD_ERR_INVALID;
    Rng rng(s->seed);
    int64_t v = 0, e = 0, ss = 0, p = 0, pv = 0;
    OneGraph g;
    for(int i = 0; i < s->n_graphs; i++) { gen_one(rng, *s, g); v += g.V; e += g.E; ss += (int64_t)g.sid.size(); p += (int64_t)g.pc.size(); pv += (int64_t)g.pv.size(); }
    if(tot_v) *tot_v = v; if(tot_e) *tot_e = e; if(tot_s) *tot_s = ss; if(tot_p) *tot_p = p; if(tot_pv) *tot_pv = pv;
    return ALD_OK;
}

int ald_synth_fill(const ald_synth_spec *s,
                   int32_t *g_nv, int32_t *g_ne, int32_t *g_np,
                   int32_t *vertex_offset, int32_t *edge_target, double *edge_weight, uint8_t *edge_strand,
                   double *edge_abd, int32_t *edge_sample_offset, int32_t *sample_id, double *sample_abd,
                   double *vertex_weight, int32_t *vertex_lpos, int32_t *vertex_rpos, int32_t *vertex_type,
                   int32_t *phasing_offset, int32_t *phasing_vertex, int32_t *phasing_count, char *graph_strand)
{
    if(!s) return ALD_ERR_INVALID;
    Rng rng(s->seed);
    OneGraph g;
    int64_t ov = 0, ovo = 0, oe = 0, oeo = 0, os = 0, op = 0, opo = 0, opv = 0;
    for(int i = 0; i < s->n_graphs; i++) {
        gen_one(rng, *s, g);
        g_nv[i] = g.V; g_ne[i] = g.E; g_np[i] = (int)g.pc.size(); graph_strand[i] = g.strand;
        memcpy(vertex_offset + ovo, g.voff.data(), sizeof(int32_t) * (g.V + 1));
        memcpy(edge_target + oe, g.etgt.data(), sizeof(int32_t) * g.E);
        memcpy(edge_weight + oe, g.ew.data(), sizeof(double) * g.E);
        memcpy(edge_strand + oe, g.estrand.data(), g.E);
        memcpy(edge_abd + oe, g.eabd.data(), sizeof(double) * g.E);
        memcpy(edge_sample_offset + oeo, g.esoff.data(), sizeof(int32_t) * (g.E + 1));
        memcpy(sample_id + os, g.sid.data(), sizeof(int32_t) * g.sid.size());
        memcpy(sample_abd + os, g.sabd.data(), sizeof(double) * g.sabd.size());
        memcpy(vertex_weight + ov, g.vw.data(), sizeof(double) * g.V);
        memcpy(vertex_lpos + ov, g.lpos.data(), sizeof(int32_t) * g.V);
        memcpy(vertex_rpos + ov, g.rpos.data(), sizeof(int32_t) * g.V);
        memcpy(vertex_type + ov, g.vtype.data(), sizeof(int32_t) * g.V);
        memcpy(phasing_offset + opo, g.poff.data(), sizeof(int32_t) * g.poff.size());
        if(!g.pv.empty()) memcpy(phasing_vertex + opv, g.pv.data(), sizeof(int32_t) * g.pv.size());
        if(!g.pc.empty()) memcpy(phasing_count + op, g.pc.data(), sizeof(int32_t) * g.pc.size());
        ov += g.V; ovo += g.V + 1; oe += g.E; oeo += g.E + 1; os += (int64_t)g.sid.size(); op += (int64_t)g.pc.size(); opo += (int64_t)g.pc.size() + 1; opv += (int64_t)g.pv.size();
    }
    return ALD_OK;
}

} // extern "C"
