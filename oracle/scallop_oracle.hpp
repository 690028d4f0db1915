/*
 * scallop_oracle.hpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the live part of the reference's per-bundle decomposition
 * ("Scallop core"), written from a reading of the reference sources; each function
 * cites the reference file:line it follows.  It deliberately keeps the reference's
 * container semantics (ordered sets / maps, list splices) so that it is an independent
 * statement of the algorithm, not a twin of the array-based HIP kernels it checks.
 *
 * PARITY PINNING: the reference has no golden vectors for scallop/router/hyper_set
 *  and its full hot path cannot be built in this image without
 * stand-ins for htslib / Boost.ICL / config.h, which the build rules forbid
 * (DESIGN.md "Oracle").  So for scallop/router/hyper_set this oracle is
 * **parity unpinned**.  What IS pinned against the real reference, compiled from its
 * own sources into oracle/_ref/ (oracle/Makefile):
 *   - subsetsum (scallop/subsetsum.cc)           -> oracle/subsetsum_oracle.hpp vs _ref/ref_subsetsum
 *   - graph-layer ordering / toposort / components (graph/*.cc) -> _ref/ref_graph, tests/golden/ref_graph.json
 *   - the result sink (rnacore/transcript_set.cc, gtf/transcript.cc) -> _ref/ref_tset, tests/golden/ref_tset.json
 *     (all three in tests/test_oracle_pins.py / tests/test_tset_cpu.py)
 * Corroboration, NOT a pin: the Router below reproduces 240 + 150 one-vertex cases computed by the reference's own scallop/router.cc
 * (oracle/_ref/ref_router, tests/golden/ref_router.json) -- but that binary takes seven splice_graph members from our driver
 * (oracle/ref_drivers/ref_router_main.cc), because rnacore/splice_graph.cc cannot be compiled here.
 * and, as a band only, the aggregates the survey measured on the real reference: paths per graph,
 * rule mix, router evaluations, graph growth (tests/test_oracle_pins.py).
 *
 * Canonical order: the reference orders edges by raw pointer; here
 * "pointer order" := creation order, and an edge's creation number IS its scallop
 * edge index (reference scallop.cc:24 get_edge_indices assigns indices in se order,
 * and every later gr.add_edge is followed at once by i2e.push_back:
 * scallop.cc:2262-2266, 2449/2479-2481, 1906-1911).
 */
#pragma once
#include <vector>
#include <set>
#include <map>
#include <tuple>
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <string>

namespace ora {

static const double SMIN = 0.00001;            // rnacore/splice_graph.h:18
enum { TRIVIAL = 0, NORMAL = 1, SPLITTABLE_SIMPLE = 2, SPLITTABLE_HYPER = 3, SPLITTABLE_PURE = 4,
       UNSPLITTABLE_SINGLE = 5, UNSPLITTABLE_MULTIPLE = 6, TRIVIAL_VERTEX = 7, SMALLEST_EDGE = 0 }; // util/constants.h:30-46
static const int EMPTY_VERTEX = -9;            // util/constants.h:53

struct AssertFail { int cls; int line; const char *what; };
#define ORA_ASSERT(cls, cond) do { if(!(cond)) throw ora::AssertFail{cls, __LINE__, #cond}; } while(0)
// assert classes mirror include/aletsch_decomp.h ALD_INV_*
enum { INV_WEIGHT = 1, INV_MERGE_EQUAL = 2, INV_COUNT = 3, INV_ROUTER = 4, INV_DEGREE = 5, INV_OTHER = 9 };

struct Params {                                 // util/parameters.cc:85-105
    double max_decompose_error_ratio[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00};
    double min_guaranteed_edge_weight = 0.01;
    double min_transcript_coverage = 2.0;
    int max_num_exons = 10000;
};

struct EdgeInfo {                               // rnacore/edge_info.h:14-35 (fields the path touches)
    int strand = 0;
    int count = 0;
    double confidence = 0;
    double abd = 0;
    std::set<int> samples;
    std::map<int, double> spAbd;
};

struct VertexInfo {                             // rnacore/vertex_info.h:20-42 (fields the path touches; the last eight only feed the feature block)
    int32_t lpos = 0, rpos = 0; int type = -1;
    double boundary_loss1 = 0, boundary_loss2 = 0, boundary_loss3 = 0, boundary_merged_loss = 0;
    int unbridge_leaving_count = 0; double unbridge_leaving_ratio = 0; int unbridge_coming_count = 0; double unbridge_coming_ratio = 0;
};

typedef std::tuple<int, int, int> EKey;        // (source, target, creation id): graph/edge_base.h:35-45

// ---------------------------------------------------------------------------------------------
// directed graph with the reference's iteration orders (graph/directed_graph.cc, graph_base.cc)
// ---------------------------------------------------------------------------------------------
struct Graph {
    std::vector<std::set<EKey>> si, so;        // vertex_base::si / so
    std::set<int> se;                          // graph_base::se  (creation order)
    std::vector<int> es, et;                   // endpoints by id
    std::vector<double> ewrt;                  // splice_graph::ewrt by id
    std::vector<EdgeInfo> einf;                // splice_graph::einf by id
    std::vector<double> vwrt;
    std::vector<VertexInfo> vinf;
    char strand = '.';
    int reads = 0, subgraph = 0;               // splice_graph::reads / subgraph (feature block only)

    int num_vertices() const { return (int)si.size(); }
    int num_edges() const { return (int)se.size(); }
    int add_vertex() { si.emplace_back(); so.emplace_back(); vwrt.push_back(0); vinf.emplace_back(); return 0; }
    int add_edge(int s, int t) {               // directed_graph.cc:38-48
        int id = (int)es.size();
        es.push_back(s); et.push_back(t); ewrt.push_back(0); einf.emplace_back();
        se.insert(id); so[s].insert(EKey(s, t, id)); si[t].insert(EKey(s, t, id));
        return id;
    }
    bool alive(int e) const { return e >= 0 && e < (int)es.size() && se.count(e) > 0; }
    void remove_edge(int e) {                  // directed_graph.cc:50-58
        if(!se.count(e)) return;
        so[es[e]].erase(EKey(es[e], et[e], e)); si[et[e]].erase(EKey(es[e], et[e], e)); se.erase(e);
    }
    void move_edge(int e, int x, int y) {      // directed_graph.cc:180-194
        so[es[e]].erase(EKey(es[e], et[e], e)); si[et[e]].erase(EKey(es[e], et[e], e));
        es[e] = x; et[e] = y;
        so[x].insert(EKey(x, y, e)); si[y].insert(EKey(x, y, e));
    }
    int in_degree(int v) const { return (int)si[v].size(); }
    int out_degree(int v) const { return (int)so[v].size(); }
    int degree(int v) const { return in_degree(v) + out_degree(v); }
    // directed_graph.cc:60-76: the stack-allocated probe compares above every heap pointer, so the
    // lookup lands on the NEWEST parallel (s,t) edge.
    int edge(int s, int t) const {
        int best = -1;
        auto it = so[s].lower_bound(EKey(s, t, INT_MIN));
        for(; it != so[s].end() && std::get<0>(*it) == s && std::get<1>(*it) == t; ++it) best = std::get<2>(*it);
        return best;
    }
    std::vector<int> in_edges(int v) const { std::vector<int> r; for(auto &k : si[v]) r.push_back(std::get<2>(k)); return r; }
    std::vector<int> out_edges(int v) const { std::vector<int> r; for(auto &k : so[v]) r.push_back(std::get<2>(k)); return r; }
    double get_in_weights(int v) const { double w = 0; for(auto &k : si[v]) w += ewrt[std::get<2>(k)]; return w; }   // splice_graph.cc:187-198
    double get_out_weights(int v) const { double w = 0; for(auto &k : so[v]) w += ewrt[std::get<2>(k)]; return w; }  // splice_graph.cc:174-185
    std::vector<int> get_strand_degree(int i) const {  // splice_graph.cc:1384-1406
        std::vector<int> vs(6, 0);
        for(auto &k : si[i]) vs[einf[std::get<2>(k)].strand]++;
        for(auto &k : so[i]) vs[einf[std::get<2>(k)].strand + 3]++;
        return vs;
    }
    bool mixed_strand_vertex(int i) const {    // splice_graph.cc:1375-1382
        std::vector<int> v = get_strand_degree(i);
        return (v[1] + v[4] >= 1) && (v[2] + v[5] >= 1);
    }
    std::vector<int> topological_sort() const { // directed_graph.cc:420-451
        std::vector<int> v, q, vd;
        for(int i = 0; i < num_vertices(); i++) { int d = in_degree(i); vd.push_back(d); if(d == 0) q.push_back(i); }
        size_t k = 0;
        while(k < q.size()) {
            int x = q[k++]; v.push_back(x);
            for(auto &key : so[x]) { int t = std::get<1>(key); vd[t]--; if(vd[t] == 0) q.push_back(t); }
        }
        return v;
    }
    // splice_graph.cc:819-885
    double compute_maximum_path_w(std::vector<int> &p) const {
        p.clear();
        int n = num_vertices(); int ss = 0, tt = n - 1;
        std::vector<double> table(n, -1); std::vector<int> back(n, -1);
        std::vector<int> tp = topological_sort();
        ORA_ASSERT(INV_OTHER, (int)tp.size() == n);
        int ssi = -1, tti = -1;
        for(int i = 0; i < n; i++) { if(tp[i] == ss) ssi = i; if(tp[i] == tt) tti = i; }
        ORA_ASSERT(INV_OTHER, ssi != -1 && tti != -1);
        table[ss] = DBL_MAX;
        for(int ii = ssi + 1; ii <= tti; ii++) {
            int i = tp[ii];
            if(degree(i) == 0) continue;
            double max_abd = 0; int max_edge = -1;
            for(auto &key : si[i]) {
                int e = std::get<2>(key); int s = std::get<0>(key);
                if(table[s] <= -1) continue;
                double xw = ewrt[e];
                double ww = xw < table[s] ? xw : table[s];
                if(ww >= max_abd) { max_abd = ww; max_edge = e; }
            }
            if(max_edge == -1) continue;
            back[i] = max_edge; table[i] = max_abd;
        }
        int x = tt;
        while(true) { int e = back[x]; if(e == -1) break; p.push_back(e); x = es[e]; }
        std::reverse(p.begin(), p.end());
        return table[tt];
    }
};

// ---------------------------------------------------------------------------------------------
// hyper_set (scallop/hyper_set.cc) -- phasing paths as edge-id lists + the edit API
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// what assembler::assemble(gx, px, sid) does before it builds the scallop object (meta/assembler.cc:1075-1086)
// PARITY UNPINNED: splice_graph.cc / graph_reviser.cc / essential.cc need config.h, Boost.ICL and htslib to build
// ---------------------------------------------------------------------------------------------
inline bool check_continuous_vertices(const Graph &gr, int x, int y) {       // essential.cc:436-446
    if(x >= y) return true;
    for(int i = x; i < y; i++) { if(gr.edge(i, i + 1) < 0) return false; if(gr.vinf[i].rpos != gr.vinf[i + 1].lpos) return false; }
    return true;
}
inline void extend_strands(Graph &gr) {                                      // splice_graph.cc:1338-1373
    for(int e : gr.se) {
        int sd = gr.einf[e].strand, s = gr.es[e], t = gr.et[e];
        int32_t p1 = gr.vinf[s].rpos, p2 = gr.vinf[t].lpos;
        if(p1 >= p2) continue;
        if(s + 2 != t) continue;
        double we = gr.ewrt[e], wv = gr.vwrt[s + 1];
        if(we <= wv) continue;
        if(gr.vinf[s + 1].lpos != p1) continue;
        if(gr.vinf[s + 1].rpos != p2) continue;
        int e1 = gr.edge(s, s + 1), e2 = gr.edge(s + 1, t);
        if(e1 >= 0) { if(gr.einf[e1].strand == 0) gr.einf[e1].strand = sd; }
        if(e2 >= 0) { if(gr.einf[e2].strand == 0) gr.einf[e2].strand = sd; }
    }
}
inline void group_start_boundaries(Graph &gr, std::map<int32_t, int32_t> &smap, int32_t max_group_boundary_distance) {     // graph_reviser.cc:916-991
    smap.clear();
    std::vector<int> v;
    for(int e : gr.out_edges(0)) v.push_back(gr.et[e]);
    if(v.size() <= 1) return;
    std::sort(v.begin(), v.end());
    int32_t p1 = gr.vinf[v[0]].lpos, p2 = p1; int k1 = v[0], k2 = k1;
    int pa = gr.edge(0, v[0]); ORA_ASSERT(INV_OTHER, pa >= 0);
    double wa = gr.ewrt[pa]; EdgeInfo ea = gr.einf[pa];
    for(size_t i = 1; i < v.size(); i++) {
        int32_t p = gr.vinf[v[i]].lpos;
        int pb = gr.edge(0, v[i]); ORA_ASSERT(INV_OTHER, pb >= 0);
        double wb = gr.ewrt[pb]; const EdgeInfo eb = gr.einf[pb];
        bool b = check_continuous_vertices(gr, k2, v[i]);
        ORA_ASSERT(INV_OTHER, p >= p2);
        if(p - p2 > max_group_boundary_distance) b = false;
        if(b == false) { p1 = p; p2 = p; k1 = v[i]; k2 = v[i]; pa = pb; wa = wb; ea = eb; }
        else {
            smap.insert(std::make_pair(p, p1));
            for(int j = k1; j < v[i]; j++) {
                int pc = gr.edge(j, j + 1); ORA_ASSERT(INV_OTHER, pc >= 0);
                double vc = gr.vwrt[j], wc = gr.ewrt[pc];
                gr.vwrt[j] = vc + wb;
                gr.einf[pc].count += eb.count;
                gr.ewrt[pc] = wc + wb;
            }
            wa += wb; ea.count += eb.count;
            gr.ewrt[pa] = wa; gr.einf[pa] = ea;
            gr.remove_edge(pb);
            k2 = v[i]; p2 = p;
        }
    }
}
inline void group_end_boundaries(Graph &gr, std::map<int32_t, int32_t> &tmap, int32_t max_group_boundary_distance) {       // graph_reviser.cc:993-1066
    tmap.clear();
    const int n = gr.num_vertices() - 1;
    std::vector<int> v;
    for(int e : gr.in_edges(n)) v.push_back(gr.es[e]);
    if(v.size() <= 1) return;
    std::sort(v.begin(), v.end(), std::greater<int>());
    int32_t p1 = gr.vinf[v[0]].rpos, p2 = p1; int k1 = v[0], k2 = k1;
    int pa = gr.edge(v[0], n); ORA_ASSERT(INV_OTHER, pa >= 0);
    double wa = gr.ewrt[pa];
    for(size_t i = 1; i < v.size(); i++) {
        int32_t p = gr.vinf[v[i]].rpos;
        int pb = gr.edge(v[i], n); ORA_ASSERT(INV_OTHER, pb >= 0);
        double wb = gr.ewrt[pb];
        bool b = check_continuous_vertices(gr, v[i], k2);
        ORA_ASSERT(INV_OTHER, p <= p2);
        if(p2 - p > max_group_boundary_distance) b = false;
        if(b == false) { p1 = p; p2 = p; k1 = v[i]; k2 = v[i]; pa = pb; wa = wb; }
        else {
            tmap.insert(std::make_pair(p, p1));
            for(int j = v[i]; j < k1; j++) {
                int pc = gr.edge(j, j + 1); ORA_ASSERT(INV_OTHER, pc >= 0);
                double wc = gr.ewrt[pc];
                gr.ewrt[pc] = wc + wb;
                gr.vwrt[j + 1] = wc + wb;                 // (yes: the edge's weight, not the vertex's -- graph_reviser.cc:1052)
            }
            wa += wb; gr.ewrt[pa] = wa;
            gr.remove_edge(pb);
            k2 = v[i]; p2 = p;
        }
    }
}
struct PhaseSet {                               // rnacore/phase_set.h:21-32
    std::map<std::vector<int32_t>, int> pmap;
    void add(const std::vector<int32_t> &v, int c) {                                                                    // phase_set.cc:12-25
        if(v.empty()) return;                                    // "error: adding empty vector to phase_set": printed and ignored (phase_set.cc:14-18), not an assert
        ORA_ASSERT(INV_OTHER, v.size() % 2 == 0);
        if(pmap.find(v) == pmap.end()) pmap.insert(std::make_pair(v, c)); else pmap[v] += c; }
    void project_boundaries(const std::map<int32_t, int32_t> &smap, const std::map<int32_t, int32_t> &tmap) {             // phase_set.cc:50-67
        PhaseSet ps;
        for(auto &x : pmap) {
            std::vector<int32_t> v = x.first; int c = x.second;
            auto is = smap.find(v.front()); auto it = tmap.find(v.back());
            if(is != smap.end()) v[0] = is->second;
            if(it != tmap.end()) v[v.size() - 1] = it->second;
            ps.add(v, c);
        }
        pmap = std::move(ps.pmap);
    }
};
inline bool build_path_from_exon_coordinates(const Graph &gr, const std::map<int32_t, int> &lindex, const std::map<int32_t, int> &rindex,
                                             const std::vector<int32_t> &v, std::vector<int> &vv) {                     // essential.cc:321-366
    vv.clear();
    if(v.size() <= 0) return true;
    int n = (int)v.size() / 2;
    std::vector<std::pair<int, int>> pp(n);
    for(int k = 0; k < n; k++) {
        int32_t p = v[2 * k + 0], q = v[2 * k + 1];
        if(p < 0 || q < 0) return false;
        if(p >= q) return false;
        if(lindex.find(p) == lindex.end()) return false;
        if(rindex.find(q) == rindex.end()) return false;
        pp[k].first = lindex.find(p)->second; pp[k].second = rindex.find(q)->second;
    }
    for(int k = 0; k < n; k++) {
        int a = pp[k].first, b = pp[k].second;
        if(a > b) return false;
        if(check_continuous_vertices(gr, a, b) == false) return false;
        for(int j = a; j <= b; j++) vv.push_back(j);
    }
    for(size_t i = 0; i + 1 < vv.size(); i++) ORA_ASSERT(INV_OTHER, vv[i] < vv[i + 1]);
    return true;
}
inline bool check_valid_path(const Graph &gr, const std::vector<int> &vv) {      // essential.cc:448-459
    int n = gr.num_vertices() - 1;
    for(size_t k = 0; k + 1 < vv.size(); k++) {
        if(vv[k] < 0 || vv[k] > n) return false;
        if(vv[k + 1] < 0 || vv[k + 1] > n) return false;
        if(gr.edge(vv[k], vv[k + 1]) < 0) return false;
    }
    return true;
}

struct HyperSet {
    std::map<std::vector<int>, int> nodes;      // MVII nodes (after ctor + filter_nodes: boundary input)
    HyperSet() {}
    HyperSet(const Graph &gr, const PhaseSet &ps) {       // hyper_set.cc:17-29; lindex / rindex: splice_graph.cc:1087-1099
        std::map<int32_t, int> lindex, rindex; int n = gr.num_vertices() - 1;
        for(int i = 0; i <= n; i++) { if(i != 0) lindex.insert(std::make_pair(gr.vinf[i].lpos, i)); if(i != n) rindex.insert(std::make_pair(gr.vinf[i].rpos, i)); }
        for(auto &x : ps.pmap) {
            std::vector<int> vv;
            if(build_path_from_exon_coordinates(gr, lindex, rindex, x.first, vv) == false) continue;
            for(size_t k = 0; k < vv.size(); k++) vv[k]--;
            add_node_list(vv, x.second);
        }
    }
    void add_node_list(const std::vector<int> &s, int c, int o = 1) {   // hyper_set.cc:40-48
        std::vector<int> v = s; std::sort(v.begin(), v.end());
        for(size_t i = 0; i < v.size(); i++) v[i] += o;
        if(nodes.find(v) == nodes.end()) nodes.insert(std::make_pair(v, c)); else nodes[v] += c;
    }
    void filter_nodes(const Graph &gr) {        // hyper_set.cc:356-371
        std::map<std::vector<int>, int> mv;
        for(auto &x : nodes) { if(x.first.size() <= 1) continue; if(check_valid_path(gr, x.first) == false) continue; mv.insert(x); }
        nodes = mv;
    }
    std::vector<std::vector<int>> edges;
    std::vector<int> ecnts;
    std::map<int, std::set<int>> e2s;

    void build(const Graph &gr) { build_edges(gr); build_index(); }   // hyper_set.cc:316-321
    void build_edges(const Graph &gr) {         // hyper_set.cc:323-354 (e2i is the identity here)
        edges.clear();
        for(auto &kv : nodes) {
            int c = kv.second;
            if(c <= 1) continue;
            const std::vector<int> &vv = kv.first;
            if(vv.size() <= 1) continue;
            std::vector<int> ve; bool b = true;
            for(size_t k = 0; k + 1 < vv.size(); k++) {
                ORA_ASSERT(INV_OTHER, vv[k] < vv[k + 1]);
                int e = gr.edge(vv[k], vv[k + 1]);
                if(e < 0) { b = false; ve.push_back(-1); } else ve.push_back(e);
            }
            if(b && ve.size() >= 2) { edges.push_back(ve); ecnts.push_back(c); }
        }
    }
    void build_index() {                        // hyper_set.cc:436-459
        e2s.clear();
        for(int i = 0; i < (int)edges.size(); i++) for(int e : edges[i]) { if(e == -1) continue; e2s[e].insert(i); }
    }
    std::set<int> get_intersection(const std::vector<int> &v) {   // hyper_set.cc:489-507
        std::set<int> ss;
        if(v.empty()) return ss;
        if(!e2s.count(v[0])) return ss;
        ss = e2s[v[0]];
        for(size_t i = 1; i < v.size(); i++) {
            if(!e2s.count(v[i])) return std::set<int>();
            const std::set<int> &s = e2s[v[i]];
            std::set<int> r;
            std::set_intersection(ss.begin(), ss.end(), s.begin(), s.end(), std::inserter(r, r.begin()));
            ss = r;
        }
        return ss;
    }
    std::map<int, int> get_successors(int e) { // hyper_set.cc:509-529
        std::map<int, int> s;
        if(!e2s.count(e)) return s;
        for(int k : e2s[e]) {
            std::vector<int> &v = edges[k]; int c = ecnts[k];
            for(size_t i = 0; i < v.size(); i++) {
                if(v[i] != e) continue;
                if(i + 1 >= v.size()) continue;
                int x = v[i + 1];
                if(x == -1) continue;
                s[x] += c;
            }
        }
        return s;
    }
    std::map<std::pair<int, int>, int> get_routes(int x, const Graph &gr) {   // hyper_set.cc:553-571
        std::map<std::pair<int, int>, int> mpi;
        for(int e : gr.in_edges(x)) { auto s = get_successors(e); for(auto &kv : s) mpi.insert({{e, kv.first}, kv.second}); }
        return mpi;
    }
    static std::vector<int> consecutive_subset(const std::vector<int> &ref, const std::vector<int> &x) {  // util/util.h:142-162
        std::vector<int> v;
        if(x.empty() || ref.empty() || x.size() > ref.size()) return v;
        for(size_t i = 0; i + x.size() <= ref.size(); i++) {
            if(ref[i] != x[0]) continue;
            bool b = true;
            for(size_t j = 0; j < x.size(); j++) if(x[j] != ref[j + i]) { b = false; break; }
            if(b) v.push_back((int)i);
        }
        return v;
    }
    void replace(int x, int e) { replace(std::vector<int>{x}, e); }                // hyper_set.cc:609-615
    void replace(int x, int y, int e) { replace(std::vector<int>{x, y}, e); }      // hyper_set.cc:617-624
    void replace(const std::vector<int> &v, int e) {                               // hyper_set.cc:626-675
        if(v.empty()) return;
        std::set<int> s = get_intersection(v);
        std::vector<int> fb;
        for(int k : s) {
            std::vector<int> &vv = edges[k];
            std::vector<int> bv = consecutive_subset(vv, v);
            if(bv.empty()) continue;
            std::sort(bv.begin(), bv.end());
            for(int j = (int)bv.size() - 1; j >= 0; j--) {
                int b = bv[j];
                vv[b] = e;
                vv.erase(vv.begin() + b + 1, vv.begin() + b + v.size());
            }
            fb.push_back(k);
            e2s[e].insert(k);
        }
        if(v.size() != 1) return;       // asymmetric index maintenance (hyper_set.cc:665), kept as is
        for(int u : v) {
            if(!e2s.count(u)) continue;
            for(int k : fb) e2s[u].erase(k);
            if(e2s[u].empty()) e2s.erase(u);
        }
    }
    void remove(int e) {                        // hyper_set.cc:787-818
        if(!e2s.count(e)) return;
        for(int k : e2s[e]) { std::vector<int> &vv = edges[k]; for(size_t i = 0; i < vv.size(); i++) if(vv[i] == e) vv[i] = -1; }
        e2s.erase(e);
    }
    void remove_pair(int x, int y) { insert_between(x, y, -1); }                   // hyper_set.cc:820-823
    void insert_between(int x, int y, int e) {  // hyper_set.cc:865-902
        if(!e2s.count(x)) return;
        std::set<int> s = e2s[x];
        for(int k : s) {
            std::vector<int> &vv = edges[k];
            for(size_t i = 0; i < vv.size(); i++) {
                if(i == vv.size() - 1) continue;
                if(vv[i] != x) continue;
                if(vv[i + 1] != y) continue;
                vv.insert(vv.begin() + i + 1, e);
                if(e == -1) continue;
                e2s[e].insert(k);
            }
        }
    }
    bool left_extend(int e) {                   // hyper_set.cc:949-965
        if(!e2s.count(e)) return false;
        for(int k : e2s[e]) { std::vector<int> &vv = edges[k]; for(size_t i = 1; i < vv.size(); i++) if(vv[i] == e && vv[i - 1] != -1) return true; }
        return false;
    }
    bool right_extend(int e) {                  // hyper_set.cc:967-983
        if(!e2s.count(e)) return false;
        for(int k : e2s[e]) { std::vector<int> &vv = edges[k]; for(size_t i = 0; i + 1 < vv.size(); i++) if(vv[i] == e && vv[i + 1] != -1) return true; }
        return false;
    }
    bool left_dominate(int e) {                 // hyper_set.cc:1003-1042
        if(!e2s.count(e)) return true;
        std::set<std::pair<int, int>> x1, x2;
        for(int k : e2s[e]) {
            std::vector<int> &vv = edges[k];
            for(int i = 0; i + 1 < (int)vv.size(); i++) {
                if(vv[i] != e) continue;
                if(vv[i + 1] == -1) continue;
                if(i == 0 || vv[i - 1] == -1) {
                    if(i + 2 < (int)vv.size()) x1.insert({vv[i + 1], vv[i + 2]}); else x1.insert({vv[i + 1], -1});
                } else {
                    x2.insert({vv[i + 1], -1});
                    if(i + 2 < (int)vv.size()) x2.insert({vv[i + 1], vv[i + 2]});
                }
            }
        }
        for(auto &p : x1) if(!x2.count(p)) return false;
        return true;
    }
    bool right_dominate(int e) {                // hyper_set.cc:1044-1082
        if(!e2s.count(e)) return true;
        std::set<std::pair<int, int>> x1, x2;
        for(int k : e2s[e]) {
            std::vector<int> &vv = edges[k];
            for(int i = 1; i < (int)vv.size(); i++) {
                if(vv[i] != e) continue;
                if(vv[i - 1] == -1) continue;
                if(i == (int)vv.size() - 1 || vv[i + 1] == -1) {
                    if(i - 2 >= 0) x1.insert({vv[i - 1], vv[i - 2]}); else x1.insert({vv[i - 1], -1});
                } else {
                    x2.insert({vv[i - 1], -1});
                    if(i - 2 >= 0) x2.insert({vv[i - 1], vv[i - 2]});
                }
            }
        }
        for(auto &p : x1) if(!x2.count(p)) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// small undirected multigraph with the reference's orders (graph/undirected_graph.cc)
// ---------------------------------------------------------------------------------------------
struct UGraph {
    std::vector<std::set<EKey>> so;
    std::set<int> se; std::vector<int> es, et;
    void clear() { so.clear(); se.clear(); es.clear(); et.clear(); }
    void add_vertex() { so.emplace_back(); }
    int num_vertices() const { return (int)so.size(); }
    int num_edges() const { return (int)se.size(); }
    int degree(int v) const { return (int)so[v].size(); }
    int add_edge(int s, int t) {               // undirected_graph.cc:38-48
        ORA_ASSERT(INV_ROUTER, s >= 0 && s < num_vertices());
        ORA_ASSERT(INV_ROUTER, t >= 0 && t < num_vertices());
        int id = (int)es.size(); es.push_back(s); et.push_back(t);
        se.insert(id); so[s].insert(EKey(s, t, id)); so[t].insert(EKey(s, t, id));
        return id;
    }
    void remove_edge(int e) { so[es[e]].erase(EKey(es[e], et[e], e)); so[et[e]].erase(EKey(es[e], et[e], e)); se.erase(e); }
    void clear_vertex(int x) {                 // graph_base.cc:51-69
        std::vector<int> v; for(auto &k : so[x]) v.push_back(std::get<2>(k));
        for(int e : v) remove_edge(e);
    }
    int neighbor(int e, int x) const { return es[e] == x ? et[e] : es[e]; }   // edge_base.cc
    std::vector<int> bfs(int s) const {        // graph_base.cc bfs(s, v): visit order irrelevant to callers (sets)
        std::vector<int> v; std::vector<char> closed(num_vertices(), 0); std::vector<int> open{s}; closed[s] = 1; size_t p = 0;
        while(p < open.size()) {
            int x = open[p++]; v.push_back(x);
            for(auto &k : so[x]) { int y = neighbor(std::get<2>(k), x); if(closed[y]) continue; closed[y] = 1; open.push_back(y); }
        }
        return v;
    }
    std::vector<std::set<int>> compute_connected_components() const {   // undirected_graph.cc:134-151
        std::vector<char> m(num_vertices(), 0); std::vector<std::set<int>> vv;
        for(int i = 0; i < num_vertices(); i++) {
            if(m[i]) continue;
            std::vector<int> v = bfs(i);
            vv.emplace_back(v.begin(), v.end());
            for(int x : v) m[x] = 1;
        }
        return vv;
    }
};

typedef std::map<std::pair<int, int>, double> MPID;
typedef std::map<std::pair<int, int>, int> MPII;

// ---------------------------------------------------------------------------------------------
// router (scallop/router.cc): live part = classify_plain_vertex + thread
// ---------------------------------------------------------------------------------------------
struct Router {
    int root; Graph &gr; const Params &cfg;
    std::vector<std::pair<int, int>> routes; std::vector<int> counts;
    std::map<int, int> e2u; std::vector<int> u2e; std::map<int, double> u2w; UGraph ug;
    int type = -1, degree = -1; double ratio = 0; MPID pe2w; std::map<int, double> econf;

    Router(int r, Graph &g, const MPII &mpi, const Params &c) : root(r), gr(g), cfg(c) {   // router.cc:25-36
        for(auto &kv : mpi) { routes.push_back(kv.first); counts.push_back(kv.second); }
    }
    void classify() {                           // router.cc:61-81
        type = -1; degree = 0;
        ORA_ASSERT(INV_ROUTER, gr.in_degree(root) >= 1 && gr.out_degree(root) >= 1);
        build_indices();
        ORA_ASSERT(INV_ROUTER, !gr.mixed_strand_vertex(root));   // router.cc:71-76 assert(false)
        classify_plain_vertex();
    }
    void build_indices() {                      // router.cc:225-248
        e2u.clear(); u2e.clear();
        for(int e : gr.in_edges(root)) { e2u.insert({e, (int)e2u.size()}); u2e.push_back(e); }
        for(int e : gr.out_edges(root)) { e2u.insert({e, (int)e2u.size()}); u2e.push_back(e); }
    }
    void build_bipartite_graph() {              // router.cc:250-325
        ug.clear(); u2w.clear();
        for(size_t i = 0; i < u2e.size(); i++) ug.add_vertex();
        std::vector<int> left, right;
        int l = gr.in_degree(root);
        for(int i = 0; i < (int)u2e.size(); i++) {
            if(gr.einf[u2e[i]].count == 0) continue;      // "Warning!(count = 0)" (router.cc:269,276)
            if(i < l) left.push_back(i); else right.push_back(i);
        }
        for(size_t i = 0; i < routes.size(); i++) {
            int e1 = routes[i].first, e2 = routes[i].second;
            ORA_ASSERT(INV_ROUTER, e2u.count(e1) && e2u.count(e2));
            int s = e2u[e1], t = e2u[e2];
            ORA_ASSERT(INV_ROUTER, s >= 0 && s < gr.in_degree(root));
            ORA_ASSERT(INV_ROUTER, t >= gr.in_degree(root) && t < gr.degree(root));
            int e = ug.add_edge(s, t);
            u2w[e] = counts[i];
        }
        std::vector<int> v1, v2;
        for(int i : left) if(ug.degree(i) == 0) v1.push_back(i);
        thread_left_isolate(v1, right);
        for(int i : right) if(ug.degree(i) == 0) v2.push_back(i);
        thread_right_isolate(v2, left);
    }
    double common_abd(const EdgeInfo &a, const EdgeInfo &b) {   // router.cc:1035-1038 / 1096-1099
        double c = 0;
        std::vector<int> common;
        std::set_intersection(a.samples.begin(), a.samples.end(), b.samples.begin(), b.samples.end(), std::back_inserter(common));
        for(int sp : common) { double x = a.spAbd.at(sp), y = b.spAbd.at(sp); c += 0.99 * std::min(x, y) + 0.01 * std::max(x, y); }
        return c;
    }
    void thread_left_isolate(std::vector<int> &left_iso, std::vector<int> &right_all) {   // router.cc:1010-1069
        for(int v : left_iso) {
            int le = u2e[v];
            int partner = -1; double max_abd = 0.0, sum_abd = 0.0;
            for(int r : right_all) {
                double c = common_abd(gr.einf[le], gr.einf[u2e[r]]);
                sum_abd += c;
                if(c > max_abd) { max_abd = c; partner = r; }
            }
            int e = ug.add_edge(v, partner);     // asserts on partner == -1 (undirected_graph.cc:40-41)
            u2w[e] = max_abd;
            econf[le] = log(max_abd / sum_abd);
        }
    }
    void thread_right_isolate(std::vector<int> &right_iso, std::vector<int> &left_all) {  // router.cc:1071-1129
        for(int v : right_iso) {
            int re = u2e[v];
            int partner = -1; double max_abd = 0, sum_abd = 0.0;
            for(int l : left_all) {
                double c = common_abd(gr.einf[u2e[l]], gr.einf[re]);
                sum_abd += c;
                if(c > max_abd) { max_abd = c; partner = l; }
            }
            int e = ug.add_edge(partner, v);
            u2w[e] = max_abd;
            econf[re] = log(max_abd / sum_abd);
        }
    }
    bool one_side_connected() {                 // router.cc:173-191
        std::vector<std::set<int>> cc = ug.compute_connected_components();
        std::vector<int> v(ug.num_vertices(), -1);
        for(int c = 0; c < (int)cc.size(); c++) for(int x : cc[c]) v[x] = c;
        bool b1 = true, b2 = true;
        for(int i = 1; i < gr.in_degree(root); i++) if(v[i] != v[0]) b1 = false;
        for(int i = gr.in_degree(root) + 1; i < gr.degree(root); i++) if(v[i] != v[gr.in_degree(root)]) b2 = false;
        return b1 || b2;
    }
    void classify_plain_vertex() {              // router.cc:116-171
        build_bipartite_graph();
        if(gr.in_degree(root) == 1 || gr.out_degree(root) == 1) { type = TRIVIAL; degree = gr.degree(root); return; }
        for(int i = 0; i < ug.num_vertices(); i++) ORA_ASSERT(INV_ROUTER, ug.degree(i) >= 1);
        std::vector<std::set<int>> vv = ug.compute_connected_components();
        if(vv.size() == 1) { type = UNSPLITTABLE_SINGLE; degree = ug.num_edges() - ug.num_vertices() + (int)vv.size() + (int)vv.size(); return; }
        ORA_ASSERT(INV_ROUTER, !one_side_connected());          // router.cc:148-154 assert(false)
        int a = 0, b = 0; type = SPLITTABLE_PURE;
        for(auto &c : vv) { if(c.size() == 1) a++; if(c.size() >= 2) b++; }
        ORA_ASSERT(INV_ROUTER, b >= 1);
        degree = b - 1 + (a + 1) / 2;
    }
    void build() {                              // router.cc:193-223
        ORA_ASSERT(INV_ROUTER, type == UNSPLITTABLE_SINGLE || type == SPLITTABLE_PURE);
        thread();
        for(auto &kv : pe2w) if(kv.second < cfg.min_guaranteed_edge_weight) kv.second = cfg.min_guaranteed_edge_weight;
    }
    std::vector<double> compute_balanced_weights_components() {  // router.cc:1248-1275
        std::vector<std::set<int>> vv = ug.compute_connected_components();
        std::vector<double> vw(u2e.size(), 0.0);
        for(auto &cc : vv) {
            double sum1 = 0, sum2 = 0;
            for(int i : cc) { double w = gr.ewrt[u2e[i]]; if(i < gr.in_degree(root)) sum1 += w; else sum2 += w; vw[i] = w; }
            double r1 = sqrt(sum2 / sum1), r2 = sqrt(sum1 / sum2);
            for(int i : cc) { if(i < gr.in_degree(root)) vw[i] *= r1; else vw[i] *= r2; }
        }
        return vw;
    }
    void thread() {                             // router.cc:738-857
        pe2w.clear();
        std::vector<double> vw = compute_balanced_weights_components();
        double weight_sum = 0;
        for(double w : vw) weight_sum += w;
        while(true) {
            if(thread_leaf(vw)) continue;
            if(!thread_turn(vw)) break;
        }
        ORA_ASSERT(INV_ROUTER, ug.num_edges() == 0);
        double weight_remain = 0;
        for(double w : vw) { if(w <= 0) continue; weight_remain += w; }
        ratio = weight_remain / weight_sum;
        for(auto &kv : econf) gr.einf[kv.first].confidence += kv.second;    // router.cc:849-855: side effect on EVERY build()
    }
    bool thread_leaf(std::vector<double> &vw) { // router.cc:859-897
        for(int e : ug.se) {
            int s = ug.es[e], t = ug.et[e];
            if(s >= t) std::swap(s, t);
            if(vw[s] < -0.5) continue;
            if(vw[t] < -0.5) continue;
            if(ug.degree(s) == 1 && vw[s] <= vw[t]) {
                pe2w.insert({{u2e[s], u2e[t]}, vw[s]});
                ug.clear_vertex(s); vw[t] -= vw[s]; vw[s] = -1; return true;
            }
            if(ug.degree(t) == 1 && vw[t] <= vw[s]) {
                pe2w.insert({{u2e[s], u2e[t]}, vw[t]});
                ug.clear_vertex(t); vw[s] -= vw[t]; vw[t] = -1; return true;
            }
        }
        return false;
    }
    bool thread_turn(std::vector<double> &vw) { // router.cc:899-936
        int x = -1;
        for(int k = 0; k < (int)vw.size(); k++) {
            if(vw[k] < -0.5) continue;
            if(ug.degree(k) <= 1) continue;
            if(x != -1 && vw[k] > vw[x]) continue;
            x = k;
        }
        if(x == -1) return false;
        double sum = 0;
        std::vector<int> oe; for(auto &k : ug.so[x]) oe.push_back(std::get<2>(k));
        for(int e : oe) { int t = ug.neighbor(e, x); sum += u2w[e]; ORA_ASSERT(INV_ROUTER, vw[t] >= vw[x]); }
        for(int e : oe) {
            int t = ug.neighbor(e, x);
            double w = vw[x] * u2w[e] / sum;
            std::pair<int, int> p = (x < t) ? std::make_pair(u2e[x], u2e[t]) : std::make_pair(u2e[t], u2e[x]);
            pe2w.insert({p, w});
            vw[t] -= w;
        }
        vw[x] = -1;
        ug.clear_vertex(x);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// outputs
// ---------------------------------------------------------------------------------------------
struct Path {                                   // rnacore/path.h
    std::vector<int> v; std::vector<std::pair<int, int>> junc;
    int length = 0; double abd = 0, weight = 0, conf = 0, reads = 0; char strand = '.'; int count = 0;
};
struct Features {                               // transcript::TrstFeatures (gtf/transcript.h:60-104), same order
    int gr_vertices = 0, gr_edges = 0, gr_reads = 0, gr_subgraph = 0, num_vertices = 0, num_edges = 0; double junc_ratio = 0; int max_mid_exon_len = 0;
    double start_loss1 = 0, start_loss2 = 0, start_loss3 = 0, end_loss1 = 0, end_loss2 = 0, end_loss3 = 0, start_merged_loss = 0, end_merged_loss = 0;
    int introns = 0, start_introns = 0, end_introns = 0; double intron_ratio = 0, start_intron_ratio = 0, end_intron_ratio = 0; int uni_junc = 0;
    double seq_min_wt = 0; int seq_min_cnt = 0; double seq_min_abd = 0, seq_min_ratio = 0, seq_max_wt = 0; int seq_max_cnt = 0; double seq_max_abd = 0, seq_max_ratio = 0;
    int unbridge_start_coming_count = 0; double unbridge_start_coming_ratio = 0; int unbridge_end_leaving_count = 0; double unbridge_end_leaving_ratio = 0;
    int start_cnt = 0; double start_weight = 0, start_abd = 0; int end_cnt = 0; double end_weight = 0, end_abd = 0;
    bool complete = false;                      // false: the reference returned early (no junction) and left the rest indeterminate
};
struct Transcript {                             // the fields build_transcript fills (essential.cc:719-748)
    double coverage = 0, conf = 0, abd = 0; int count1 = 0; char strand = '.';
    std::vector<std::pair<int32_t, int32_t>> exons;
    Features features;
};
struct TraceEvent { int code, a, b; double val; };
enum { OP_BROKEN = 1, OP_TRIVIAL_FAST = 2, OP_TRIVIAL_NOW = 3, OP_TRIVIAL_BEST = 4, OP_SMALL_NOW = 5, OP_SMALLEST = 6,
       OP_UNSPLIT_NOW = 7, OP_UNSPLIT_BEST = 8, OP_GREEDY = 9, OP_COLLECT = 10 };

struct Stats { int max_live_edges = 0, max_vertices = 0, total_edge_ids = 0, iterations = 0, router_builds = 0, max_mev = 0, cut_short = 0; };

// ---------------------------------------------------------------------------------------------
// scallop (scallop/scallop.cc)
// ---------------------------------------------------------------------------------------------
struct Scallop {
    const Params &cfg; Graph &gr; HyperSet &hs;
    std::vector<std::vector<int>> mev; std::vector<double> med; std::vector<int> mei;
    std::vector<int> v2v; std::set<int> nonzeroset;
    std::vector<Path> paths; std::vector<Transcript> trsts;
    std::vector<TraceEvent> *trace = nullptr; Stats st;
    Graph gr_ori;
    int feature_assert = 0;                     // line of the first assert update_trst_features would have hit (0: none); the paths stand either way

    Scallop(Graph &g, HyperSet &h, const Params &c) : cfg(c), gr(g), hs(h) {       // scallop.cc:19-32
        hs.build(gr);
        // init_super_edges / init_inner_weights (weight copy only feeds dead fields) / init_vertex_map / init_nonzeroset: 1621-1673
        mev.assign(gr.es.size(), std::vector<int>()); med.assign(gr.es.size(), 0); mei.assign(gr.es.size(), 0);
        for(int i = 0; i < gr.num_vertices(); i++) v2v.push_back(i);
        for(int i = 1; i < gr.num_vertices() - 1; i++) { if(gr.degree(i) <= 0) continue; nonzeroset.insert(i); }
    }
    void ev(int code, int a, int b, double v) { if(trace) trace->push_back({code, a, b, v}); st.iterations++; }
    void track() { st.max_live_edges = std::max(st.max_live_edges, gr.num_edges()); st.max_vertices = std::max(st.max_vertices, gr.num_vertices()); st.total_edge_ids = (int)gr.es.size(); }
    void grow_edge_maps() { size_t n = gr.es.size(); if(mev.size() < n) { mev.resize(n); med.resize(n, 0); mei.resize(n, 0); } }

    int assemble() {                            // scallop.cc:38-188
        gr_ori = gr;
        while(true) {
            if(gr.num_vertices() > cfg.max_num_exons) { st.cut_short = 1; break; }     // also when the graph GROWS past the limit mid-run
            track();
            if(resolve_broken_vertex()) continue;
            if(resolve_trivial_vertex_fast(cfg.max_decompose_error_ratio[TRIVIAL_VERTEX])) continue;
            if(resolve_trivial_vertex(1, true, cfg.max_decompose_error_ratio[TRIVIAL_VERTEX])) continue;
            if(resolve_smallest_edges(cfg.max_decompose_error_ratio[SMALLEST_EDGE])) continue;
            if(resolve_unsplittable_vertex(UNSPLITTABLE_SINGLE, 1, 0.01)) continue;
            if(resolve_unsplittable_vertex(SPLITTABLE_PURE, 1, 0.01)) continue;
            if(resolve_unsplittable_vertex(UNSPLITTABLE_SINGLE, INT_MAX, cfg.max_decompose_error_ratio[UNSPLITTABLE_SINGLE])) continue;
            if(resolve_unsplittable_vertex(SPLITTABLE_PURE, INT_MAX, cfg.max_decompose_error_ratio[SPLITTABLE_PURE])) continue;
            if(resolve_unsplittable_vertex(UNSPLITTABLE_SINGLE, INT_MAX, DBL_MAX)) continue;
            if(resolve_unsplittable_vertex(SPLITTABLE_PURE, INT_MAX, DBL_MAX)) continue;
            if(resolve_trivial_vertex(2, true, cfg.max_decompose_error_ratio[TRIVIAL_VERTEX])) continue;
            break;
        }
        track();
        collect_existing_st_paths();
        greedy_decompose();
        track();
        build_transcripts();
        return 0;
    }

    bool resolve_broken_vertex() {              // scallop.cc:190-236
        std::vector<int> vv(nonzeroset.begin(), nonzeroset.end());
        int x = -1;
        for(int i : vv) {
            if(i == 0) continue;
            if(i == gr.num_vertices() - 1) continue;
            if(gr.in_degree(i) >= 1 && gr.out_degree(i) >= 1) continue;
            x = i; break;
        }
        if(x == -1) return false;
        std::vector<int> ve = gr.in_edges(x);
        for(int e : gr.out_edges(x)) ve.push_back(e);
        ORA_ASSERT(INV_OTHER, ve.size() >= 1);
        ev(OP_BROKEN, x, (int)ve.size(), 0);
        for(int e : ve) { remove_edge(e); hs.remove(e); }
        ORA_ASSERT(INV_DEGREE, gr.degree(x) == 0);
        nonzeroset.erase(x);
        return true;
    }

    bool resolve_single_trivial_vertex(int i, double jump_ratio) {   // scallop.cc:1236-1254
        if(gr.in_degree(i) <= 0) return false;
        if(gr.out_degree(i) <= 0) return false;
        if(gr.in_degree(i) >= 2 && gr.out_degree(i) >= 2) return false;
        if(gr.mixed_strand_vertex(i)) return false;
        if(classify_trivial_vertex(i, false) != 1) return false;
        double r = compute_balance_ratio(i);
        if(r >= jump_ratio) return false;
        ev(OP_TRIVIAL_FAST, i, 0, r);
        decompose_trivial_vertex(i);
        ORA_ASSERT(INV_DEGREE, gr.degree(i) == 0);
        return true;
    }
    bool resolve_trivial_vertex_fast(double jump_ratio) {            // scallop.cc:1256-1270
        bool flag = false;
        std::vector<int> vv(nonzeroset.begin(), nonzeroset.end());
        for(int i : vv) if(resolve_single_trivial_vertex(i, jump_ratio)) flag = true;
        return flag;
    }
    bool resolve_trivial_vertex(int type, bool fast, double jump_ratio) {   // scallop.cc:1180-1234
        int root = -1; double ratio = DBL_MAX; bool flag = false;
        std::vector<int> vv(nonzeroset.begin(), nonzeroset.end());
        for(int i : vv) {
            if(gr.in_degree(i) <= 0) continue;
            if(gr.out_degree(i) <= 0) continue;
            if(gr.mixed_strand_vertex(i)) continue;
            if(gr.in_degree(i) >= 2 && gr.out_degree(i) >= 2) continue;
            if(classify_trivial_vertex(i, fast) != type) continue;
            double r = compute_balance_ratio(i);
            if(r < 1.02) {
                ev(OP_TRIVIAL_NOW, i, type, r);
                decompose_trivial_vertex(i);
                flag = true;
                continue;
            }
            if(ratio < r) continue;
            root = i; ratio = r;
            if(ratio < jump_ratio) break;
        }
        if(flag) return true;
        if(root == -1) return false;
        ev(OP_TRIVIAL_BEST, root, type, ratio);
        decompose_trivial_vertex(root);
        ORA_ASSERT(INV_DEGREE, gr.degree(root) == 0);
        return true;
    }
    int compute_smallest_in_edge(int x, double &ratio) {   // scallop.cc:2967-2986
        int e = -1; double sum1 = 0, minw = DBL_MAX;
        for(int id : gr.in_edges(x)) { double w = gr.ewrt[id]; sum1 += w; if(w > minw) continue; minw = w; e = id; }
        if(e == -1) return -1;
        ORA_ASSERT(INV_WEIGHT, sum1 >= SMIN);
        ratio = minw / sum1; return e;
    }
    int compute_smallest_out_edge(int x, double &ratio) {  // scallop.cc:2988-3007
        int e = -1; double sum1 = 0, minw = DBL_MAX;
        for(int id : gr.out_edges(x)) { double w = gr.ewrt[id]; sum1 += w; if(w > minw) continue; minw = w; e = id; }
        if(e == -1) return -1;
        ORA_ASSERT(INV_WEIGHT, sum1 >= SMIN);
        ratio = minw / sum1; return e;
    }
    int compute_smallest_edge(int x, double &ratio) {      // scallop.cc:3009-3030
        double r1, r2;
        int e1 = compute_smallest_in_edge(x, r1), e2 = compute_smallest_out_edge(x, r2);
        if(e1 < 0 || e2 < 0) return -1;
        if(r1 < r2) { ratio = r1; return e1; }
        ratio = r2; return e2;
    }
    bool resolve_smallest_edges(double max_ratio) {        // scallop.cc:844-945
        int se = -1, root = -1; double ratio = max_ratio; bool flag = false;
        std::vector<int> vv(nonzeroset.begin(), nonzeroset.end());
        for(int i : vv) {
            if(gr.in_degree(i) <= 1) continue;
            if(gr.out_degree(i) <= 1) continue;
            double r; int e = compute_smallest_edge(i, r);
            if(e == -1) continue;
            int s = gr.es[e], t = gr.et[e];
            ORA_ASSERT(INV_OTHER, s == i || t == i);
            if(gr.out_degree(s) <= 1) continue;
            if(gr.in_degree(t) <= 1) continue;
            if(hs.right_extend(e) && hs.left_extend(e)) continue;
            if(t == i && hs.right_extend(e)) continue;
            if(s == i && hs.left_extend(e)) continue;
            std::vector<int> vs = gr.get_strand_degree(i);
            int z = gr.einf[e].strand;
            if(s == i && z >= 1 && vs[0] + vs[z + 0] <= 1) continue;
            if(t == i && z >= 1 && vs[3] + vs[z + 3] <= 1) continue;
            if(r < 0.01) {
                ev(OP_SMALL_NOW, e, i, r);
                remove_edge(e); hs.remove(e);
                flag = true; continue;
            }
            if(ratio < r) continue;
            ratio = r; se = e; root = i;
        }
        if(flag) return true;
        if(se == -1) return false;
        ev(OP_SMALLEST, se, root, ratio);
        remove_edge(se); hs.remove(se);
        return true;
    }
    bool resolve_unsplittable_vertex(int type, int degree, double max_ratio) {   // scallop.cc:1004-1060
        int root = -1; MPID pe2w; double ratio = max_ratio; bool flag = false;
        std::vector<int> vv(nonzeroset.begin(), nonzeroset.end());
        for(int i : vv) {
            if(gr.in_degree(i) <= 1) continue;
            if(gr.out_degree(i) <= 1) continue;
            MPII mpi = hs.get_routes(i, gr);
            Router rt(i, gr, mpi, cfg);
            rt.classify();
            if(rt.type != type) continue;
            if(rt.degree > degree) continue;
            rt.build(); st.router_builds++;
            if(rt.ratio < 0.01) {
                ev(OP_UNSPLIT_NOW, i, type, rt.ratio);
                decompose_vertex_extend(i, rt.pe2w);
                flag = true; continue;
            }
            if(rt.ratio > ratio) continue;
            root = i; ratio = rt.ratio; pe2w = rt.pe2w;
        }
        if(flag) return true;
        if(root == -1) return false;
        ev(OP_UNSPLIT_BEST, root, type, ratio);
        decompose_vertex_extend(root, pe2w);
        return true;
    }

    void decompose_vertex_extend(int root, MPID &pe2w) {   // scallop.cc:1675-1986
        std::map<int, int> mdegree;
        for(auto &kv : pe2w) { mdegree[kv.first.first]++; mdegree[kv.first.second]++; }
        double total_weight = 0; std::map<int, double> mweight;
        for(auto &kv : pe2w) {
            double w = kv.second;
            ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN);
            total_weight += w;
            if(!mweight.count(kv.first.first)) mweight[kv.first.first] = w; else mweight[kv.first.first] += w;
            if(!mweight.count(kv.first.second)) mweight[kv.first.second] = w; else mweight[kv.first.second] += w;
        }
        VertexInfo root_info = gr.vinf[root];
        double vertex_weight = gr.vwrt[root] * (root_info.rpos - root_info.lpos);
        for(auto &kv : mweight) kv.second = kv.second / total_weight * vertex_weight;

        int m = gr.num_vertices() - 1, n = m;
        std::map<int, int> ev1, ev2;
        for(int ei : gr.in_edges(root)) { ORA_ASSERT(INV_OTHER, mdegree.count(ei)); if(mdegree[ei] >= 2) ev1.insert({ei, n++}); }
        for(int ei : gr.out_edges(root)) { ORA_ASSERT(INV_OTHER, mdegree.count(ei)); if(mdegree[ei] >= 2) ev2.insert({ei, n++}); }
        for(auto &kv : pe2w) {
            int e1 = kv.first.first, e2 = kv.first.second;
            if(mdegree[e1] == 1 && mdegree[e2] == 1) { ORA_ASSERT(INV_OTHER, gr.et[e1] == root); ev1.insert({e1, n++}); }
        }
        for(int i = m; i < n; i++) { gr.add_vertex(); ORA_ASSERT(INV_OTHER, !nonzeroset.count(i)); nonzeroset.insert(i); v2v.push_back(-1); }
        if(m != n) { v2v[n] = v2v[m]; gr.vinf[n] = gr.vinf[m]; gr.vwrt[n] = gr.vwrt[n]; exchange_sink(m, n); }
        for(auto &kv : ev1) {
            int e = kv.first, k = kv.second;
            VertexInfo vi; vi.lpos = gr.vinf[gr.es[e]].rpos; vi.rpos = gr.vinf[gr.es[e]].rpos;
            gr.move_edge(e, gr.es[e], k); gr.vinf[k] = vi; gr.vwrt[k] = 0; v2v[k] = -2;
        }
        for(auto &kv : ev2) {
            int e = kv.first, k = kv.second;
            VertexInfo vi; vi.lpos = gr.vinf[gr.et[e]].lpos; vi.rpos = gr.vinf[gr.et[e]].lpos;
            gr.move_edge(e, k, gr.et[e]); gr.vinf[k] = vi; gr.vwrt[k] = 0; v2v[k] = -2;
        }
        for(auto &kv : pe2w) {
            int e1 = kv.first.first, e2 = kv.first.second; double w = kv.second;
            ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN);
            if(mdegree[e1] == 1 && mdegree[e2] >= 2) {
                ORA_ASSERT(INV_OTHER, !ev1.count(e1) && ev2.count(e2));
                borrow_edge_strand(e1, e2);
                int v1 = gr.es[e1], v2 = ev2[e2];
                gr.move_edge(e1, v1, v2);
                mev[e1].push_back(root);
                med[e1] += mweight[e1]; mei[e1] += root_info.rpos - root_info.lpos;
            } else if(mdegree[e2] == 1) {
                ORA_ASSERT(INV_OTHER, ev1.count(e1) && !ev2.count(e2));
                borrow_edge_strand(e2, e1);
                int v1 = ev1[e1], v2 = gr.et[e2];
                gr.move_edge(e2, v1, v2);
                mev[e2].insert(mev[e2].begin(), root);
                med[e2] += mweight[e2]; mei[e2] += root_info.rpos - root_info.lpos;
            } else {
                ORA_ASSERT(INV_OTHER, mdegree[e1] >= 2 && mdegree[e2] >= 2 && ev1.count(e1) && ev2.count(e2));
                int v1 = ev1[e1], v2 = ev2[e2];
                int z = gr.add_edge(v1, v2); grow_edge_maps();
                gr.ewrt[z] = w;
                EdgeInfo ei; const EdgeInfo &ei1 = gr.einf[e1], &ei2 = gr.einf[e2];
                ORA_ASSERT(INV_COUNT, ei1.count > 0 && ei2.count > 0);
                std::set_intersection(ei1.samples.begin(), ei1.samples.end(), ei2.samples.begin(), ei2.samples.end(), std::inserter(ei.samples, ei.samples.begin()));
                ei.count = (int)ei.samples.size();
                ORA_ASSERT(INV_COUNT, ei.count > 0);
                ei.abd = 0;
                for(int sp : ei.samples) { double common = std::min(ei1.spAbd.at(sp), ei2.spAbd.at(sp)); ei.spAbd[sp] = common; ei.abd += common; }
                gr.einf[z] = ei;            // strand 0, confidence 0 (fresh edge_info)
                mev[z] = std::vector<int>{root};
                med[z] = w / total_weight * vertex_weight;
                mei[z] = root_info.rpos - root_info.lpos;
                borrow_edge_strand(z, e1); borrow_edge_strand(z, e2);
                hs.insert_between(e1, e2, z);
            }
        }
        ORA_ASSERT(INV_DEGREE, gr.degree(root) == 0);
        nonzeroset.erase(root);
        for(auto &kv : ev1) resolve_single_trivial_vertex(kv.second, cfg.max_decompose_error_ratio[TRIVIAL_VERTEX]);
        for(auto &kv : ev2) resolve_single_trivial_vertex(kv.second, cfg.max_decompose_error_ratio[TRIVIAL_VERTEX]);
    }
    void borrow_edge_strand(int e1, int e2) {   // scallop.cc:1997-2007
        int s2 = gr.einf[e2].strand; if(s2 == 0) return; gr.einf[e1].strand = s2;
    }
    void exchange_sink(int old_sink, int new_sink) {       // scallop.cc:2198-2215
        std::vector<int> ve = gr.in_edges(old_sink);
        for(int e : ve) gr.move_edge(e, gr.es[e], new_sink);
        ORA_ASSERT(INV_DEGREE, gr.degree(old_sink) == 0);
    }
    void decompose_vertex_replace(int root, MPID &pe2w) {  // scallop.cc:2009-2142
        std::map<int, double> md;
        for(auto &kv : pe2w) {
            double w = kv.second;
            ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN);
            if(!md.count(kv.first.first)) md[kv.first.first] = w; else md[kv.first.first] += w;
            if(!md.count(kv.first.second)) md[kv.first.second] = w; else md[kv.first.second] += w;
        }
        for(auto &kv : md) gr.ewrt[kv.first] = kv.second;
        for(int e : gr.in_edges(root)) ORA_ASSERT(INV_OTHER, md.count(e));
        for(int e : gr.out_edges(root)) ORA_ASSERT(INV_OTHER, md.count(e));
        MPII mpi = hs.get_routes(root, gr);
        for(auto &kv : mpi) ORA_ASSERT(INV_OTHER, pe2w.count(kv.first));
        std::map<int, int> m;
        for(auto &kv : pe2w) { m[kv.first.first]++; m[kv.first.second]++; }
        for(auto &kv : pe2w) {
            int e1 = kv.first.first, e2 = kv.first.second; double w = kv.second;
            int e = merge_adjacent_edges(e1, e2, w);
            hs.replace(e1, e2, e);
            if(m[e1] == 1) hs.replace(e1, e);
            if(m[e2] == 1) hs.replace(e2, e);
        }
        for(auto &kv : pe2w) { hs.remove(kv.first.first); hs.remove(kv.first.second); }
        ORA_ASSERT(INV_DEGREE, gr.degree(root) == 0);
        nonzeroset.erase(root);
    }
    void decompose_trivial_vertex(int x) {      // scallop.cc:2144-2167
        balance_vertex(x);
        MPID pe2w;
        for(int e1 : gr.in_edges(x)) { double w1 = gr.ewrt[e1];
            for(int e2 : gr.out_edges(x)) { double w2 = gr.ewrt[e2]; pe2w.insert({{e1, e2}, w1 <= w2 ? w1 : w2}); } }
        decompose_vertex_replace(x, pe2w);
    }
    int classify_trivial_vertex(int x, bool fast) {        // scallop.cc:2169-2196
        int d1 = gr.in_degree(x), d2 = gr.out_degree(x);
        if(d1 != 1 && d2 != 1) return -1;
        int e1 = std::get<2>(*gr.si[x].begin()), e2 = std::get<2>(*gr.so[x].begin());
        if(d1 == 1) { int s = gr.es[e1]; if(gr.out_degree(s) == 1) return 1; if(fast && hs.right_dominate(e1)) return 1; }
        if(d2 == 1) { int t = gr.et[e2]; if(gr.in_degree(t) == 1) return 1; if(fast && hs.left_dominate(e2)) return 1; }
        return 2;
    }
    int split_merge_path(const std::vector<int> &p, double ww) {   // scallop.cc:2230-2240
        if(p.empty()) return -1;
        int ee = split_edge(p[0], ww);
        for(size_t i = 1; i < p.size(); i++) { int x = split_edge(p[i], ww); ee = merge_adjacent_equal_edges(ee, x); }
        return ee;
    }
    int merge_adjacent_equal_edges(int x, int y) {         // scallop.cc:2242-2378
        if(!gr.alive(x) || !gr.alive(y)) return -1;
        int xs = gr.es[x], xt = gr.et[x], ys = gr.es[y], yt = gr.et[y];
        if(xt != ys && yt != xs) return -1;
        if(yt == xs) return merge_adjacent_equal_edges(y, x);
        int n = gr.add_edge(xs, yt); grow_edge_maps();
        double wx0 = gr.ewrt[x], wy0 = gr.ewrt[y];
        ORA_ASSERT(INV_MERGE_EQUAL, fabs(wx0 - wy0) <= SMIN);
        gr.ewrt[n] = wx0 * 0.5 + wy0 * 0.5;
        EdgeInfo ei; const EdgeInfo ei1 = gr.einf[x], ei2 = gr.einf[y];
        ORA_ASSERT(INV_COUNT, ei1.count > 0 && ei2.count > 0);
        std::set_intersection(ei1.samples.begin(), ei1.samples.end(), ei2.samples.begin(), ei2.samples.end(), std::inserter(ei.samples, ei.samples.begin()));
        ei.count = (int)ei.samples.size();
        ei.abd = 0;
        for(int sp : ei.samples) { double common = std::min(ei1.spAbd.at(sp), ei2.spAbd.at(sp)); ei.spAbd[sp] = common; ei.abd += common; }
        ei.confidence = ei1.confidence + ei2.confidence;
        gr.einf[n] = ei;
        borrow_edge_strand(n, x); borrow_edge_strand(n, y);
        std::vector<int> v = mev[x]; v.push_back(xt); v.insert(v.end(), mev[y].begin(), mev[y].end());
        mev[n] = v; st.max_mev = std::max(st.max_mev, (int)v.size());
        double sum1 = gr.get_in_weights(xt), sum2 = gr.get_out_weights(xt);
        double sum = (sum1 + sum2) * 0.5;
        double r1 = gr.vwrt[xt] * (wx0 + wy0) * 0.5 / sum;
        double r2 = gr.vwrt[xt] - r1;
        gr.vwrt[xt] = r2;
        int mi = gr.vinf[xt].rpos - gr.vinf[xt].lpos + mei[x] + mei[y];
        double mdv = mi * r1 + med[x] + med[y];
        med[n] = mdv; mei[n] = mi;
        remove_edge(x); remove_edge(y);
        if(gr.in_degree(xt) == 0 && gr.out_degree(xt) == 0) nonzeroset.erase(xt);
        return n;
    }
    void remove_edge(int e) { ORA_ASSERT(INV_OTHER, gr.alive(e)); gr.remove_edge(e); }   // scallop.cc:2380-2392
    int merge_adjacent_edges(int x, int y, double ww) {    // scallop.cc:2394-2416
        ORA_ASSERT(INV_WEIGHT, ww >= cfg.min_guaranteed_edge_weight - SMIN);
        if(!gr.alive(x) || !gr.alive(y)) return -1;
        if(gr.et[x] != gr.es[y]) return merge_adjacent_edges(y, x, ww);
        int x1 = split_edge(x, ww), y1 = split_edge(y, ww);
        return merge_adjacent_equal_edges(x1, y1);
    }
    int split_edge(int ei, double w) {          // scallop.cc:2433-2484
        ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN);
        ORA_ASSERT(INV_OTHER, gr.alive(ei));
        double ww = gr.ewrt[ei];
        if(fabs(ww - w) <= SMIN) return ei;
        int s = gr.es[ei], t = gr.et[ei];
        int p2 = gr.add_edge(s, t); grow_edge_maps();
        double www = ww - w;
        if(www <= cfg.min_guaranteed_edge_weight) www = cfg.min_guaranteed_edge_weight;
        gr.ewrt[ei] = www; gr.ewrt[p2] = w; gr.einf[p2] = gr.einf[ei];
        mev[p2] = mev[ei];
        mei[p2] = mei[ei]; med[p2] = med[ei] * w / ww;
        return p2;
    }
    void balance_vertex(int v) {                // scallop.cc:2486-2576
        if(gr.in_degree(v) <= 0 || gr.out_degree(v) <= 0) return;
        std::vector<int> ve1 = gr.in_edges(v), ve2 = gr.out_edges(v);
        if(gr.degree(v) <= 0 || ve1.empty() || ve2.empty()) return;
        double w1 = 0, w2 = 0;
        for(int e : ve1) { double w = gr.ewrt[e]; ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN); w1 += w; }
        for(int e : ve2) { double w = gr.ewrt[e]; ORA_ASSERT(INV_WEIGHT, w >= cfg.min_guaranteed_edge_weight - SMIN); w2 += w; }
        double ww = sqrt(w1 * w2);
        double r1 = ww / w1, r2 = ww / w2;
        double m1 = 0, m2 = 0;
        for(int e : ve1) { double wy = gr.ewrt[e] * r1; if(wy < cfg.min_guaranteed_edge_weight) { m1 += cfg.min_guaranteed_edge_weight - wy; wy = cfg.min_guaranteed_edge_weight; } gr.ewrt[e] = wy; }
        for(int e : ve2) { double wy = gr.ewrt[e] * r2; if(wy < cfg.min_guaranteed_edge_weight) { m2 += cfg.min_guaranteed_edge_weight - wy; wy = cfg.min_guaranteed_edge_weight; } gr.ewrt[e] = wy; }
        if(m1 > m2) { int e = ve2.front(); gr.ewrt[e] = gr.ewrt[e] + m1 - m2; }
        else if(m1 < m2) { int e = ve1.front(); gr.ewrt[e] = gr.ewrt[e] + m2 - m1; }
    }
    double compute_balance_ratio(int v) {       // scallop.cc:2578-2602
        double w1 = gr.get_in_weights(v), w2 = gr.get_out_weights(v);
        ORA_ASSERT(INV_WEIGHT, w1 >= SMIN); ORA_ASSERT(INV_WEIGHT, w2 >= SMIN);
        if(w1 >= w2) return w1 / w2; else return w2 / w1;
    }
    void collect_existing_st_paths() {          // scallop.cc:2742-2752
        for(int i = 0; i < (int)gr.es.size(); i++) {
            if(!gr.alive(i)) continue;
            if(gr.es[i] != 0) continue;
            if(gr.et[i] != gr.num_vertices() - 1) continue;
            collect_path(i);
        }
    }
    void collect_path(int e) {                  // scallop.cc:2766-2834
        std::vector<int> v0 = mev[e], v; int mi = 0;
        for(int x : v0) { if(v2v[x] < 0) continue; v.push_back(v2v[x]); const VertexInfo &vi = gr.vinf[v2v[x]]; mi += vi.rpos - vi.lpos; }
        ORA_ASSERT(INV_OTHER, mei[e] == mi);
        std::sort(v.begin(), v.end());
        int n = v2v[gr.num_vertices() - 1];
        ORA_ASSERT(INV_OTHER, !v.empty() && v[0] > 0 && v.back() < n);   // v[0] on an empty vector is UB in the reference
        v.insert(v.begin(), 0); v.push_back(n);
        bool empty = false;
        for(int x : v) { if(gr.vinf[x].type == EMPTY_VERTEX) empty = true; if(empty) break; }
        if(!empty) {
            Path p; const EdgeInfo &ei = gr.einf[e];
            p.length = mi; p.weight = gr.ewrt[e]; p.abd = ei.abd; p.conf = exp(ei.confidence); p.reads = med[e]; p.v = v; p.count = ei.count;
            for(size_t i = 2; i + 1 < v.size(); i++) if(gr.vinf[v[i]].lpos != gr.vinf[v[i - 1]].rpos) p.junc.push_back({v[i - 1], v[i]});
            if(ei.strand == 1) p.strand = '+';
            if(ei.strand == 2) p.strand = '-';
            if(p.strand == '.') p.strand = gr.strand;
            paths.push_back(p);
            if(trace) trace->push_back({OP_COLLECT, e, (int)v.size(), p.weight});
        }
        gr.remove_edge(e);
    }
    void greedy_decompose() {                   // scallop.cc:2874-2897
        if(gr.num_edges() == 0) return;
        for(int i = 1; i < gr.num_vertices() - 1; i++) balance_vertex(i);
        for(int i = 1; i < gr.num_vertices() - 1; i++) balance_vertex(i);
        while(true) {
            std::vector<int> v;
            double w = gr.compute_maximum_path_w(v);
            if(w < 0) break;
            if(w <= cfg.min_transcript_coverage) break;
            if(trace) trace->push_back({OP_GREEDY, (int)v.size(), 0, w});
            int e = split_merge_path(v, w);
            collect_path(e);
            track();
        }
    }
    // scallop::unique_junc (scallop.cc:3472-3497)
    int unique_junc(int i) const {
        std::map<std::pair<int, int>, int> juncUni;
        for(size_t idx = 0; idx < paths.size(); ++idx) for(const auto &pr : paths[idx].junc) {
            if(juncUni.find(pr) == juncUni.end()) juncUni[pr] = (int)idx;
            else if(juncUni[pr] != (int)idx && juncUni[pr] != -1) juncUni[pr] = -1;
        }
        int uniqueCount = 0;
        for(const auto &pr : paths[i].junc) if(juncUni.find(pr) != juncUni.end() && juncUni[pr] == i) uniqueCount++;
        return uniqueCount;
    }
    // scallop::update_trst_features (scallop.cc:3268-3451) on the pre-decomposition copy of the graph
    void update_trst_features(const Graph &g, Features &f, int pid) {
        const Path &p = paths[pid];
        int n = (int)p.v.size();
        ORA_ASSERT(INV_OTHER, n >= 3);
        f.num_vertices = n - 2; f.num_edges = n - 3; f.gr_vertices = g.num_vertices(); f.gr_edges = g.num_edges(); f.gr_reads = g.reads; f.gr_subgraph = g.subgraph;
        f.max_mid_exon_len = 0;
        int junc = (int)p.junc.size();
        if(junc == 0) return;                   // ignore single exon
        int start_splicing_v = p.junc.front().first, end_splicing_v = p.junc.back().second;
        auto it_s = std::lower_bound(p.v.begin(), p.v.end(), start_splicing_v), it_t = std::lower_bound(p.v.begin(), p.v.end(), end_splicing_v);
        ORA_ASSERT(INV_OTHER, !(it_s == p.v.end() || *it_s != start_splicing_v || it_t == p.v.end() || *it_t != end_splicing_v));
        f.junc_ratio = 1.0 * junc / (it_t - it_s);
        for(int i = 1; i < junc; i++) { int exon_len = g.vinf[p.junc[i].first].rpos - g.vinf[p.junc[i - 1].second].lpos; f.max_mid_exon_len = std::max(f.max_mid_exon_len, exon_len); }
        const VertexInfo &svi = g.vinf[p.v[1]], &evi = g.vinf[p.v[n - 2]];
        f.start_loss1 = svi.boundary_loss1; f.start_loss2 = svi.boundary_loss2; f.start_loss3 = svi.boundary_loss3;
        f.end_loss1 = evi.boundary_loss1; f.end_loss2 = evi.boundary_loss2; f.end_loss3 = evi.boundary_loss3;
        f.start_merged_loss = svi.boundary_merged_loss; f.end_merged_loss = evi.boundary_merged_loss;
        f.uni_junc = unique_junc(pid);
        auto ratio = [&](int v1, int v2) {      // the three edge lookups the reference asserts on, then junction weight / smaller flank
            int e = g.edge(v1, v2), e1 = g.edge(v1, v1 + 1), e2 = g.edge(v2 - 1, v2);
            ORA_ASSERT(INV_OTHER, e >= 0); ORA_ASSERT(INV_OTHER, e1 >= 0); ORA_ASSERT(INV_OTHER, e2 >= 0);
            return g.ewrt[e] / std::min(g.ewrt[e1], g.ewrt[e2]);
        };
        for(int o = 0; o < (int)paths.size(); o++) {
            if(o == pid) continue;
            const std::vector<std::pair<int, int>> &junc1 = p.junc, junc2 = paths[o].junc;
            if(junc1.size() < 2 || junc2.size() < 1) continue;
            int intron_cnt = 0, start_intron = 0, end_intron = 0;
            for(size_t i = 0; i < junc1.size(); ++i) for(size_t j = 0; j < junc2.size(); ++j) {
                if(i == 0) {
                    if(junc2[j].first >= p.v[1] && junc2[j].second <= junc1[0].first) { start_intron++; f.start_intron_ratio = std::max(f.start_intron_ratio, ratio(junc2[j].first, junc2[j].second)); }
                }
                else if(junc2[j].second <= junc1[i].first && junc2[j].first >= junc1[i - 1].second) { intron_cnt++; f.intron_ratio = std::max(f.intron_ratio, ratio(junc2[j].first, junc2[j].second)); }
                if(i == junc1.size() - 1) {
                    if(junc2[j].first >= junc1[i].second && junc2[j].second <= p.v[n - 2]) { end_intron++; f.end_intron_ratio = std::max(f.end_intron_ratio, ratio(junc2[j].first, junc2[j].second)); }
                }
            }
            f.introns = std::max(f.introns, intron_cnt); f.start_introns = std::max(f.start_introns, start_intron); f.end_introns = std::max(f.end_introns, end_intron);
        }
        f.seq_min_wt = DBL_MAX; f.seq_min_cnt = INT_MAX; f.seq_min_abd = DBL_MAX; f.seq_min_ratio = 1.0;
        f.seq_max_wt = 0; f.seq_max_cnt = 0; f.seq_max_abd = 0; f.seq_max_ratio = 0;
        for(int i = 1; i < n; i++) {
            int v1 = p.v[i - 1], v2 = p.v[i];
            int e = g.edge(v1, v2);
            ORA_ASSERT(INV_OTHER, e >= 0);
            const EdgeInfo &ei = g.einf[e]; const VertexInfo &vi2 = g.vinf[v2];
            double w = g.ewrt[e], r = w / std::max(g.get_in_weights(v2), g.get_out_weights(v1));
            f.seq_min_wt = std::min(f.seq_min_wt, w); f.seq_min_cnt = std::min(f.seq_min_cnt, ei.count); f.seq_min_abd = std::min(f.seq_min_abd, ei.abd); f.seq_min_ratio = std::min(f.seq_min_ratio, r);
            f.seq_max_wt = std::max(f.seq_max_wt, w); f.seq_max_cnt = std::max(f.seq_max_cnt, ei.count); f.seq_max_abd = std::max(f.seq_max_abd, ei.abd); f.seq_max_ratio = std::max(f.seq_max_ratio, r);
            if(i == 1) { f.unbridge_start_coming_count = vi2.unbridge_coming_count; f.unbridge_start_coming_ratio = vi2.unbridge_coming_ratio; f.start_cnt = ei.count; f.start_weight = w; f.start_abd = ei.abd; }
            else if(i == n - 2) { f.unbridge_end_leaving_count = vi2.unbridge_leaving_count; f.unbridge_end_leaving_ratio = vi2.unbridge_leaving_ratio; }
            else if(i == n - 1) { f.end_cnt = ei.count; f.end_weight = w; f.end_abd = ei.abd; }
        }
        f.complete = true;
    }
    void build_transcripts() {                  // scallop.cc:3250-3266 + essential.cc:719-748 (features first, then the exon join)
        trsts.clear();
        for(size_t pi = 0; pi < paths.size(); pi++) {
            const Path &p = paths[pi];
            Transcript t;
            // the reference would abort on an assert in here; the oracle records it and keeps the decomposition (the product computes
            // the features in a separate, optional call)
            try { update_trst_features(gr_ori, t.features, (int)pi); } catch(const AssertFail &a) { if(!feature_assert) feature_assert = a.line; }
            t.coverage = log(1.0 + p.weight); t.strand = p.strand; t.conf = p.conf; t.abd = p.abd; t.count1 = p.count;
            // join_interval_map: add [lpos,rpos) with value 1; touching equal-valued intervals join, empty intervals vanish
            for(size_t k = 1; k + 1 < p.v.size(); k++) {
                int32_t p1 = gr_ori.vinf[p.v[k]].lpos, p2 = gr_ori.vinf[p.v[k]].rpos;
                if(p1 >= p2) continue;
                if(!t.exons.empty() && t.exons.back().second == p1) t.exons.back().second = p2; else t.exons.push_back({p1, p2});
            }
            trsts.push_back(t);
        }
    }
};

} // namespace ora
