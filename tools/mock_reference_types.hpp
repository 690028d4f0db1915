// mock_reference_types.hpp -- reference-shaped stand-ins for the benches of the host adapter (tools/dispatch_bench.cc, tools/convert_bench.cc):
// a splice graph whose edges are heap objects and whose edge_info carries std::set / unordered_map members like the reference's
// (rnacore/edge_info.h), filled from the bench workload's generator.  Diagnostic code, not part of the product.
#pragma once
#include "../include/aletsch_decomp.h"
#include <vector>
#include <map>
#include <set>
#include <unordered_map>
#include <cstdint>

struct mock_edge { int s, t, id; int source() const { return s; } int target() const { return t; } };
struct mock_edge_info { int strand = 0, count = 0; double abd = 0; std::set<int> samples; std::unordered_map<int, double> spAbd; };
struct mock_vertex_info { int32_t lpos = 0, rpos = 0; int type = -1; };
struct mock_graph {
    std::vector<mock_edge*> es; std::vector<double> ew; std::vector<mock_edge_info> ei; std::vector<double> vw; std::vector<mock_vertex_info> vi; char strand = '.';
    size_t num_vertices() const { return vw.size(); }
    std::pair<std::vector<mock_edge*>::iterator, std::vector<mock_edge*>::iterator> edges() { return {es.begin(), es.end()}; }
    double get_edge_weight(const mock_edge *e) const { return ew[(size_t)e->id]; }
    const mock_edge_info &get_edge_info(const mock_edge *e) const { return ei[(size_t)e->id]; }
    double get_vertex_weight(int v) const { return vw[(size_t)v]; }
    const mock_vertex_info &get_vertex_info(int v) const { return vi[(size_t)v]; }
};
struct mock_hyper_set { std::map<std::vector<int>, int> nodes; };
struct mock_parameters { double max_decompose_error_ratio[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00}; double min_guaranteed_edge_weight = 0.01, min_transcript_coverage = 2.0; int max_num_exons = 10000; };


// the bench workload (64 vertices / 256 edges, U[1,100) weights, one sample), unpacked into reference-shaped objects
inline bool make_mock_graphs(int N, std::vector<mock_graph> &G, std::vector<mock_hyper_set> &H)
{
    ald_synth_spec sp{}; sp.seed = 1002; sp.n_graphs = N; sp.v_min = 64; sp.v_max = 64; sp.fixed_edges = 256; sp.n_samples = 1;
    int64_t tv, te, ts, tp, tpv;
    if(ald_synth_sizes(&sp, &tv, &te, &ts, &tp, &tpv) != ALD_OK) return false;
    std::vector<int32_t> g_nv(N), g_ne(N), g_np(N), voff((size_t)(tv + N)), et((size_t)te), eso((size_t)(te + N)), sid((size_t)ts + 1), lp((size_t)tv), rp((size_t)tv), vt((size_t)tv), po((size_t)(tp + N)), pv((size_t)tpv + 1), pc((size_t)tp + 1);
    std::vector<double> ew((size_t)te), ea((size_t)te), sa((size_t)ts + 1), vw((size_t)tv); std::vector<uint8_t> est((size_t)te); std::vector<char> gs((size_t)N);
    if(ald_synth_fill(&sp, g_nv.data(), g_ne.data(), g_np.data(), voff.data(), et.data(), ew.data(), est.data(), ea.data(), eso.data(), sid.data(), sa.data(), vw.data(), lp.data(), rp.data(), vt.data(), po.data(), pv.data(), pc.data(), gs.data()) != ALD_OK) return false;
    G.assign((size_t)N, mock_graph()); H.assign((size_t)N, mock_hyper_set());
    int64_t ov = 0, oe = 0, oo = 0;
    for(int n = 0; n < N; n++) {
        mock_graph &g = G[(size_t)n]; const int V = g_nv[n];
        for(int i = 0; i < V; i++) { g.vw.push_back(vw[(size_t)(ov + i)]); mock_vertex_info vi; vi.lpos = lp[(size_t)(ov + i)]; vi.rpos = rp[(size_t)(ov + i)]; vi.type = vt[(size_t)(ov + i)]; g.vi.push_back(vi); }
        int k = 0;
        for(int s = 0; s < V; s++) for(int q = voff[(size_t)(oo + s)]; q < voff[(size_t)(oo + s + 1)]; q++, k++) {
            g.es.push_back(new mock_edge{s, et[(size_t)(oe + q)], k}); g.ew.push_back(ew[(size_t)(oe + q)]);
            mock_edge_info ei; ei.count = 1; ei.abd = ew[(size_t)(oe + q)]; ei.samples.insert(0); ei.spAbd[0] = ew[(size_t)(oe + q)]; g.ei.push_back(ei);
        }
        ov += V; oe += g_ne[n]; oo += V + 1;
    }
    return true;
}
