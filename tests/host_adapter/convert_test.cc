// convert_test.cc -- the submitting thread's conversion (aletsch::packed_chunk::append_graph: one walk over gr.edges(), counting sort into
// CSR rows, prefetching of the reference's heap objects) against the plain two-step form it replaces (stage_graph + packed_chunk::append) on
// reference-shaped objects: random DAGs whose edges are created in a random order, with parallel edges, edges of zero, one and several
// supporting samples, sample sets that do not match the abundance map, strands, counts that differ from the number of samples, and phasing
// nodes.  No GPU: the arrays of the two chunks must be identical.  Run by tests/test_abi_cpu.py (also under ASan + UBSan).
#include "../../aletsch_amd/host/gpu_dispatch.hpp"
#include <cstdio>
#include <cstdlib>
#include <set>
#include <unordered_map>
#include <random>

struct mock_edge { int s, t, id; int source() const { return s; } int target() const { return t; } };
struct mock_edge_info { int strand = 0, count = 0; double abd = 0; std::set<int> samples; std::unordered_map<int, double> spAbd; };
struct mock_vertex_info { int32_t lpos = 0, rpos = 0; int type = -1; };
struct mock_graph {
    std::vector<mock_edge*> es; std::vector<double> ew; std::vector<mock_edge_info> ei; std::vector<double> vw; std::vector<mock_vertex_info> vi; char strand = '.';
    ~mock_graph() { for(mock_edge *e : es) delete e; }
    size_t num_vertices() const { return vw.size(); }
    std::pair<std::vector<mock_edge*>::iterator, std::vector<mock_edge*>::iterator> edges() { return {es.begin(), es.end()}; }
    double get_edge_weight(const mock_edge *e) const { return ew[(size_t)e->id]; }
    const mock_edge_info &get_edge_info(const mock_edge *e) const { return ei[(size_t)e->id]; }
    double get_vertex_weight(int v) const { return vw[(size_t)v]; }
    const mock_vertex_info &get_vertex_info(int v) const { return vi[(size_t)v]; }
};
struct mock_hyper_set { std::map<std::vector<int>, int> nodes; };

template<class T> static bool same(const char *name, const std::vector<T> &a, const std::vector<T> &b)
{
    if(a == b) return true;
    printf("MISMATCH in %s (%zu vs %zu entries)\n", name, a.size(), b.size());
    return false;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(12345);
    auto uni = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    aletsch::packed_chunk a, b; aletsch::packed_chunk::scratch tmp;
    long edges = 0;
    for(int r = 0; r < rounds; r++) {
        mock_graph g; mock_hyper_set h;
        const int V = uni(2, 40), E = V < 3 ? 1 : uni(1, 4 * V);
        g.strand = "+-."[uni(0, 2)];
        for(int i = 0; i < V; i++) { g.vw.push_back((double)uni(0, 1000) / 7.0); mock_vertex_info vi; vi.lpos = uni(0, 100000); vi.rpos = vi.lpos + uni(0, 500); vi.type = uni(-1, 3); g.vi.push_back(vi); }
        for(int k = 0; k < E; k++) {                                      // creation order is random: sources and targets in no order, parallel edges allowed
            const int s = uni(0, V - 2), t = uni(s + 1, V - 1);
            g.es.push_back(new mock_edge{s, t, k}); g.ew.push_back((double)uni(1, 100000) / 13.0);
            mock_edge_info ei; ei.strand = uni(0, 2); ei.abd = (double)uni(0, 5000) / 3.0;
            const int ns = uni(0, 9) < 6 ? 1 : uni(0, 4);
            for(int q = 0; q < ns; q++) { const int sp = uni(0, 11); ei.samples.insert(sp); if(uni(0, 9) < 9) ei.spAbd[sp] = (double)uni(1, 9000) / 11.0; }
            if(uni(0, 9) == 0) ei.spAbd[uni(20, 30)] = 1.5;                 // an abundance without its sample
            ei.count = (int)ei.samples.size() + uni(0, 2);
            g.ei.push_back(ei);
        }
        for(int p = uni(0, 5); p > 0; p--) { std::vector<int> v; for(int q = uni(2, 6); q > 0; q--) v.push_back(uni(0, V - 1)); h.nodes[v] += uni(1, 9); }
        edges += E;
        a.append_graph(g, h, r % 5, tmp);
        b.append(aletsch::stage_graph(g, h), r % 5);
    }
    bool ok = same("g_nv", a.g_nv, b.g_nv) & same("g_ne", a.g_ne, b.g_ne) & same("g_np", a.g_np, b.g_np) & same("vertex_offset", a.vertex_offset, b.vertex_offset)
            & same("edge_target", a.edge_target, b.edge_target) & same("edge_weight", a.edge_weight, b.edge_weight) & same("edge_strand", a.edge_strand, b.edge_strand)
            & same("edge_abd", a.edge_abd, b.edge_abd) & same("edge_sample_offset", a.edge_sample_offset, b.edge_sample_offset) & same("sample_id", a.sample_id, b.sample_id)
            & same("sample_abd", a.sample_abd, b.sample_abd) & same("vertex_weight", a.vertex_weight, b.vertex_weight) & same("vertex_lpos", a.vertex_lpos, b.vertex_lpos)
            & same("vertex_rpos", a.vertex_rpos, b.vertex_rpos) & same("vertex_type", a.vertex_type, b.vertex_type) & same("phasing_offset", a.phasing_offset, b.phasing_offset)
            & same("phasing_vertex", a.phasing_vertex, b.phasing_vertex) & same("phasing_count", a.phasing_count, b.phasing_count) & same("edge_count", a.edge_count, b.edge_count)
            & same("edge_rank", a.edge_rank, b.edge_rank) & same("sid", a.sid, b.sid) & same("graph_strand", a.graph_strand, b.graph_strand) & same("raw_dist", a.raw_dist, b.raw_dist);
    printf("%s: %d graphs, %ld edges\n", ok ? "identical" : "DIFFERENT", rounds, edges);
    return ok ? 0 : 1;
}
