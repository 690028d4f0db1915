// adapter_test.cc -- exercises aletsch_amd/host/gpu_scallop.hpp (the reference-shaped C++ surface over the C ABI) with
// mock types that carry the same member names as the reference's splice_graph / hyper_set / parameters / path.
// Reads a tiny text graph from stdin, decomposes it on the GPU, prints the paths.  Driven by tests/test_gpu_adapter.py.
#include "../../aletsch_amd/host/gpu_scallop.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <set>
#include <unordered_map>

struct mock_edge { int s, t; int source() const { return s; } int target() const { return t; } };
struct mock_edge_info { int strand = 0, count = 0; double abd = 0; std::set<int> samples; std::unordered_map<int, double> spAbd; };
struct mock_vertex_info { int32_t lpos = 0, rpos = 0; int type = -1; };
struct mock_graph {
    std::vector<mock_edge*> es; std::vector<double> ew; std::vector<mock_edge_info> ei; std::vector<double> vw; std::vector<mock_vertex_info> vi; char strand = '.';
    size_t num_vertices() const { return vw.size(); }
    std::pair<std::vector<mock_edge*>::iterator, std::vector<mock_edge*>::iterator> edges() { return {es.begin(), es.end()}; }
    int idx(mock_edge *e) const { for(size_t i = 0; i < es.size(); i++) if(es[i] == e) return (int)i; return -1; }
    double get_edge_weight(mock_edge *e) const { return ew[idx(e)]; }
    const mock_edge_info &get_edge_info(mock_edge *e) const { return ei[idx(e)]; }
    double get_vertex_weight(int v) const { return vw[v]; }
    const mock_vertex_info &get_vertex_info(int v) const { return vi[v]; }
};
struct mock_hyper_set { std::map<std::vector<int>, int> nodes; };
struct mock_phase_set { std::map<std::vector<int32_t>, int> pmap; };       // rnacore/phase_set.h:24
struct mock_parameters { double max_decompose_error_ratio[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00}; double min_guaranteed_edge_weight = 0.01, min_transcript_coverage = 2.0; int max_num_exons = 10000; };
struct mock_path { std::vector<int> v; std::vector<std::pair<int, int>> junc; int length = 0; double abd = 0, weight = 0, conf = 0, reads = 0; char strand = '.'; int count = 0; };

static bool read_graph(mock_graph &g, mock_hyper_set &hs)
{
    int V, E, P;
    if(scanf("%d %d %d", &V, &E, &P) != 3) return false;
    for(int i = 0; i < V; i++) { double w; int l, r; if(scanf("%lf %d %d", &w, &l, &r) != 3) return false; g.vw.push_back(w); mock_vertex_info vi; vi.lpos = l; vi.rpos = r; g.vi.push_back(vi); }
    for(int k = 0; k < E; k++) { int s, t; double w; if(scanf("%d %d %lf", &s, &t, &w) != 3) return false; g.es.push_back(new mock_edge{s, t}); g.ew.push_back(w); mock_edge_info ei; ei.count = 1; ei.abd = w; ei.samples.insert(0); ei.spAbd[0] = w; g.ei.push_back(ei); }
    for(int p = 0; p < P; p++) { int len, c; if(scanf("%d %d", &len, &c) != 2) return false; std::vector<int> v(len); for(int &x : v) if(scanf("%d", &x) != 1) return false; hs.nodes[v] += c; }
    return true;
}
static void print_paths(int status, const std::vector<mock_path> &paths)
{
    printf("status %d paths %zu\n", status, paths.size());
    for(auto &p : paths) { printf("%.17g %.17g %.17g %d %d %c %zu :", p.weight, p.abd, p.reads, p.length, p.count, p.strand, p.junc.size()); for(int x : p.v) printf(" %d", x); printf("\n"); }
}

// "raw N": N graphs as assembler::assemble(gx, px, sid) receives them (the P lines after the edges are PHASES in exon coordinates:
//          "len count c0 c1 ..."), through gpu_scallop_batch::enqueue_raw -- the library runs the pre-steps -- then ONE flush
// no argument: one graph through aletsch::gpu_scallop (ctor + assemble() + .paths);
// "batch N": N graphs through aletsch::gpu_scallop_batch -- enqueue all, ONE flush, paths(i) per ticket, then clear() and a second round
int main(int argc, char **argv)
{
    mock_parameters cfg;
    try {
        if(argc >= 3 && std::string(argv[1]) == "raw") {
            const int N = atoi(argv[2]);
            std::vector<mock_graph> gs((size_t)N); std::vector<mock_phase_set> ps((size_t)N);
            for(int n = 0; n < N; n++) {
                mock_hyper_set as_lists;                                   // read_graph parses "len count v..." lines: here they are coordinates
                if(!read_graph(gs[(size_t)n], as_lists)) return 2;
                for(auto &kv : as_lists.nodes) ps[(size_t)n].pmap[std::vector<int32_t>(kv.first.begin(), kv.first.end())] += kv.second;
            }
            aletsch::gpu_scallop_batch<mock_graph, mock_hyper_set, mock_parameters, mock_path> batch(cfg, 0);
            std::vector<int> ticket;
            for(int n = 0; n < N; n++) ticket.push_back(batch.enqueue_raw(gs[(size_t)n], ps[(size_t)n], 10000));
            batch.flush();
            for(int n = 0; n < N; n++) { if(ticket[(size_t)n] < 0) printf("status %d paths 0\n", -1 - ticket[(size_t)n]); else print_paths(batch.status(ticket[(size_t)n]), batch.paths(ticket[(size_t)n])); }
            return 0;
        }
        if(argc >= 3 && std::string(argv[1]) == "batch") {
            const int N = atoi(argv[2]);
            std::vector<mock_graph> gs((size_t)N); std::vector<mock_hyper_set> hs((size_t)N);
            for(int n = 0; n < N; n++) if(!read_graph(gs[(size_t)n], hs[(size_t)n])) return 2;
            aletsch::gpu_scallop_batch<mock_graph, mock_hyper_set, mock_parameters, mock_path> batch(cfg, 0);
            for(int round = 0; round < 2; round++) {
                std::vector<int> ticket;
                for(int n = 0; n < N; n++) ticket.push_back(batch.enqueue(gs[(size_t)n], hs[(size_t)n]));
                batch.flush();
                if(round == 1) for(int n = 0; n < N; n++) print_paths(batch.status(ticket[(size_t)n]), batch.paths(ticket[(size_t)n]));
                batch.clear();
            }
            return 0;
        }
        mock_graph g; mock_hyper_set hs;
        if(!read_graph(g, hs)) return 2;
        aletsch::gpu_scallop<mock_graph, mock_hyper_set, mock_parameters, mock_path> sx(g, hs, cfg, false);
        sx.assemble();
        print_paths(sx.status, sx.paths);
    } catch(const std::exception &e) { printf("EXCEPTION %s\n", e.what()); return 1; }
    return 0;
}
