// dispatch_test.cc -- exercises aletsch_amd/host/gpu_dispatch.hpp (queue + flusher over the C ABI) with the mock types of
// adapter_test.cc.  stdin: "N T B S D" (graphs, submitting threads, graphs per batch, slots per device, device slots -- all on HIP
// device 0: D > 1 runs the several-devices-in-one-process form with D GPU threads), then N graphs in adapter_test's format.
// Thread t submits graphs t, t + T, ...; sample id of graph g = g % 3.  Prints the merged sink.  Driven by tests/test_gpu_adapter.py.
#include "../../aletsch_amd/host/gpu_dispatch.hpp"
#include <cstdio>
#include <set>
#include <unordered_map>

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

int main()
{
    int N, T, B, S, D;
    if(scanf("%d %d %d %d %d", &N, &T, &B, &S, &D) != 5) return 2;
    std::vector<mock_graph> gs((size_t)N); std::vector<mock_hyper_set> hs((size_t)N); mock_parameters cfg;
    for(int n = 0; n < N; n++) {
        int V, E, P; mock_graph &g = gs[(size_t)n];
        if(scanf("%d %d %d", &V, &E, &P) != 3) return 2;
        for(int i = 0; i < V; i++) { double w; int l, r; if(scanf("%lf %d %d", &w, &l, &r) != 3) return 2; g.vw.push_back(w); mock_vertex_info vi; vi.lpos = l; vi.rpos = r; g.vi.push_back(vi); }
        for(int k = 0; k < E; k++) { int s, t; double w; if(scanf("%d %d %lf", &s, &t, &w) != 3) return 2; g.es.push_back(new mock_edge{s, t, k}); g.ew.push_back(w); mock_edge_info ei; ei.count = 1; ei.abd = w; ei.samples.insert(0); ei.spAbd[0] = w; g.ei.push_back(ei); }
        for(int p = 0; p < P; p++) { int len, c; if(scanf("%d %d", &len, &c) != 2) return 2; std::vector<int> v((size_t)len); for(int &x : v) if(scanf("%d", &x) != 1) return 2; hs[(size_t)n].nodes[v] += c; }
    }
    ald_tset *tm = nullptr;
    if(ald_tset_create(0.8, &tm) != ALD_OK) return 3;
    try {
        aletsch::gpu_assembly_queue<mock_graph, mock_hyper_set, mock_parameters> q(cfg, tm, false, std::vector<int>((size_t)D, 0), B, S);
        std::vector<std::thread> th;
        for(int t = 0; t < T; t++) th.emplace_back([&, t] { for(int n = t; n < N; n += T) q.submit(gs[(size_t)n], hs[(size_t)n], n % 3); });
        for(auto &x : th) x.join();
        q.drain();
        printf("submitted %ld failed %ld batches %ld\n", q.submitted(), q.failed_graphs(), q.batches());
        // a second round through the same queue (slots are reused after a drain)
        q.drain();
    } catch(const std::exception &e) { printf("EXCEPTION %s\n", e.what()); return 1; }
    int64_t n = 0, ne = 0, ns = 0;
    ald_tset_size(tm, &n, &ne, &ns);
    std::vector<uint64_t> h((size_t)n + 1); std::vector<int32_t> cnt((size_t)n + 1), c1((size_t)n + 1), c2((size_t)n + 1), lr(2 * (size_t)ne + 2), ssid((size_t)ns + 1), sc1((size_t)ns + 1);
    std::vector<char> st((size_t)n + 1); std::vector<double> cov((size_t)n + 1), cov2((size_t)n + 1), conf((size_t)n + 1), abd((size_t)n + 1), scov2((size_t)ns + 1), sconf((size_t)ns + 1), sabd((size_t)ns + 1);
    std::vector<int64_t> tid((size_t)n + 1), eo((size_t)n + 2), so((size_t)n + 2);
    if(ald_tset_export(tm, h.data(), cnt.data(), st.data(), cov.data(), cov2.data(), conf.data(), abd.data(), c1.data(), c2.data(), tid.data(), eo.data(), lr.data(),
                       so.data(), ssid.data(), scov2.data(), sconf.data(), sabd.data(), sc1.data()) != ALD_OK) return 4;
    for(int64_t i = 0; i < n; i++) {
        printf("%llu %d %c %.17g %.17g %.17g %.17g %d %d %lld :", (unsigned long long)h[(size_t)i], cnt[(size_t)i], st[(size_t)i], cov[(size_t)i], cov2[(size_t)i], conf[(size_t)i], abd[(size_t)i], c1[(size_t)i], c2[(size_t)i], (long long)tid[(size_t)i]);
        for(int64_t k = eo[(size_t)i]; k < eo[(size_t)i + 1]; k++) printf(" %d-%d", lr[2 * (size_t)k], lr[2 * (size_t)k + 1]);
        printf(" :");
        for(int64_t k = so[(size_t)i]; k < so[(size_t)i + 1]; k++) printf(" %d,%.17g,%.17g,%.17g,%d", ssid[(size_t)k], scov2[(size_t)k], sconf[(size_t)k], sabd[(size_t)k], sc1[(size_t)k]);
        printf("\n");
    }
    ald_tset_destroy(tm);
    return 0;
}
