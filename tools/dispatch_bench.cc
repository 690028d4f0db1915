// dispatch_bench.cc -- throughput of aletsch::gpu_assembly_queue (aletsch_amd/host/gpu_dispatch.hpp) fed by T submitting threads
// with reference-shaped objects (mock splice_graph / hyper_set carrying std::set / unordered_map members like the reference's
// edge_info), i.e. what a pool of `assembler::assemble` tasks would hand over.  Diagnostic, run on the GPU box:
//   g++ -std=c++11 -O2 -pthread -Iinclude tools/dispatch_bench.cc -o gpurun_out/dispatch_bench -Laletsch_amd/lib -laletsch_decomp -Wl,-rpath,$PWD/aletsch_amd/lib
//   gpurun_out/dispatch_bench <distinct graphs> <rounds> <threads> <graphs per batch> <slots>
#include "../aletsch_amd/host/gpu_dispatch.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>
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

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 20000, R = argc > 2 ? atoi(argv[2]) : 5, T = argc > 3 ? atoi(argv[3]) : 8, B = argc > 4 ? atoi(argv[4]) : 32768, S = argc > 5 ? atoi(argv[5]) : 3;
    // the bench workload's generator (64 vertices / 256 edges, U[1,100) weights), unpacked into reference-shaped objects
    ald_synth_spec sp{}; sp.seed = 1002; sp.n_graphs = N; sp.v_min = 64; sp.v_max = 64; sp.fixed_edges = 256; sp.n_samples = 1;
    int64_t tv, te, ts, tp, tpv;
    if(ald_synth_sizes(&sp, &tv, &te, &ts, &tp, &tpv) != ALD_OK) return 2;
    std::vector<int32_t> g_nv(N), g_ne(N), g_np(N), voff((size_t)(tv + N)), et((size_t)te), eso((size_t)(te + N)), sid((size_t)ts + 1), lp((size_t)tv), rp((size_t)tv), vt((size_t)tv), po((size_t)(tp + N)), pv((size_t)tpv + 1), pc((size_t)tp + 1);
    std::vector<double> ew((size_t)te), ea((size_t)te), sa((size_t)ts + 1), vw((size_t)tv); std::vector<uint8_t> est((size_t)te); std::vector<char> gs((size_t)N);
    if(ald_synth_fill(&sp, g_nv.data(), g_ne.data(), g_np.data(), voff.data(), et.data(), ew.data(), est.data(), ea.data(), eso.data(), sid.data(), sa.data(), vw.data(), lp.data(), rp.data(), vt.data(), po.data(), pv.data(), pc.data(), gs.data()) != ALD_OK) return 2;
    std::vector<mock_graph> G((size_t)N); std::vector<mock_hyper_set> H((size_t)N);
    { int64_t ov = 0, oe = 0, oo = 0;
      for(int n = 0; n < N; n++) {
        mock_graph &g = G[(size_t)n]; const int V = g_nv[n];
        for(int i = 0; i < V; i++) { g.vw.push_back(vw[(size_t)(ov + i)]); mock_vertex_info vi; vi.lpos = lp[(size_t)(ov + i)]; vi.rpos = rp[(size_t)(ov + i)]; vi.type = vt[(size_t)(ov + i)]; g.vi.push_back(vi); }
        int k = 0;
        for(int s = 0; s < V; s++) for(int q = voff[(size_t)(oo + s)]; q < voff[(size_t)(oo + s + 1)]; q++, k++) {
            g.es.push_back(new mock_edge{s, et[(size_t)(oe + q)], k}); g.ew.push_back(ew[(size_t)(oe + q)]);
            mock_edge_info ei; ei.count = 1; ei.abd = ew[(size_t)(oe + q)]; ei.samples.insert(0); ei.spAbd[0] = ew[(size_t)(oe + q)]; g.ei.push_back(ei);
        }
        ov += V; oe += g_ne[n]; oo += V + 1;
      } }
    ald_tset *tm = nullptr; if(ald_tset_create(0.8, &tm) != ALD_OK) return 3;
    mock_parameters cfg;
    try {
        aletsch::gpu_assembly_queue<mock_graph, mock_hyper_set, mock_parameters> q(cfg, tm, false, 0, B, S);
        double p0 = 0, g0 = 0, m0 = 0;
        for(int pass = 0; pass < 2; pass++) {                     // pass 0 warms the buffers (pinned / device allocations) of EVERY slot
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            const int rounds = pass == 0 ? (int)(((long)B * (S + 2) + N - 1) / N) : R;
            for(int t = 0; t < T; t++) th.emplace_back([&, t] { for(int r = 0; r < rounds; r++) for(int n = t; n < N; n += T) q.submit(G[(size_t)n], H[(size_t)n], n % 4); });
            for(auto &x : th) x.join();
            auto t1 = std::chrono::steady_clock::now();
            q.drain();
            auto t2 = std::chrono::steady_clock::now();
            const double s_sub = std::chrono::duration<double>(t1 - t0).count(), s_all = std::chrono::duration<double>(t2 - t0).count();
            double tp, tg, tm2; q.stage_seconds(tp, tg, tm2);
            printf("   busy seconds of the stages in this pass: pack %.3f s, upload+kernel+download %.3f s, merge %.3f s\n", tp - p0, tg - g0, tm2 - m0);
            p0 = tp; g0 = tg; m0 = tm2;
            printf("%s: %ld graphs, %d submitters, batches of %d, %d slots: submit %.3f s, drained %.3f s -> %.0f graphs/s (failed %ld, batches %ld)\n",
                   pass == 0 ? "warm-up" : "measured", (long)N * rounds, T, B, S, s_sub, s_all, (double)N * rounds / s_all, q.failed_graphs(), q.batches());
        }
    } catch(const std::exception &e) { printf("EXCEPTION %s\n", e.what()); return 1; }
    int64_t n = 0, ne = 0, ns = 0; ald_tset_size(tm, &n, &ne, &ns);
    printf("sink: %lld items, %lld exons, %lld sample records\n", (long long)n, (long long)ne, (long long)ns);
    ald_tset_destroy(tm);
    return 0;
}
