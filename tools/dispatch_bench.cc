// dispatch_bench.cc -- throughput of aletsch::gpu_assembly_queue (aletsch_amd/host/gpu_dispatch.hpp) fed by T submitting threads
// with reference-shaped objects (mock splice_graph / hyper_set carrying std::set / unordered_map members like the reference's
// edge_info), i.e. what a pool of `assembler::assemble` tasks would hand over.  Diagnostic, run on the GPU box:
//   g++ -std=c++11 -O2 -pthread -Iinclude tools/dispatch_bench.cc -o gpurun_out/dispatch_bench -Laletsch_amd/lib -laletsch_decomp -Wl,-rpath,$PWD/aletsch_amd/lib
//   gpurun_out/dispatch_bench <distinct graphs> <rounds> <threads> <graphs per batch> <slots per GPU thread> <GPU threads on device 0> <pack threads>
#include "../aletsch_amd/host/gpu_dispatch.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "mock_reference_types.hpp"

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 20000, R = argc > 2 ? atoi(argv[2]) : 5, T = argc > 3 ? atoi(argv[3]) : 8, B = argc > 4 ? atoi(argv[4]) : 32768, S = argc > 5 ? atoi(argv[5]) : 3,
              GT = argc > 6 ? atoi(argv[6]) : 1, PT = argc > 7 ? atoi(argv[7]) : 2;
    std::vector<mock_graph> G; std::vector<mock_hyper_set> H;
    if(!make_mock_graphs(N, G, H)) return 2;
    ald_tset *tm = nullptr; if(ald_tset_create(0.8, &tm) != ALD_OK) return 3;
    mock_parameters cfg;
    try {
        aletsch::gpu_assembly_queue<mock_graph, mock_hyper_set, mock_parameters> q(cfg, tm, false, std::vector<int>((size_t)(GT < 1 ? 1 : GT), 0), B, S, 2048, PT);
        double p0 = 0, g0 = 0, m0 = 0;
        for(int pass = 0; pass < 2; pass++) {                     // pass 0 warms the buffers (pinned / device allocations) of EVERY slot
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            const int rounds = pass == 0 ? (int)(((long)B * (S * (GT < 1 ? 1 : GT) + 2) + N - 1) / N) : R;
            for(int t = 0; t < T; t++) th.emplace_back([&, t] { for(int r = 0; r < rounds; r++) for(int n = t; n < N; n += T) q.submit(G[(size_t)n], H[(size_t)n], n % 4); });
            for(auto &x : th) x.join();
            auto t1 = std::chrono::steady_clock::now();
            q.drain();
            auto t2 = std::chrono::steady_clock::now();
            const double s_sub = std::chrono::duration<double>(t1 - t0).count(), s_all = std::chrono::duration<double>(t2 - t0).count();
            double tp, tg, tm2; q.stage_seconds(tp, tg, tm2);
            printf("   busy seconds of the stages in this pass: pack %.3f s, upload+kernel+download %.3f s, merge %.3f s\n", tp - p0, tg - g0, tm2 - m0);
            p0 = tp; g0 = tg; m0 = tm2;
            printf("%s: %ld graphs, %d submitters, batches of %d, %d slots x %d GPU threads, %d pack threads: submit %.3f s, drained %.3f s -> %.0f graphs/s (failed %ld, batches %ld)\n",
                   pass == 0 ? "warm-up" : "measured", (long)N * rounds, T, B, S, GT, PT, s_sub, s_all, (double)N * rounds / s_all, q.failed_graphs(), q.batches());
        }
    } catch(const std::exception &e) { printf("EXCEPTION %s\n", e.what()); return 1; }
    int64_t n = 0, ne = 0, ns = 0; ald_tset_size(tm, &n, &ne, &ns);
    printf("sink: %lld items, %lld exons, %lld sample records\n", (long long)n, (long long)ne, (long long)ns);
    ald_tset_destroy(tm);
    return 0;
}
