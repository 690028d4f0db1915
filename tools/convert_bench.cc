// convert_bench.cc -- the submitting thread's share of aletsch::gpu_assembly_queue alone (no GPU): reference-shaped objects ->
// packed_chunk::append_graph, T threads, each over its own graphs.  What bounds the dispatcher is this walk over the containers.
//   g++ -std=c++11 -O2 -pthread -Iinclude tools/convert_bench.cc aletsch_amd/csrc/synth.cpp -o /tmp/convert_bench && /tmp/convert_bench [graphs] [rounds] [threads]
#include "../aletsch_amd/host/gpu_dispatch.hpp"
#include "mock_reference_types.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 20000, R = argc > 2 ? atoi(argv[2]) : 5, T = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<mock_graph> G; std::vector<mock_hyper_set> H;
    if(!make_mock_graphs(N, G, H)) return 2;
    for(int pass = 0; pass < 2; pass++) {
        std::vector<long> sums((size_t)T, 0);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for(int t = 0; t < T; t++) th.emplace_back([&, t] {
            aletsch::packed_chunk c, prev; aletsch::packed_chunk::scratch tmp; long s = 0;
            for(int r = 0; r < R; r++) for(int n = t; n < N; n += T) {
                c.append_graph(G[(size_t)n], H[(size_t)n], n % 4, tmp);
                if(c.n() >= 2048) { s += (long)c.edge_target.size() + (long)c.sample_id.size(); std::swap(prev, c); c = aletsch::packed_chunk(); c.reserve_like(prev); }
            }
            sums[(size_t)t] = s + (long)c.edge_target.size();
        });
        for(auto &x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long tot = 0; for(long x : sums) tot += x;
        printf("%s: %ld graphs on %d threads in %.3f s -> %.0f graphs/s, %.2f us per graph per thread (checksum %ld)\n", pass ? "measured" : "warm-up", (long)N * R, T, s, (double)N * R / s, s * T / ((double)N * R) * 1e6, tot);
    }
    return 0;
}
