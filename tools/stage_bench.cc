// Host staging throughput (no GPU): synthetic batch -> HostBatch::add_packed -> layout -> pack_into a buffer.
//   g++ -O2 -std=c++17 -I include -I aletsch_amd/csrc tools/stage_bench.cc aletsch_amd/csrc/synth.cpp -o /tmp/stage_bench -lpthread && /tmp/stage_bench
#include "host_pack.h"
#include <chrono>
#include <cstdio>
using namespace ald;
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 100000;
    ald_synth_spec sp{}; sp.seed = 1002; sp.n_graphs = n; sp.v_min = 64; sp.v_max = 64; sp.fixed_edges = 256; sp.n_samples = 1;
    int64_t tv, te, ts, tp, tpv; ald_synth_sizes(&sp, &tv, &te, &ts, &tp, &tpv);
    std::vector<int32_t> nv(n), ne(n), np(n), voff(tv + n), etgt(te), esoff(te + n), sid(ts), lpos(tv), rpos(tv), vtype(tv), poff(tp + n), pv(tpv + 1), pc(tp + 1);
    std::vector<double> ew(te), eabd(te), sabd(ts), vw(tv); std::vector<uint8_t> es(te); std::vector<char> gs(n);
    ald_synth_fill(&sp, nv.data(), ne.data(), np.data(), voff.data(), etgt.data(), ew.data(), es.data(), eabd.data(), esoff.data(), sid.data(), sabd.data(),
                   vw.data(), lpos.data(), rpos.data(), vtype.data(), poff.data(), pv.data(), pc.data(), gs.data());
    for(int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        HostBatch B;
        int rc = B.add_packed(n, nv.data(), ne.data(), np.data(), voff.data(), etgt.data(), ew.data(), es.data(), eabd.data(), esoff.data(), sid.data(), sabd.data(),
                              vw.data(), lpos.data(), rpos.data(), vtype.data(), poff.data(), pv.data(), pc.data(), gs.data());
        auto t1 = std::chrono::steady_clock::now();
        HostBatch::Section sec[HostBatch::S_COUNT]; uint64_t bytes = B.layout(sec);
        std::vector<uint8_t> buf(bytes);
        auto t2 = std::chrono::steady_clock::now();
        B.pack_into(buf.data(), sec);
        auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        printf("rc %d  add_packed %.1f ms  alloc %.1f ms  pack %.1f ms  (%d graphs, %.1f MB)  -> %.0f graphs/s staging\n", rc, ms(t0, t1), ms(t1, t2), ms(t2, t3), n, bytes / 1e6, n / ((ms(t0, t1) + ms(t2, t3)) / 1e3));
    }
    return 0;
}
