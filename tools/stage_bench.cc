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
    // per-graph entry point (what the C++ adapter uses): same arrays as the bulk form, and its own timing
    {
        HostBatch B1, B2;
        for(int round = 0; round < 2; round++) {
        B1.clear(); B2.clear();
        B1.add_packed(n, nv.data(), ne.data(), np.data(), voff.data(), etgt.data(), ew.data(), es.data(), eabd.data(), esoff.data(), sid.data(), sabd.data(),
                      vw.data(), lpos.data(), rpos.data(), vtype.data(), poff.data(), pv.data(), pc.data(), gs.data());
        auto t0 = std::chrono::steady_clock::now();
        int64_t ov = 0, ovo = 0, oe = 0, oeo = 0, os = 0, op = 0, opo = 0, opv = 0;
        for(int i = 0; i < n; i++) {
            ald_graph_view g; memset(&g, 0, sizeof(g));
            g.num_vertices = nv[i]; g.num_edges = ne[i]; g.num_phasing = np[i];
            g.vertex_offset = voff.data() + ovo; g.edge_target = etgt.data() + oe; g.edge_weight = ew.data() + oe; g.edge_strand = es.data() + oe; g.edge_abd = eabd.data() + oe;
            g.edge_sample_offset = esoff.data() + oeo; g.sample_id = sid.data() + os; g.sample_abd = sabd.data() + os;
            g.vertex_weight = vw.data() + ov; g.vertex_lpos = lpos.data() + ov; g.vertex_rpos = rpos.data() + ov; g.vertex_type = vtype.data() + ov;
            g.phasing_offset = poff.data() + opo; g.phasing_vertex = pv.data() + opv; g.phasing_count = pc.data() + op; g.strand = gs[i];
            int64_t ns = ne[i] > 0 ? g.edge_sample_offset[ne[i]] : 0, npv = np[i] > 0 ? g.phasing_offset[np[i]] : 0;
            if(B2.add_graph(g) != ALD_OK) { printf("add_graph failed: %s\n", B2.err.c_str()); return 1; }
            ov += nv[i]; ovo += nv[i] + 1; oe += ne[i]; oeo += ne[i] + 1; os += ns; op += np[i]; opo += np[i] + 1; opv += npv;
        }
        auto t1 = std::chrono::steady_clock::now();
        auto same = [](const auto &a, const auto &b) { return a.size() == b.size() && (a.size() == 0 || memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0); };
        bool ok = same(B1.edge_target, B2.edge_target) && same(B1.edge_weight, B2.edge_weight) && same(B1.edge_strand, B2.edge_strand) && same(B1.edge_abd, B2.edge_abd) && same(B1.edge_count, B2.edge_count)
               && same(B1.edge_sample_offset, B2.edge_sample_offset) && same(B1.sample_id, B2.sample_id) && same(B1.sample_abd, B2.sample_abd) && same(B1.vertex_offset, B2.vertex_offset)
               && same(B1.vertex_weight, B2.vertex_weight) && same(B1.vertex_lpos, B2.vertex_lpos) && same(B1.vertex_rpos, B2.vertex_rpos) && same(B1.vertex_type, B2.vertex_type)
               && same(B1.in_offset, B2.in_offset) && same(B1.in_edge, B2.in_edge) && same(B1.phasing_offset, B2.phasing_offset) && same(B1.phasing_vertex, B2.phasing_vertex) && same(B1.phasing_count, B2.phasing_count)
               && same(B1.g_nv, B2.g_nv) && same(B1.g_ne, B2.g_ne) && same(B1.g_np, B2.g_np) && same(B1.off_e, B2.off_e) && same(B1.off_s, B2.off_s) && same(B1.off_pv, B2.off_pv) && same(B1.graph_strand, B2.graph_strand);
        printf("add_graph x %d: %.1f ms (%.0f graphs/s on one thread); arrays identical to add_packed: %s\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count(),
               n / std::chrono::duration<double>(t1 - t0).count(), ok ? "yes" : "NO");
        if(!ok) return 1;
        }
    }
    return 0;
}
