// replay_dump.cc -- bundles dumped by an Aletsch build (splice_graph::write + hyper_set::write, rnacore/splice_graph.cc:422-477,
// scallop/hyper_set.cc:1109-1128) replayed through the MI355X path: dump -> aletsch::read_bundle_dump -> aletsch::gpu_assembly_queue
// (the queue + flusher that stands where the reference calls assembler::assemble per graph: meta/assembler.cc:296-347,370,
// meta/incubator.cc:553-577,609-637) -> the merged transcript set -> GTF by ald_gtf_format_transcript, one record per merged
// transcript as incubator::write prints the meta set (meta/incubator.cc:713-740: t.write(ss, -1, #samples)).
//
//   replay_dump [-b graphs_per_batch] [-s sample_id] [--keep-single-exon] [--echo] < bundles.dump > transcripts.gtf
//     --echo: write the dump back (aletsch::write_bundle_dump) instead of decomposing -- a byte-for-byte round trip check
//     --from-listing: stdin is not a dump but graphs listed edge by edge IN CREATION ORDER (the order gr.edges() would iterate in):
//           N, then per graph "gid chrm strand V E P", V lines "weight lpos rpos", E lines "s t weight", P lines "len count v..." --
//           and the dump of each is written: how a graph with parallel edges / an arbitrary creation order is printed
// Build: g++ -std=c++11 -O2 -Iinclude tools/replay_dump.cc -o replay_dump -Laletsch_amd/lib -laletsch_decomp -pthread
#include "../aletsch_amd/host/gpu_dispatch.hpp"
#include "../aletsch_amd/host/graph_io.hpp"
#include <iostream>
#include <cstring>

namespace {
struct no_graph {}; struct no_hyper_set {};
struct run_parameters { double max_decompose_error_ratio[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00}; double min_guaranteed_edge_weight = 0.01, min_transcript_coverage = 2.0; int max_num_exons = 10000; };   // util/parameters.cc:85-105
}

int main(int argc, char **argv)
{
    int batch = 65536, sid = 0; bool skip_single = true, echo = false;                     // skip_single_exon_transcripts = true (util/parameters.cc:34)
    for(int i = 1; i < argc; i++) {
        if(!strcmp(argv[i], "-b") && i + 1 < argc) batch = atoi(argv[++i]);
        else if(!strcmp(argv[i], "-s") && i + 1 < argc) sid = atoi(argv[++i]);
        else if(!strcmp(argv[i], "--keep-single-exon")) skip_single = false;
        else if(!strcmp(argv[i], "--echo")) echo = true;
        else if(!strcmp(argv[i], "--from-listing")) {
            int N = 0; if(!(std::cin >> N)) return 2;
            for(int g = 0; g < N; g++) {
                std::string gid, chrm, strand; int V, E, P;
                if(!(std::cin >> gid >> chrm >> strand >> V >> E >> P)) return 2;
                aletsch::staged_graph s; s.strand = strand[0];
                for(int v = 0; v < V; v++) { double w; int32_t l, r; if(!(std::cin >> w >> l >> r)) return 2; s.vertex_weight.push_back(w); s.vertex_lpos.push_back(l); s.vertex_rpos.push_back(r); s.vertex_type.push_back(-1); }
                struct edge { int s, t; double w; }; std::vector<edge> es((size_t)E);
                for(int k = 0; k < E; k++) if(!(std::cin >> es[(size_t)k].s >> es[(size_t)k].t >> es[(size_t)k].w)) return 2;
                std::vector<int> order((size_t)E); for(int k = 0; k < E; k++) order[(size_t)k] = k;
                std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return es[(size_t)a].s != es[(size_t)c].s ? es[(size_t)a].s < es[(size_t)c].s : es[(size_t)a].t < es[(size_t)c].t; });
                s.vertex_offset.assign((size_t)V + 1, 0);
                for(const edge &e : es) s.vertex_offset[(size_t)e.s + 1]++;
                for(int v = 0; v < V; v++) s.vertex_offset[(size_t)v + 1] += s.vertex_offset[(size_t)v];
                for(int k : order) { s.edge_target.push_back(es[(size_t)k].t); s.edge_weight.push_back(es[(size_t)k].w); s.edge_rank.push_back(k); }
                s.phasing_offset.push_back(0);
                for(int p = 0; p < P; p++) { int len, c; if(!(std::cin >> len >> c)) return 2; for(int q = 0; q < len; q++) { int x; if(!(std::cin >> x)) return 2; s.phasing_vertex.push_back(x); } s.phasing_offset.push_back((int32_t)s.phasing_vertex.size()); s.phasing_count.push_back(c); }
                aletsch::write_bundle_dump(std::cout, s, gid, chrm);
            }
            return 0;
        }
        else { fprintf(stderr, "usage: replay_dump [-b graphs_per_batch] [-s sample_id] [--keep-single-exon] [--echo] < bundles.dump > out.gtf\n"); return 2; }
    }
    std::vector<aletsch::dump_block> blocks;
    try { blocks = aletsch::read_bundle_dump(std::cin); }
    catch(const std::exception &e) { fprintf(stderr, "replay_dump: %s\n", e.what()); return 2; }
    if(echo) { for(const auto &b : blocks) aletsch::write_bundle_dump(std::cout, b.g, b.gid, b.chrm); return 0; }
    run_parameters cfg;
    ald_tset *tm = nullptr;
    if(ald_tset_create(0.8, &tm) != ALD_OK) return 3;
    long failed = 0;
    try {
        aletsch::gpu_assembly_queue<no_graph, no_hyper_set, run_parameters> q(cfg, tm, skip_single, 0, batch, 2);
        for(const auto &b : blocks) q.submit_staged(b.g, sid);           // one submitting thread: ticket k == block k
        q.drain();
        failed = q.failed_graphs();
    } catch(const std::exception &e) { fprintf(stderr, "replay_dump: %s\n", e.what()); return 1; }
    int64_t n = 0, ne = 0, ns = 0;
    ald_tset_size(tm, &n, &ne, &ns);
    std::vector<uint64_t> h((size_t)n + 1); std::vector<int32_t> cnt((size_t)n + 1), c1((size_t)n + 1), c2((size_t)n + 1), lr(2 * (size_t)ne + 2), ssid((size_t)ns + 1), sc1((size_t)ns + 1);
    std::vector<char> st((size_t)n + 1); std::vector<double> cov((size_t)n + 1), cov2((size_t)n + 1), conf((size_t)n + 1), abd((size_t)n + 1), scov2((size_t)ns + 1), sconf((size_t)ns + 1), sabd((size_t)ns + 1);
    std::vector<int64_t> tid((size_t)n + 1), eo((size_t)n + 2), so((size_t)n + 2);
    if(ald_tset_export(tm, h.data(), cnt.data(), st.data(), cov.data(), cov2.data(), conf.data(), abd.data(), c1.data(), c2.data(), tid.data(), eo.data(), lr.data(),
                       so.data(), ssid.data(), scov2.data(), sconf.data(), sabd.data(), sc1.data()) != ALD_OK) return 4;
    std::vector<char> buf(1 << 16);
    for(int64_t i = 0; i < n; i++) {
        const int64_t ticket = tid[(size_t)i] >> 20; const int path = (int)(tid[(size_t)i] & ((1 << 20) - 1));
        if(ticket < 0 || ticket >= (int64_t)blocks.size()) return 5;
        const aletsch::dump_block &b = blocks[(size_t)ticket];
        char id[512]; ald_transcript_id(id, sizeof(id), b.chrm.c_str(), b.gid.c_str(), path);            // "chr<chrm>.<gid>.<path>" (scallop.cc:3258)
        const int32_t nex = (int32_t)(eo[(size_t)i + 1] - eo[(size_t)i]);
        int64_t need = ald_gtf_format_transcript(buf.data(), (int64_t)buf.size(), b.chrm.c_str(), "aletsch", b.gid.c_str(), id, "", "", st[(size_t)i], cov[(size_t)i], -1.0, c2[(size_t)i], nex, lr.data() + 2 * eo[(size_t)i]);
        if(need + 1 > (int64_t)buf.size()) { buf.resize((size_t)need + 1); ald_gtf_format_transcript(buf.data(), (int64_t)buf.size(), b.chrm.c_str(), "aletsch", b.gid.c_str(), id, "", "", st[(size_t)i], cov[(size_t)i], -1.0, c2[(size_t)i], nex, lr.data() + 2 * eo[(size_t)i]); }
        fwrite(buf.data(), 1, (size_t)need, stdout);
    }
    fprintf(stderr, "replay_dump: %zu bundles, %ld not decomposed, %lld transcripts\n", blocks.size(), failed, (long long)n);
    ald_tset_destroy(tm);
    return 0;
}
