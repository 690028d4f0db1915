// graph_io.hpp -- the reference's text form of a bundle (graph + phasing set) <-> the staged arrays of the C ABI, in C++ beside
// gpu_scallop.hpp (graphs dumped by an Aletsch build elsewhere replayed through the decomposition kernel).
//
// The *bundle dump* is what splice_graph::write(ostream&) (rnacore/splice_graph.cc:422-477) followed by hyper_set::write(ostream&)
// (scallop/hyper_set.cc:1109-1128) print:
//       # <gid> <chrm> <strand>
//       region <lpos> <rpos> <weight>         internal vertices with lpos < rpos, ascending
//       sbound <lpos of target> <weight> 1    edges source -> v, in out_edges(0) order (target, creation)
//       tbound <rpos of source> <weight> 1    edges v -> sink, in in_edges(n) order (source, creation)
//       junction <rpos of s> <lpos of t> <weight> 1     every other edge with rpos(s) < lpos(t), in gr.edges() (creation) order
//       path <n> <v0> ... <vn-1> <count> 1    hyper_set::nodes (a std::map: lexicographic) entries with more than two vertices
// numbers `fixed`, two decimals.  The dump is lossy by design (no edge between touching regions, no edge_info); nothing in the live
// reference reads it back.  The dead meta-scallop loader (meta/combined_graph.cc:355-498) shows the intended reading, which
// read_bundle_dump follows: vertices = source + regions + sink; edges created in file order (sbounds, tbounds, junctions: that order is
// their creation rank); then one edge between every pair of touching consecutive regions, weight = the weight of the region with
// fewer edges on the touching side (ties: the right one), at least min_guaranteed_edge_weight; every edge gets count = the trailing
// field and the single sample {0: weight}.
//
// write_bundle_dump prints a staged graph exactly as the reference's two writers would print the graph it was staged from -- the line
// ORDER is pinned against the reference's own graph containers (oracle/_ref/ref_graph dumps out_edges(0) / in_edges(n) / edges() for
// graphs built in a given creation order; tests/test_graphio_cpu.py).  C++11, header only.
#pragma once
#include "gpu_scallop.hpp"
#include <istream>
#include <ostream>
#include <sstream>
#include <iomanip>
#include <cstdio>

namespace aletsch {

struct dump_block { std::string gid, chrm; char strand = '.'; staged_graph g; };

class dump_error : public std::runtime_error { public: explicit dump_error(const std::string &s) : std::runtime_error("bundle dump: " + s) {} };

// every `# gid chrm strand` block of the stream as one staged graph
inline std::vector<dump_block> read_bundle_dump(std::istream &in, double min_guaranteed_edge_weight = 0.01)
{
    struct region { int32_t l, r; double w; };
    struct bound { int32_t p; double w; int c; };
    struct junc { int32_t p1, p2; double w; int c; };
    struct raw { std::string gid, chrm, strand; std::vector<region> R; std::vector<bound> sb, tb; std::vector<junc> jn; std::vector<std::pair<std::vector<int>, int> > paths; };
    std::vector<raw> blocks;
    std::string line;
    while(std::getline(in, line)) {
        std::istringstream ss(line); std::string key;
        if(!(ss >> key)) continue;
        if(key == "#") { raw b; ss >> b.gid >> b.chrm >> b.strand; blocks.push_back(b); continue; }
        if(blocks.empty()) throw dump_error("does not start with a '# gid chrm strand' line");
        raw &b = blocks.back();
        if(key == "region") { region x; if(!(ss >> x.l >> x.r >> x.w)) throw dump_error("malformed region line"); b.R.push_back(x); }
        else if(key == "sbound" || key == "tbound") { bound x; x.c = 1; if(!(ss >> x.p >> x.w)) throw dump_error("malformed boundary line"); ss >> x.c; (key == "sbound" ? b.sb : b.tb).push_back(x); }
        else if(key == "junction") { junc x; x.c = 1; if(!(ss >> x.p1 >> x.p2 >> x.w)) throw dump_error("malformed junction line"); ss >> x.c; b.jn.push_back(x); }
        else if(key == "path") { int n = 0; if(!(ss >> n) || n < 0) throw dump_error("malformed path line"); std::vector<int> v((size_t)n); for(int k = 0; k < n; k++) if(!(ss >> v[(size_t)k])) throw dump_error("malformed path line"); int c = 0; ss >> c; b.paths.push_back(std::make_pair(v, c)); }
    }
    std::vector<dump_block> out;
    for(const raw &b : blocks) {
        const int nr = (int)b.R.size(), V = nr + 2;
        for(int i = 0; i < nr; i++) if(!(b.R[(size_t)i].l < b.R[(size_t)i].r) || (i && b.R[(size_t)i - 1].r > b.R[(size_t)i].l)) throw dump_error("regions of " + b.gid + " are not ascending, disjoint, non-empty intervals");
        std::map<int32_t, int> lindex, rindex;
        for(int i = 0; i < nr; i++) { lindex[b.R[(size_t)i].l] = i + 1; rindex[b.R[(size_t)i].r] = i + 1; }
        struct edge { int s, t; double w; int c; };
        std::vector<edge> E;                                    // listing order == creation order
        for(const bound &x : b.sb) { auto f = lindex.find(x.p); if(f == lindex.end()) throw dump_error("sbound of " + b.gid + " names no region"); E.push_back(edge{0, f->second, x.w, x.c}); }
        for(const bound &x : b.tb) { auto f = rindex.find(x.p); if(f == rindex.end()) throw dump_error("tbound of " + b.gid + " names no region"); E.push_back(edge{f->second, V - 1, x.w, x.c}); }
        for(const junc &x : b.jn) { auto f1 = rindex.find(x.p1); auto f2 = lindex.find(x.p2); if(f1 == rindex.end() || f2 == lindex.end()) continue; E.push_back(edge{f1->second, f2->second, x.w, x.c}); }   // combined_graph.cc:452-453
        std::vector<int> outd((size_t)V, 0), ind((size_t)V, 0);
        for(const edge &e : E) { outd[(size_t)e.s]++; ind[(size_t)e.t]++; }
        for(int i = 1; i < nr; i++) {                           // touching consecutive regions (combined_graph.cc:467-497)
            if(b.R[(size_t)i - 1].r != b.R[(size_t)i].l) continue;
            double w = outd[(size_t)i] < ind[(size_t)i + 1] ? b.R[(size_t)i - 1].w : b.R[(size_t)i].w;
            if(w < min_guaranteed_edge_weight) w = min_guaranteed_edge_weight;
            E.push_back(edge{i, i + 1, w, 1}); outd[(size_t)i]++; ind[(size_t)i + 1]++;
        }
        int32_t left = nr ? b.R[0].l : 0, right = nr ? b.R[(size_t)nr - 1].r : 0;
        for(const bound &x : b.sb) if(x.p < left) left = x.p;
        for(const bound &x : b.tb) if(x.p > right) right = x.p;
        dump_block D; D.gid = b.gid; D.chrm = b.chrm; D.strand = b.strand.empty() ? '.' : b.strand[0];
        staged_graph &s = D.g;
        for(const edge &e : E) if(!(0 <= e.s && e.s < e.t && e.t < V)) throw dump_error("an edge of " + b.gid + " does not go from a lower to a higher vertex");
        std::vector<int> order(E.size()); for(size_t k = 0; k < E.size(); k++) order[k] = (int)k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return E[(size_t)a].s != E[(size_t)c].s ? E[(size_t)a].s < E[(size_t)c].s : E[(size_t)a].t < E[(size_t)c].t; });
        s.vertex_offset.assign((size_t)V + 1, 0);
        for(const edge &e : E) s.vertex_offset[(size_t)e.s + 1]++;
        for(int i = 0; i < V; i++) s.vertex_offset[(size_t)i + 1] += s.vertex_offset[(size_t)i];
        s.edge_sample_offset.push_back(0);
        for(int k : order) {
            const edge &e = E[(size_t)k];
            s.edge_target.push_back(e.t); s.edge_weight.push_back(e.w); s.edge_strand.push_back(0); s.edge_abd.push_back(e.w); s.edge_count.push_back(e.c); s.edge_rank.push_back(k);
            s.sample_id.push_back(0); s.sample_abd.push_back(e.w); s.edge_sample_offset.push_back((int32_t)s.sample_id.size());
        }
        s.vertex_weight.push_back(0.0); s.vertex_lpos.push_back(left); s.vertex_rpos.push_back(left); s.vertex_type.push_back(-1);
        for(const region &x : b.R) { s.vertex_weight.push_back(x.w); s.vertex_lpos.push_back(x.l); s.vertex_rpos.push_back(x.r); s.vertex_type.push_back(-1); }
        s.vertex_weight.push_back(0.0); s.vertex_lpos.push_back(right); s.vertex_rpos.push_back(right); s.vertex_type.push_back(-1);
        // hyper_set::nodes: sorted lists, equal ones add up (add_node_list, hyper_set.cc:40-48), walked in key order
        std::map<std::vector<int>, int> nodes;
        for(const auto &pc : b.paths) { std::vector<int> v = pc.first; std::sort(v.begin(), v.end()); nodes[v] += pc.second; }
        s.phasing_offset.push_back(0);
        for(const auto &kv : nodes) { for(int x : kv.first) s.phasing_vertex.push_back(x); s.phasing_offset.push_back((int32_t)s.phasing_vertex.size()); s.phasing_count.push_back(kv.second); }
        s.strand = D.strand;
        out.push_back(std::move(D));
    }
    return out;
}

// splice_graph::write(os) + hyper_set::write(os) for a staged graph (CSR rows by (source, target, creation), edge_rank = position in
// gr.edges(); empty edge_rank: the CSR position)
inline void write_bundle_dump(std::ostream &os, const staged_graph &s, const std::string &gid, const std::string &chrm)
{
    const int V = (int)s.vertex_weight.size(), n = V - 1, E = (int)s.edge_target.size();
    auto num = [](double w) { char buf[64]; snprintf(buf, sizeof(buf), "%.2f", w); return std::string(buf); };     // os << fixed << setprecision(2)
    os << "# " << gid << " " << chrm << " " << s.strand << "\n";
    for(int i = 1; i < n; i++) { if(s.vertex_lpos[(size_t)i] >= s.vertex_rpos[(size_t)i]) continue; os << "region " << s.vertex_lpos[(size_t)i] << " " << s.vertex_rpos[(size_t)i] << " " << num(s.vertex_weight[(size_t)i]) << "\n"; }
    std::vector<int> src((size_t)E);
    for(int v = 0; v < V; v++) for(int k = s.vertex_offset[(size_t)v]; k < s.vertex_offset[(size_t)v + 1]; k++) src[(size_t)k] = v;
    auto rank = [&](int k) { return s.edge_rank.empty() ? k : (int)s.edge_rank[(size_t)k]; };
    // out_edges(0): the vertex's edge set orders by (source, target, pointer == creation) (graph/edge_base.h:35-45)
    { std::vector<int> ks; for(int k = 0; k < E; k++) if(src[(size_t)k] == 0 && s.edge_target[(size_t)k] != n) ks.push_back(k);
      std::sort(ks.begin(), ks.end(), [&](int a, int c) { return s.edge_target[(size_t)a] != s.edge_target[(size_t)c] ? s.edge_target[(size_t)a] < s.edge_target[(size_t)c] : rank(a) < rank(c); });
      for(int k : ks) os << "sbound " << s.vertex_lpos[(size_t)s.edge_target[(size_t)k]] << " " << num(s.edge_weight[(size_t)k]) << " 1\n"; }
    { std::vector<int> ks; for(int k = 0; k < E; k++) if(s.edge_target[(size_t)k] == n && src[(size_t)k] != 0) ks.push_back(k);
      std::sort(ks.begin(), ks.end(), [&](int a, int c) { return src[(size_t)a] != src[(size_t)c] ? src[(size_t)a] < src[(size_t)c] : rank(a) < rank(c); });
      for(int k : ks) os << "tbound " << s.vertex_rpos[(size_t)src[(size_t)k]] << " " << num(s.edge_weight[(size_t)k]) << " 1\n"; }
    { std::vector<int> ks((size_t)E); for(int k = 0; k < E; k++) ks[(size_t)k] = k;                                   // edges(): creation order
      std::sort(ks.begin(), ks.end(), [&](int a, int c) { return rank(a) < rank(c); });
      for(int k : ks) {
          const int a = src[(size_t)k], t = s.edge_target[(size_t)k];
          if(a == 0 || t == n) continue;
          const int32_t p1 = s.vertex_rpos[(size_t)a], p2 = s.vertex_lpos[(size_t)t];
          if(p1 >= p2) continue;
          os << "junction " << p1 << " " << p2 << " " << num(s.edge_weight[(size_t)k]) << " 1\n";
      } }
    std::map<std::vector<int>, int> nodes;
    for(size_t p = 0; p < s.phasing_count.size(); p++) { std::vector<int> v(s.phasing_vertex.begin() + s.phasing_offset[p], s.phasing_vertex.begin() + s.phasing_offset[p + 1]); std::sort(v.begin(), v.end()); nodes[v] += s.phasing_count[p]; }
    for(const auto &kv : nodes) { if(kv.first.size() <= 2) continue; os << "path " << kv.first.size(); for(int x : kv.first) os << " " << x; os << " " << kv.second << " 1\n"; }
}

} // namespace aletsch
