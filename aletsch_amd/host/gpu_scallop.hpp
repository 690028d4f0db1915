// gpu_scallop.hpp -- C++ host adapter over the C ABI (include/aletsch_decomp.h) with the reference's call surface.
//
// The reference decomposes one graph with (meta/assembler.cc:1110-1121, scallop/scallop.h:31-51):
//
//     scallop sx(gx, hx, pa, false);     // ctor: splice_graph&, hyper_set&, const parameters&, bool random_ordering
//     sx.assemble();                     // int, always 0
//     for(transcript &t : sx.trsts) ...  // public: vector<path> paths; vector<transcript> trsts;
//
// aletsch::gpu_scallop keeps exactly that two-call shape (single graph = a batch of one), and
// aletsch::gpu_scallop_batch lifts it to many graphs per launch, which is where the MI355X pays off: the dispatch
// loops (meta/incubator.cc:609-637, meta/assembler.cc:296-347,370) enqueue (graph, phasing set) pairs, flush() runs
// them in one batch, and result(i) hands back what `sx.paths` would have held for graph i.
//
// The adapter is a template over the reference's own types so that it compiles INSIDE the reference tree with its
// headers (splice_graph.h, hyper_set.h, parameters.h, path.h) and needs none of them here.  Duck-typed members used:
//   SpliceGraph: num_vertices(), edges() -> pair<edge_iterator,edge_iterator> over edge_descriptor (with ->source(),
//                ->target()), get_edge_weight(e), get_edge_info(e) {.strand,.count,.abd,.samples,.spAbd},
//                get_vertex_weight(v), get_vertex_info(v) {.lpos,.rpos,.type}, .strand
//   HyperSet:    .nodes  (map<vector<int>, int>: vertex lists -> count, after the ctor + filter_nodes)
//   Parameters:  .max_decompose_error_ratio[8], .min_guaranteed_edge_weight, .min_transcript_coverage, .max_num_exons
//   Path:        .v, .junc, .length, .abd, .weight, .conf, .reads, .strand, .count      (rnacore/path.h)
//
// Canonical edge order: the reference's order is raw-pointer order, which is what gr.edges() iterates in and what
// scallop's ctor turns into edge indices (scallop.cc:24, graph_base.cc:139-153).  The adapter lays the edges out as CSR by
// (source, target, position in gr.edges()) -- the order each vertex's out-edge set iterates in -- and hands the position in
// gr.edges() over as ald_graph_view.edge_creation_rank, so that the kernel's edge ids ARE the reference's e2i values.
// The consumed-input contract is relaxed: gx / hx are left untouched (the reference empties them; its callers never read
// them again: assembler.cc:346-355).
#pragma once
#include "../../include/aletsch_decomp.h"
#include <vector>
#include <map>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <cstdint>
#include <type_traits>

namespace aletsch {

struct staged_graph {                       // owning arrays behind one ald_graph_view
    std::vector<int32_t> vertex_offset, edge_target, edge_sample_offset, sample_id, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, edge_count, edge_rank;
    std::vector<double> edge_weight, edge_abd, sample_abd, vertex_weight; std::vector<uint8_t> edge_strand; char strand = '.';
    ald_graph_view view() const {
        ald_graph_view g{};
        g.num_vertices = (int32_t)vertex_weight.size(); g.num_edges = (int32_t)edge_target.size();
        g.vertex_offset = vertex_offset.data(); g.edge_target = edge_target.data(); g.edge_weight = edge_weight.data(); g.edge_strand = edge_strand.data(); g.edge_abd = edge_abd.data();
        g.edge_sample_offset = edge_sample_offset.data(); g.sample_id = sample_id.data(); g.sample_abd = sample_abd.data();
        g.vertex_weight = vertex_weight.data(); g.vertex_lpos = vertex_lpos.data(); g.vertex_rpos = vertex_rpos.data(); g.vertex_type = vertex_type.data();
        g.num_phasing = (int32_t)phasing_count.size(); g.phasing_offset = phasing_offset.data(); g.phasing_vertex = phasing_vertex.data(); g.phasing_count = phasing_count.data();
        g.strand = strand; g.edge_count = edge_count.data(); g.edge_creation_rank = edge_rank.empty() ? nullptr : edge_rank.data();
        return g;
    }
};

template<class SpliceGraph, class HyperSet>
staged_graph stage_graph(SpliceGraph &gr, const HyperSet &hs)
{
    staged_graph s;
    const int V = (int)gr.num_vertices();
    typedef typename std::decay<decltype(*gr.edges().first)>::type edge_t;     // edge_descriptor
    struct E { int src, dst, ord; edge_t e; };
    std::vector<E> es; es.reserve(256);
    { auto pe = gr.edges(); int k = 0; for(auto it = pe.first; it != pe.second; ++it, ++k) es.push_back(E{(*it)->source(), (*it)->target(), k, *it}); }
    std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.src != b.src ? a.src < b.src : (a.dst != b.dst ? a.dst < b.dst : a.ord < b.ord); });
    s.vertex_offset.assign(V + 1, 0);
    for(const E &x : es) s.vertex_offset[x.src + 1]++;
    for(int i = 0; i < V; i++) s.vertex_offset[i + 1] += s.vertex_offset[i];
    { const size_t ne = es.size();                              // one allocation per array instead of a doubling series
      s.edge_target.reserve(ne); s.edge_weight.reserve(ne); s.edge_strand.reserve(ne); s.edge_abd.reserve(ne); s.edge_count.reserve(ne); s.edge_rank.reserve(ne);
      s.edge_sample_offset.reserve(ne + 1); s.sample_id.reserve(ne); s.sample_abd.reserve(ne);
      s.vertex_weight.reserve((size_t)V); s.vertex_lpos.reserve((size_t)V); s.vertex_rpos.reserve((size_t)V); s.vertex_type.reserve((size_t)V); }
    s.edge_sample_offset.push_back(0);
    for(const E &x : es) {
        const auto &ei = gr.get_edge_info(x.e);
        s.edge_target.push_back(x.dst); s.edge_weight.push_back(gr.get_edge_weight(x.e)); s.edge_strand.push_back((uint8_t)ei.strand); s.edge_abd.push_back(ei.abd);
        // edge_info.count travels as its own field: it starts as the number of supporting samples (meta/assembler.cc:202-231) but
        // group_start_boundaries ADDS counts along a grouped boundary (graph_reviser.cc:965-975) without touching the sample sets
        s.edge_count.push_back((int32_t)ei.count);
        s.edge_rank.push_back((int32_t)x.ord);          // scallop's edge index of this edge: get_edge_indices numbers gr.edges() in iteration order (graph_base.cc:139-153)
        for(int sp : ei.samples) { s.sample_id.push_back(sp); auto f = ei.spAbd.find(sp); s.sample_abd.push_back(f == ei.spAbd.end() ? 0.0 : f->second); }
        s.edge_sample_offset.push_back((int32_t)s.sample_id.size());
    }
    for(int i = 0; i < V; i++) { const auto &vi = gr.get_vertex_info(i); s.vertex_weight.push_back(gr.get_vertex_weight(i)); s.vertex_lpos.push_back(vi.lpos); s.vertex_rpos.push_back(vi.rpos); s.vertex_type.push_back(vi.type); }
    s.phasing_offset.push_back(0);
    for(const auto &kv : hs.nodes) { for(int x : kv.first) s.phasing_vertex.push_back(x); s.phasing_offset.push_back((int32_t)s.phasing_vertex.size()); s.phasing_count.push_back(kv.second); }
    s.strand = gr.strand;
    return s;
}

class gpu_error : public std::runtime_error {
public:
    int code;
    gpu_error(int c, const char *what) : std::runtime_error(std::string("aletsch_decomp: ") + what + " (" + std::to_string(c) + "): " + ald_last_error()), code(c) {}
};

// assembler::assemble(gx, px, sid) up to `scallop sx(gx, hx, pa)` (meta/assembler.cc:1075-1086): extend_strands, boundary grouping,
// phase projection, hyper_set(gx, px), filter_nodes -- done by the library (ald_pre_assemble) on the staged arrays; gx and px are left
// untouched.  PhaseSet: .pmap (map<vector<int32_t>, int>, rnacore/phase_set.h:24).  `status` receives 0, or the positive status word
// where the reference would have asserted in those steps (the returned graph is empty then).
template<class SpliceGraph, class PhaseSet>
staged_graph stage_raw(SpliceGraph &gx, const PhaseSet &px, int max_group_boundary_distance, int &status)
{
    struct no_hyper_set { std::map<std::vector<int>, int> nodes; } none;
    staged_graph in = stage_graph(gx, none);
    std::vector<int32_t> off(1, 0), coord, cnt;
    for(const auto &kv : px.pmap) { coord.insert(coord.end(), kv.first.begin(), kv.first.end()); off.push_back((int32_t)coord.size()); cnt.push_back((int32_t)kv.second); }
    static const int32_t zero = 0;
    ald_phase_view pv; pv.num_phases = (int32_t)cnt.size(); pv.phase_offset = off.data(); pv.phase_coord = coord.empty() ? &zero : coord.data(); pv.phase_count = cnt.empty() ? &zero : cnt.data();
    ald_graph_view gv = in.view();
    ald_staged *S = nullptr;
    status = ald_pre_assemble(&gv, &pv, max_group_boundary_distance, &S);
    if(status < 0) throw gpu_error(status, "ald_pre_assemble");
    staged_graph out;
    if(status > 0) return out;
    ald_graph_view v; ald_staged_view(S, &v);
    const int V = v.num_vertices, E = v.num_edges, P = v.num_phasing;
    const int NS = E > 0 ? v.edge_sample_offset[E] : 0, NPV = P > 0 ? v.phasing_offset[P] : 0;
    out.vertex_offset.assign(v.vertex_offset, v.vertex_offset + V + 1); out.edge_target.assign(v.edge_target, v.edge_target + E); out.edge_weight.assign(v.edge_weight, v.edge_weight + E);
    out.edge_strand.assign(v.edge_strand, v.edge_strand + E); out.edge_abd.assign(v.edge_abd, v.edge_abd + E); out.edge_sample_offset.assign(v.edge_sample_offset, v.edge_sample_offset + E + 1);
    out.sample_id.assign(v.sample_id, v.sample_id + NS); out.sample_abd.assign(v.sample_abd, v.sample_abd + NS);
    out.vertex_weight.assign(v.vertex_weight, v.vertex_weight + V); out.vertex_lpos.assign(v.vertex_lpos, v.vertex_lpos + V); out.vertex_rpos.assign(v.vertex_rpos, v.vertex_rpos + V);
    out.vertex_type.assign(v.vertex_type, v.vertex_type + V); out.phasing_offset.assign(v.phasing_offset, v.phasing_offset + P + 1); out.phasing_vertex.assign(v.phasing_vertex, v.phasing_vertex + NPV);
    out.phasing_count.assign(v.phasing_count, v.phasing_count + P); out.edge_count.assign(v.edge_count, v.edge_count + E);
    if(v.edge_creation_rank) out.edge_rank.assign(v.edge_creation_rank, v.edge_creation_rank + E);
    out.strand = v.strand;
    ald_staged_free(S);
    return out;
}

template<class Parameters>
ald_params stage_params(const Parameters &cfg)
{
    ald_params p;
    for(int i = 0; i < 8; i++) p.max_decompose_error_ratio[i] = cfg.max_decompose_error_ratio[i];
    p.min_guaranteed_edge_weight = cfg.min_guaranteed_edge_weight; p.min_transcript_coverage = cfg.min_transcript_coverage;
    p.max_num_exons = cfg.max_num_exons; p.reserved = 0;
    return p;
}

// what scallop::collect_path fills (scallop.cc:2797-2822), including the junction list derived from the ORIGINAL coordinates
template<class Path>
Path make_path(const ald_path_view &pv, const std::vector<int32_t> &lpos, const std::vector<int32_t> &rpos)
{
    Path p;
    p.v.assign(pv.vertices, pv.vertices + pv.num_vertices);
    p.length = pv.length; p.weight = pv.weight; p.abd = pv.abd; p.conf = pv.conf; p.reads = pv.reads; p.count = pv.count; p.strand = pv.strand;
    p.junc.clear();
    for(int i = 2; i + 1 < pv.num_vertices; i++) if(lpos[p.v[i]] != rpos[p.v[i - 1]]) p.junc.push_back(std::make_pair(p.v[i - 1], p.v[i]));
    return p;
}

// many graphs per launch
template<class SpliceGraph, class HyperSet, class Parameters, class Path>
class gpu_scallop_batch {
public:
    explicit gpu_scallop_batch(const Parameters &cfg, int device = 0) {
        ald_params p = stage_params(cfg);
        int rc = ald_batch_create(&p, device, &b_);
        if(rc != ALD_OK) throw gpu_error(rc, "ald_batch_create");
    }
    ~gpu_scallop_batch() { if(b_) ald_batch_destroy(b_); }
    gpu_scallop_batch(const gpu_scallop_batch &) = delete;
    gpu_scallop_batch &operator=(const gpu_scallop_batch &) = delete;

    // returns the index under which the graph's result will be available after flush()
    int enqueue(SpliceGraph &gr, const HyperSet &hs) {
        staged_graph s = stage_graph(gr, hs);
        ald_graph_view g = s.view();
        int rc = ald_batch_add_graph(b_, &g);
        if(rc != ALD_OK) throw gpu_error(rc, "ald_batch_add_graph");
        lpos_.push_back(std::move(s.vertex_lpos)); rpos_.push_back(std::move(s.vertex_rpos));
        return (int)lpos_.size() - 1;
    }
    // the same for a graph as assembler::assemble(gx, px, sid) receives it: its pre-steps (meta/assembler.cc:1075-1086) run on the device,
    // in the wave that loads the graph.  Returns the ticket; where the reference would have asserted in them status(ticket) says so.
    template<class PhaseSet>
    int enqueue_raw(SpliceGraph &gx, const PhaseSet &px, int max_group_boundary_distance = 10000) {
        struct no_hyper_set { std::map<std::vector<int>, int> nodes; } none;
        staged_graph s = stage_graph(gx, none);                      // the graph as it is; the pre-steps run on the device (ald_batch_add_graph_raw)
        std::vector<int32_t> off(1, 0), coord, cnt;
        for(const auto &kv : px.pmap) { coord.insert(coord.end(), kv.first.begin(), kv.first.end()); off.push_back((int32_t)coord.size()); cnt.push_back((int32_t)kv.second); }
        static const int32_t zero = 0;
        ald_phase_view pv; pv.num_phases = (int32_t)cnt.size(); pv.phase_offset = off.data(); pv.phase_coord = coord.empty() ? &zero : coord.data(); pv.phase_count = cnt.empty() ? &zero : cnt.data();
        ald_graph_view g = s.view();
        int rc = ald_batch_add_graph_raw(b_, &g, &pv, max_group_boundary_distance);
        if(rc != ALD_OK) throw gpu_error(rc, "ald_batch_add_graph_raw");
        lpos_.push_back(std::move(s.vertex_lpos)); rpos_.push_back(std::move(s.vertex_rpos));
        return (int)lpos_.size() - 1;
    }
    void flush() {
        int rc;
        if((rc = ald_batch_upload(b_)) != ALD_OK) throw gpu_error(rc, "ald_batch_upload");
        if((rc = ald_batch_run(b_)) != ALD_OK) throw gpu_error(rc, "ald_batch_run");
        if((rc = ald_batch_download(b_)) != ALD_OK) throw gpu_error(rc, "ald_batch_download");
    }
    // per-graph status word (ALD_ST_*): the reference would have aborted on an assert where this is >= ALD_ST_INVARIANT
    int status(int i) const { ald_result_view r; if(ald_batch_get_result(b_, i, &r) != ALD_OK) return -1; return r.status; }
    std::vector<Path> paths(int i) const {
        ald_result_view r;
        int rc = ald_batch_get_result(b_, i, &r);
        if(rc != ALD_OK) throw gpu_error(rc, "ald_batch_get_result");
        std::vector<Path> out;
        for(int k = 0; k < r.num_paths; k++) { ald_path_view pv; ald_batch_get_path(b_, i, k, &pv); out.push_back(make_path<Path>(pv, lpos_[i], rpos_[i])); }
        return out;
    }
    void clear() { ald_batch_clear(b_); lpos_.clear(); rpos_.clear(); }
    ald_batch *handle() const { return b_; }        // for the calls that take the batch itself (ald_tset_add_batch, ald_batch_export_transcripts, ...)
private:
    ald_batch *b_ = nullptr;
    std::vector<std::vector<int32_t>> lpos_, rpos_;
};

// the reference's single-graph shape: ctor + assemble() + .paths
template<class SpliceGraph, class HyperSet, class Parameters, class Path>
class gpu_scallop {
public:
    gpu_scallop(SpliceGraph &gr, HyperSet &hs, const Parameters &cfg, bool random_ordering = false, int device = 0)
        : gr_(gr), hs_(hs), batch_(cfg, device) { if(random_ordering) throw std::invalid_argument("random_ordering is not supported (always false in the reference: assembler.cc:1110)"); }
    int assemble() {
        int i = batch_.enqueue(gr_, hs_);
        batch_.flush();
        status = batch_.status(i);
        paths = batch_.paths(i);
        return 0;
    }
    std::vector<Path> paths;               // scallop::paths
    int status = 0;                        // ALD_ST_*: >= 100 where the reference would have asserted
private:
    SpliceGraph &gr_; HyperSet &hs_;
    gpu_scallop_batch<SpliceGraph, HyperSet, Parameters, Path> batch_;
};

} // namespace aletsch
