// host_pack.h -- host-side staging of splice-graph batches into the wire format (one contiguous buffer), and
// parsing of the packed path records the kernels emit.  Plain C++; shared by the C ABI (ald_abi.cpp) and by the
// tests' single-lane emulation harness (tests/kernel_emu).
//
// Staging normalises what the reference gets for free from its containers:
//   * CSR rows sorted by target (stable)  -> edge id == position == the canonical creation order;
//   * per-edge sample lists ascending     -> std::set<int> iteration order (rnacore/edge_info.h:32);
//   * phasing lists sorted lexicographically, duplicates merged by adding counts -> hyper_set::add_node_list
//     (scallop/hyper_set.cc:40-48) into the std::map `nodes`;
//   * an in-CSR (edge ids ordered by (target, source, id)) so that the kernel links in-lists without sorting.
#pragma once
#include "decomp_common.h"
#include "../host/transcript_sink.hpp"      // sink_transcript::chain_key: the bucket key of a transcript, taken while its record is decoded
#include <vector>
#include <algorithm>
#include <numeric>
#include <cstring>
#include <string>
#include <map>
#include <thread>
#include <cstdlib>
#include <cmath>
#include <new>

namespace ald {

// std::vector that does not zero-fill on resize(): the staging arrays are sized once and then written by block copies from
// several threads (zero-filling 1.3 GB on one thread first costs more than the copies themselves)
template<class T> struct raw_init_alloc : std::allocator<T> {
    template<class U> struct rebind { typedef raw_init_alloc<U> other; };
    raw_init_alloc() = default;
    template<class U> raw_init_alloc(const raw_init_alloc<U> &) {}
    template<class U, class... A> void construct(U *p, A &&... a) { if(sizeof...(A) == 0) ::new((void*)p) U; else ::new((void*)p) U(std::forward<A>(a)...); }
};
template<class T> using rvec = std::vector<T, raw_init_alloc<T>>;

// The arrays of a batch that travel to the device.  A batch of the GPU library keeps them in PINNED memory (HostBatch(true): ald_abi.cpp
// installs the two hooks when the library is loaded), so that ald_batch_upload hands them to the copy engine as they are -- no pack into a
// second buffer; every other HostBatch (the emulation, host-only utilities) uses plain memory.  The choice is a property of the vector's
// allocator, fixed when the batch is constructed.
struct wire_hooks_t { void *(*alloc)(size_t) = nullptr; void (*release)(void*) = nullptr; };
inline wire_hooks_t &wire_hooks() { static wire_hooks_t h; return h; }
template<class T> struct wire_alloc : raw_init_alloc<T> {
    template<class U> struct rebind { typedef wire_alloc<U> other; };
    typedef std::true_type propagate_on_container_move_assignment; typedef std::true_type propagate_on_container_copy_assignment; typedef std::true_type propagate_on_container_swap;
    bool pinned = false;
    wire_alloc() = default;
    explicit wire_alloc(bool p) : pinned(p && wire_hooks().alloc && wire_hooks().release) {}
    template<class U> wire_alloc(const wire_alloc<U> &o) : pinned(o.pinned) {}
    T *allocate(size_t n) { const size_t bytes = n * sizeof(T); void *p = pinned ? wire_hooks().alloc(bytes) : ::operator new(bytes); if(!p) throw std::bad_alloc(); return static_cast<T*>(p); }
    void deallocate(T *p, size_t) { if(pinned) wire_hooks().release((void*)p); else ::operator delete((void*)p); }
    template<class U> bool operator==(const wire_alloc<U> &o) const { return pinned == o.pinned; }
    template<class U> bool operator!=(const wire_alloc<U> &o) const { return pinned != o.pinned; }
};
template<class T> using wvec = std::vector<T, wire_alloc<T>>;

struct HostBatch {
    wvec<int32_t> g_nv, g_ne, g_np;
    wvec<int64_t> off_v, off_e, off_s, off_p, off_pv;            // [n+1] prefixes, {0} for an empty batch
    wvec<int32_t> vertex_offset, edge_target; wvec<double> edge_weight; wvec<uint8_t> edge_strand; wvec<double> edge_abd;
    wvec<int32_t> edge_sample_offset, sample_id; wvec<double> sample_abd;
    wvec<double> vertex_weight; wvec<int32_t> vertex_lpos, vertex_rpos, vertex_type;
    wvec<int32_t> in_offset, in_edge;
    wvec<int32_t> phasing_offset, phasing_vertex, phasing_count; wvec<char> graph_strand;
    wvec<int32_t> edge_count;
    wvec<int32_t> edge_rank; bool has_rank = false;   // creation rank of every input edge (optional: empty until a caller supplies one, then identity-filled for the rest)
    // RAW graphs (row f1): a graph as assembler::assemble(gx, px, sid) receives it + its phase set in exon coordinates; the pre-steps
    // (extend_strands, boundary grouping, phase projection, hyper_set ctor, filter_nodes) run in the kernel's load phase.
    // g_rawdist[g] = -1: an ordinary (staged) graph, else max_group_boundary_distance of a raw one.  Kept for every graph (a few bytes);
    // the sections only travel when the batch holds a raw graph.
    wvec<int32_t> g_rawdist; wvec<int64_t> off_rp, off_rc;           // [n] / [n+1] prefix of raw phases / of their coordinates
    wvec<int32_t> rphase_offset, rphase_coord, rphase_count;          // per graph a local CSR of np + 1 entries (at off_rp[g] + g), coordinates, counts
    explicit HostBatch(bool pinned = false)
        : g_nv(wire_alloc<int32_t>(pinned)), g_ne(wire_alloc<int32_t>(pinned)), g_np(wire_alloc<int32_t>(pinned)),
          off_v(wire_alloc<int64_t>(pinned)), off_e(wire_alloc<int64_t>(pinned)), off_s(wire_alloc<int64_t>(pinned)), off_p(wire_alloc<int64_t>(pinned)), off_pv(wire_alloc<int64_t>(pinned)),
          vertex_offset(wire_alloc<int32_t>(pinned)), edge_target(wire_alloc<int32_t>(pinned)), edge_weight(wire_alloc<double>(pinned)), edge_strand(wire_alloc<uint8_t>(pinned)), edge_abd(wire_alloc<double>(pinned)),
          edge_sample_offset(wire_alloc<int32_t>(pinned)), sample_id(wire_alloc<int32_t>(pinned)), sample_abd(wire_alloc<double>(pinned)),
          vertex_weight(wire_alloc<double>(pinned)), vertex_lpos(wire_alloc<int32_t>(pinned)), vertex_rpos(wire_alloc<int32_t>(pinned)), vertex_type(wire_alloc<int32_t>(pinned)),
          in_offset(wire_alloc<int32_t>(pinned)), in_edge(wire_alloc<int32_t>(pinned)),
          phasing_offset(wire_alloc<int32_t>(pinned)), phasing_vertex(wire_alloc<int32_t>(pinned)), phasing_count(wire_alloc<int32_t>(pinned)), graph_strand(wire_alloc<char>(pinned)),
          edge_count(wire_alloc<int32_t>(pinned)), edge_rank(wire_alloc<int32_t>(pinned)),
          g_rawdist(wire_alloc<int32_t>(pinned)), off_rp(wire_alloc<int64_t>(pinned)), off_rc(wire_alloc<int64_t>(pinned)),
          rphase_offset(wire_alloc<int32_t>(pinned)), rphase_coord(wire_alloc<int32_t>(pinned)), rphase_count(wire_alloc<int32_t>(pinned))
    { off_v.assign(1, 0); off_e.assign(1, 0); off_s.assign(1, 0); off_p.assign(1, 0); off_pv.assign(1, 0); off_rp.assign(1, 0); off_rc.assign(1, 0); }
    bool has_raw = false;
    std::vector<int32_t> cur_, perm_; std::vector<uint8_t> seen_;          // scratch of add_graph, kept across calls
    std::string err;

    int n() const { return (int)g_nv.size(); }
    void clear()        // keeps every array's capacity: the next batch of the same shape is staged without touching the allocator
    {
        g_nv.clear(); g_ne.clear(); g_np.clear();
        off_v.assign(1, 0); off_e.assign(1, 0); off_s.assign(1, 0); off_p.assign(1, 0); off_pv.assign(1, 0);
        vertex_offset.clear(); edge_target.clear(); edge_weight.clear(); edge_strand.clear(); edge_abd.clear();
        edge_sample_offset.clear(); sample_id.clear(); sample_abd.clear();
        vertex_weight.clear(); vertex_lpos.clear(); vertex_rpos.clear(); vertex_type.clear();
        in_offset.clear(); in_edge.clear(); phasing_offset.clear(); phasing_vertex.clear(); phasing_count.clear(); graph_strand.clear(); edge_count.clear(); edge_rank.clear(); has_rank = false;
        g_rawdist.clear(); off_rp.assign(1, 0); off_rc.assign(1, 0); rphase_offset.clear(); rphase_coord.clear(); rphase_count.clear(); has_raw = false;
        err.clear();
    }

    // sizes of every array: a rejected graph must leave the batch exactly as it found it (the element-wise path appends while it checks)
    struct Mark { size_t a[34]; bool has_rank, has_raw; };
    Mark mark() const
    {
        Mark m = {{ g_nv.size(), g_ne.size(), g_np.size(), off_v.size(), off_e.size(), off_s.size(), off_p.size(), off_pv.size(),
                    vertex_offset.size(), edge_target.size(), edge_weight.size(), edge_strand.size(), edge_abd.size(), edge_sample_offset.size(), sample_id.size(), sample_abd.size(),
                    vertex_weight.size(), vertex_lpos.size(), vertex_rpos.size(), vertex_type.size(), in_offset.size(), in_edge.size(),
                    phasing_offset.size(), phasing_vertex.size(), phasing_count.size(), graph_strand.size(), edge_count.size(), edge_rank.size(),
                    g_rawdist.size(), off_rp.size(), off_rc.size(), rphase_offset.size(), rphase_coord.size(), rphase_count.size() }, has_rank, has_raw};
        return m;
    }
    void rollback(const Mark &m)
    {
        g_nv.resize(m.a[0]); g_ne.resize(m.a[1]); g_np.resize(m.a[2]); off_v.resize(m.a[3]); off_e.resize(m.a[4]); off_s.resize(m.a[5]); off_p.resize(m.a[6]); off_pv.resize(m.a[7]);
        vertex_offset.resize(m.a[8]); edge_target.resize(m.a[9]); edge_weight.resize(m.a[10]); edge_strand.resize(m.a[11]); edge_abd.resize(m.a[12]); edge_sample_offset.resize(m.a[13]);
        sample_id.resize(m.a[14]); sample_abd.resize(m.a[15]); vertex_weight.resize(m.a[16]); vertex_lpos.resize(m.a[17]); vertex_rpos.resize(m.a[18]); vertex_type.resize(m.a[19]);
        in_offset.resize(m.a[20]); in_edge.resize(m.a[21]); phasing_offset.resize(m.a[22]); phasing_vertex.resize(m.a[23]); phasing_count.resize(m.a[24]); graph_strand.resize(m.a[25]);
        edge_count.resize(m.a[26]); edge_rank.resize(m.a[27]); has_rank = m.has_rank;
        g_rawdist.resize(m.a[28]); off_rp.resize(m.a[29]); off_rc.resize(m.a[30]); rphase_offset.resize(m.a[31]); rphase_coord.resize(m.a[32]); rphase_count.resize(m.a[33]); has_raw = m.has_raw;
    }
    // the first graph that brings a creation rank turns the column on: every edge staged before it gets its CSR position
    void enable_rank()
    {
        if(has_rank) return;
        has_rank = true; edge_rank.resize(edge_target.size());
        for(int g = 0; g < n(); g++) { const int64_t o = off_e[g]; for(int k = 0; k < g_ne[g]; k++) edge_rank[o + k] = k; }
    }
    // is r[0..E) a permutation of 0..E-1 ?
    bool rank_is_permutation(const int32_t *r, int E, std::vector<uint8_t> &seen) const
    {
        seen.assign((size_t)E, 0);
        for(int k = 0; k < E; k++) { const int32_t x = r[k]; if(x < 0 || x >= E || seen[x]) return false; seen[x] = 1; }
        return true;
    }
    int add_graph(const ald_graph_view &g)
    {
        const Mark m = mark();
        const int rc = add_graph_body(g);
        if(rc != ALD_OK) rollback(m);
        return rc;
    }
    int add_graph_body(const ald_graph_view &g)
    {
        const int V = g.num_vertices, E = g.num_edges, P = g.num_phasing;
        if(V < 2 || E < 0 || P < 0 || !g.vertex_offset || (E > 0 && (!g.edge_target || !g.edge_weight || !g.edge_sample_offset)) || !g.vertex_weight || !g.vertex_lpos || !g.vertex_rpos) { err = "null or negative field"; return ALD_ERR_INVALID; }
        if(g.vertex_offset[0] != 0 || g.vertex_offset[V] != E) { err = "vertex_offset does not span the edges"; return ALD_ERR_INVALID; }
        for(int i = 0; i < V; i++) if(g.vertex_offset[i + 1] < g.vertex_offset[i]) { err = "vertex_offset not monotone"; return ALD_ERR_INVALID; }
        if(E > 0 && g.edge_sample_offset[0] != 0) { err = "edge_sample_offset[0] != 0"; return ALD_ERR_INVALID; }
        // ---- edges: per-row order by target (stable) ----
        std::vector<int32_t> &perm = perm_; perm.resize(E); std::iota(perm.begin(), perm.end(), 0);
        const int32_t *rk = g.edge_creation_rank;
        if(rk && E > 0 && !rank_is_permutation(rk, E, seen_)) { err = "edge_creation_rank is not a permutation of 0..E-1"; return ALD_ERR_INVALID; }
        if(rk) { bool ident = true; for(int k = 0; k < E && ident; k++) ident = rk[k] == k; if(ident) rk = nullptr; }     // the default order: nothing to carry
        for(int s = 0; s < V; s++) {
            int a = g.vertex_offset[s], b = g.vertex_offset[s + 1];
            bool sorted = true;
            for(int k = a; k < b; k++) { int t = g.edge_target[k]; if(t <= s || t >= V) { err = "edge target out of range (edges go from a lower to a higher vertex index, below V)"; return ALD_ERR_INVALID; }
                                         if(k > a && (g.edge_target[k - 1] > t || (rk && g.edge_target[k - 1] == t && rk[k - 1] > rk[k]))) sorted = false; }
            // parallel edges iterate in creation order (graph/edge_base.h:35-45): by rank when the caller gives one, else by position
            if(!sorted) std::stable_sort(perm.begin() + a, perm.begin() + b, [&](int x, int y) { return g.edge_target[x] != g.edge_target[y] ? g.edge_target[x] < g.edge_target[y] : (rk ? rk[x] < rk[y] : false); });
        }
        if(rk) enable_rank();
        size_t e0 = edge_target.size(), s0 = sample_id.size();
        // canonical input (rows sorted, sample lists ascending, strands in range: what the reference's containers produce) is appended
        // by ranges; anything else goes through the element-wise path below, which sorts and reports defects
        bool canonical = true;
        for(int k = 0; k < E && canonical; k++) if(perm[k] != k) canonical = false;
        if(canonical && E > 0) {
            const int32_t NS = g.edge_sample_offset[E];
            if(NS < 0) canonical = false;
            for(int k = 0; k < E && canonical; k++) {
                const int a = g.edge_sample_offset[k], b = g.edge_sample_offset[k + 1];
                if(b < a) { canonical = false; break; }
                if(g.edge_strand && g.edge_strand[k] > 2) { canonical = false; break; }
                if(g.edge_count && g.edge_count[k] < 0) { canonical = false; break; }
                for(int j = a + 1; j < b; j++) if(g.sample_id[j - 1] >= g.sample_id[j]) { canonical = false; break; }
            }
        }
        if(canonical && E > 0) {
            const int32_t NS = g.edge_sample_offset[E];
            edge_target.insert(edge_target.end(), g.edge_target, g.edge_target + E);
            edge_weight.insert(edge_weight.end(), g.edge_weight, g.edge_weight + E);
            if(g.edge_strand) edge_strand.insert(edge_strand.end(), g.edge_strand, g.edge_strand + E); else edge_strand.resize(e0 + E, 0);
            sample_id.insert(sample_id.end(), g.sample_id, g.sample_id + NS);
            sample_abd.insert(sample_abd.end(), g.sample_abd, g.sample_abd + NS);
            edge_sample_offset.insert(edge_sample_offset.end(), g.edge_sample_offset, g.edge_sample_offset + E + 1);
            if(g.edge_abd) edge_abd.insert(edge_abd.end(), g.edge_abd, g.edge_abd + E);
            else { edge_abd.resize(e0 + E); for(int k = 0; k < E; k++) { double sum = 0; for(int j = g.edge_sample_offset[k]; j < g.edge_sample_offset[k + 1]; j++) sum += g.sample_abd[j]; edge_abd[e0 + k] = sum; } }
            if(g.edge_count) edge_count.insert(edge_count.end(), g.edge_count, g.edge_count + E);
            else { edge_count.resize(e0 + E); for(int k = 0; k < E; k++) edge_count[e0 + k] = g.edge_sample_offset[k + 1] - g.edge_sample_offset[k]; }
            if(has_rank) { if(rk) edge_rank.insert(edge_rank.end(), rk, rk + E); else { edge_rank.resize(e0 + E); for(int k = 0; k < E; k++) edge_rank[e0 + k] = k; } }
        } else {
        edge_sample_offset.push_back(0);
        for(int k = 0; k < E; k++) {
            int q = perm[k];
            edge_target.push_back(g.edge_target[q]); edge_weight.push_back(g.edge_weight[q]);
            int st = g.edge_strand ? g.edge_strand[q] : 0;
            if(st < 0 || st > 2) { err = "edge strand not in 0..2"; return ALD_ERR_INVALID; }
            edge_strand.push_back((uint8_t)st);
            int a = g.edge_sample_offset[q], b = g.edge_sample_offset[q + 1];
            if(b < a) { err = "edge_sample_offset not monotone"; return ALD_ERR_INVALID; }
            size_t base = sample_id.size(); double sum = 0; bool asc = true;
            for(int j = a; j < b; j++) { sample_id.push_back(g.sample_id[j]); sample_abd.push_back(g.sample_abd[j]); sum += g.sample_abd[j]; if(j > a && g.sample_id[j - 1] >= g.sample_id[j]) asc = false; }
            if(!asc) {
                std::vector<std::pair<int32_t, double>> t; for(size_t j = base; j < sample_id.size(); j++) t.push_back({sample_id[j], sample_abd[j]});
                std::sort(t.begin(), t.end());
                for(size_t j = 1; j < t.size(); j++) if(t[j].first == t[j - 1].first) { err = "duplicate sample id on an edge"; return ALD_ERR_INVALID; }
                for(size_t j = 0; j < t.size(); j++) { sample_id[base + j] = t[j].first; sample_abd[base + j] = t[j].second; }
            }
            edge_abd.push_back(g.edge_abd ? g.edge_abd[q] : sum);
            if(g.edge_count && g.edge_count[q] < 0) { err = "negative edge count"; return ALD_ERR_INVALID; }
            edge_count.push_back(g.edge_count ? g.edge_count[q] : (int32_t)(b - a));
            if(has_rank) edge_rank.push_back(rk ? rk[q] : k);
            edge_sample_offset.push_back((int32_t)(sample_id.size() - s0));
        }
        }
        vertex_offset.insert(vertex_offset.end(), g.vertex_offset, g.vertex_offset + V + 1);
        // ---- in-CSR: counting sort of edge ids by target; ids ascend within a target == (source, id) order ----
        {
            size_t b0 = in_offset.size(); in_offset.resize(b0 + V + 1, 0);
            int32_t *io = in_offset.data() + b0;
            for(int k = 0; k < E; k++) io[edge_target[e0 + k] + 1]++;
            for(int i = 0; i < V; i++) io[i + 1] += io[i];
            size_t ie0 = in_edge.size(); in_edge.resize(ie0 + E);
            cur_.assign(io, io + V);
            for(int k = 0; k < E; k++) in_edge[ie0 + cur_[edge_target[e0 + k]]++] = k;
        }
        vertex_weight.insert(vertex_weight.end(), g.vertex_weight, g.vertex_weight + V); vertex_lpos.insert(vertex_lpos.end(), g.vertex_lpos, g.vertex_lpos + V); vertex_rpos.insert(vertex_rpos.end(), g.vertex_rpos, g.vertex_rpos + V);
        if(g.vertex_type) vertex_type.insert(vertex_type.end(), g.vertex_type, g.vertex_type + V); else vertex_type.resize(vertex_type.size() + V, -1);
        // ---- phasing lists -> the std::map `nodes` ----
        int np_kept = 0; size_t pv0 = phasing_vertex.size();
        phasing_offset.push_back(0);
        if(P > 0) {
            if(!g.phasing_offset || !g.phasing_count || (g.phasing_offset[P] > 0 && !g.phasing_vertex)) { err = "null phasing arrays"; return ALD_ERR_INVALID; }
            bool need_map = false;
            for(int p = 0; p < P && !need_map; p++) {
                int a = g.phasing_offset[p], b = g.phasing_offset[p + 1];
                if(b < a) { err = "phasing_offset not monotone"; return ALD_ERR_INVALID; }
                for(int k = a + 1; k < b; k++) if(g.phasing_vertex[k - 1] >= g.phasing_vertex[k]) need_map = true;   // add_node_list sorts
                if(p > 0) { int a0 = g.phasing_offset[p - 1];
                    if(!std::lexicographical_compare(g.phasing_vertex + a0, g.phasing_vertex + a, g.phasing_vertex + a, g.phasing_vertex + b)) need_map = true; }
            }
            if(!need_map) {
                for(int p = 0; p < P; p++) { for(int k = g.phasing_offset[p]; k < g.phasing_offset[p + 1]; k++) phasing_vertex.push_back(g.phasing_vertex[k]); phasing_offset.push_back((int32_t)(phasing_vertex.size() - pv0)); phasing_count.push_back(g.phasing_count[p]); np_kept++; }
            } else {
                std::map<std::vector<int32_t>, int> nodes;
                for(int p = 0; p < P; p++) { std::vector<int32_t> v(g.phasing_vertex + g.phasing_offset[p], g.phasing_vertex + g.phasing_offset[p + 1]); std::sort(v.begin(), v.end()); nodes[v] += g.phasing_count[p]; }
                for(auto &kv : nodes) { for(int x : kv.first) phasing_vertex.push_back(x); phasing_offset.push_back((int32_t)(phasing_vertex.size() - pv0)); phasing_count.push_back(kv.second); np_kept++; }
            }
        }
        g_nv.push_back(V); g_ne.push_back(E); g_np.push_back(np_kept); graph_strand.push_back(g.strand ? g.strand : '.');
        off_v.push_back(off_v.back() + V); off_e.push_back(off_e.back() + E); off_s.push_back(off_s.back() + (int64_t)(sample_id.size() - s0));
        off_p.push_back(off_p.back() + np_kept); off_pv.push_back(off_pv.back() + (int64_t)(phasing_vertex.size() - pv0));
        g_rawdist.push_back(-1); off_rp.push_back(off_rp.back()); off_rc.push_back(off_rc.back()); rphase_offset.push_back(0);
        return ALD_OK;
    }
    // A graph BEFORE the pre-steps of assemble(gx, px, sid) + its phase set (meta/assembler.cc:1075-1086).  The graph is staged like any
    // other (CSR rows sorted, creation ranks carried); its own phasing arrays are ignored, the phases travel as exon coordinates and the
    // kernel derives the phasing lists from them.  What the reference's grouping is undefined on is refused here, as ald_pre_assemble does.
    int add_graph_raw(const ald_graph_view &g0, const ald_phase_view *ph, int32_t max_group_boundary_distance)
    {
        if(max_group_boundary_distance < 0) { err = "negative max_group_boundary_distance"; return ALD_ERR_INVALID; }
        const int P = ph ? ph->num_phases : 0;
        if(P < 0 || (P > 0 && (!ph->phase_offset || !ph->phase_count || (ph->phase_offset[P] > 0 && !ph->phase_coord)))) { err = "null phase arrays"; return ALD_ERR_INVALID; }
        for(int p = 0; p < P; p++) { const int a = ph->phase_offset[p], b = ph->phase_offset[p + 1]; if(b <= a || ((b - a) & 1)) { err = "a phase is a non-empty list of exon coordinate PAIRS"; return ALD_ERR_INVALID; } }   // phase_set::add asserts
        const Mark m = mark();
        ald_graph_view g = g0; g.num_phasing = 0; g.phasing_offset = nullptr; g.phasing_vertex = nullptr; g.phasing_count = nullptr;
        int rc = add_graph_body(g);
        if(rc != ALD_OK) { rollback(m); return rc; }
        const int gi = n() - 1, V = g_nv[gi], E = g_ne[gi]; const int64_t oe = off_e[gi], ovo = off_v[gi] + gi;
        // parallel edges out of the source / into the sink: the rows are sorted by target, the in-CSR by source
        for(int k = vertex_offset[ovo] + 1; k < vertex_offset[ovo + 1]; k++) if(edge_target[oe + k] == edge_target[oe + k - 1]) { rollback(m); err = "parallel edges out of the source: the reference's boundary grouping is undefined on them"; return ALD_ERR_INVALID; }
        { int last = -1; for(int k = in_offset[ovo + V - 1]; k < in_offset[ovo + V]; k++) { const int e = in_edge[oe + k]; int s = 0; while(vertex_offset[ovo + s + 1] <= e) s++; if(s == last) { rollback(m); err = "parallel edges into the sink: the reference's boundary grouping is undefined on them"; return ALD_ERR_INVALID; } last = s; } }
        (void)E;
        g_rawdist[(size_t)gi] = max_group_boundary_distance; has_raw = true;
        const size_t c0 = rphase_coord.size();
        for(int p = 0; p < P; p++) { for(int k = ph->phase_offset[p]; k < ph->phase_offset[p + 1]; k++) rphase_coord.push_back(ph->phase_coord[k]); rphase_offset.push_back((int32_t)(rphase_coord.size() - c0)); rphase_count.push_back(ph->phase_count[p]); }
        off_rp.back() += P; off_rc.back() += (int64_t)(rphase_coord.size() - c0);
        return ALD_OK;
    }

    int add_packed_serial(int32_t n, const int32_t *nv, const int32_t *ne, const int32_t *np,
                   const int32_t *voff, const int32_t *etgt, const double *ew, const uint8_t *estrand, const double *eabd,
                   const int32_t *esoff, const int32_t *sid, const double *sabd,
                   const double *vw, const int32_t *lpos, const int32_t *rpos, const int32_t *vtype,
                   const int32_t *poff, const int32_t *pv, const int32_t *pc, const char *gstrand, const int32_t *ecount = nullptr, const int32_t *erank = nullptr)
    {
        int64_t ov = 0, ovo = 0, oe = 0, oeo = 0, os = 0, op = 0, opo = 0, opv = 0;
        const Mark m0 = mark();                     // all or nothing: a defect in graph i also takes graphs 0..i-1 of this call back out
        for(int i = 0; i < n; i++) {
            ald_graph_view g; memset(&g, 0, sizeof(g));
            int V = nv[i], E = ne[i], P = np ? np[i] : 0;
            g.num_vertices = V; g.num_edges = E; g.num_phasing = P;
            g.vertex_offset = voff + ovo; g.edge_target = etgt + oe; g.edge_weight = ew + oe; g.edge_strand = estrand ? estrand + oe : nullptr; g.edge_abd = eabd ? eabd + oe : nullptr;
            g.edge_sample_offset = esoff + oeo; g.sample_id = sid + os; g.sample_abd = sabd + os;
            g.vertex_weight = vw + ov; g.vertex_lpos = lpos + ov; g.vertex_rpos = rpos + ov; g.vertex_type = vtype ? vtype + ov : nullptr;
            g.phasing_offset = poff ? poff + opo : nullptr; g.phasing_vertex = pv ? pv + opv : nullptr; g.phasing_count = pc ? pc + op : nullptr;
            g.strand = gstrand ? gstrand[i] : '.'; g.edge_count = ecount ? ecount + oe : nullptr; g.edge_creation_rank = erank ? erank + oe : nullptr;
            if(V < 2 || E < 0) { err = "bad graph size"; rollback(m0); return ALD_ERR_INVALID; }
            int64_t ns = E > 0 ? g.edge_sample_offset[E] : 0, npv = (P > 0 && poff) ? g.phasing_offset[P] : 0;
            int rc = add_graph(g);
            if(rc != ALD_OK) { rollback(m0); err = "graph " + std::to_string(i) + " of the call: " + err; return rc; }
            ov += V; ovo += V + 1; oe += E; oeo += E + 1; os += ns; op += P; opo += P + 1; opv += npv;
        }
        return ALD_OK;
    }

    // Bulk form.  A batch whose graphs are already canonical -- CSR rows sorted by target, sample lists ascending, phasing lists
    // strictly ascending and in lexicographic order: what the reference's containers give and what every generator here emits --
    // is staged by block copies, the only per-edge work being the in-CSR counting sort, with the graphs split over host threads
    // (all sizes are known up front, so every thread writes its own disjoint ranges).  Anything else takes the per-graph path
    // above, which normalises.
    int add_packed(int32_t n, const int32_t *nv, const int32_t *ne, const int32_t *np,
                   const int32_t *voff, const int32_t *etgt, const double *ew, const uint8_t *estrand, const double *eabd,
                   const int32_t *esoff, const int32_t *sid, const double *sabd,
                   const double *vw, const int32_t *lpos, const int32_t *rpos, const int32_t *vtype,
                   const int32_t *poff, const int32_t *pv, const int32_t *pc, const char *gstrand, const int32_t *ecount = nullptr, const int32_t *erank = nullptr)
    {
        if(n <= 0) return ALD_OK;
        // offsets of every graph inside the caller's concatenated arrays
        std::vector<int64_t> iv(n + 1, 0), ie(n + 1, 0), is(n + 1, 0), ip(n + 1, 0), ipv(n + 1, 0);
        for(int i = 0; i < n; i++) {
            const int V = nv[i], E = ne[i], P = np ? np[i] : 0;
            if(V < 2 || E < 0 || P < 0) { err = "bad graph size"; return ALD_ERR_INVALID; }
            if(P > 0 && (!poff || !pc)) { err = "null phasing arrays"; return ALD_ERR_INVALID; }
            iv[i + 1] = iv[i] + V; ie[i + 1] = ie[i] + E; ip[i + 1] = ip[i] + P;
            const int64_t ns = E > 0 ? esoff[ie[i] + i + E] : 0, npv = P > 0 ? poff[ip[i] + i + P] : 0;
            if(ns < 0 || npv < 0) { err = "negative offsets"; return ALD_ERR_INVALID; }
            is[i + 1] = is[i] + ns; ipv[i + 1] = ipv[i] + npv;
        }
        const int64_t TV = iv[n], TE = ie[n], TS = is[n], TP = ip[n], TPV = ipv[n];
        if(TPV > 0 && !pv) { err = "null phasing arrays"; return ALD_ERR_INVALID; }
        unsigned nthr = std::thread::hardware_concurrency(); if(nthr == 0) nthr = 1; if(nthr > 16) nthr = 16;
        if(const char *ev = getenv("ALD_STAGE_THREADS")) { int k = atoi(ev); if(k >= 1 && k <= 64) nthr = (unsigned)k; }
        if((int64_t)nthr > (TE >> 16) + 1) nthr = (unsigned)((TE >> 16) + 1);                      // not worth a thread below ~64k edges
        // ---- ONE parallel pass: every thread checks a graph (canonical and valid?) and, while its arrays are still in the cache, copies
        // it to its place (all sizes are known up front, so every thread writes its own disjoint ranges).  A graph that is not canonical
        // or not valid ends the pass: the batch goes back to what it was and the per-graph path normalises, or names the defect.
        const Mark m0 = mark();
        if(erank) enable_rank();                     // carried tentatively; dropped again below when it is the identity everywhere
        const size_t n0 = g_nv.size(), v0 = vertex_weight.size(), vo0 = vertex_offset.size(), e0 = edge_target.size(), eo0 = edge_sample_offset.size(), s0 = sample_id.size(),
                     p0 = phasing_count.size(), po0 = phasing_offset.size(), pv0 = phasing_vertex.size();
        g_nv.insert(g_nv.end(), nv, nv + n); g_ne.insert(g_ne.end(), ne, ne + n);
        if(np) g_np.insert(g_np.end(), np, np + n); else g_np.resize(n0 + n, 0);
        graph_strand.resize(n0 + n); for(int i = 0; i < n; i++) graph_strand[n0 + i] = (gstrand && gstrand[i]) ? gstrand[i] : '.';
        off_v.resize(n0 + n + 1); off_e.resize(n0 + n + 1); off_s.resize(n0 + n + 1); off_p.resize(n0 + n + 1); off_pv.resize(n0 + n + 1);
        { const int64_t rp = off_rp.back(), rc = off_rc.back(); g_rawdist.resize(n0 + n); off_rp.resize(n0 + n + 1); off_rc.resize(n0 + n + 1); const size_t r0 = rphase_offset.size(); rphase_offset.resize(r0 + n);
          for(int i = 0; i < n; i++) { g_rawdist[n0 + i] = -1; off_rp[n0 + i + 1] = rp; off_rc[n0 + i + 1] = rc; rphase_offset[r0 + i] = 0; } }
        for(int i = 0; i < n; i++) { off_v[n0 + i + 1] = off_v[n0] + iv[i + 1]; off_e[n0 + i + 1] = off_e[n0] + ie[i + 1]; off_s[n0 + i + 1] = off_s[n0] + is[i + 1]; off_p[n0 + i + 1] = off_p[n0] + ip[i + 1]; off_pv[n0 + i + 1] = off_pv[n0] + ipv[i + 1]; }
        vertex_offset.resize(vo0 + TV + n); in_offset.resize(vo0 + TV + n);
        edge_target.resize(e0 + TE); edge_weight.resize(e0 + TE); edge_strand.resize(e0 + TE); edge_abd.resize(e0 + TE); edge_count.resize(e0 + TE); in_edge.resize(e0 + TE); if(has_rank) edge_rank.resize(e0 + TE);
        edge_sample_offset.resize(eo0 + TE + n); sample_id.resize(s0 + TS); sample_abd.resize(s0 + TS);
        vertex_weight.resize(v0 + TV); vertex_lpos.resize(v0 + TV); vertex_rpos.resize(v0 + TV); vertex_type.resize(v0 + TV);
        phasing_offset.resize(po0 + TP + n); phasing_vertex.resize(pv0 + TPV); phasing_count.resize(p0 + TP);
        std::vector<int> verdict(nthr, 0);           // 0 ok, 1 not canonical, 2 invalid
        std::vector<int> permuted(nthr, 0);          // some graph of the slice carries a creation rank other than the CSR position
        auto work = [&](unsigned t) {
            const int g0 = (int)((int64_t)n * t / nthr), g1 = (int)((int64_t)n * (t + 1) / nthr);
            int worst = 0; std::vector<uint8_t> seen; std::vector<int32_t> cur;
            for(int g = g0; g < g1; g++) {
                const int V = nv[g], E = ne[g], P = np ? np[g] : 0;
                const int32_t *vo = voff + iv[g] + g, *tg = etgt + ie[g], *so = esoff + ie[g] + g;
                // ---- the checks ----
                if(vo[0] != 0 || vo[V] != E || (E > 0 && so[0] != 0)) { worst = 2; break; }
                for(int s = 0; s < V && worst < 2; s++) {
                    if(vo[s + 1] < vo[s]) { worst = 2; break; }
                    for(int k = vo[s]; k < vo[s + 1]; k++) { int t2 = tg[k]; if(t2 <= s || t2 >= V) { worst = 2; break; } if(k > vo[s] && (tg[k - 1] > t2 || (erank && tg[k - 1] == t2 && erank[ie[g] + k - 1] > erank[ie[g] + k]))) worst = 1; }
                }
                if(erank && worst < 2 && E > 0 && !rank_is_permutation(erank + ie[g], E, seen)) { worst = 2; break; }
                if(erank && !permuted[t]) for(int k = 0; k < E; k++) if(erank[ie[g] + k] != k) { permuted[t] = 1; break; }
                for(int k = 0; k < E && worst < 2; k++) {
                    if(so[k + 1] < so[k]) { worst = 2; break; }
                    if(estrand && estrand[ie[g] + k] > 2) { worst = 2; break; }
                    if(ecount && ecount[ie[g] + k] < 0) { worst = 2; break; }
                    for(int j = so[k] + 1; j < so[k + 1]; j++) if(sid[is[g] + j - 1] >= sid[is[g] + j]) worst = 1;
                }
                if(P > 0 && worst < 2) {
                    const int32_t *po = poff + ip[g] + g, *v = pv + ipv[g];
                    if(po[0] != 0) { worst = 2; break; }
                    for(int p = 0; p < P && worst < 2; p++) {
                        if(po[p + 1] < po[p]) { worst = 2; break; }
                        for(int k = po[p] + 1; k < po[p + 1]; k++) if(v[k - 1] >= v[k]) worst = 1;
                        if(p > 0 && !std::lexicographical_compare(v + po[p - 1], v + po[p], v + po[p], v + po[p + 1])) worst = 1;
                    }
                }
                if(worst) break;                                    // the batch is rolled back: nothing more to stage here
                // ---- the copy: graph g to its place ----
                const int64_t a_v = iv[g], a_e = ie[g], a_s = is[g], a_p = ip[g], a_pv = ipv[g]; const int64_t NS = is[g + 1] - a_s, NPV = ipv[g + 1] - a_pv;
                memcpy(&vertex_offset[vo0 + a_v + g], vo, 4 * ((size_t)V + 1));
                memcpy(&edge_sample_offset[eo0 + a_e + g], so, 4 * ((size_t)E + 1));
                if(E > 0) {
                    memcpy(&edge_target[e0 + a_e], tg, 4 * (size_t)E); memcpy(&edge_weight[e0 + a_e], ew + a_e, 8 * (size_t)E);
                    if(estrand) memcpy(&edge_strand[e0 + a_e], estrand + a_e, (size_t)E); else memset(&edge_strand[e0 + a_e], 0, (size_t)E);
                    if(eabd) memcpy(&edge_abd[e0 + a_e], eabd + a_e, 8 * (size_t)E);
                    if(ecount) memcpy(&edge_count[e0 + a_e], ecount + a_e, 4 * (size_t)E);
                    if(has_rank && erank) memcpy(&edge_rank[e0 + a_e], erank + a_e, 4 * (size_t)E);
                }
                if(NS > 0) { memcpy(&sample_id[s0 + a_s], sid + a_s, 4 * (size_t)NS); memcpy(&sample_abd[s0 + a_s], sabd + a_s, 8 * (size_t)NS); }
                memcpy(&vertex_weight[v0 + a_v], vw + a_v, 8 * (size_t)V); memcpy(&vertex_lpos[v0 + a_v], lpos + a_v, 4 * (size_t)V); memcpy(&vertex_rpos[v0 + a_v], rpos + a_v, 4 * (size_t)V);
                if(vtype) memcpy(&vertex_type[v0 + a_v], vtype + a_v, 4 * (size_t)V); else for(int64_t k = a_v; k < a_v + V; k++) vertex_type[v0 + k] = -1;
                if(poff) memcpy(&phasing_offset[po0 + a_p + g], poff + a_p + g, 4 * ((size_t)P + 1)); else memset(&phasing_offset[po0 + a_p + g], 0, 4 * ((size_t)P + 1));
                if(NPV > 0) memcpy(&phasing_vertex[pv0 + a_pv], pv + a_pv, 4 * (size_t)NPV);
                if(P > 0) memcpy(&phasing_count[p0 + a_p], pc + a_p, 4 * (size_t)P);
                // defaults that depend on the sample lists, and the in-CSR (counting sort of edge ids by target)
                if(!eabd) for(int k = 0; k < E; k++) { double sum = 0; for(int j = so[k]; j < so[k + 1]; j++) sum += sabd[is[g] + j]; edge_abd[e0 + ie[g] + k] = sum; }
                if(!ecount) for(int k = 0; k < E; k++) edge_count[e0 + ie[g] + k] = so[k + 1] - so[k];
                if(has_rank && !erank) for(int k = 0; k < E; k++) edge_rank[e0 + ie[g] + k] = k;
                int32_t *io = &in_offset[vo0 + iv[g] + g]; int32_t *ied = E > 0 ? &in_edge[e0 + ie[g]] : nullptr;
                for(int i = 0; i <= V; i++) io[i] = 0;
                for(int k = 0; k < E; k++) io[tg[k] + 1]++;
                for(int i = 0; i < V; i++) io[i + 1] += io[i];
                cur.assign(io, io + V);
                for(int k = 0; k < E; k++) ied[cur[tg[k]]++] = k;
            }
            verdict[t] = worst;
        };
        run_threads(nthr, work);
        int worst = 0; for(unsigned t = 0; t < nthr; t++) worst = std::max(worst, verdict[t]);
        if(worst != 0) {      // the per-graph path normalises, or names the defect
            rollback(m0);
            return add_packed_serial(n, nv, ne, np, voff, etgt, ew, estrand, eabd, esoff, sid, sabd, vw, lpos, rpos, vtype, poff, pv, pc, gstrand, ecount, erank);
        }
        if(erank && !m0.has_rank) { bool any = false; for(unsigned t = 0; t < nthr; t++) any = any || permuted[t]; if(!any) { edge_rank.clear(); has_rank = false; } }     // identity everywhere: nothing to carry
        return ALD_OK;
    }
    // add_packed for a run of graphs some (or all) of which are RAW: raw_dist[i] >= 0 marks graph i as one (its max_group_boundary_distance),
    // such a graph brings phases (nph[i] of them: local CSR rpoff with nph[i] + 1 entries per graph, coordinates, counts) and no phasing lists
    int add_packed_raw(int32_t n, const int32_t *nv, const int32_t *ne, const int32_t *np,
                       const int32_t *voff, const int32_t *etgt, const double *ew, const uint8_t *estrand, const double *eabd,
                       const int32_t *esoff, const int32_t *sid, const double *sabd,
                       const double *vw, const int32_t *lpos, const int32_t *rpos, const int32_t *vtype,
                       const int32_t *poff, const int32_t *pv, const int32_t *pc, const char *gstrand, const int32_t *ecount, const int32_t *erank,
                       const int32_t *raw_dist, const int32_t *nph, const int32_t *rpoff, const int32_t *rpcoord, const int32_t *rpcount)
    {
        if(n <= 0) return ALD_OK;
        if(!raw_dist) return add_packed(n, nv, ne, np, voff, etgt, ew, estrand, eabd, esoff, sid, sabd, vw, lpos, rpos, vtype, poff, pv, pc, gstrand, ecount, erank);
        const Mark m = mark(); const size_t n0 = (size_t)this->n();
        int rc = add_packed(n, nv, ne, np, voff, etgt, ew, estrand, eabd, esoff, sid, sabd, vw, lpos, rpos, vtype, poff, pv, pc, gstrand, ecount, erank);
        if(rc != ALD_OK) return rc;
        // phases: offsets of every graph inside the caller's arrays
        std::vector<int64_t> ip((size_t)n + 1, 0), ic((size_t)n + 1, 0);
        for(int i = 0; i < n; i++) {
            const int P = (raw_dist[i] >= 0 && nph) ? nph[i] : 0;
            if(P < 0 || (P > 0 && (!rpoff || !rpcount))) { rollback(m); err = "null phase arrays"; return ALD_ERR_INVALID; }
            ip[(size_t)i + 1] = ip[(size_t)i] + P;
            const int64_t nc = P > 0 ? rpoff[ip[(size_t)i] + i + P] : 0;
            if(nc < 0 || (nc > 0 && !rpcoord)) { rollback(m); err = "null phase arrays"; return ALD_ERR_INVALID; }
            ic[(size_t)i + 1] = ic[(size_t)i] + nc;
        }
        unsigned nthr = std::thread::hardware_concurrency(); if(nthr == 0) nthr = 1; if(nthr > 16) nthr = 16;
        if(const char *ev = getenv("ALD_STAGE_THREADS")) { int k = atoi(ev); if(k >= 1 && k <= 64) nthr = (unsigned)k; }
        if(n < 4096) nthr = 1;
        std::vector<int> verdict(nthr, 0);            // 1: a raw graph with phasing lists, 2: malformed phase, 3 / 4: parallel source / sink edges
        run_threads(nthr, [&](unsigned t) {
            int worst = 0;
            for(int i = (int)((int64_t)n * t / nthr); i < (int)((int64_t)n * (t + 1) / nthr) && !worst; i++) {
                if(raw_dist[i] < 0) continue;
                const size_t g = n0 + (size_t)i; const int V = g_nv[g]; const int64_t oe = off_e[g], ovo = off_v[g] + (int64_t)g;
                if(g_np[g] != 0) { worst = 1; break; }
                const int P = (int)(ip[(size_t)i + 1] - ip[(size_t)i]); const int32_t *po = rpoff ? rpoff + ip[(size_t)i] + i : nullptr;
                if(P > 0 && po[0] != 0) { worst = 2; break; }
                for(int p = 0; p < P; p++) { const int a = po[p], b = po[p + 1]; if(b <= a || ((b - a) & 1)) { worst = 2; break; } }      // phase_set::add asserts
                if(worst) break;
                for(int k = vertex_offset[ovo] + 1; k < vertex_offset[ovo + 1]; k++) if(edge_target[oe + k] == edge_target[oe + k - 1]) { worst = 3; break; }
                if(worst) break;
                int s = 0, last = -1;
                for(int k = in_offset[ovo + V - 1]; k < in_offset[ovo + V]; k++) { const int e = in_edge[oe + k]; while(vertex_offset[ovo + s + 1] <= e) s++; if(s == last) { worst = 4; break; } last = s; }   // in-CSR ids ascend: so do their sources
            }
            verdict[t] = worst;
        });
        int worst = 0; for(unsigned t = 0; t < nthr; t++) worst = std::max(worst, verdict[t]);
        if(worst) { rollback(m); err = worst == 1 ? "a raw graph brings phases, not phasing lists" : worst == 2 ? "a phase is a non-empty list of exon coordinate PAIRS"
                                       : worst == 3 ? "parallel edges out of the source: the reference's boundary grouping is undefined on them" : "parallel edges into the sink: the reference's boundary grouping is undefined on them"; return ALD_ERR_INVALID; }
        // the raw bookkeeping of the new graphs, laid out again with their phases
        const int64_t rp0 = off_rp[n0], rc0 = off_rc[n0];
        rphase_offset.resize(m.a[31] + (size_t)ip[(size_t)n] + (size_t)n); rphase_coord.resize(m.a[32] + (size_t)ic[(size_t)n]); rphase_count.resize(m.a[33] + (size_t)ip[(size_t)n]);
        for(int i = 0; i < n; i++) {
            g_rawdist[n0 + (size_t)i] = raw_dist[i] >= 0 ? raw_dist[i] : -1;
            off_rp[n0 + (size_t)i + 1] = rp0 + ip[(size_t)i + 1]; off_rc[n0 + (size_t)i + 1] = rc0 + ic[(size_t)i + 1];
            const int P = (int)(ip[(size_t)i + 1] - ip[(size_t)i]);
            int32_t *dst = &rphase_offset[m.a[31] + (size_t)ip[(size_t)i] + (size_t)i];
            if(P > 0) memcpy(dst, rpoff + ip[(size_t)i] + i, 4 * ((size_t)P + 1)); else dst[0] = 0;
            if(P > 0) memcpy(&rphase_count[m.a[33] + (size_t)ip[(size_t)i]], rpcount + ip[(size_t)i], 4 * (size_t)P);
            const int64_t nc = ic[(size_t)i + 1] - ic[(size_t)i];
            if(nc > 0) memcpy(&rphase_coord[m.a[32] + (size_t)ic[(size_t)i]], rpcoord + ic[(size_t)i], 4 * (size_t)nc);
            if(raw_dist[i] >= 0) has_raw = true;
        }
        return ALD_OK;
    }
    template<class F> static void run_threads(unsigned nthr, F &&f)
    {
        if(nthr <= 1) { f(0u); return; }
        std::vector<std::thread> th; th.reserve(nthr - 1);
        for(unsigned t = 1; t < nthr; t++) th.emplace_back([&f, t]() { f(t); });
        f(0u);
        for(auto &x : th) x.join();
    }

    // ---- one contiguous buffer; section offsets are 256-byte aligned ----
    struct Section { const void *src; uint64_t bytes; uint64_t off; };
    enum { S_NV, S_NE, S_NP, S_OFFV, S_OFFE, S_OFFS, S_OFFP, S_OFFPV, S_VOFF, S_ETGT, S_EW, S_ESTRAND, S_EABD, S_ESOFF, S_SID, S_SABD,
           S_VW, S_LPOS, S_RPOS, S_VTYPE, S_INOFF, S_INEDGE, S_POFF, S_PV, S_PC, S_GSTRAND, S_ECOUNT, S_ERANK,
           S_RAWDIST, S_OFFRP, S_OFFRC, S_RPOFF, S_RPCOORD, S_RPCOUNT, S_COUNT };
    uint64_t layout(Section sec[S_COUNT]) const
    {
        auto set = [&](int i, const void *p, uint64_t b) { sec[i].src = p; sec[i].bytes = b; };
        set(S_NV, g_nv.data(), 4ull * g_nv.size()); set(S_NE, g_ne.data(), 4ull * g_ne.size()); set(S_NP, g_np.data(), 4ull * g_np.size());
        set(S_OFFV, off_v.data(), 8ull * off_v.size()); set(S_OFFE, off_e.data(), 8ull * off_e.size()); set(S_OFFS, off_s.data(), 8ull * off_s.size());
        set(S_OFFP, off_p.data(), 8ull * off_p.size()); set(S_OFFPV, off_pv.data(), 8ull * off_pv.size());
        set(S_VOFF, vertex_offset.data(), 4ull * vertex_offset.size()); set(S_ETGT, edge_target.data(), 4ull * edge_target.size());
        set(S_EW, edge_weight.data(), 8ull * edge_weight.size()); set(S_ESTRAND, edge_strand.data(), edge_strand.size()); set(S_EABD, edge_abd.data(), 8ull * edge_abd.size());
        set(S_ESOFF, edge_sample_offset.data(), 4ull * edge_sample_offset.size()); set(S_SID, sample_id.data(), 4ull * sample_id.size()); set(S_SABD, sample_abd.data(), 8ull * sample_abd.size());
        set(S_VW, vertex_weight.data(), 8ull * vertex_weight.size()); set(S_LPOS, vertex_lpos.data(), 4ull * vertex_lpos.size()); set(S_RPOS, vertex_rpos.data(), 4ull * vertex_rpos.size());
        set(S_VTYPE, vertex_type.data(), 4ull * vertex_type.size()); set(S_INOFF, in_offset.data(), 4ull * in_offset.size()); set(S_INEDGE, in_edge.data(), 4ull * in_edge.size());
        set(S_POFF, phasing_offset.data(), 4ull * phasing_offset.size()); set(S_PV, phasing_vertex.data(), 4ull * phasing_vertex.size()); set(S_PC, phasing_count.data(), 4ull * phasing_count.size());
        set(S_GSTRAND, graph_strand.data(), graph_strand.size()); set(S_ECOUNT, edge_count.data(), 4ull * edge_count.size());
        set(S_ERANK, edge_rank.data(), has_rank ? 4ull * edge_rank.size() : 0);      // travels only when some caller supplied a creation rank
        set(S_RAWDIST, g_rawdist.data(), has_raw ? 4ull * g_rawdist.size() : 0); set(S_OFFRP, off_rp.data(), has_raw ? 8ull * off_rp.size() : 0); set(S_OFFRC, off_rc.data(), has_raw ? 8ull * off_rc.size() : 0);
        set(S_RPOFF, rphase_offset.data(), has_raw ? 4ull * rphase_offset.size() : 0); set(S_RPCOORD, rphase_coord.data(), has_raw ? 4ull * rphase_coord.size() : 0); set(S_RPCOUNT, rphase_count.data(), has_raw ? 4ull * rphase_count.size() : 0);
        uint64_t o = 0;
        for(int i = 0; i < S_COUNT; i++) { sec[i].off = o; o = (o + sec[i].bytes + 255) / 256 * 256; }
        return o < 256 ? 256 : o;
    }
    void pack_into(uint8_t *dst, const Section sec[S_COUNT]) const { pack_range(dst, sec, 0, ~0ull); }
    // the bytes [lo, hi) of the wire buffer only (sections are laid out back to back): ald_batch_upload packs the buffer chunk by chunk
    // and sends every chunk on its way as soon as it is complete, so that packing chunk k+1 runs under the H2D copy of chunk k
    void pack_range(uint8_t *dst, const Section sec[S_COUNT], uint64_t lo, uint64_t hi) const
    {
        struct Piece { uint64_t off; const uint8_t *src; uint64_t len; };
        Piece pc[S_COUNT]; int np = 0; uint64_t tot = 0;
        for(int i = 0; i < S_COUNT; i++) {
            const uint64_t a = std::max<uint64_t>(lo, sec[i].off), b = std::min<uint64_t>(hi, sec[i].off + sec[i].bytes);
            if(!sec[i].bytes || b <= a) continue;
            pc[np++] = Piece{a, (const uint8_t*)sec[i].src + (a - sec[i].off), b - a}; tot += b - a;
        }
        if(!tot) return;
        unsigned nthr = std::thread::hardware_concurrency(); if(nthr == 0) nthr = 1; if(nthr > 16) nthr = 16;
        if(const char *ev = getenv("ALD_STAGE_THREADS")) { int k = atoi(ev); if(k >= 1 && k <= 64) nthr = (unsigned)k; }
        if(tot < (8u << 20)) nthr = 1;
        run_threads(nthr, [&](unsigned t) {                     // thread t copies bytes [tot * t / nthr, tot * (t + 1) / nthr) of the pieces laid end to end
            uint64_t a = tot * t / nthr, b = tot * (t + 1) / nthr, base = 0;
            for(int i = 0; i < np && a < b; i++) {
                const uint64_t end = base + pc[i].len;
                if(a < end) { const uint64_t x = a - base, n = std::min(b, end) - a; memcpy(dst + pc[i].off + x, pc[i].src + x, n); a += n; }
                base = end;
            }
        });
    }
    // BatchIn whose pointers are `base + section offset` (base = device address, or the host buffer for the emulation)
    BatchIn make_batch_in(uint8_t *base, const Section sec[S_COUNT]) const
    {
        BatchIn b; memset(&b, 0, sizeof(b));
        b.n_graphs = n();
#define ALD_P(T, i) ((ALD_GLOBAL const T*)(base + sec[i].off))
        b.g_nv = ALD_P(int32_t, S_NV); b.g_ne = ALD_P(int32_t, S_NE); b.g_np = ALD_P(int32_t, S_NP);
        b.off_v = ALD_P(int64_t, S_OFFV); b.off_e = ALD_P(int64_t, S_OFFE); b.off_s = ALD_P(int64_t, S_OFFS); b.off_p = ALD_P(int64_t, S_OFFP); b.off_pv = ALD_P(int64_t, S_OFFPV);
        b.vertex_offset = ALD_P(int32_t, S_VOFF); b.edge_target = ALD_P(int32_t, S_ETGT); b.edge_weight = ALD_P(double, S_EW); b.edge_strand = ALD_P(uint8_t, S_ESTRAND); b.edge_abd = ALD_P(double, S_EABD);
        b.edge_sample_offset = ALD_P(int32_t, S_ESOFF); b.sample_id = ALD_P(int32_t, S_SID); b.sample_abd = ALD_P(double, S_SABD);
        b.vertex_weight = ALD_P(double, S_VW); b.vertex_lpos = ALD_P(int32_t, S_LPOS); b.vertex_rpos = ALD_P(int32_t, S_RPOS); b.vertex_type = ALD_P(int32_t, S_VTYPE);
        b.in_offset = ALD_P(int32_t, S_INOFF); b.in_edge = ALD_P(int32_t, S_INEDGE);
        b.phasing_offset = ALD_P(int32_t, S_POFF); b.phasing_vertex = ALD_P(int32_t, S_PV); b.phasing_count = ALD_P(int32_t, S_PC); b.graph_strand = ALD_P(char, S_GSTRAND); b.edge_count = ALD_P(int32_t, S_ECOUNT);
        b.edge_rank = has_rank ? ALD_P(int32_t, S_ERANK) : nullptr;
        b.g_rawdist = has_raw ? ALD_P(int32_t, S_RAWDIST) : nullptr; b.off_rp = ALD_P(int64_t, S_OFFRP); b.off_rc = ALD_P(int64_t, S_OFFRC);
        b.rphase_offset = ALD_P(int32_t, S_RPOFF); b.rphase_coord = ALD_P(int32_t, S_RPCOORD); b.rphase_count = ALD_P(int32_t, S_RPCOUNT);
#undef ALD_P
        return b;
    }
    // algorithmic input bytes per SURVEY.md 8d: out-CSR offsets + targets + weights + strand + vertex weight + lpos/rpos + samples + phasing
    int64_t algorithmic_in_bytes() const
    {
        int64_t V = off_v.back(), E = off_e.back(), S = off_s.back(), P = off_p.back(), PV = off_pv.back(), N = n();
        return 4 * (V + N) + 4 * E + 8 * E + E + 8 * V + 8 * V + 12 * S + 4 * PV + 4 * P;
    }
};

// ---- results ----
// One decoded path record (a VIEW, made on demand from the record words: HostResults::path).  vert_off: word offset of its vertex list
// in the record pool; the exon words of the transcript it becomes follow the vertices (decomp_common.h: record layout);
// coverage = log(1 + weight) (essential.cc:725), taken with the host's libm.
struct PathRec { int32_t graph, index, nv, length, count, nexw; char strand; int attempt; double weight, abd, conf, reads, coverage; uint64_t vert_off; };
struct HostResults {
    std::vector<int32_t> status, n_iters, attempt;       // per graph (attempt = pass that produced the final answer)
    std::vector<int64_t> path_begin;                     // [n+1] into rec_off / coverage (paths sorted by graph, index)
    rvec<uint64_t> rec_off;                              // [paths] pool offset of the record of every path
    rvec<double> coverage;                               // [paths] log(1 + weight)
    rvec<uint32_t> chain_key;                            // [paths] bucket of the transcript in a transcript_set (transcript.cc:183-201): the decode has the exons in hand
    rvec<uint16_t> n_exon_words;                         // [paths] min(#exon words, 65535): who is a single-exon transcript without touching the record again
    std::vector<uint32_t> pool;                          // raw record words (vertex and exon lists are read in place) ...
    const uint32_t *ext_pool = nullptr; uint64_t ext_words = 0;   // ... or a borrowed buffer (the batch's pinned D2H landing area)
    int64_t out_bytes = 0;                               // algorithmic output bytes: sum(4*len + 40)
    void clear() { status.clear(); n_iters.clear(); attempt.clear(); path_begin.clear(); rec_off.clear(); coverage.clear(); chain_key.clear(); n_exon_words.clear(); pool.clear(); ext_pool = nullptr; ext_words = 0; out_bytes = 0; }
    const uint32_t *pool_data() const { return ext_pool ? ext_pool : pool.data(); }
    uint64_t pool_size() const { return ext_pool ? ext_words : (uint64_t)pool.size(); }
    int64_t n_paths() const { return (int64_t)rec_off.size(); }
    const uint32_t *rec(int64_t i) const { return pool_data() + rec_off[(size_t)i]; }
    const uint32_t *vertices(const PathRec &p) const { return pool_data() + p.vert_off; }
    const int32_t *exons(const PathRec &p) const { return (const int32_t*)(pool_data() + p.vert_off + p.nv); }
    PathRec path(int64_t i) const
    {
        const uint32_t *r = rec(i); PathRec p;
        p.graph = (int32_t)r[0]; p.index = (int32_t)r[1]; p.nv = (int32_t)r[2]; p.length = (int32_t)r[3]; p.count = (int32_t)r[4]; p.nexw = (int32_t)r[REC_NEXW];
        p.strand = (char)(r[5] & 0xFF); p.attempt = (int)((r[5] >> 8) & 0xFF);
        memcpy(&p.weight, r + 6, 8); memcpy(&p.abd, r + 8, 8); memcpy(&p.conf, r + 10, 8); memcpy(&p.reads, r + 12, 8);
        p.coverage = coverage[(size_t)i]; p.vert_off = rec_off[(size_t)i] + REC_HDR_WORDS;
        return p;
    }

    // The path table of a batch from the index the KERNEL wrote: index[graph_first[g] + p] = pool offset of record (g, p), published
    // only by graphs that ended well (records of abandoned attempts are never looked at).  No walk over the pool, no counting pass,
    // no copy of the records' fields: per path one offset (checked against the record it names) and the coverage.
    int build(int n, const int32_t *n_paths_dev, const unsigned long long *index, uint64_t index_n, const long long *graph_first)
    {
        rec_off.clear(); coverage.clear(); chain_key.clear(); n_exon_words.clear(); out_bytes = 0;
        const uint64_t W = pool_size(); const uint32_t *pw = pool_data();
        path_begin.assign((size_t)n + 1, 0); attempt.assign((size_t)n, 0);
        for(int g = 0; g < n; g++) {
            const bool ok = status[g] == ALD_ST_OK || status[g] == ALD_ST_SKIPPED_LARGE;
            const int64_t c = ok ? n_paths_dev[g] : 0;
            if(c < 0) return -1;
            if(c > 0 && (graph_first[g] < 0 || (uint64_t)graph_first[g] + (uint64_t)c > index_n)) return -3;
            path_begin[(size_t)g + 1] = path_begin[(size_t)g] + c;
        }
        const int64_t total = path_begin[(size_t)n];
        rec_off.resize((size_t)total); coverage.resize((size_t)total); chain_key.resize((size_t)total); n_exon_words.resize((size_t)total);
        unsigned nthr = std::thread::hardware_concurrency(); if(nthr == 0) nthr = 1; if(nthr > 16) nthr = 16;
        if(const char *ev = getenv("ALD_STAGE_THREADS")) { int k = atoi(ev); if(k >= 1 && k <= 64) nthr = (unsigned)k; }
        if(total < 50000) nthr = 1;
        std::vector<int> bad(nthr, 0); std::vector<int64_t> ob(nthr, 0);
        HostBatch::run_threads(nthr, [&](unsigned t) {
            const int64_t lo = total * t / nthr, hi = total * (t + 1) / nthr;            // slices of equal path counts
            int g = (int)(std::upper_bound(path_begin.begin(), path_begin.end(), lo) - path_begin.begin()) - 1; if(g < 0) g = 0;
            int64_t obt = 0;
            int g2 = g;                                                               // the record eight paths ahead is asked for now: records lie in the pool
            for(int64_t i = lo; i < hi; i++) {                                        // in the order the waves emitted them, every one is a miss of its own
                while(path_begin[(size_t)g + 1] <= i) g++;
                if(i + 8 < hi) { while(path_begin[(size_t)g2 + 1] <= i + 8) g2++; const uint64_t o2 = index[(uint64_t)graph_first[g2] + (uint64_t)(i + 8 - path_begin[(size_t)g2])]; if(o2 + REC_HDR_WORDS <= W) { __builtin_prefetch(pw + o2); __builtin_prefetch(pw + o2 + 16); } }
                const int32_t idx = (int32_t)(i - path_begin[(size_t)g]);
                const uint64_t o = index[(uint64_t)graph_first[g] + (uint64_t)idx];
                if(o + REC_HDR_WORDS > W) { bad[t] = 1; return; }
                const uint32_t *r = pw + o;
                const uint32_t nv = r[2], nexw = r[REC_NEXW];
                if(nv < 2 || (nexw & 1) || nexw > 2 * nv || o + rec_words(nv, nexw) > W || (int32_t)r[0] != g || (int32_t)r[1] != idx) { bad[t] = 2; return; }
                double w; memcpy(&w, r + 6, 8);
                rec_off[(size_t)i] = o; coverage[(size_t)i] = log(1.0 + w);
                chain_key[(size_t)i] = (uint32_t)aletsch::sink_transcript::chain_key((const int32_t*)(r + REC_HDR_WORDS + nv), (size_t)nexw); n_exon_words[(size_t)i] = (uint16_t)(nexw < 65535u ? nexw : 65535u);
                if(idx == 0) attempt[(size_t)g] = (int)((r[5] >> 8) & 0xFF);
                obt += 4ll * nv + 40;
            }
            ob[t] = obt;
        });
        for(unsigned t = 0; t < nthr; t++) { if(bad[t]) return -2; out_bytes += ob[t]; }
        return 0;
    }
};

static inline int export_results(const HostResults &R, int n, int64_t *total_paths, int64_t *total_path_vertices,
                                 int32_t *status, int32_t *path_offset, double *weight, double *abd, double *conf, double *reads,
                                 int32_t *length, int32_t *count, char *strand, int64_t *pv_offset, int32_t *path_vertices)
{
    int64_t tp = R.n_paths(), tv = 0;
    for(int64_t i = 0; i < tp; i++) tv += (int64_t)R.rec(i)[2];
    if(total_paths) *total_paths = tp;
    if(total_path_vertices) *total_path_vertices = tv;
    if(!status) return ALD_OK;
    int64_t iv = 0;
    for(int g = 0; g < n; g++) { status[g] = R.status[g]; path_offset[g] = (int32_t)R.path_begin[g]; }
    path_offset[n] = (int32_t)R.path_begin[n];
    for(int64_t i = 0; i < tp; i++) {
        const PathRec p = R.path(i);
        weight[i] = p.weight; abd[i] = p.abd; conf[i] = p.conf; reads[i] = p.reads; length[i] = p.length; count[i] = p.count; strand[i] = p.strand;
        pv_offset[i] = iv;
        const uint32_t *v = R.vertices(p);
        for(int k = 0; k < p.nv; k++) path_vertices[iv++] = (int32_t)v[k];
    }
    pv_offset[tp] = iv;
    return ALD_OK;
}

static inline void params_from_abi(const ald_params *p, Params &q)
{
    ald_params d;
    if(!p) { const double r[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00}; for(int i = 0; i < 8; i++) d.max_decompose_error_ratio[i] = r[i]; d.min_guaranteed_edge_weight = 0.01; d.min_transcript_coverage = 2.0; d.max_num_exons = 10000; d.reserved = 0; p = &d; }
    for(int i = 0; i < 8; i++) q.max_ratio[i] = p->max_decompose_error_ratio[i];
    q.min_w = p->min_guaranteed_edge_weight; q.min_cov = p->min_transcript_coverage; q.max_num_exons = p->max_num_exons; q.pad = 0;
}

} // namespace ald
