// pre_steps.cpp -- host side of the ABI: what assembler::assemble(gx, px, sid) does to a graph and its phase set BEFORE it builds the
// scallop object (meta/assembler.cc:1075-1086), so that the boundary can sit at assemble() itself:
//
//   gx.extend_strands()                               rnacore/splice_graph.cc:1338-1373
//   group_start_boundaries / group_end_boundaries     rnacore/graph_reviser.cc:916-1066   (+ check_continuous_vertices, essential.cc:436-446)
//   px.project_boundaries(smap, tmap)                 rnacore/phase_set.cc:50-67
//   hyper_set hx(gx, px)                              scallop/hyper_set.cc:17-29 -> build_path_from_exon_coordinates, essential.cc:321-366
//   hx.filter_nodes(gx)                               scallop/hyper_set.cc:356-371 -> check_valid_path, essential.cc:448-459
//
// O(V + E + sum of phase lengths) per graph, sequential by nature (the boundary grouping folds weights left to right), and it has to
// run before the graph's wire arrays exist -- so it is host code, on flat arrays: edges in creation order with a per-(source,
// target) "newest parallel edge" lookup standing in for directed_graph::edge (directed_graph.cc:60-76).
#include "ald_internal.h"
#include <map>
#include <utility>

struct ald_staged {
    std::vector<int32_t> vertex_offset, edge_target, edge_sample_offset, sample_id, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, edge_count, edge_rank;
    std::vector<double> edge_weight, edge_abd, sample_abd, vertex_weight; std::vector<uint8_t> edge_strand; char strand = '.';
    std::vector<int32_t> smap, tmap;            // (from, to) pairs of the two boundary maps, ascending by `from` (diagnostic)
    int32_t removed_edges = 0;
};

namespace {

struct RawEdge { int s, t; double w; int strand, count; double abd; int so, sn; int rank; bool alive; };

struct RawGraph {
    int V = 0; std::vector<RawEdge> E;          // creation order
    std::vector<double> vw; std::vector<int32_t> lpos, rpos;
    // directed_graph::edge(s, t): the NEWEST live parallel edge (directed_graph.cc:60-76), or -1
    int edge(int s, int t) const { int best = -1; for(int k = 0; k < (int)E.size(); k++) if(E[k].alive && E[k].s == s && E[k].t == t) best = k; return best; }
    // check_continuous_vertices (essential.cc:436-446)
    bool continuous(int x, int y) const { if(x >= y) return true; for(int i = x; i < y; i++) { if(edge(i, i + 1) < 0) return false; if(rpos[i] != lpos[i + 1]) return false; } return true; }
};

// per-(s,t) lookup without the O(E) scan: newest live edge per pair, rebuilt when an edge dies
struct PairIndex {
    std::map<std::pair<int, int>, std::vector<int>> m;
    explicit PairIndex(const RawGraph &g) { for(int k = 0; k < (int)g.E.size(); k++) m[std::make_pair(g.E[k].s, g.E[k].t)].push_back(k); }
    int edge(const RawGraph &g, int s, int t) const { auto it = m.find(std::make_pair(s, t)); if(it == m.end()) return -1; for(size_t i = it->second.size(); i-- > 0; ) if(g.E[it->second[i]].alive) return it->second[i]; return -1; }
    bool continuous(const RawGraph &g, int x, int y) const { if(x >= y) return true; for(int i = x; i < y; i++) { if(edge(g, i, i + 1) < 0) return false; if(g.rpos[i] != g.lpos[i + 1]) return false; } return true; }
};

} // namespace

extern "C" {

int ald_pre_assemble(const ald_graph_view *g, const ald_phase_view *ph, int32_t max_group_boundary_distance, ald_staged **out)
{
    if(!g || !out || g->num_vertices < 2 || g->num_edges < 0 || !g->vertex_offset || !g->vertex_weight || !g->vertex_lpos || !g->vertex_rpos) return ald_set_err(ALD_ERR_INVALID, "null or negative field");
    const int V = g->num_vertices, NE = g->num_edges, n = V - 1;
    if(NE > 0 && (!g->edge_target || !g->edge_weight || !g->edge_sample_offset)) return ald_set_err(ALD_ERR_INVALID, "null edge arrays");
    if(g->vertex_offset[0] != 0 || g->vertex_offset[V] != NE) return ald_set_err(ALD_ERR_INVALID, "vertex_offset does not span the edges");
    RawGraph G; G.V = V; G.vw.assign(g->vertex_weight, g->vertex_weight + V); G.lpos.assign(g->vertex_lpos, g->vertex_lpos + V); G.rpos.assign(g->vertex_rpos, g->vertex_rpos + V);
    {   // edges in creation order: by rank when the caller gives one, else CSR position
        std::vector<int> src(NE);
        for(int s = 0; s < V; s++) { if(g->vertex_offset[s + 1] < g->vertex_offset[s]) return ald_set_err(ALD_ERR_INVALID, "vertex_offset not monotone"); for(int k = g->vertex_offset[s]; k < g->vertex_offset[s + 1]; k++) src[k] = s; }
        std::vector<int> at(NE);
        if(g->edge_creation_rank) { std::vector<char> seen(NE, 0); for(int k = 0; k < NE; k++) { int r = g->edge_creation_rank[k]; if(r < 0 || r >= NE || seen[r]) return ald_set_err(ALD_ERR_INVALID, "edge_creation_rank is not a permutation of 0..E-1"); seen[r] = 1; at[r] = k; } }
        else for(int k = 0; k < NE; k++) at[k] = k;
        G.E.resize(NE);
        for(int r = 0; r < NE; r++) {
            const int k = at[r]; RawEdge &e = G.E[r];
            e.s = src[k]; e.t = g->edge_target[k];
            if(e.t <= e.s || e.t >= V) return ald_set_err(ALD_ERR_INVALID, "edge target out of range (edges go from a lower to a higher vertex index, below V)");
            e.w = g->edge_weight[k]; e.strand = g->edge_strand ? g->edge_strand[k] : 0;
            e.so = g->edge_sample_offset[k]; e.sn = g->edge_sample_offset[k + 1] - e.so;
            if(e.sn < 0 || e.strand > 2) return ald_set_err(ALD_ERR_INVALID, "bad sample offsets or strand");
            double sum = 0; for(int j = 0; j < e.sn; j++) sum += g->sample_abd[e.so + j];
            e.abd = g->edge_abd ? g->edge_abd[k] : sum; e.count = g->edge_count ? g->edge_count[k] : e.sn; e.rank = r; e.alive = true;
        }
    }
    PairIndex PI(G);
    // ---- splice_graph::extend_strands (splice_graph.cc:1338-1373): a junction s -> s+2 that jumps exactly over vertex s+1 and outweighs
    // it lends its strand to the two edges through s+1, if they have none -- visited in creation order
    for(const RawEdge &e : G.E) {
        const int s = e.s, t = e.t, p1 = G.rpos[s], p2 = G.lpos[t];
        if(p1 >= p2 || s + 2 != t) continue;
        if(e.w <= G.vw[s + 1]) continue;
        if(G.lpos[s + 1] != p1 || G.rpos[s + 1] != p2) continue;
        const int e1 = PI.edge(G, s, s + 1), e2 = PI.edge(G, s + 1, t);
        if(e1 >= 0 && G.E[e1].strand == 0) G.E[e1].strand = e.strand;
        if(e2 >= 0 && G.E[e2].strand == 0) G.E[e2].strand = e.strand;
    }
    ald_staged *S = new ald_staged();
    std::map<int32_t, int32_t> smap, tmap;
    const int32_t dist = max_group_boundary_distance;
    // ---- group_start_boundaries (graph_reviser.cc:916-991): start boundaries that reach the same run of touching vertices within `dist`
    // fold into the leftmost one: its source edge takes their weight and count, every edge j -> j+1 and vertex j on the way as well
    {
        std::vector<int> v; for(const RawEdge &e : G.E) if(e.alive && e.s == 0) v.push_back(e.t);
        std::sort(v.begin(), v.end());
        for(size_t i = 1; i < v.size(); i++) if(v[i] == v[i - 1]) { delete S; return ald_set_err(ALD_ERR_INVALID, "parallel edges out of the source: the reference's boundary grouping is undefined on them"); }
        if(v.size() > 1) {
            int32_t p1 = G.lpos[v[0]], p2 = p1; int k1 = v[0], k2 = k1; int pa = PI.edge(G, 0, v[0]);
            for(size_t i = 1; i < v.size(); i++) {
                const int32_t p = G.lpos[v[i]]; const int pb = PI.edge(G, 0, v[i]);
                const double wb = G.E[pb].w; const int cb = G.E[pb].count;
                bool b = PI.continuous(G, k2, v[i]);
                if(p < p2) { delete S; return ALD_ST_INVARIANT + ALD_INV_OTHER; }                          // assert(p >= p2)
                if(p - p2 > dist) b = false;
                if(!b) { p1 = p; p2 = p; k1 = v[i]; k2 = v[i]; pa = pb; continue; }
                smap.insert(std::make_pair(p, p1));
                for(int j = k1; j < v[i]; j++) {
                    const int pc = PI.edge(G, j, j + 1);
                    if(pc < 0) { delete S; return ALD_ST_INVARIANT + ALD_INV_OTHER; }                      // assert(pc.second == true)
                    G.vw[j] = G.vw[j] + wb; G.E[pc].count += cb; G.E[pc].w = G.E[pc].w + wb;
                }
                G.E[pa].w += wb; G.E[pa].count += cb;
                G.E[pb].alive = false; S->removed_edges++;
                k2 = v[i]; p2 = p;
            }
        }
    }
    // ---- group_end_boundaries (graph_reviser.cc:993-1066): the mirror image, from the right -- with the reference's own asymmetries: the
    // vertex on the way takes (edge weight + wb), not (vertex weight + wb), and no count moves
    {
        std::vector<int> v; for(const RawEdge &e : G.E) if(e.alive && e.t == n) v.push_back(e.s);
        std::sort(v.begin(), v.end(), [](int a, int b) { return a > b; });
        for(size_t i = 1; i < v.size(); i++) if(v[i] == v[i - 1]) { delete S; return ald_set_err(ALD_ERR_INVALID, "parallel edges into the sink: the reference's boundary grouping is undefined on them"); }
        if(v.size() > 1) {
            int32_t p1 = G.rpos[v[0]], p2 = p1; int k1 = v[0], k2 = k1; int pa = PI.edge(G, v[0], n);
            for(size_t i = 1; i < v.size(); i++) {
                const int32_t p = G.rpos[v[i]]; const int pb = PI.edge(G, v[i], n);
                const double wb = G.E[pb].w;
                bool b = PI.continuous(G, v[i], k2);
                if(p > p2) { delete S; return ALD_ST_INVARIANT + ALD_INV_OTHER; }                          // assert(p <= p2)
                if(p2 - p > dist) b = false;
                if(!b) { p1 = p; p2 = p; k1 = v[i]; k2 = v[i]; pa = pb; continue; }
                tmap.insert(std::make_pair(p, p1));
                for(int j = v[i]; j < k1; j++) {
                    const int pc = PI.edge(G, j, j + 1);
                    if(pc < 0) { delete S; return ALD_ST_INVARIANT + ALD_INV_OTHER; }
                    const double wc = G.E[pc].w;
                    G.E[pc].w = wc + wb; G.vw[j + 1] = wc + wb;
                }
                G.E[pa].w += wb;
                G.E[pb].alive = false; S->removed_edges++;
                k2 = v[i]; p2 = p;
            }
        }
    }
    for(auto &x : smap) { S->smap.push_back(x.first); S->smap.push_back(x.second); }
    for(auto &x : tmap) { S->tmap.push_back(x.first); S->tmap.push_back(x.second); }
    // ---- phase_set::project_boundaries (phase_set.cc:50-67): first / last coordinate through the two maps; equal lists merge
    std::map<std::vector<int32_t>, int> pmap;
    if(ph && ph->num_phases > 0) {
        if(!ph->phase_offset || !ph->phase_count || (ph->phase_offset[ph->num_phases] > 0 && !ph->phase_coord)) { delete S; return ald_set_err(ALD_ERR_INVALID, "null phase arrays"); }
        std::map<std::vector<int32_t>, int> raw;                                         // phase_set::pmap itself: a std::map, so the projection walks it in key order
        for(int p = 0; p < ph->num_phases; p++) {
            const int a = ph->phase_offset[p], b = ph->phase_offset[p + 1];
            if(b <= a || ((b - a) & 1)) { delete S; return ald_set_err(ALD_ERR_INVALID, "a phase is a non-empty list of exon coordinate PAIRS"); }     // phase_set::add asserts
            raw[std::vector<int32_t>(ph->phase_coord + a, ph->phase_coord + b)] += ph->phase_count[p];
        }
        for(auto &x : raw) {
            std::vector<int32_t> v = x.first;
            auto is = smap.find(v.front()); auto it = tmap.find(v.back());
            if(is != smap.end()) v[0] = is->second;
            if(it != tmap.end()) v[v.size() - 1] = it->second;
            pmap[v] += x.second;
        }
    }
    // ---- hyper_set(gx, px) (hyper_set.cc:17-29) + filter_nodes (:356-371): coordinates -> vertex lists through lindex / rindex
    // (build_vertex_index, splice_graph.cc:1087-1099: the first vertex with that coordinate wins), each exon a run of touching vertices
    std::map<int32_t, int> lindex, rindex;
    for(int i = 0; i <= n; i++) { if(i != 0) lindex.insert(std::make_pair(G.lpos[i], i)); if(i != n) rindex.insert(std::make_pair(G.rpos[i], i)); }
    std::map<std::vector<int>, int> nodes;
    for(auto &x : pmap) {
        const std::vector<int32_t> &v = x.first; std::vector<int> vv; bool ok = true;
        const int ne = (int)v.size() / 2;
        std::vector<std::pair<int, int>> pp((size_t)ne);
        for(int k = 0; k < ne && ok; k++) {                                              // build_path_from_exon_coordinates (essential.cc:321-366)
            const int32_t p = v[2 * k], q = v[2 * k + 1];
            if(p < 0 || q < 0 || p >= q) { ok = false; break; }
            auto a = lindex.find(p); auto b = rindex.find(q);
            if(a == lindex.end() || b == rindex.end()) { ok = false; break; }
            pp[(size_t)k] = std::make_pair(a->second, b->second);
        }
        for(int k = 0; k < ne && ok; k++) {
            const int a = pp[(size_t)k].first, b = pp[(size_t)k].second;
            if(a > b || !PI.continuous(G, a, b)) { ok = false; break; }
            for(int j = a; j <= b; j++) vv.push_back(j);
        }
        if(!ok) continue;
        for(size_t i = 0; i + 1 < vv.size(); i++) if(!(vv[i] < vv[i + 1])) { delete S; return ALD_ST_INVARIANT + ALD_INV_OTHER; }      // assert(vv[i] < vv[i + 1])
        std::sort(vv.begin(), vv.end());                                                 // add_node_list (the -1 / +1 shifts cancel)
        nodes[vv] += x.second;
    }
    // ---- the staged graph: live edges as CSR by (source, target, creation), ranks compacted
    std::vector<int> order; for(int k = 0; k < NE; k++) if(G.E[k].alive) order.push_back(k);
    std::vector<int> newrank(NE, -1); for(size_t r = 0; r < order.size(); r++) newrank[order[r]] = (int)r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return G.E[a].s != G.E[b].s ? G.E[a].s < G.E[b].s : G.E[a].t < G.E[b].t; });
    S->vertex_offset.assign(V + 1, 0);
    for(int k : order) S->vertex_offset[G.E[k].s + 1]++;
    for(int i = 0; i < V; i++) S->vertex_offset[i + 1] += S->vertex_offset[i];
    S->edge_sample_offset.push_back(0);
    for(int k : order) {
        const RawEdge &e = G.E[k];
        S->edge_target.push_back(e.t); S->edge_weight.push_back(e.w); S->edge_strand.push_back((uint8_t)e.strand); S->edge_abd.push_back(e.abd);
        S->edge_count.push_back(e.count); S->edge_rank.push_back(newrank[k]);
        for(int j = 0; j < e.sn; j++) { S->sample_id.push_back(g->sample_id[e.so + j]); S->sample_abd.push_back(g->sample_abd[e.so + j]); }
        S->edge_sample_offset.push_back((int32_t)S->sample_id.size());
    }
    S->vertex_weight = G.vw; S->vertex_lpos = G.lpos; S->vertex_rpos = G.rpos;
    if(g->vertex_type) S->vertex_type.assign(g->vertex_type, g->vertex_type + V); else S->vertex_type.assign(V, -1);
    S->strand = g->strand ? g->strand : '.';
    // filter_nodes: at least two vertices, every consecutive pair an edge of the (grouped) graph
    S->phasing_offset.push_back(0);
    for(auto &x : nodes) {
        const std::vector<int> &vv = x.first;
        if(vv.size() <= 1) continue;
        bool ok = true;
        for(size_t k = 0; k + 1 < vv.size() && ok; k++) { if(vv[k] < 0 || vv[k] > n || vv[k + 1] < 0 || vv[k + 1] > n || PI.edge(G, vv[k], vv[k + 1]) < 0) ok = false; }
        if(!ok) continue;
        for(int q : vv) S->phasing_vertex.push_back(q);
        S->phasing_offset.push_back((int32_t)S->phasing_vertex.size()); S->phasing_count.push_back(x.second);
    }
    *out = S;
    return ALD_OK;
}

int ald_staged_view(const ald_staged *S, ald_graph_view *g)
{
    if(!S || !g) return ALD_ERR_INVALID;
    static const int32_t zi = 0; static const double zd = 0; static const uint8_t zb = 0;
    memset(g, 0, sizeof(*g));
    g->num_vertices = (int32_t)S->vertex_weight.size(); g->num_edges = (int32_t)S->edge_target.size();
    g->vertex_offset = S->vertex_offset.data(); g->edge_target = S->edge_target.empty() ? &zi : S->edge_target.data(); g->edge_weight = S->edge_weight.empty() ? &zd : S->edge_weight.data();
    g->edge_strand = S->edge_strand.empty() ? &zb : S->edge_strand.data(); g->edge_abd = S->edge_abd.empty() ? &zd : S->edge_abd.data();
    g->edge_sample_offset = S->edge_sample_offset.data(); g->sample_id = S->sample_id.empty() ? &zi : S->sample_id.data(); g->sample_abd = S->sample_abd.empty() ? &zd : S->sample_abd.data();
    g->vertex_weight = S->vertex_weight.data(); g->vertex_lpos = S->vertex_lpos.data(); g->vertex_rpos = S->vertex_rpos.data(); g->vertex_type = S->vertex_type.data();
    g->num_phasing = (int32_t)S->phasing_count.size(); g->phasing_offset = S->phasing_offset.data();
    g->phasing_vertex = S->phasing_vertex.empty() ? &zi : S->phasing_vertex.data(); g->phasing_count = S->phasing_count.empty() ? &zi : S->phasing_count.data();
    g->strand = S->strand; g->edge_count = S->edge_count.empty() ? &zi : S->edge_count.data(); g->edge_creation_rank = S->edge_rank.empty() ? nullptr : S->edge_rank.data();
    return ALD_OK;
}

int ald_staged_boundary_maps(const ald_staged *S, int32_t *n_smap, const int32_t **smap_pairs, int32_t *n_tmap, const int32_t **tmap_pairs)
{
    if(!S) return ALD_ERR_INVALID;
    if(n_smap) *n_smap = (int32_t)(S->smap.size() / 2); if(smap_pairs) *smap_pairs = S->smap.data();
    if(n_tmap) *n_tmap = (int32_t)(S->tmap.size() / 2); if(tmap_pairs) *tmap_pairs = S->tmap.data();
    return ALD_OK;
}

int ald_staged_free(ald_staged *S) { delete S; return ALD_OK; }

// The graph goes into the batch AS IT IS, with its phases in exon coordinates: the pre-steps run on the device, in the wave that loads the
// graph (decomp_device.h: pre_assemble_device).  Where the reference would have asserted in them the graph ends with status
// ALD_ST_INVARIANT + ALD_INV_OTHER, like any other assert on the path.  (ALD_RAW_ON_HOST=1: the host routine above stages the graph
// instead -- a debugging aid; a positive return then means that assert, and nothing is added.)
int ald_batch_add_graph_raw(ald_batch *b, const ald_graph_view *g, const ald_phase_view *phases, int32_t max_group_boundary_distance)
{
    if(!b || !g) return ALD_ERR_INVALID;
    if(getenv("ALD_RAW_ON_HOST")) {
        ald_staged *S = nullptr;
        int rc = ald_pre_assemble(g, phases, max_group_boundary_distance, &S);
        if(rc != ALD_OK) return rc;
        ald_graph_view v; ald_staged_view(S, &v);
        rc = ald_batch_add_graph(b, &v);
        delete S;
        return rc;
    }
    b->uploaded = b->ran = b->downloaded = false;
    int rc;
    try { rc = b->hb.add_graph_raw(*g, phases, max_group_boundary_distance); }
    catch(const std::bad_alloc &) { try { b->hb.clear(); } catch(...) {} return ald_set_err(ALD_ERR_NOMEM, "out of (pinned) host memory while staging: the batch was cleared"); }
    if(rc != ALD_OK) return ald_set_err(rc, b->hb.err);
    return ALD_OK;
}

} // extern "C"
