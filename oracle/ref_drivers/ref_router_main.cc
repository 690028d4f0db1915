// Driver (ours) over the REFERENCE's router: scallop/router.cc compiled unmodified from /root/reference, with the graph layer
// (graph/*.cc), edge_info.cc, vertex_info.cc, equation.cc, subsetsum.cc and util it links against -- all the reference's own files.
//
// What is NOT the reference's here, and why: router.cc reads the graph through five splice_graph accessors (get_edge_weight,
// get_edge_info, get_editable_edge_info, get_strand_degree, mixed_strand_vertex).  Their home, rnacore/splice_graph.cc, cannot be
// compiled in this image (it includes the autoconf-generated config.h), so this driver defines those five members and the class's
// constructor / destructor (+ two never-called virtual overrides the vtable names) itself: two map lookups, one map lookup returning a reference, a 12-line count of strands and a
// two-line test on it (splice_graph.cc:101-120, 1375-1406).  Everything the router COMPUTES -- build_indices, build_bipartite_graph,
// thread_left/right_isolate, classify_plain_vertex, one_side_connected, compute_balanced_weights_components, thread, thread_leaf,
// thread_turn, the confidence side effect, the clamp of build() -- is the reference's own object code.
//
// A global monotonic operator new makes pointer order == creation order.
//
// stdin, any number of cases:
//   R <n_vertices> <root> <n_edges> <n_routes> <min_guaranteed_edge_weight>
//   <s> <t> <weight> <strand> <count> <n_samples> (<sample id> <abundance>)*      n_edges lines, IN CREATION ORDER (edge index = line number)
//   <e1> <e2> <count>                                                             n_routes lines (edge indices: an in-edge and an out-edge of root)
// stdout per case (lines starting with '@'; the router's own printf output is left in between):
//   @case type <type> degree <degree>
//   @ratio <%.17g>                     (when build() ran: type UNSPLITTABLE_SINGLE or SPLITTABLE_PURE)
//   @pair <e1> <e2> <%.17g>            pe2w in map order
//   @conf <edge> <%.17g>               edge_info.confidence of every edge after build()
//   @end
#include "router.h"
#include "constants.h"
#include <cstdio>
#include <cstdlib>
#include <new>

static char *arena = NULL; static size_t arena_off = 0; static const size_t ARENA = (size_t)1 << 30;
void *operator new(size_t n) { if(!arena) arena = (char*)malloc(ARENA); n = (n + 15) & ~(size_t)15; if(arena_off + n > ARENA) abort(); void *p = arena + arena_off; arena_off += n; return p; }
void operator delete(void *) noexcept {}
void operator delete(void *, size_t) noexcept {}

// ---- the seven members of splice_graph this driver has to supply (see the header comment)
splice_graph::splice_graph() {}
splice_graph::~splice_graph() {}
double splice_graph::get_edge_weight(edge_base *e) const { MED::const_iterator it = ewrt.find(e); if(it == ewrt.end()) abort(); return it->second; }
const edge_info &splice_graph::get_edge_info(edge_base *e) const { MEIF::const_iterator it = einf.find(e); if(it == einf.end()) abort(); return it->second; }
edge_info &splice_graph::get_editable_edge_info(edge_base *e) { MEIF::iterator it = einf.find(e); if(it == einf.end()) abort(); return it->second; }
vector<int> splice_graph::get_strand_degree(int i)
{
	vector<int> vs(6, 0);
	PEEI pei = in_edges(i);
	for(edge_iterator it = pei.first; it != pei.second; it++) vs[get_edge_info(*it).strand]++;
	pei = out_edges(i);
	for(edge_iterator it = pei.first; it != pei.second; it++) vs[get_edge_info(*it).strand + 3]++;
	return vs;
}
bool splice_graph::mixed_strand_vertex(int i) { vector<int> v = get_strand_degree(i); return (v[1] + v[4] >= 1) && (v[2] + v[5] >= 1); }
// two virtuals of the base class that splice_graph overrides: the vtable emitted with the constructor above names them; never called here
int splice_graph::clear() { abort(); }
int splice_graph::draw(const string &, const MIS &, const MES &, double, bool) { abort(); }

int main()
{
	int nv, root, ne, nr; double minw;
	while(scanf(" R %d %d %d %d %lf", &nv, &root, &ne, &nr, &minw) == 5)
	{
		splice_graph *gr = new splice_graph();
		for(int i = 0; i < nv; i++) gr->add_vertex();
		MEI e2i; VE i2e;
		for(int k = 0; k < ne; k++)
		{
			int s, t, strand, count, ns; double w;
			if(scanf("%d %d %lf %d %d %d", &s, &t, &w, &strand, &count, &ns) != 6) return 2;
			edge_descriptor e = gr->add_edge(s, t);
			edge_info ei; ei.strand = strand; ei.count = count; ei.confidence = 0; ei.abd = 0;
			for(int j = 0; j < ns; j++) { int sp; double a; if(scanf("%d %lf", &sp, &a) != 2) return 2; ei.samples.insert(sp); ei.spAbd[sp] = a; ei.abd += a; }
			gr->ewrt.insert(PED(e, w)); gr->einf.insert(PEIF(e, ei));
			e2i.insert(PEI(e, k)); i2e.push_back(e);
		}
		MPII mpi;
		for(int k = 0; k < nr; k++) { int a, b, c; if(scanf("%d %d %d", &a, &b, &c) != 3) return 2; mpi[PI(a, b)] += c; }
		parameters cfg; cfg.min_guaranteed_edge_weight = minw; cfg.verbose = 0;
		router rt(root, *gr, e2i, i2e, mpi, cfg);
		rt.classify();
		printf("@case type %d degree %d\n", rt.type, rt.degree);
		if(rt.type == UNSPLITTABLE_SINGLE || rt.type == SPLITTABLE_PURE)
		{
			rt.build();
			printf("@ratio %.17g\n", rt.ratio);
			for(MPID::iterator it = rt.pe2w.begin(); it != rt.pe2w.end(); it++) printf("@pair %d %d %.17g\n", it->first.first, it->first.second, it->second);
			for(int k = 0; k < ne; k++) printf("@conf %d %.17g\n", k, gr->get_edge_info(i2e[k]).confidence);
		}
		printf("@end\n");
	}
	return 0;
}
