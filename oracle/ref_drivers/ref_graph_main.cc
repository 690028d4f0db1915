// Driver (ours) over the REFERENCE's graph layer (graph/directed_graph.cc, undirected_graph.cc),
// compiled against /root/reference sources.  It replays a script of graph edits and dumps the
// iteration orders the decomposition depends on, so that the oracle's "creation order" restatement
// of the pointer-ordered containers can be pinned.
//
// A global monotonic operator new makes pointer order == creation order (nothing is ever reused),
// which is the canonical order this repo defines.
//
// stdin script:
//   D n            start a directed graph with n vertices        U n   start an undirected graph
//   a s t          add_edge(s,t)  (edge handles are numbered 0,1,2,... in creation order)
//   r k            remove_edge(handle k)
//   m k x y        move_edge(handle k, x, y)   (directed only)
//   c v            clear_vertex(v)
//   q              dump: for every vertex "in:" handles / "out:" handles in iteration order,
//                  "edges:" global order, "topo:" topological_sort (directed), "cc:" components (undirected),
//                  "edge s t:" for all s<t with an edge -> handle returned by edge(s,t) (directed)
#include "directed_graph.h"
#include "undirected_graph.h"
#include <cstdio>
#include <cstdlib>
#include <new>
#include <map>
#include <vector>

static char *arena = NULL; static size_t arena_off = 0; static const size_t ARENA = (size_t)1 << 30;
void *operator new(size_t n) { if(!arena) arena = (char*)malloc(ARENA); n = (n + 15) & ~(size_t)15; if(arena_off + n > ARENA) abort(); void *p = arena + arena_off; arena_off += n; return p; }
void operator delete(void *) noexcept {}
void operator delete(void *, size_t) noexcept {}

int main()
{
	directed_graph *dg = NULL; undirected_graph *ug = NULL;
	std::vector<edge_descriptor> h; std::map<edge_descriptor, int> hid;
	char op[8];
	while(scanf("%7s", op) == 1)
	{
		if(op[0] == 'D') { int n; scanf("%d", &n); dg = new directed_graph(); ug = NULL; h.clear(); hid.clear(); for(int i = 0; i < n; i++) dg->add_vertex(); }
		else if(op[0] == 'U') { int n; scanf("%d", &n); ug = new undirected_graph(); dg = NULL; h.clear(); hid.clear(); for(int i = 0; i < n; i++) ug->add_vertex(); }
		else if(op[0] == 'a') { int s, t; scanf("%d %d", &s, &t); edge_descriptor e = dg ? dg->add_edge(s, t) : ug->add_edge(s, t); hid[e] = (int)h.size(); h.push_back(e); }
		else if(op[0] == 'r') { int k; scanf("%d", &k); if(dg) dg->remove_edge(h[k]); else ug->remove_edge(h[k]); hid.erase(h[k]); h[k] = NULL; }
		else if(op[0] == 'm') { int k, x, y; scanf("%d %d %d", &k, &x, &y); dg->move_edge(h[k], x, y); }
		else if(op[0] == 'c') { int v; scanf("%d", &v); graph_base *g = dg ? (graph_base*)dg : (graph_base*)ug;
			for(size_t i = 0; i < h.size(); i++) if(h[i] && (h[i]->source() == v || h[i]->target() == v)) { hid.erase(h[i]); h[i] = NULL; }
			g->clear_vertex(v); }
		else if(op[0] == 'q')
		{
			graph_base *g = dg ? (graph_base*)dg : (graph_base*)ug;
			int n = g->num_vertices();
			for(int v = 0; v < n; v++)
			{
				if(dg) { printf("in %d:", v); PEEI p = dg->in_edges(v); for(edge_iterator it = p.first; it != p.second; it++) printf(" %d", hid[*it]); printf("\n"); }
				printf("out %d:", v); PEEI p = g->out_edges(v); for(edge_iterator it = p.first; it != p.second; it++) printf(" %d", hid[*it]); printf("\n");
			}
			printf("edges:"); PEEI p = g->edges(); for(edge_iterator it = p.first; it != p.second; it++) printf(" %d", hid[*it]); printf("\n");
			if(dg)
			{
				vector<int> tp = dg->topological_sort();
				printf("topo:"); for(size_t i = 0; i < tp.size(); i++) printf(" %d", tp[i]); printf("\n");
				for(int s = 0; s < n; s++) for(int t = 0; t < n; t++) { if(s == t) continue; PEB e = dg->edge(s, t); if(e.second) printf("edge %d %d: %d\n", s, t, hid[e.first]); }
			}
			else
			{
				vector< set<int> > cc = ug->compute_connected_components();
				for(size_t i = 0; i < cc.size(); i++) { printf("cc:"); for(set<int>::iterator it = cc[i].begin(); it != cc[i].end(); it++) printf(" %d", *it); printf("\n"); }
			}
			printf("end\n");
		}
	}
	return 0;
}
