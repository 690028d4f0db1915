// Driver (ours) over the REFERENCE's subsetsum class, compiled against /root/reference sources.
// stdin:  T instances, each "ns nt" then ns (value label) pairs then nt pairs.
// stdout: per instance "e |S| s... |T| t..." with e printed as %.17g.
// The class's result lives in the public member eqn (scallop/subsetsum.h:36).
#include "subsetsum.h"
#include <cstdio>
int main()
{
	int T;
	if(scanf("%d", &T) != 1) return 1;
	for(int k = 0; k < T; k++)
	{
		int ns, nt;
		if(scanf("%d %d", &ns, &nt) != 2) return 1;
		vector<PI> s, t;
		for(int i = 0; i < ns; i++) { int v, l; if(scanf("%d %d", &v, &l) != 2) return 1; s.push_back(PI(v, l)); }
		for(int i = 0; i < nt; i++) { int v, l; if(scanf("%d %d", &v, &l) != 2) return 1; t.push_back(PI(v, l)); }
		subsetsum sss(s, t);
		sss.solve();
		printf("%.17g %lu", sss.eqn.e, sss.eqn.s.size());
		for(size_t i = 0; i < sss.eqn.s.size(); i++) printf(" %d", sss.eqn.s[i]);
		printf(" %lu", sss.eqn.t.size());
		for(size_t i = 0; i < sss.eqn.t.size(); i++) printf(" %d", sss.eqn.t[i]);
		printf("\n");
	}
	return 0;
}
