// Driver (ours) over the REFERENCE's transcript_set (rnacore/transcript_set.cc + gtf/transcript.cc), compiled against
// /root/reference sources.  It replays what assembler::assemble(gx, px, sid) does with the transcripts of each graph
// (meta/assembler.cc:1105-1133): per graph a local set `ts`, ts.add(t, 1, sid, ADD) for every transcript, then tm.add(ts, ADD).
// stdin:  G  then per graph "sid n" and n lines "strand coverage conf abd count1 nexons l r ... tid"
// stdout: every item of tm in map order:  hash count coverage cov2 conf abd count1 count2 tid nexons l r ... | nsamples {sid coverage cov2 conf abd count1 count2}
#include "transcript_set.h"
#include "constants.h"
#include <cstdio>
#include <string>
int main()
{
	int G;
	if(scanf("%d", &G) != 1) return 1;
	transcript_set tm("1", 0, 0.8);
	for(int g = 0; g < G; g++)
	{
		int sid, n;
		if(scanf("%d %d", &sid, &n) != 2) return 1;
		transcript_set ts("1", tm.rid, 0.8);
		for(int i = 0; i < n; i++)
		{
			char st; double cov, conf, abd; int c1, ne; char tid[64];
			if(scanf(" %c %lf %lf %lf %d %d", &st, &cov, &conf, &abd, &c1, &ne) != 6) return 1;
			transcript t;
			t.clear();
			t.seqname = "1"; t.strand = st; t.coverage = cov; t.cov2 = cov; t.conf = conf; t.abd = abd; t.count1 = c1; t.count2 = 1;
			for(int k = 0; k < ne; k++) { int l, r; if(scanf("%d %d", &l, &r) != 2) return 1; t.add_exon(l, r); }
			if(scanf("%63s", tid) != 1) return 1;
			t.transcript_id = tid; t.meta_tid = tid; t.RPKM = 0;
			ts.add(t, 1, sid, TRANSCRIPT_COUNT_ADD_COVERAGE_ADD);
		}
		tm.add(ts, TRANSCRIPT_COUNT_ADD_COVERAGE_ADD);
	}
	for(auto &x : tm.mt) for(auto &z : x.second)
	{
		const transcript &t = z.trst;
		printf("%lu %d %.17g %.17g %.17g %.17g %d %d %s %lu", x.first, z.count, t.coverage, t.cov2, t.conf, t.abd, t.count1, t.count2, t.transcript_id.c_str(), t.exons.size());
		for(auto &e : t.exons) printf(" %d %d", e.first, e.second);
		printf(" | %lu", z.samples.size());
		for(auto &s : z.samples) printf(" { %d %.17g %.17g %.17g %.17g %d %d }", s.first, s.second.coverage, s.second.cov2, s.second.conf, s.second.abd, s.second.count1, s.second.count2);
		printf("\n");
	}
	return 0;
}
