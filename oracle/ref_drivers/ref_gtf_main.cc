// Driver (ours) over the REFERENCE's GTF / feature writers (gtf/transcript.cc:318-494), compiled against /root/reference sources.
// It fills `transcript` objects from stdin and prints what the reference writes for them:
//   transcript::write(ostream, cov2, count)             -- the *.gtf records (meta/incubator.cc:732,780,811)
//   transcript::write_features(ostream)                  -- the per-sample *.trstFeature.csv rows, default stream state (incubator.cc:781)
//   transcript::write_features(int sample_id = -1)       -- the file form: fixed, precision 2, appended to meta.trstFeature.csv (incubator.cc:813)
// argv[1]: scratch directory the file form may write into.
// stdin:  N, then per transcript
//   seqname source gene_id transcript_id meta_tid gene_type transcript_type strand coverage cov2 conf abd count1 count2 w_cov2 w_count nexons l r ...   ("-" = empty string)
//   41 feature values in the order of transcript::TrstFeatures (transcript.h:60-104)
// stdout: per transcript "@T\n<write>@F\n<write_features(ostream)>@G\n<write_features(-1)>"
#include "transcript.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <fstream>
#include <sstream>
#include <string>
#include <unistd.h>
using namespace std;
static string tok() { string s; cin >> s; return s == "-" ? string() : s; }
int main(int argc, char **argv)
{
	if(argc < 2 || chdir(argv[1]) != 0) return 2;
	int N; if(!(cin >> N)) return 1;
	for(int i = 0; i < N; i++)
	{
		transcript t; t.clear();
		t.seqname = tok(); t.source = tok(); t.gene_id = tok(); t.transcript_id = tok(); t.meta_tid = tok(); t.gene_type = tok(); t.transcript_type = tok();
		double wcov2; int wcount, ne;
		cin >> t.strand >> t.coverage >> t.cov2 >> t.conf >> t.abd >> t.count1 >> t.count2 >> wcov2 >> wcount >> ne;
		for(int k = 0; k < ne; k++) { int l, r; cin >> l >> r; t.add_exon(l, r); }
		transcript::TrstFeatures &f = t.features;
		cin >> f.gr_vertices >> f.gr_edges >> f.gr_reads >> f.gr_subgraph >> f.num_vertices >> f.num_edges >> f.junc_ratio >> f.max_mid_exon_len
		    >> f.start_loss1 >> f.start_loss2 >> f.start_loss3 >> f.end_loss1 >> f.end_loss2 >> f.end_loss3 >> f.start_merged_loss >> f.end_merged_loss
		    >> f.introns >> f.start_introns >> f.end_introns >> f.intron_ratio >> f.start_intron_ratio >> f.end_intron_ratio >> f.uni_junc
		    >> f.seq_min_wt >> f.seq_min_cnt >> f.seq_min_abd >> f.seq_min_ratio >> f.seq_max_wt >> f.seq_max_cnt >> f.seq_max_abd >> f.seq_max_ratio
		    >> f.unbridge_start_coming_count >> f.unbridge_start_coming_ratio >> f.unbridge_end_leaving_count >> f.unbridge_end_leaving_ratio
		    >> f.start_cnt >> f.start_weight >> f.start_abd >> f.end_cnt >> f.end_weight >> f.end_abd;
		if(!cin) return 1;
		stringstream ss, sf;
		t.write(ss, wcov2, wcount);
		t.write_features(sf);
		remove("meta.trstFeature.csv");
		t.write_features(-1);
		ifstream fin("meta.trstFeature.csv"); stringstream sg; sg << fin.rdbuf(); fin.close();
		remove("meta.trstFeature.csv");
		cout << "@T\n" << ss.str() << "@F\n" << sf.str() << "@G\n" << sg.str();
	}
	return 0;
}
