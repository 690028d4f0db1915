/*
 * aletsch_decomp.h -- C ABI of the MI355X splice-graph decomposition path.
 *
 * This is the drop-in boundary for ONE hot path of Shao-Group/aletsch: the
 * per-bundle "Scallop core" (reference meta/assembler.cc:1110-1111,
 *   scallop sx(gx, hx, pa, false); sx.assemble();   -> sx.paths / sx.trsts
 * reference scallop/scallop.h:31-35,50-51).  A maintainer binds these entry
 * points from the reference's C++ host code (INTEGRATION.md shows the 2-line
 * swap through aletsch_amd/host/gpu_scallop.hpp).
 *
 * Plain C: pointers + sizes only, no C++ / torch types.  Every function returns
 * an int status (0 = ALD_OK); nothing throws across this boundary and nothing
 * aborts the process (the reference aborts on assert; here an invariant
 * violation becomes a per-graph status word, see ALD_ST_*).
 *
 * Canonical order: the reference orders edges by raw
 * pointer value; this ABI defines the order as "edge creation sequence".  The
 * caller may hand over the creation sequence of the input edges explicitly
 * (ald_graph_view.edge_creation_rank: the position of each edge in the
 * reference's gr.edges() / get_edge_indices order, graph/graph_base.cc:139-153);
 * without it the creation sequence is the CSR position.
 */
#ifndef ALETSCH_DECOMP_H
#define ALETSCH_DECOMP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared between this push and the pop at the end of the
 * file are exported (tests/test_abi_cpu.py checks both directions against `nm -D`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- library-level status codes ---- */
#define ALD_OK                 0
#define ALD_ERR_INVALID       -1   /* bad argument / malformed graph            */
#define ALD_ERR_NO_DEVICE     -2   /* no HIP device: the product path has NO CPU fallback */
#define ALD_ERR_HIP           -3   /* a HIP runtime call failed (see ald_last_error) */
#define ALD_ERR_STATE         -4   /* call sequence error (e.g. run before upload)    */
#define ALD_ERR_NOMEM         -5

/* ---- per-graph status words (ald_result_view.status) ---- */
#define ALD_ST_OK              0
#define ALD_ST_SKIPPED_LARGE   1   /* the rule loop left through |V| > max_num_exons (scallop.cc:49: at once, or after the graph grew past it); greedy still ran */
#define ALD_ST_CAPACITY        2   /* the graph fit a size class at hand-over, but its working set outgrew the largest class while it ran */
#define ALD_ST_POOL_FULL       3   /* the batch's path-record pool was exhausted while this graph emitted a path; ald_batch_download grows the pool
                                      and decomposes the batch again, so a caller only sees this word if the device cannot hold a larger pool */
#define ALD_ST_TOO_LARGE       4   /* never run: the graph is beyond the largest size class at hand-over (more than 10 240 vertices or 58 752 edges,
                                      or sample supports / phasing lists beyond that class's pools).  The reference itself leaves its rule loop
                                      above max_num_exons = 10 000 vertices */
#define ALD_ST_INVARIANT     100   /* 100+n: the reference would have hit assert class n      */

/* assert classes (status = ALD_ST_INVARIANT + class) */
#define ALD_INV_WEIGHT         1   /* w >= min_guaranteed_edge_weight - SMIN  (scallop.cc:1727,2396,2435) */
#define ALD_INV_MERGE_EQUAL    2   /* |wx-wy| <= SMIN before merge             (scallop.cc:2270) */
#define ALD_INV_COUNT          3   /* edge_info.count > 0 / empty sample intersection (scallop.cc:1915,2300) */
#define ALD_INV_ROUTER         4   /* router precondition (mixed strand, isolated w/o partner, one-side connected) */
#define ALD_INV_DEGREE         5   /* degree(root)==0 after decomposition      (scallop.cc:1973,2139) */
#define ALD_INV_OTHER          9

/* Parameters the path reads (reference util/parameters.cc:85-105). */
typedef struct ald_params {
    double  max_decompose_error_ratio[8]; /* {.30,0,1.10,1.10,.75,.30,0,1.00} */
    double  min_guaranteed_edge_weight;   /* 0.01 */
    double  min_transcript_coverage;      /* 2.0  */
    int32_t max_num_exons;                /* 10000 */
    int32_t reserved;
} ald_params;

/* One splice graph, caller-owned arrays, borrowed for the duration of the call.
 * Vertex 0 is the source, vertex V-1 the sink (reference rnacore/splice_graph.h).
 * Edges are CSR by source vertex; edge id == CSR position == creation order. */
typedef struct ald_graph_view {
    int32_t        num_vertices;        /* V >= 2                                        */
    int32_t        num_edges;           /* E                                             */
    const int32_t *vertex_offset;       /* [V+1] out-CSR                                 */
    const int32_t *edge_target;         /* [E]   target > source (DAG, forward edges)    */
    const double  *edge_weight;         /* [E]   splice_graph::ewrt                      */
    const uint8_t *edge_strand;         /* [E]   edge_info.strand 0/1/2; NULL => all 0   */
    const double  *edge_abd;            /* [E]   edge_info.abd;      NULL => sum(sample_abd) */
    const int32_t *edge_sample_offset;  /* [E+1] per-edge slice into sample_id/sample_abd (edge_info.samples / spAbd)   */
    const int32_t *sample_id;           /* [S]   edge_info.samples, ascending per edge   */
    const double  *sample_abd;          /* [S]   edge_info.spAbd                         */
    const double  *vertex_weight;       /* [V]   splice_graph::vwrt                      */
    const int32_t *vertex_lpos;         /* [V]   vertex_info.lpos                        */
    const int32_t *vertex_rpos;         /* [V]   vertex_info.rpos                        */
    const int32_t *vertex_type;         /* [V]   vertex_info.type; NULL => -1 (EMPTY_VERTEX is -9) */
    int32_t        num_phasing;         /* P: entries of hyper_set::nodes after filter_nodes */
    const int32_t *phasing_offset;      /* [P+1]                                         */
    const int32_t *phasing_vertex;      /* vertex lists, ascending                       */
    const int32_t *phasing_count;       /* [P]                                           */
    char           strand;              /* splice_graph::strand '+','-','.'              */
    const int32_t *edge_count;          /* [E]   edge_info.count; NULL => number of supporting samples.  The callers of assemble() hand over graphs
                                          *       whose counts were ADDED along grouped boundaries (graph_reviser.cc:965-975): count != |samples| there */
    const int32_t *edge_creation_rank;  /* [E]   a permutation of 0..E-1: the scallop edge index e2i of each edge, i.e. its position in the reference's
                                          *       gr.edges() (get_edge_indices walks `se` in pointer order, graph/graph_base.cc:139-153, scallop.cc:24).
                                          *       Ids are behaviour: pe2w / routes are ordered by (id, id) (router.h:23), parallel edges and
                                          *       thread_leaf's scan by id (graph/edge_base.h:35-45, router.cc:861).  NULL => CSR position */
} ald_graph_view;

/* One decomposed s-t path (reference rnacore/path.h:14-35, scallop.cc:2766-2834). */
typedef struct ald_path_view {
    int32_t        num_vertices;   /* includes source 0 and the sink (original index V-1) */
    const int32_t *vertices;       /* ascending original vertex indices                    */
    double         weight;         /* path.weight  */
    double         abd;            /* path.abd     */
    double         conf;           /* path.conf = exp(edge_info.confidence) */
    double         reads;          /* path.reads (med) */
    int32_t        length;         /* path.length (mei) */
    int32_t        count;          /* path.count   */
    char           strand;         /* '+','-','.'  */
} ald_path_view;

typedef struct ald_result_view {
    int32_t status;        /* ALD_ST_*                                */
    int32_t num_paths;     /* in reference order (scallop::paths)     */
    int32_t num_iterations;/* main-loop rule firings (diagnostic)     */
    int32_t reserved;
} ald_result_view;

typedef struct ald_batch ald_batch;   /* opaque */

/* ---- lifecycle (replaces per-graph `scallop sx(...)` construction, scallop.cc:19-32) ---- */
int  ald_default_params(ald_params *p);
int  ald_batch_create(const ald_params *p, int device, ald_batch **out);
int  ald_batch_destroy(ald_batch *b);
int  ald_batch_clear(ald_batch *b);                       /* forget graphs, keep buffers */

/* ---- staging: copy one graph / many packed graphs into the batch's host arrays (pinned memory) ----
 * A malformed graph (edge not from a lower to a higher vertex index below V, offsets that do not span their arrays, strand
 * outside 0..2, negative count, duplicate sample id on an edge) is refused with ALD_ERR_INVALID and a message in ald_last_error();
 * the batch is left exactly as it was -- for the bulk form all or nothing: no graph of a refused call is added. */
int  ald_batch_add_graph(ald_batch *b, const ald_graph_view *g);
/* (A malformed edge_creation_rank -- not a permutation of 0..E-1 -- is refused the same way.)
 * Bulk form for n graphs concatenated: every per-vertex / per-edge / per-sample array is the
 * concatenation over graphs; g_nv[n], g_ne[n], g_np[n] give sizes; vertex_offset, edge_sample_offset
 * and phasing_offset are per-graph LOCAL (each restarts at 0; lengths V+1, E+1, P+1). */
int  ald_batch_add_packed(ald_batch *b, int32_t n,
                          const int32_t *g_nv, const int32_t *g_ne, const int32_t *g_np,
                          const int32_t *vertex_offset, const int32_t *edge_target,
                          const double *edge_weight, const uint8_t *edge_strand, const double *edge_abd,
                          const int32_t *edge_sample_offset, const int32_t *sample_id, const double *sample_abd,
                          const double *vertex_weight, const int32_t *vertex_lpos, const int32_t *vertex_rpos,
                          const int32_t *vertex_type,
                          const int32_t *phasing_offset, const int32_t *phasing_vertex, const int32_t *phasing_count,
                          const char *graph_strand, const int32_t *edge_count /* NULL => per-edge sample count */,
                          const int32_t *edge_creation_rank /* NULL => CSR position; else per graph a permutation of 0..E-1 */);
int  ald_batch_num_graphs(const ald_batch *b);

/* ---- the pre-steps of assembler::assemble(gx, px, sid) (meta/assembler.cc:1075-1086), so that the boundary can sit at assemble() itself ----
 * extend_strands (splice_graph.cc:1338-1373), group_start_boundaries / group_end_boundaries (graph_reviser.cc:916-1066: weights and
 * counts fold along grouped boundaries, the folded boundary edges disappear), phase_set::project_boundaries (phase_set.cc:50-67),
 * hyper_set(gx, px) (hyper_set.cc:17-29 via build_path_from_exon_coordinates, essential.cc:321-366) and filter_nodes (hyper_set.cc:356-371).
 * Host code: O(V + E + phase lengths), sequential by nature, and it runs before the graph's wire arrays exist. */
typedef struct ald_phase_view {            /* phase_set::pmap (rnacore/phase_set.h:21-32) */
    int32_t        num_phases;
    const int32_t *phase_offset;           /* [P+1]                                                          */
    const int32_t *phase_coord;            /* exon coordinates l0 r0 l1 r1 ... per phase (even, non-zero length) */
    const int32_t *phase_count;            /* [P]                                                            */
} ald_phase_view;
typedef struct ald_staged ald_staged;      /* the graph + phasing lists as assemble() hands them to scallop  */
/* g->num_phasing / phasing_* are ignored: the phasing lists come out of `phases`.  Returns ALD_OK, ALD_ERR_INVALID (malformed input,
 * incl. parallel source / sink edges, on which the reference's grouping is undefined), or ALD_ST_INVARIANT + ALD_INV_OTHER (> 0) where
 * the reference would have asserted.  No device is involved. */
int  ald_pre_assemble(const ald_graph_view *g, const ald_phase_view *phases, int32_t max_group_boundary_distance /* parameters.cc:77: 10000 */, ald_staged **out);
int  ald_staged_view(const ald_staged *s, ald_graph_view *out);      /* borrowed pointers into s; edge_creation_rank is set */
int  ald_staged_boundary_maps(const ald_staged *s, int32_t *n_smap, const int32_t **smap_pairs, int32_t *n_tmap, const int32_t **tmap_pairs);   /* (from, to) pairs */
int  ald_staged_free(ald_staged *s);
/* The batched form of `assemble(gx, px, sid)` (meta/assembler.cc:1075): the graph goes into the batch AS RECEIVED, its phase set in exon
 * coordinates, and the pre-steps named above run ON THE DEVICE, in the wave that loads the graph (decomp_device.h: pre_assemble_device),
 * before that wave decomposes it -- no host pass per graph.  g->num_phasing / phasing_* are ignored.  Returns ALD_OK or ALD_ERR_INVALID
 * (malformed input, incl. parallel source / sink edges); where the reference would have asserted in the pre-steps the graph ENDS with
 * status ALD_ST_INVARIANT + ALD_INV_OTHER, like any other assert on the path. */
int  ald_batch_add_graph_raw(ald_batch *b, const ald_graph_view *g, const ald_phase_view *phases, int32_t max_group_boundary_distance);
/* The bulk form: ald_batch_add_packed's arrays + per graph raw_max_group_boundary_distance[i] (>= 0: graph i is raw and brings g_nphase[i]
 * phases -- phase_offset is a local CSR of g_nphase[i] + 1 entries per graph, phase_coord / phase_count concatenated --; < 0: graph i is
 * an ordinary staged graph with its phasing lists).  Replaces the same call, one graph after the other. */
int  ald_batch_add_packed_raw(ald_batch *b, int32_t n, const int32_t *g_nv, const int32_t *g_ne, const int32_t *g_np,
                              const int32_t *vertex_offset, const int32_t *edge_target, const double *edge_weight, const uint8_t *edge_strand, const double *edge_abd,
                              const int32_t *edge_sample_offset, const int32_t *sample_id, const double *sample_abd,
                              const double *vertex_weight, const int32_t *vertex_lpos, const int32_t *vertex_rpos, const int32_t *vertex_type,
                              const int32_t *phasing_offset, const int32_t *phasing_vertex, const int32_t *phasing_count, const char *graph_strand, const int32_t *edge_count,
                              const int32_t *edge_creation_rank,
                              const int32_t *raw_max_group_boundary_distance, const int32_t *g_nphase, const int32_t *phase_offset, const int32_t *phase_coord, const int32_t *phase_count);

/* ---- execution (replaces `sx.assemble()`, scallop.cc:38-188) ---- */
int  ald_batch_upload(ald_batch *b);      /* H2D: the batch's arrays to ONE device buffer (a copy per array, no packing pass) */
int  ald_batch_run(ald_batch *b);         /* launch decomposition kernels on the batch stream */
int  ald_batch_sync(ald_batch *b);        /* wait for the stream                              */
int  ald_batch_download(ald_batch *b);    /* D2H of status, path records (vertices + joined exons) and the kernel-written index; decode */
/* device milliseconds of the last ald_batch_run (hipEvents on the batch stream); <0 if n/a */
double ald_batch_last_kernel_ms(const ald_batch *b);
/* host milliseconds of the stages of the last ald_batch_download (diagnostic; nothing in the reference corresponds): waiting for the
 * kernels, status words + class retries, the D2H copies, decoding the records through the index; bytes moved to the host */
int  ald_batch_last_download_ms(const ald_batch *b, double *wait_kernel_ms, double *status_retries_ms, double *copy_ms, double *decode_ms, int64_t *bytes_to_host);
/* algorithmic bytes: packed input bytes + packed path-record bytes of the last run */
int  ald_batch_algorithmic_bytes(const ald_batch *b, int64_t *in_bytes, int64_t *out_bytes);

/* ---- results (replaces reading sx.paths, scallop.h:50) ---- */
int  ald_batch_get_result(const ald_batch *b, int32_t graph, ald_result_view *out);
int  ald_batch_get_path(const ald_batch *b, int32_t graph, int32_t path, ald_path_view *out);
int  ald_batch_export_iterations(const ald_batch *b, int32_t *num_iterations /* [graphs] */);
/* Bulk export: fills caller arrays. path_offset[n+1] (paths per graph prefix), then per path the
 * scalar fields and pv_offset[num_paths_total+1] into path_vertices. Pass NULL to query sizes. */
int  ald_batch_export(const ald_batch *b, int64_t *total_paths, int64_t *total_path_vertices,
                      int32_t *status, int32_t *path_offset,
                      double *weight, double *abd, double *conf, double *reads,
                      int32_t *length, int32_t *count, char *strand,
                      int64_t *pv_offset, int32_t *path_vertices);

/* ---- transcripts (replaces scallop::build_transcripts -> build_transcript, scallop.cc:3250-3266, rnacore/essential.cc:719-748) ----
 * One transcript per path: exons = the path's internal vertices' [lpos, rpos) intervals with touching intervals joined (the
 * reference's Boost.ICL join_interval_map), coverage = log(1 + path.weight).  The ML feature block of update_trst_features
 * (scallop.cc:3268-3451) comes from ald_batch_features below. */
typedef struct ald_transcript_view {
    int32_t        num_exons;
    const int32_t *exons;          /* 2 * num_exons values: l0, r0, l1, r1, ...; valid until the next call on this thread */
    double         coverage;       /* transcript.coverage = cov2 = log(1 + weight) */
    double         conf, abd;      /* transcript.conf / abd                         */
    int32_t        count1;         /* transcript.count1 = path.count                */
    char           strand;
} ald_transcript_view;
int  ald_batch_get_transcript(const ald_batch *b, int32_t graph, int32_t path, ald_transcript_view *out);
/* bulk: coverage[total_paths], exon_offset[total_paths+1], exon_lr[2*total_exons]; pass NULL arrays to query *total_exons */
int  ald_batch_export_transcripts(const ald_batch *b, int64_t *total_exons, double *coverage, int64_t *exon_offset, int32_t *exon_lr);

/* ---- transcript features (replaces scallop::update_trst_features, scallop.cc:3268-3451, and unique_junc, :3472-3497) ----
 * Host side, read from the ORIGINAL graph as staged (the reference's gr_ori copy, scallop.cc:42,179) and the graph's path set.
 * Fields and order are transcript::TrstFeatures (gtf/transcript.h:60-104). */
typedef struct ald_trst_features {
    int32_t gr_vertices, gr_edges, gr_reads, gr_subgraph;
    int32_t num_vertices, num_edges;
    double  junc_ratio;
    int32_t max_mid_exon_len;
    double  start_loss1, start_loss2, start_loss3, end_loss1, end_loss2, end_loss3, start_merged_loss, end_merged_loss;
    int32_t introns, start_introns, end_introns;
    double  intron_ratio, start_intron_ratio, end_intron_ratio;
    int32_t uni_junc;
    double  seq_min_wt; int32_t seq_min_cnt; double seq_min_abd, seq_min_ratio;
    double  seq_max_wt; int32_t seq_max_cnt; double seq_max_abd, seq_max_ratio;
    int32_t unbridge_start_coming_count; double unbridge_start_coming_ratio;
    int32_t unbridge_end_leaving_count;  double unbridge_end_leaving_ratio;
    int32_t start_cnt; double start_weight, start_abd;
    int32_t end_cnt;   double end_weight, end_abd;
} ald_trst_features;
/* what the features read beyond the decomposition's own inputs: vertex_info's boundary / unbridged-read fields (rnacore/vertex_info.h:35-42)
 * and splice_graph::reads / subgraph; every array is [V] of the graph and may be NULL (= zeros) */
typedef struct ald_graph_extras {
    const double  *boundary_loss1, *boundary_loss2, *boundary_loss3, *boundary_merged_loss;
    const int32_t *unbridge_leaving_count; const double *unbridge_leaving_ratio;
    const int32_t *unbridge_coming_count;  const double *unbridge_coming_ratio;
    int32_t gr_reads, gr_subgraph;
} ald_graph_extras;
/* features[k] for every path k of `graph` (ald_batch_get_result(...).num_paths entries); complete[k] = 1 when every field was set, 0 for a
 * path without a junction (single exon): the reference leaves all but the first seven fields of such a transcript indeterminate (it
 * never writes them: incubator.cc:781,813), here they are zero.  Returns ALD_OK, or ALD_ST_INVARIANT + ALD_INV_OTHER (> 0) where the
 * reference would have hit one of the asserts of update_trst_features (an edge it looks up does not exist in the original graph). */
int  ald_batch_features(const ald_batch *b, int32_t graph, const ald_graph_extras *extras, ald_trst_features *features, int32_t *complete);

/* ---- GTF / feature-table writers (replace transcript::write, gtf/transcript.cc:318-360, and transcript::write_features, :362-494) ----
 * snprintf-style: write at most cap bytes (NUL-terminated when cap > 0), return the number of bytes the full text needs.
 * Byte for byte what the reference's writers put on their streams (pinned: oracle/_ref/ref_gtf, tests/golden/ref_gtf.json). */
int64_t ald_gtf_format_transcript(char *buf, int64_t cap, const char *seqname, const char *source, const char *gene_id, const char *transcript_id,
                                  const char *gene_type /* "" or NULL: omitted */, const char *transcript_type, char strand,
                                  double coverage, double cov2 /* < -0.5: omitted */, int32_t count /* < 0: omitted */,
                                  int32_t n_exons, const int32_t *exon_lr);
/* one row of *.trstFeature.csv; fixed2 = 0: the stream form (default ostream state, 6 significant digits: incubator.cc:781),
 * fixed2 = 1: the file form (fixed, 2 decimals: transcript.cc:430-434, incubator.cc:813) */
int64_t ald_gtf_format_features(char *buf, int64_t cap, int32_t fixed2, const char *transcript_id, const char *meta_tid, const char *seqname,
                                double coverage, double cov2, double abd, double conf, int32_t count1, int32_t count2, int32_t n_exons,
                                const ald_trst_features *f);
/* "chr<chrm>.<gid>.<path index>" (scallop.cc:3258) */
int64_t ald_transcript_id(char *buf, int64_t cap, const char *chrm, const char *gid, int32_t path_index);

/* ---- result sink (replaces transcript_set::add / trans_item::merge, rnacore/transcript_set.cc:38-175) ----
 * Host-side: hash-bucketed merge of transcripts across graphs / samples, as assembler::assemble does with `ts` / `tm`
 * (meta/assembler.cc:1105-1133).  Transcripts of one group (= one graph) first merge among themselves, then into the sink. */
typedef struct ald_tset ald_tset;
int  ald_tset_create(double single_exon_overlap, ald_tset **out);
int  ald_tset_destroy(ald_tset *t);
/* n transcripts, grouped: group_offset[n_groups+1] into the transcript arrays, group_sid[n_groups] = sample id of the group;
 * exon_offset[n+1] into exon_lr (l, r pairs); tid[n] = caller-chosen transcript ids (the first inserted wins a merge) */
int  ald_tset_add(ald_tset *t, int32_t n_groups, const int64_t *group_offset, const int32_t *group_sid,
                  const char *strand, const double *coverage, const double *conf, const double *abd, const int32_t *count1,
                  const int64_t *tid, const int64_t *exon_offset, const int32_t *exon_lr, int32_t skip_single_exon);
/* every graph of a downloaded batch, ascending graph id; sid[graphs] gives each graph's sample id (NULL => -1);
 * tid = tid_base + (graph << 20 | path index) */
int  ald_tset_add_batch(ald_tset *t, const ald_batch *b, const int32_t *sid, int64_t tid_base, int32_t skip_single_exon);
/* Finished transcripts of a downloaded batch as ONE self-contained stream -- what ranks exchange in the multi-GPU gather and
 * what a host driving several devices merges.  4-byte words, transcripts in ascending (graph, path index) order:
 *   [graph, path index, sid, strand, count1, n_exons, weight f64, conf f64, abd f64, (l, r) * n_exons]
 * Only graphs that ended ALD_ST_OK / ALD_ST_SKIPPED_LARGE contribute, records of abandoned capacity attempts are gone, exons are joined,
 * single-exon transcripts are left out when skip_single_exon (assembler.cc:1117).  Valid until the next call on this batch. */
int  ald_batch_transcript_stream(const ald_batch *b, const int32_t *sid, int32_t skip_single_exon, const uint32_t **words, int64_t *n_words);
/* The same stream, word for word, built by kernels and left in DEVICE memory (exon join, lengths, prefix sum, fill): for the RCCL
 * exchange of a multi-process host (ald_comm_gather_streams takes host or device pointers alike; torch.distributed a zero-copy view),
 * so that the finished transcripts travel HBM -> xGMI -> HBM of rank 0 without a detour through this rank's host memory.
 * Valid until the next run / reduction / stream call on this batch. */
int  ald_batch_device_transcript_stream(const ald_batch *b, const int32_t *sid, int32_t skip_single_exon, void **dev_words, int64_t *n_words);
/* Merge such a stream, graph by graph in stream order (assembler.cc:1105-1133): coverage = log(1 + weight) is taken here, on the host;
 * tid = tid_base + ((graph + graph_offset) << 20 | path index), i.e. what ald_tset_add_batch gives the same graph in an unsharded batch */
int  ald_tset_add_stream(ald_tset *t, const uint32_t *words, int64_t n_words, int32_t graph_offset, int64_t tid_base);
/* ---- transcript-set reduction of a whole batch on the GPU (replaces the per-graph loop `ts.add(t, 1, sid)` ... `tm.add(ts)` of
 * meta/assembler.cc:1105-1133 over rnacore/transcript_set.cc:38-175 for one region's graphs) ----
 * The transcripts of a downloaded batch merged exactly as the reference merges them into an EMPTY transcript_set, graph by graph in
 * ascending graph id: transcripts with two or more exons on the device (exon join, intron-chain hash, stable radix sort by group,
 * one lane per group folding its members in the reference's order of floating-point additions), single-exon transcripts -- whose
 * overlap rule depends on the order of the comparisons -- on the host.  The result is FLAT (arrays in the reference's iteration order:
 * ascending bucket hash, bucket order inside), item for item and bit for bit what ald_tset_add_batch into an empty set followed by
 * ald_tset_export gives; ald_tset_add_flat merges it into a persistent set as transcript_set::add(transcript_set&) does. */
typedef struct ald_tset_flat ald_tset_flat;
int  ald_batch_reduce_transcripts(const ald_batch *b, const int32_t *sid /* [graphs] or NULL => -1 */, int64_t tid_base, int32_t skip_single_exon,
                                  double single_exon_overlap, ald_tset_flat **out);
/* The same reduction for transcripts that arrive as a TRANSCRIPT STREAM (format of ald_batch_transcript_stream: groups = runs of
 * equal graph id, ascending) instead of a decomposed batch -- a stream gathered from another rank, or transcripts produced elsewhere:
 * they go to `device` and through the very kernels a batch's records go through.  coverage[i] / tid[i] (optional, one per transcript
 * of the stream in stream order): coverage to merge (NULL: log(1 + weight), essential.cc:725) and transcript id (NULL: tid_base +
 * (graph << 20 | path)).  Replaces the same loop, meta/assembler.cc:1105-1133 over rnacore/transcript_set.cc:38-175. */
int  ald_tset_reduce_stream(int32_t device, const uint32_t *words, int64_t n_words, const double *coverage, const int64_t *tid, int64_t tid_base,
                            int32_t skip_single_exon, double single_exon_overlap, ald_tset_flat **out);
int  ald_tset_flat_size(const ald_tset_flat *f, int64_t *n_items, int64_t *n_exons, int64_t *n_samples);
int  ald_tset_flat_export(const ald_tset_flat *f, uint64_t *hash, int32_t *count, char *strand, double *coverage, double *cov2, double *conf, double *abd,
                          int32_t *count1, int32_t *count2, int64_t *tid, int64_t *exon_offset, int32_t *exon_lr,
                          int64_t *sample_offset, int32_t *sample_sid, double *sample_cov2, double *sample_conf, double *sample_abd, int32_t *sample_count1);
/* device milliseconds (HIP events: kernels, sorts and copies of the reduction), wall milliseconds of the whole call, groups folded on the device, items merged on the host */
int  ald_tset_flat_stats(const ald_tset_flat *f, double *device_ms, double *total_ms, int64_t *device_groups, int64_t *host_items);
int  ald_tset_add_flat(ald_tset *t, const ald_tset_flat *f);
int  ald_tset_flat_free(ald_tset_flat *f);

/* ---- the exchange step for a multi-PROCESS host (one process per MI355X): transcript streams -> rank 0 over RCCL / xGMI ----
 * (A host that drives all its devices from one process -- aletsch::gpu_assembly_queue over a device list -- needs none of this.)
 * RCCL is loaded on first use.  Bootstrap: rank 0 calls ald_comm_unique_id and ships the 128 bytes to the other ranks by its own
 * means; then every rank calls ald_comm_create.  ald_comm_gather_streams is a collective: every rank passes its stream and the
 * global id of its first graph; rank 0 receives all streams back to back in rank order (== ascending global graph id), ready for
 * ald_tset_add_stream(t, all_words + offsets[r], offsets[r + 1] - offsets[r], graph_offsets[r], tid_base). */
typedef struct ald_comm ald_comm;
int  ald_comm_unique_id(uint8_t id[128]);
int  ald_comm_create(const uint8_t id[128], int32_t world, int32_t rank, int32_t device, ald_comm **out);
int  ald_comm_gather_streams(ald_comm *c, const uint32_t *words, int64_t n_words, int32_t graph_offset,
                             const uint32_t **all_words /* rank 0 */, const int64_t **offsets /* [world + 1] */, const int32_t **graph_offsets /* [world] */);
/* The same collective in two halves, with the receive buffers held twice.  _begin exchanges the sizes, ENQUEUES the payloads and, on rank
 * 0, one copy to pinned host memory per received stream (an event behind each) and returns; _wait(upto) blocks until the streams of ranks
 * 0..upto have landed (-1: all; on a rank other than 0: until its own send has left -- its source may be reused then) and hands out the
 * pointers.  Rank 0 so merges rank r's stream (ald_tset_add_stream) while rank r + 1's is still on its way, and may begin gather k + 1
 * before it has merged gather k: what a _wait handed out stays valid until the SECOND next _begin on this communicator. */
int  ald_comm_gather_begin(ald_comm *c, const uint32_t *words, int64_t n_words, int32_t graph_offset);
int  ald_comm_gather_wait(ald_comm *c, int32_t upto, const uint32_t **all_words /* rank 0 */, const int64_t **offsets /* [world + 1] */, const int32_t **graph_offsets /* [world] */);
int  ald_comm_destroy(ald_comm *c);
/* transcript_set::add(transcript_set&) (transcript_set.cc:156-175): every bucket of src zipped into dst; src is left empty */
int  ald_tset_merge(ald_tset *dst, ald_tset *src);
int  ald_tset_size(const ald_tset *t, int64_t *n_items, int64_t *n_exons, int64_t *n_samples);
/* items in the reference's iteration order (hash ascending, then bucket order) */
int  ald_tset_export(const ald_tset *t, uint64_t *hash, int32_t *count, char *strand, double *coverage, double *cov2, double *conf, double *abd,
                     int32_t *count1, int32_t *count2, int64_t *tid, int64_t *exon_offset, int32_t *exon_lr,
                     int64_t *sample_offset, int32_t *sample_sid, double *sample_cov2, double *sample_conf, double *sample_abd, int32_t *sample_count1);

/* the same record stream where the kernels wrote it, in DEVICE memory (valid from ald_batch_download until the next ald_batch_run):
 * lets a multi-GPU caller hand the records to RCCL without a host round trip */
int  ald_batch_device_records(const ald_batch *b, const void **device_words, int64_t *n_words);
/* host helper for the rank that receives several record streams: adds `graph_offset` to the graph word of every record */
int  ald_records_add_graph_offset(uint32_t *words, int64_t n_words, int32_t graph_offset);

/* raw packed path-record pool of the last download: 4-byte words, record = [graph, path index, #vertices, length, count,
 * strand | attempt<<8, weight f64, abd f64, conf f64, reads f64, #exon words, 0, vertices..., exon words (l, r)*..., pad to even]:
 * scallop::paths (scallop.h:50) AND what scallop::build_transcripts makes of them (scallop.cc:3250-3266; exon join of
 * rnacore/essential.cc:735-746 done by the kernel).  The pool also holds the records of abandoned capacity attempts; the records that
 * count are the ones the index names. */
int  ald_batch_raw_records(const ald_batch *b, const uint32_t **words, int64_t *n_words);
/* the result index the decomposition kernel wrote (replaces walking sx.paths / sx.trsts, scallop.h:50-51): index[graph_first[g] + p] =
 * word offset of record (g, p) in the pool above, p < num_paths of g; graph_first[g] = -1 for a graph without paths or one that did
 * not end well.  Host copies of the last download, valid until the next one. */
int  ald_batch_result_index(const ald_batch *b, const uint64_t **index, int64_t *n_entries, const int64_t **graph_first);
/* diagnostics: size class `cls` (0..4): capacities, resident workgroups per CU, grid of the last run, graphs assigned */
int  ald_batch_class_info(ald_batch *b, int32_t cls, int32_t *maxv, int32_t *maxe, int32_t *blocks_per_cu, int32_t *blocks_last_run,
                          int64_t *slab_bytes, int32_t *n_graphs);

/* test hook: the slab address class `cls` was launched with in the last run, and the address of the slab buffer the batch owns now */
int  ald_batch_debug_slab(const ald_batch *b, int32_t cls, const void **launched_with, const void **owned);

/* optional per-graph operation trace (debug builds of the parity tests): rule id, vertex/edge, ratio */
int  ald_batch_enable_trace(ald_batch *b, int32_t max_events_per_graph);
int  ald_batch_get_trace(const ald_batch *b, int32_t graph, int32_t *n_events,
                         const int32_t **codes /* 3 ints per event */, const double **values);

/* ---- subset-sum (reference scallop/subsetsum.cc:20-206; dead in the live path, KAT-pinned) ----
 * n instances; instance i has ns[i] source and nt[i] target integers (values concatenated).
 * Outputs per instance: error e (eqn.e), and the chosen item labels (eqn.s / eqn.t), max 64 each. */
int  ald_subsetsum_batch(int device, int32_t n,
                         const int32_t *ns, const int32_t *nt,
                         const int32_t *src_val, const int32_t *src_lab,
                         const int32_t *tgt_val, const int32_t *tgt_lab,
                         double *err, int32_t *out_ns, int32_t *out_nt,
                         int32_t *out_s /* [n*64] */, int32_t *out_t /* [n*64] */);

/* ---- deterministic synthetic batches (SURVEY 8d generator; used by bench and tests) ---- */
typedef struct ald_synth_spec {
    uint64_t seed;
    int32_t  n_graphs;
    int32_t  v_min, v_max;      /* V ~ U{v_min..v_max}                       */
    int32_t  edges_per_vertex;  /* E = V * edges_per_vertex, unless ...      */
    int32_t  fixed_edges;       /* ... fixed_edges > 0: E = fixed_edges      */
    int32_t  weight_mode;       /* 0 = U[1,100) f64, 1 = integer U{1..100}, 2 = flow-conserving (sum of s-t paths) */
    int32_t  n_samples;         /* samples per edge drawn from {0..n_samples-1}; 1 => {0} */
    int32_t  phasing_per_graph; /* number of random phasing paths (count in 2..10)   */
    int32_t  strand_mode;       /* 0 = all '.', 1 = per-graph single strand on ~70% of edges  */
    int32_t  layout_mode;       /* 0 = exons 1000 apart (every edge a junction), 1 = ~30% of consecutive vertices touch */
} ald_synth_spec;
/* Two-pass: call with all output pointers NULL to get totals, then with buffers. */
int  ald_synth_sizes(const ald_synth_spec *s, int64_t *tot_v, int64_t *tot_e, int64_t *tot_s, int64_t *tot_p, int64_t *tot_pv);
int  ald_synth_fill(const ald_synth_spec *s,
                    int32_t *g_nv, int32_t *g_ne, int32_t *g_np,
                    int32_t *vertex_offset, int32_t *edge_target, double *edge_weight, uint8_t *edge_strand,
                    double *edge_abd, int32_t *edge_sample_offset, int32_t *sample_id, double *sample_abd,
                    double *vertex_weight, int32_t *vertex_lpos, int32_t *vertex_rpos, int32_t *vertex_type,
                    int32_t *phasing_offset, int32_t *phasing_vertex, int32_t *phasing_count, char *graph_strand);

const char *ald_last_error(void);
const char *ald_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ALETSCH_DECOMP_H */
