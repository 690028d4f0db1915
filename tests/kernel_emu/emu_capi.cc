// emu_capi.cc -- TEST-ONLY driver of the single-lane engine emulation: same staging (host_pack.h), same class
// selection / retry policy and same record parsing as the GPU ABI (aletsch_amd/csrc/ald_abi.cpp), with the
// kernel launch replaced by a direct call.
#ifndef ALD_EMU
#error "emulation build only"
#endif
#include "../../aletsch_amd/csrc/host_pack.h"
#include <cstdlib>

extern "C" {
#define ALD_DECL(ID) void emu_run_class_##ID(const ald::KernelArgs *);
ALD_FOR_EACH_PICK_CLASS(ALD_DECL)
ALD_DECL(13)
#undef ALD_DECL
}
using namespace ald;

struct emu_result { HostResults R; int n = 0; std::vector<int32_t> trace_n, trace_codes; std::vector<double> trace_vals; int trace_cap = 0; std::vector<int32_t> cls; };

extern "C" {

static int emu_run_batch(HostBatch &B, const ald_params *prm, int32_t trace_cap, int32_t force_class, emu_result **out);
int emu_run_packed(int32_t n, const int32_t *nv, const int32_t *ne, const int32_t *np,
                   const int32_t *voff, const int32_t *etgt, const double *ew, const uint8_t *estrand, const double *eabd,
                   const int32_t *esoff, const int32_t *sid, const double *sabd,
                   const double *vw, const int32_t *lpos, const int32_t *rpos, const int32_t *vtype,
                   const int32_t *poff, const int32_t *pv, const int32_t *pc, const char *gstrand, const int32_t *ecount, const int32_t *erank,
                   const ald_params *prm, int32_t trace_cap, int32_t force_class, emu_result **out)
{
    HostBatch B;
    int rc = B.add_packed(n, nv, ne, np, voff, etgt, ew, estrand, eabd, esoff, sid, sabd, vw, lpos, rpos, vtype, poff, pv, pc, gstrand, ecount, erank);
    if(rc != ALD_OK) { fprintf(stderr, "emu: add_packed failed: %s\n", B.err.c_str()); return rc; }
    return emu_run_batch(B, prm, trace_cap, force_class, out);
}
/* a batch built graph by graph, raw graphs (pre-steps in the engine's load phase) included: the staging calls are the product's own */
HostBatch *emu_batch_new() { return new HostBatch(); }
void emu_batch_free(HostBatch *B) { delete B; }
int emu_batch_add_raw(HostBatch *B, const ald_graph_view *g, const ald_phase_view *ph, int32_t dist) { return B->add_graph_raw(*g, ph, dist); }
int emu_batch_add(HostBatch *B, const ald_graph_view *g) { return B->add_graph(*g); }
int emu_batch_run(HostBatch *B, const ald_params *prm, int32_t trace_cap, int32_t force_class, emu_result **out) { return emu_run_batch(*B, prm, trace_cap, force_class, out); }
static int emu_run_batch(HostBatch &B, const ald_params *prm, int32_t trace_cap, int32_t force_class, emu_result **out)
{
    const int n = B.n(); int rc = 0;
    HostBatch::Section sec[HostBatch::S_COUNT];
    uint64_t bytes = B.layout(sec);
    std::vector<uint8_t> buf(bytes);
    B.pack_into(buf.data(), sec);
    KernelArgs A; memset(&A, 0, sizeof(A));
    A.in = B.make_batch_in(buf.data(), sec);
    params_from_abi(prm, A.prm);
    std::vector<int32_t> status(n, 0), n_paths(n, 0), n_iters(n, 0);
    uint64_t pool_cap = 0; for(int g = 0; g < n; g++) pool_cap += 16ull * B.g_ne[g] + 256;
    if(const char *ev = getenv("ALD_DEBUG_POOL_WORDS")) { const long long k = atoll(ev); if(k > 0 && (uint64_t)k < pool_cap) pool_cap = (uint64_t)k; }     // same knob as ald_batch_upload
    std::vector<uint32_t> pool(pool_cap); unsigned long long counters[2] = {0, 0};      // [0] pool words used, [1] index entries used (as in the batch's counter buffer)
    unsigned long long &pool_used = counters[0], &index_used = counters[1];
    uint64_t index_cap = pool_cap / (REC_HDR_WORDS + 2) + 1;
    std::vector<unsigned long long> index(index_cap); std::vector<long long> graph_first(n, -1);
    emu_result *E = new emu_result(); E->n = n; E->trace_cap = trace_cap;
    if(trace_cap > 0) { E->trace_n.assign(n, 0); E->trace_codes.assign(3ull * n * trace_cap, 0); E->trace_vals.assign((size_t)n * trace_cap, 0); }
    A.out.status = status.data(); A.out.n_paths = n_paths.data(); A.out.n_iters = n_iters.data();
    A.out.pool_used = &pool_used; A.out.pool = pool.data(); A.out.pool_cap = pool_cap;
    A.out.index_used = &index_used; A.out.index = index.data(); A.out.index_cap = index_cap; A.out.graph_first = graph_first.data();
    A.out.trace_cap = trace_cap; A.out.trace_n = E->trace_n.data(); A.out.trace_codes = E->trace_codes.data(); A.out.trace_vals = E->trace_vals.data();
    std::vector<int32_t> cls(n), attempt(n, 0);
  for(int regrow = 0; ; regrow++) {          // the host policy of ald_batch_download: a full record pool grows and the batch runs again
    std::vector<int32_t> work[ALD_NUM_CLASSES];
    std::fill(status.begin(), status.end(), 0); std::fill(n_paths.begin(), n_paths.end(), 0); std::fill(attempt.begin(), attempt.end(), 0); pool_used = 0; index_used = 0; std::fill(graph_first.begin(), graph_first.end(), -1);
    if(trace_cap > 0) std::fill(E->trace_n.begin(), E->trace_n.end(), 0);
    for(int g = 0; g < n; g++) {
        int64_t ns = B.off_s[g + 1] - B.off_s[g], npv = B.off_pv[g + 1] - B.off_pv[g];
        if(B.g_rawdist[g] >= 0) npv += 2 * (B.off_rc[g + 1] - B.off_rc[g]);
        cls[g] = debug_underclass(pick_class(B.g_nv[g], B.g_ne[g], ns, npv, force_class));
        if(cls[g] < 0) status[g] = ALD_ST_TOO_LARGE; else work[cls[g]].push_back(g);
    }
    typedef void (*run_fn)(const KernelArgs *);
#define ALD_R(ID) emu_run_class_##ID,
    run_fn runs[ALD_NUM_CLASSES] = { ALD_FOR_EACH_PICK_CLASS(ALD_R) nullptr, nullptr, emu_run_class_13 };      // (the twins 11 / 12 are a placement choice of the GPU host)
#undef ALD_R
    for(int pass = 0; pass < ALD_NUM_CLASSES + 1; pass++) {
        bool any = false;
        for(int c = 0; c < ALD_NUM_CLASSES; c++) {
            if(work[c].empty()) continue;
            any = true;
            ClassInfo ci = class_info(c);
            std::vector<uint8_t> slab(ci.slab_bytes);
            int32_t counter = 0;
            A.work = work[c].data(); A.n_work = (int32_t)work[c].size(); A.attempt = pass; A.counter = &counter; A.slabs = slab.data(); A.slab_stride = ci.slab_bytes;
            runs[c](&A);
        }
        if(!any) break;
        std::vector<int32_t> next[ALD_NUM_CLASSES];
        for(int c = 0; c < ALD_NUM_CLASSES; c++) for(int g : work[c]) {
            attempt[g] = pass;
            const int up = class_retry_up(c);
            if(status[g] == ALD_ST_CAPACITY && up >= 0) { cls[g] = up; next[up].push_back(g); }
        }
        for(int c = 0; c < ALD_NUM_CLASSES; c++) work[c].swap(next[c]);
    }
    bool pool_full = false; for(int g = 0; g < n; g++) if(status[g] == ALD_ST_POOL_FULL) pool_full = true;
    if(!pool_full || regrow >= 8) break;
    pool_cap = std::max<uint64_t>(2 * pool_cap, pool_used + pool_used / 4 + 4096); pool.assign(pool_cap, 0);
    A.out.pool = pool.data(); A.out.pool_cap = pool_cap;
    index_cap = pool_cap / (REC_HDR_WORDS + 2) + 1; index.assign(index_cap, 0); A.out.index = index.data(); A.out.index_cap = index_cap;
  }
    E->R.status = status; E->R.n_iters = n_iters;
    E->R.pool.assign(pool.begin(), pool.begin() + std::min<uint64_t>(pool_used, pool_cap));
    E->cls = cls;
    rc = E->R.build(n, n_paths.data(), index.data(), std::min<uint64_t>(index_used, index_cap), graph_first.data());
    if(rc != 0) { fprintf(stderr, "emu: record parse failed rc=%d\n", rc); delete E; return ALD_ERR_STATE; }
    *out = E;
    return ALD_OK;
}

int emu_result_export(const emu_result *E, int64_t *total_paths, int64_t *total_path_vertices, int32_t *status, int32_t *path_offset,
                      double *weight, double *abd, double *conf, double *reads, int32_t *length, int32_t *count, char *strand,
                      int64_t *pv_offset, int32_t *path_vertices)
{
    return export_results(E->R, E->n, total_paths, total_path_vertices, status, path_offset, weight, abd, conf, reads, length, count, strand, pv_offset, path_vertices);
}
/* the transcripts the records carry (exons joined by the engine, coverage by HostResults::build): same signature as ald_batch_export_transcripts */
int emu_result_export_transcripts(const emu_result *E, int64_t *total_exons, double *coverage, int64_t *exon_offset, int32_t *exon_lr)
{
    int64_t te = 0; const int64_t np = E->R.n_paths();
    for(int64_t i = 0; i < np; i++) {
        const PathRec p = E->R.path((int64_t)((size_t)i));
        if(coverage) { coverage[i] = p.coverage; exon_offset[i] = te; if(p.nexw) memcpy(exon_lr + 2 * te, E->R.exons(p), 4 * (size_t)p.nexw); }
        te += p.nexw / 2;
    }
    if(coverage) exon_offset[np] = te;
    if(total_exons) *total_exons = te;
    return 0;
}
int emu_result_iters(const emu_result *E, int32_t *iters, int32_t *cls) { for(int g = 0; g < E->n; g++) { iters[g] = E->R.n_iters[g]; cls[g] = E->cls[g]; } return 0; }
int emu_result_trace(const emu_result *E, int32_t graph, int32_t *n_events, int32_t *codes3, double *values, int32_t cap)
{
    if(E->trace_cap <= 0) { *n_events = 0; return 0; }
    int n = E->trace_n[graph]; *n_events = n;
    if(!codes3) return 0;
    for(int i = 0; i < n && i < cap && i < E->trace_cap; i++) { size_t o = (size_t)graph * E->trace_cap + i; codes3[3 * i] = E->trace_codes[3 * o]; codes3[3 * i + 1] = E->trace_codes[3 * o + 1]; codes3[3 * i + 2] = E->trace_codes[3 * o + 2]; values[i] = E->trace_vals[o]; }
    return 0;
}
void emu_result_free(emu_result *E) { delete E; }

} // extern "C"
