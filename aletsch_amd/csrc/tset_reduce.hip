// tset_reduce.hip -- transcript-set reduction of a whole batch on the GPU.
//
// What it replaces: the loop of assembler::assemble over the graphs of one region (meta/assembler.cc:1105-1133) -- per graph a local
// transcript_set `ts` filled by ts.add(t, 1, sid) (rnacore/transcript_set.cc:149-154), then tm.add(ts) (:156-175) into the region's
// set -- i.e. bucketing by intron-chain hash (gtf/transcript.cc:183-201), ordering / equality by compare1 (:269-300) and
// trans_item::merge (transcript_set.cc:38-81), for the case the reference runs it in: a set that starts EMPTY and takes the graphs in
// ascending order.  (Merging two finished sets is a separate, cheap step: ald_tset_add_flat below = transcript_set::add(set).)
//
// Why it splits the way it does.  For transcripts with two or more exons compare1 is a plain lexicographic order on (number of
// exons, strand, the compared coordinates), so "equal" is an equivalence relation and everything a merge computes is independent of
// the order of the merges -- counts add, cov2 / conf / abd / count1 and the per-sample copies take maxima, the outer bounds widen --
// except ONE thing: the floating-point sum of the coverages, which the reference forms as  ((s1 + s2) + s3) + ...  over the graphs
// that contribute, s_k itself being the left-to-right sum inside graph k.  So: sort the transcripts by group (stable, they start in
// (graph, path) order), and let one lane per group fold its members in exactly that nesting.  Single-exon transcripts merge by an
// overlap test that is not transitive (transcript.cc:283-291): their result depends on the sequence of comparisons, they are few,
// and they go through the host sink (aletsch_amd/host/transcript_sink.hpp) as before.
//
// Kernels (HBM-bound streaming passes over ~2 M transcripts of ~80 B; the sorts are hipCUB radix sorts):
//   tx_build      1 lane / path     record -> joined exons (essential.cc:735-746) -> bucket hash, group key
//   tx_heads      1 lane / sorted   first member of a group?  (keys equal AND the compared words equal: a key collision splits, never fuses)
//   tx_fold       1 lane / group    coverage in the reference's nesting, count, maxima, bounds
//   tx_skeys / tx_sheads / tx_sfold    the per-sample copies: (group, sample) runs after a second stable sort, maxima per run
#include "ald_internal.h"
#include "../host/transcript_sink.hpp"
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <thread>
#include <memory>
#include <chrono>

namespace {

#define HCHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) return ald_set_err(ALD_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while(0)

enum { TX_BLOCK = 256 };
static const uint64_t TX_HOST = ~0ull;                 // key of a transcript the host merges (fewer than two exons)

// The paths of a batch as the kernels see them: record offsets in (graph, path) order + the record pool.  A record carries the joined
// exons of its transcript behind the vertex list (decomp_common.h), written by the decomposition kernel, so nothing here needs the
// staged graphs or a host-side parse.
struct TxIn { const unsigned long long *roff; const uint32_t *pool; int64_t np; };
__host__ __device__ inline const int32_t *rec_exons(const uint32_t *r) { return (const int32_t*)(r + REC_HDR_WORDS + r[2]); }

// ---- the result index in (graph, path) order: prefix of the per-graph path counts, then one lane per graph copies its entries
__global__ void ix_order(const int32_t *n_paths, const int64_t *pbegin, const long long *graph_first, const unsigned long long *index, int n, unsigned long long *ordoff)
{
    const int g = (int)((int64_t)blockIdx.x * TX_BLOCK + threadIdx.x);
    if(g >= n) return;
    const int c = n_paths[g]; const long long f = graph_first[g]; const int64_t o = pbegin[g];
    if(f < 0) return;
    for(int p = 0; p < c; p++) ordoff[o + p] = index[f + p];
}
__global__ void ix_widen(const int32_t *n_paths, int n, int64_t *len) { const int g = (int)((int64_t)blockIdx.x * TX_BLOCK + threadIdx.x); if(g <= n) len[g] = g < n ? (int64_t)n_paths[g] : 0; }

// transcript::get_intron_chain_hashing (transcript.cc:183-201, util.cc:38-46) over the flat exon words; 64-bit size_t arithmetic as on the host
__host__ __device__ inline uint64_t chain_key_dev(const int32_t *x, int n_words)
{
    uint64_t h = (uint64_t)(n_words - 2);
    for(int k = 1; k + 1 < n_words; k++) h ^= (uint64_t)(int64_t)x[k] + 0x9e3779b9ull + (h << 6) + (h >> 2);
    return (h & 0x7FFFFFFFull) + 1;
}

__global__ void tx_build(TxIn in, int32_t *nwords, uint64_t *key, int32_t *graph_of)
{
    const int64_t p = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(p >= in.np) return;
    const uint32_t *r = in.pool + in.roff[p];
    const int g = (int)r[0], k = (int)r[REC_NEXW]; const int strand = (int)(r[5] & 0xFF);
    const int32_t *ex = rec_exons(r);
    nwords[p] = k; graph_of[p] = g;
    if(k <= 2) { key[p] = TX_HOST; return; }
    const uint64_t bucket = chain_key_dev(ex, k);
    // group = (bucket, exons, strand, the words compare1 looks at: 1, 2 .. k-5, k-2); 32 bits of it ride in the sort key
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
    mix((uint64_t)k); mix((uint64_t)strand); mix((uint64_t)(uint32_t)ex[1]);
    for(int q = 2; q + 5 <= k; q++) mix((uint64_t)(uint32_t)ex[q]);
    mix((uint64_t)(uint32_t)ex[k - 2]);
    uint64_t kk = (bucket << 32) | (h & 0xFFFFFFFFull);
    if(kk == TX_HOST) kk--;
    key[p] = kk;
}

// same group as the previous sorted element?  equal key AND equal (exon count, strand, compared words, bucket)
__device__ inline bool same_group(const TxIn &in, int64_t a, int64_t b)
{
    const uint32_t *ra = in.pool + in.roff[a], *rb = in.pool + in.roff[b];
    const int ka = (int)ra[REC_NEXW], kb = (int)rb[REC_NEXW];
    if(ka != kb) return false;
    if((ra[5] & 0xFF) != (rb[5] & 0xFF)) return false;
    const int32_t *xa = rec_exons(ra), *xb = rec_exons(rb);
    if(xa[1] != xb[1] || xa[ka - 2] != xb[ka - 2]) return false;
    for(int q = 2; q + 5 <= ka; q++) if(xa[q] != xb[q]) return false;
    return chain_key_dev(xa, ka) == chain_key_dev(xb, kb);
}

__global__ void tx_heads(TxIn in, const uint64_t *skey, const int64_t *sidx, int64_t n_dev, int32_t *head)
{
    const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(i >= n_dev) return;
    head[i] = (i == 0 || skey[i] != skey[i - 1] || !same_group(in, sidx[i - 1], sidx[i])) ? 1 : 0;
}

// first: path (in (graph, path) order) whose record the item takes its exons, strand and id from
struct TxGroup { int64_t first; unsigned long long first_off; int32_t count, count1, lo, hi; double coverage, cov2, conf, abd; uint32_t bucket; int32_t nw, graph, path, strand, pad; };

__global__ void tx_fold(TxIn in, const uint64_t *skey, const int64_t *sidx, const int32_t *head, const int32_t *gid, int64_t n_dev, const double *cov, TxGroup *out)
{
    const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(i >= n_dev || !head[i]) return;
    TxGroup G; G.first = sidx[i]; G.first_off = in.roff[sidx[i]]; G.bucket = (uint32_t)(skey[i] >> 32); G.count = 0; G.count1 = 0; G.coverage = 0; G.cov2 = 0; G.conf = 0; G.abd = 0; G.lo = 0; G.hi = 0; G.pad = 0;
    { const uint32_t *r0 = in.pool + G.first_off; G.nw = (int32_t)r0[REC_NEXW]; G.graph = (int32_t)r0[0]; G.path = (int32_t)r0[1]; G.strand = (int32_t)(r0[5] & 0xFF); }
    double inner = 0; int cur_g = -1; bool any = false;
    for(int64_t m = i; m < n_dev && (m == i || !head[m]); m++) {
        const int64_t p = sidx[m]; const uint32_t *r = in.pool + in.roff[p];
        const int g = (int)r[0]; const double c = cov[p];
        double conf, abd; memcpy(&abd, r + 8, 8); memcpy(&conf, r + 10, 8);
        const int32_t *x = rec_exons(r); const int k = (int)r[REC_NEXW];
        if(g != cur_g) {                                 // the previous graph's own sum enters the set: moved in if the set had none, else added
            if(cur_g >= 0) { G.coverage = any ? G.coverage + inner : inner; any = true; }
            inner = c; cur_g = g;
        } else inner += c;                               // trans_item::merge inside the graph's own set (coverage adds for >= 2 exons)
        if(m == i) { G.lo = x[0]; G.hi = x[k - 1]; G.cov2 = c; G.conf = conf; G.abd = abd; G.count1 = (int32_t)r[4]; }
        else {
            if(x[0] < G.lo) G.lo = x[0]; if(x[k - 1] > G.hi) G.hi = x[k - 1];
            if(G.cov2 < c) G.cov2 = c; if(G.conf < conf) G.conf = conf; if(G.abd < abd) G.abd = abd; if(G.count1 < (int32_t)r[4]) G.count1 = (int32_t)r[4];
        }
        G.count++;
    }
    G.coverage = any ? G.coverage + inner : inner;
    out[gid[i] - 1] = G;                                  // gid: inclusive scan of the head flags, 1-based
}

__global__ void tx_skeys(const int64_t *sidx, const int32_t *gid_incl, const int32_t *graph_of, const int32_t *sid, int64_t n_dev, uint64_t *key2)
{
    const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(i >= n_dev) return;
    const int s = sid ? sid[graph_of[sidx[i]]] : -1;
    key2[i] = ((uint64_t)(uint32_t)gid_incl[i] << 32) | (uint64_t)(uint32_t)(s + 1);      // sample ids ascend as the reference's std::map does (-1 first)
}
__global__ void tx_sheads(const uint64_t *skey2, int64_t n_dev, int32_t *head2)
{
    const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(i >= n_dev) return;
    head2[i] = (i == 0 || skey2[i] != skey2[i - 1]) ? 1 : 0;
}
struct TxSample { int32_t gid, sid, count1, pad; double cov2, conf, abd; };
__global__ void tx_sfold(TxIn in, const uint64_t *skey2, const int64_t *spos, const int64_t *sidx, const int32_t *head2, const int32_t *rid, int64_t n_dev, const double *cov, TxSample *out)
{
    const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(i >= n_dev || !head2[i]) return;
    TxSample S; S.gid = (int32_t)(skey2[i] >> 32) - 1; S.sid = (int32_t)(uint32_t)(skey2[i] & 0xFFFFFFFFull) - 1; S.count1 = 0; S.pad = 0; S.cov2 = 0; S.conf = 0; S.abd = 0;
    for(int64_t m = i; m < n_dev && (m == i || !head2[m]); m++) {
        const int64_t p = sidx[spos[m]]; const uint32_t *r = in.pool + in.roff[p];
        double conf, abd; memcpy(&abd, r + 8, 8); memcpy(&conf, r + 10, 8); const double c = cov[p];
        if(m == i) { S.cov2 = c; S.conf = conf; S.abd = abd; S.count1 = (int32_t)r[4]; }
        else { if(S.cov2 < c) S.cov2 = c; if(S.conf < conf) S.conf = conf; if(S.abd < abd) S.abd = abd; if(S.count1 < (int32_t)r[4]) S.count1 = (int32_t)r[4]; }
    }
    out[rid[i] - 1] = S;
}
__global__ void tx_iota(int64_t *v, int64_t n) { const int64_t i = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x; if(i < n) v[i] = i; }

// ---- the finished transcripts of a batch as one stream IN DEVICE MEMORY (ald_batch_device_transcript_stream): what ranks exchange
__global__ void ts_len(TxIn in, int skip_single, int64_t *len)
{
    const int64_t p = (int64_t)blockIdx.x * TX_BLOCK + threadIdx.x;
    if(p > in.np) return;
    if(p == in.np) { len[p] = 0; return; }                // (the exclusive scan over np + 1 entries leaves the total in the last one)
    const int k = (int)in.pool[in.roff[p] + REC_NEXW];
    len[p] = (k <= 2 && skip_single) ? 0 : (int64_t)ALD_TS_HDR + k;
}
// one 16-lane group per transcript: header by the first lanes, exon words copied with consecutive lanes on consecutive words
__global__ void ts_emit(TxIn in, const int64_t *at, const int32_t *sid, uint32_t *out)
{
    const int64_t p = ((int64_t)blockIdx.x * TX_BLOCK + threadIdx.x) / 16; const int l = (int)(threadIdx.x & 15);
    if(p >= in.np) return;
    const int64_t o = at[p];
    if(at[p + 1] == o) return;
    const uint32_t *r = in.pool + in.roff[p]; const int g = (int)r[0], k = (int)r[REC_NEXW];
    uint32_t *w = out + o;
    if(l < ALD_TS_HDR) {
        uint32_t v;
        switch(l) { case 0: v = (uint32_t)g; break; case 1: v = r[1]; break; case 2: v = (uint32_t)(sid ? sid[g] : -1); break; case 3: v = r[5] & 0xFF; break; case 4: v = r[4]; break; case 5: v = (uint32_t)(k / 2); break;
                    case 6: v = r[6]; break; case 7: v = r[7]; break;          /* weight */
                    case 8: v = r[10]; break; case 9: v = r[11]; break;        /* conf   */
                    case 10: v = r[8]; break; default: v = r[9]; break; }      /* abd    */
        w[l] = v;
    }
    const uint32_t *x = (const uint32_t*)rec_exons(r);
    for(int q = l; q < k; q += 16) w[ALD_TS_HDR + q] = x[q];
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + TX_BLOCK - 1) / TX_BLOCK); }

// scratch of a reduction (owned by a batch and kept across calls, or temporary for the stream entry point)
struct RedScratch { DevBuf *red; PinBuf *pin; hipStream_t st; };

} // namespace

// The result index of a downloaded batch in (graph, path) order, in DEVICE memory: d_ordoff[path_begin[g] + p] = pool offset of record
// (g, p).  Everything comes from what the decomposition kernel left in HBM -- the per-graph path counts, graph_first and the index
// entries -- so the transcript stream and the set reduction start without any host-side table.
static int device_path_table(ald_batch *b)
{
    if(b->paths_on_device) return ALD_OK;
    const int n = b->hb.n(); const int64_t np = b->total_paths;
    if(n == 0 || np == 0) { b->paths_on_device = true; return ALD_OK; }
    DevBuf &d_len = b->dts[0], &d_tmp = b->red[14];
    if(b->d_pbegin.ensure(8 * (size_t)n + 8) || b->d_ordoff.ensure(8 * (size_t)np + 8) || d_len.ensure(8 * (size_t)n + 8)) return ald_set_err(ALD_ERR_NOMEM, "device path table");
    hipStream_t st = b->stream;
    hipLaunchKernelGGL(ix_widen, dim3(grid_for(n + 1)), dim3(TX_BLOCK), 0, st, (const int32_t*)b->d_npaths.p, n, (int64_t*)d_len.p);
    size_t scan_bytes = 0;
    HCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const int64_t*)d_len.p, (int64_t*)b->d_pbegin.p, n + 1, st));
    if(d_tmp.ensure(scan_bytes + 256)) return ald_set_err(ALD_ERR_NOMEM, "scan scratch");
    HCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, scan_bytes, (const int64_t*)d_len.p, (int64_t*)b->d_pbegin.p, n + 1, st));
    hipLaunchKernelGGL(ix_order, dim3(grid_for(n)), dim3(TX_BLOCK), 0, st, (const int32_t*)b->d_npaths.p, (const int64_t*)b->d_pbegin.p, (const long long*)b->d_gfirst.p, (const unsigned long long*)b->d_index.p, n, (unsigned long long*)b->d_ordoff.p);
    if(hipGetLastError() != hipSuccess) return ald_set_err(ALD_ERR_HIP, "a path-table kernel failed to launch");
    b->paths_on_device = true;
    return ALD_OK;
}

// the reduced set of one batch, flat, in the reference's iteration order (ascending bucket hash, bucket order inside)
struct ald_tset_flat {      // (rvec: sized once, every element written by the parallel fill -- no zero pass over ~200 MB first)
    rvec<uint64_t> hash; rvec<int32_t> count, count1, count2; rvec<char> strand; rvec<double> coverage, cov2, conf, abd; rvec<int64_t> tid; std::vector<int64_t> exon_offset, sample_offset;
    rvec<int32_t> exon_lr, sample_sid, sample_count1; rvec<double> sample_cov2, sample_conf, sample_abd;
    double device_ms = 0, host_ms = 0; int64_t n_device_groups = 0, n_host_items = 0;
};

namespace {
struct EventPair { hipEvent_t a = nullptr, b = nullptr; ~EventPair() { if(a) hipEventDestroy(a); if(b) hipEventDestroy(b); } };

// The reduction proper.  d_pool / d_roff: record pool and record offsets in (graph, path) order in HBM; h_pool / h_roff: the same on
// the host (pinned landing areas); h_cov: log(1 + weight) per path (host libm: these values are summed, and the sums are compared bit
// for bit with the host sink); sid: sample of every graph or null; label: graph id that goes into the transcript ids (null: the
// graph index itself); h_tid: explicit transcript id per path (null: tid_base + (label << 20 | path index)).
int reduce_core(RedScratch S, const uint32_t *d_pool, const unsigned long long *d_roff, const uint32_t *h_pool, const unsigned long long *h_roff, const double *h_cov, const int64_t *h_tid,
                int64_t np, int n_graphs, const int32_t *sid, const int64_t *label, int64_t tid_base, int32_t skip_single_exon, double single_exon_overlap, ald_tset_flat **out)
{
    const auto T0 = std::chrono::steady_clock::now();
    std::unique_ptr<ald_tset_flat> Fp(new ald_tset_flat()); ald_tset_flat *F = Fp.get();
    const auto T1 = std::chrono::steady_clock::now();
    // what comes back from the device lands in pinned buffers kept across calls (pageable targets would halve the copy rate, and the
    // device scratch is not reallocated call after call either)
    int64_t n_dev = 0; size_t NGd = 0, NSd = 0; std::vector<int64_t> host_paths;
    const TxGroup *groups = nullptr; const TxSample *samples = nullptr; const uint64_t *h_key = nullptr;
    if(np > 0) {
        DevBuf &d_cov = S.red[2], &d_nw = S.red[4], &d_key = S.red[5], &d_key2 = S.red[6], &d_idx = S.red[7], &d_idx2 = S.red[8], &d_graph = S.red[9],
               &d_sid = S.red[10], &d_head = S.red[11], &d_gid = S.red[12], &d_groups = S.red[13], &d_tmp = S.red[14], &d_head2 = S.red[15], &d_rid = S.red[16], &d_samples = S.red[17], &d_pos = S.red[18], &d_pos2 = S.red[19];
        PinBuf &p_key = S.pin[0], &p_groups = S.pin[1], &p_samples = S.pin[2];
        if(p_key.ensure(8 * (size_t)np, true)) return ald_set_err(ALD_ERR_NOMEM, "pinned reduction buffers");
        if(d_cov.ensure(8 * (size_t)np) || d_nw.ensure(4 * (size_t)np) || d_key.ensure(8 * (size_t)np)
           || d_key2.ensure(8 * (size_t)np) || d_idx.ensure(8 * (size_t)np) || d_idx2.ensure(8 * (size_t)np) || d_graph.ensure(4 * (size_t)np) || d_head.ensure(4 * (size_t)np) || d_gid.ensure(4 * (size_t)np)
           || d_head2.ensure(4 * (size_t)np) || d_rid.ensure(4 * (size_t)np) || d_pos.ensure(8 * (size_t)np) || d_pos2.ensure(8 * (size_t)np) || (sid && d_sid.ensure(4 * (size_t)n_graphs + 4))) return ald_set_err(ALD_ERR_NOMEM, "reduction buffers");
        hipStream_t st = S.st;
        HCHK(hipMemcpyAsync(d_cov.p, h_cov, 8 * (size_t)np, hipMemcpyHostToDevice, st));
        if(sid) HCHK(hipMemcpyAsync(d_sid.p, sid, 4 * (size_t)n_graphs, hipMemcpyHostToDevice, st));
        EventPair ev; HCHK(hipEventCreate(&ev.a)); HCHK(hipEventCreate(&ev.b));
        HCHK(hipEventRecord(ev.a, st));
        TxIn in; in.roff = d_roff; in.pool = d_pool; in.np = np;
        hipLaunchKernelGGL(tx_build, dim3(grid_for(np)), dim3(TX_BLOCK), 0, st, in, (int32_t*)d_nw.p, (uint64_t*)d_key.p, (int32_t*)d_graph.p);
        hipLaunchKernelGGL(tx_iota, dim3(grid_for(np)), dim3(TX_BLOCK), 0, st, (int64_t*)d_idx.p, np);
        // stable sort by group key: members of a group stay in (graph, path) order; host-side transcripts (key = ~0) sink to the end
        size_t tmp_bytes = 0;
        HCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t*)d_key.p, (uint64_t*)d_key2.p, (const int64_t*)d_idx.p, (int64_t*)d_idx2.p, (int)np, 0, 64, st));
        size_t scan_bytes = 0;
        HCHK(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (const int32_t*)d_head.p, (int32_t*)d_gid.p, (int)np, st));
        if(d_tmp.ensure(std::max(tmp_bytes, scan_bytes) + 256)) return ald_set_err(ALD_ERR_NOMEM, "sort scratch");
        HCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, (const uint64_t*)d_key.p, (uint64_t*)d_key2.p, (const int64_t*)d_idx.p, (int64_t*)d_idx2.p, (int)np, 0, 64, st));
        // how many went to the device: the sorted keys below TX_HOST
        h_key = (const uint64_t*)p_key.p;
        HCHK(hipMemcpyAsync(p_key.p, d_key2.p, 8 * (size_t)np, hipMemcpyDeviceToHost, st));
        HCHK(hipStreamSynchronize(st));
        n_dev = (int64_t)(std::lower_bound(h_key, h_key + np, TX_HOST) - h_key);
        int32_t n_groups = 0, n_runs = 0;
        if(n_dev > 0) {
            const uint64_t *skey = (const uint64_t*)d_key2.p; const int64_t *sidx = (const int64_t*)d_idx2.p;
            hipLaunchKernelGGL(tx_heads, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, in, skey, sidx, n_dev, (int32_t*)d_head.p);
            HCHK(hipcub::DeviceScan::InclusiveSum(d_tmp.p, scan_bytes, (const int32_t*)d_head.p, (int32_t*)d_gid.p, (int)n_dev, st));
            HCHK(hipMemcpyAsync(&n_groups, (int32_t*)d_gid.p + (n_dev - 1), 4, hipMemcpyDeviceToHost, st));
            HCHK(hipStreamSynchronize(st));
            if(d_groups.ensure(sizeof(TxGroup) * (size_t)n_groups)) return ald_set_err(ALD_ERR_NOMEM, "group records");
            hipLaunchKernelGGL(tx_fold, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, in, skey, sidx, (const int32_t*)d_head.p, (const int32_t*)d_gid.p, n_dev, (const double*)d_cov.p, (TxGroup*)d_groups.p);
            // per-sample copies: second stable sort by (group, sample)
            hipLaunchKernelGGL(tx_skeys, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, sidx, (const int32_t*)d_gid.p, (const int32_t*)d_graph.p, sid ? (const int32_t*)d_sid.p : (const int32_t*)nullptr, n_dev, (uint64_t*)d_key.p);
            hipLaunchKernelGGL(tx_iota, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, (int64_t*)d_pos.p, n_dev);
            HCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, (const uint64_t*)d_key.p, (uint64_t*)d_idx.p /* sorted key2 */, (const int64_t*)d_pos.p, (int64_t*)d_pos2.p, (int)n_dev, 0, 64, st));
            hipLaunchKernelGGL(tx_sheads, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, (const uint64_t*)d_idx.p, n_dev, (int32_t*)d_head2.p);
            HCHK(hipcub::DeviceScan::InclusiveSum(d_tmp.p, scan_bytes, (const int32_t*)d_head2.p, (int32_t*)d_rid.p, (int)n_dev, st));
            HCHK(hipMemcpyAsync(&n_runs, (int32_t*)d_rid.p + (n_dev - 1), 4, hipMemcpyDeviceToHost, st));
            HCHK(hipStreamSynchronize(st));
            if(d_samples.ensure(sizeof(TxSample) * (size_t)n_runs)) return ald_set_err(ALD_ERR_NOMEM, "sample records");
            hipLaunchKernelGGL(tx_sfold, dim3(grid_for(n_dev)), dim3(TX_BLOCK), 0, st, in, (const uint64_t*)d_idx.p, (const int64_t*)d_pos2.p, sidx, (const int32_t*)d_head2.p, (const int32_t*)d_rid.p, n_dev,
                               (const double*)d_cov.p, (TxSample*)d_samples.p);
            if(p_groups.ensure(sizeof(TxGroup) * (size_t)n_groups, true) || p_samples.ensure(sizeof(TxSample) * (size_t)n_runs, true)) return ald_set_err(ALD_ERR_NOMEM, "pinned reduction buffers");
            NGd = (size_t)n_groups; NSd = (size_t)n_runs; groups = (const TxGroup*)p_groups.p; samples = (const TxSample*)p_samples.p;
            HCHK(hipMemcpyAsync(p_groups.p, d_groups.p, sizeof(TxGroup) * (size_t)n_groups, hipMemcpyDeviceToHost, st));
            HCHK(hipMemcpyAsync(p_samples.p, d_samples.p, sizeof(TxSample) * (size_t)n_runs, hipMemcpyDeviceToHost, st));
        }
        // the transcripts left to the host: the tail of the sorted order (key TX_HOST; the stable sort kept them in (graph, path) order)
        host_paths.resize((size_t)(np - n_dev));
        if(np > n_dev) HCHK(hipMemcpyAsync(host_paths.data(), (const int64_t*)d_idx2.p + n_dev, 8 * (size_t)(np - n_dev), hipMemcpyDeviceToHost, st));
        HCHK(hipEventRecord(ev.b, st));
        HCHK(hipStreamSynchronize(st));
        float ms = 0; hipEventElapsedTime(&ms, ev.a, ev.b); F->device_ms = ms;
        if(hipGetLastError() != hipSuccess) return ald_set_err(ALD_ERR_HIP, "a reduction kernel failed to launch");
    }
    const auto T2 = std::chrono::steady_clock::now();
    auto graph_label = [&](int g) -> int64_t { return label ? label[g] : (int64_t)g; };
    // ---- host: transcripts with fewer than two exons through the sink (their overlap rule depends on the order of the comparisons;
    // one without any exon lands in bucket 0, as get_intron_chain_hashing puts it)
    aletsch::transcript_sink single(single_exon_overlap);
    {
        aletsch::sink_transcript x;
        auto fill = [&](int64_t p) {
            const uint32_t *r = h_pool + h_roff[(size_t)p]; const int g = (int)r[0];
            double conf, abd; memcpy(&abd, r + 8, 8); memcpy(&conf, r + 10, 8);
            x.strand = (char)(r[5] & 0xFF); x.coverage = h_cov[(size_t)p]; x.top.cov2 = x.coverage; x.top.conf = conf; x.top.abd = abd; x.top.count1 = (int32_t)r[4]; x.count2 = 1;
            x.tid = h_tid ? h_tid[(size_t)p] : tid_base + ((graph_label(g) << 20) | (int64_t)r[1]);
            const int32_t *ex = rec_exons(r); x.xs.assign(ex, ex + r[REC_NEXW]);
        };
        for(size_t a = 0; a < host_paths.size() && !skip_single_exon; ) {          // one per-graph set per graph that has any (assembler.cc:1105-1133)
            const int g = (int)h_pool[h_roff[(size_t)host_paths[a]]];
            size_t e = a; while(e < host_paths.size() && (int)h_pool[h_roff[(size_t)host_paths[e]]] == g) e++;
            if(e - a == 1) { fill(host_paths[a]); single.add(x, 1, sid ? sid[g] : -1); }   // merging a one-item set is the same as adding the item (transcript_set.cc:149-175)
            else {
                aletsch::transcript_sink ts(single_exon_overlap);
                for(size_t q = a; q < e; q++) { fill(host_paths[q]); ts.add(x, 1, sid ? sid[g] : -1); }
                single.add(ts);
            }
            a = e;
        }
    }
    const auto T3 = std::chrono::steady_clock::now();
    // ---- host: the flat set.  Device groups arrive in ascending (bucket, key) order, i.e. in the reference's iteration order already;
    // what is left is (i) groups that share a bucket (a 31-bit hash collision): compare1 order among themselves, (ii) the few host
    // items: spliced in by hash, ahead of the device groups of the same bucket (fewer exons sort first, transcript.cc:271)
    const size_t NG = NGd;
    std::vector<int64_t> samp_begin(NG + 1, 0);                    // sample records are sorted by (group, sample id): group k owns [samp_begin[k], samp_begin[k + 1])
    for(size_t q = 0; q < NSd; q++) samp_begin[(size_t)samples[q].gid + 1]++;
    for(size_t k = 0; k < NG; k++) samp_begin[k + 1] += samp_begin[k];
    std::vector<int32_t> order(NG); for(size_t k = 0; k < NG; k++) order[k] = (int32_t)k;
    auto group_exons = [&](const TxGroup &G) { return rec_exons(h_pool + G.first_off); };
    auto group_tx = [&](int k, aletsch::sink_transcript &t) {
        const TxGroup &G = groups[(size_t)k]; const int32_t *x = group_exons(G);
        t.strand = (char)G.strand; t.xs.assign(x, x + G.nw); t.xs.front() = G.lo; t.xs.back() = G.hi;
    };
    for(size_t i = 0; i + 1 < NG; ) {
        size_t j = i + 1; while(j < NG && groups[j].bucket == groups[i].bucket) j++;
        if(j - i > 1) std::stable_sort(order.begin() + (long)i, order.begin() + (long)j, [&](int a, int c) { aletsch::sink_transcript ta, tc; group_tx(a, ta); group_tx(c, tc); return ta.order_against(tc, single_exon_overlap) == +1; });
        i = j;
    }
    struct HostItem { uint64_t hash; const aletsch::sink_item *z; };
    std::vector<HostItem> hosts;
    { std::vector<size_t> keys = single.sorted_keys(); for(size_t key : keys) for(auto &z : single.mt.find(key)->second) hosts.push_back(HostItem{(uint64_t)key, &z}); }
    const size_t NH = hosts.size(), NT = NG + NH;
    // final position of every item: device group at sorted rank q goes behind the host items whose hash does not exceed its bucket
    std::vector<int64_t> pos_dev(NG), pos_host(NH);
    { size_t h = 0; for(size_t q = 0; q < NG; q++) { const uint64_t bk = groups[(size_t)order[q]].bucket; while(h < NH && hosts[h].hash <= bk) { pos_host[h] = (int64_t)(q + h); h++; } pos_dev[q] = (int64_t)(q + h); } while(h < NH) { pos_host[h] = (int64_t)(NG + h); h++; } }
    F->n_device_groups = (int64_t)NG; F->n_host_items = (int64_t)NH;
    F->hash.resize(NT); F->count.resize(NT); F->strand.resize(NT); F->coverage.resize(NT); F->cov2.resize(NT); F->conf.resize(NT); F->abd.resize(NT); F->count1.resize(NT); F->count2.resize(NT); F->tid.resize(NT);
    F->exon_offset.assign(NT + 1, 0); F->sample_offset.assign(NT + 1, 0);
    for(size_t q = 0; q < NG; q++) { const int k = order[q]; F->exon_offset[(size_t)pos_dev[q] + 1] = groups[(size_t)k].nw / 2; F->sample_offset[(size_t)pos_dev[q] + 1] = samp_begin[(size_t)k + 1] - samp_begin[(size_t)k]; }
    for(size_t h = 0; h < NH; h++) { F->exon_offset[(size_t)pos_host[h] + 1] = (int64_t)hosts[h].z->trst.n_exons(); F->sample_offset[(size_t)pos_host[h] + 1] = (int64_t)hosts[h].z->samples.size(); }
    for(size_t i = 0; i < NT; i++) { F->exon_offset[i + 1] += F->exon_offset[i]; F->sample_offset[i + 1] += F->sample_offset[i]; }
    F->exon_lr.resize(2 * (size_t)F->exon_offset[NT]); const size_t NS = (size_t)F->sample_offset[NT];
    F->sample_sid.resize(NS); F->sample_count1.resize(NS); F->sample_cov2.resize(NS); F->sample_conf.resize(NS); F->sample_abd.resize(NS);
    const auto T4 = std::chrono::steady_clock::now();
    unsigned fthr = std::thread::hardware_concurrency(); if(fthr == 0) fthr = 1; if(fthr > 16) fthr = 16; if(NG < 50000) fthr = 1;
    HostBatch::run_threads(fthr, [&](unsigned t) {
        for(size_t q = NG * t / fthr; q < NG * (t + 1) / fthr; q++) {
            const int k = order[q]; const TxGroup &G = groups[(size_t)k]; const size_t i = (size_t)pos_dev[q];
            F->hash[i] = G.bucket; F->count[i] = G.count; F->strand[i] = (char)G.strand; F->coverage[i] = G.coverage; F->cov2[i] = G.cov2; F->conf[i] = G.conf; F->abd[i] = G.abd; F->count1[i] = G.count1;
            F->tid[i] = h_tid ? h_tid[(size_t)G.first] : tid_base + ((graph_label(G.graph) << 20) | (int64_t)G.path);
            int32_t *e = &F->exon_lr[2 * (size_t)F->exon_offset[i]];
            memcpy(e, group_exons(G), 4 * (size_t)G.nw); e[0] = G.lo; e[G.nw - 1] = G.hi;
            size_t so = (size_t)F->sample_offset[i]; int c2 = 0;
            for(int64_t sx = samp_begin[(size_t)k]; sx < samp_begin[(size_t)k + 1]; sx++, so++, c2++) {
                const TxSample &Sm = samples[(size_t)sx];
                F->sample_sid[so] = Sm.sid; F->sample_cov2[so] = Sm.cov2; F->sample_conf[so] = Sm.conf; F->sample_abd[so] = Sm.abd; F->sample_count1[so] = Sm.count1;
            }
            F->count2[i] = c2;
        }
    });
    for(size_t h = 0; h < NH; h++) {
        const aletsch::sink_item &z = *hosts[h].z; const aletsch::sink_transcript &r = z.trst; const size_t i = (size_t)pos_host[h];
        F->hash[i] = hosts[h].hash; F->count[i] = z.count; F->strand[i] = r.strand; F->coverage[i] = r.coverage; F->cov2[i] = r.top.cov2; F->conf[i] = r.top.conf; F->abd[i] = r.top.abd;
        F->count1[i] = r.top.count1; F->count2[i] = r.count2; F->tid[i] = r.tid;
        if(!r.xs.empty()) memcpy(&F->exon_lr[2 * (size_t)F->exon_offset[i]], r.xs.data(), 4 * r.xs.size());
        size_t so = (size_t)F->sample_offset[i];
        for(auto &q : z.samples) { F->sample_sid[so] = q.first; F->sample_cov2[so] = q.second.top.cov2; F->sample_conf[so] = q.second.top.conf; F->sample_abd[so] = q.second.top.abd; F->sample_count1[so] = q.second.top.count1; so++; }
    }
    F->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
    if(getenv("ALD_SINK_PROF")) { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        fprintf(stderr, "[reduce] setup %.1f ms, device section %.1f ms (events %.1f), single-exon host %.1f ms, order + offsets %.1f ms, fill %.1f ms\n", ms(T0, T1), ms(T1, T2), F->device_ms, ms(T2, T3), ms(T3, T4), ms(T4, std::chrono::steady_clock::now())); }
    *out = Fp.release();
    return ALD_OK;
}
} // namespace

extern "C" {

int ald_batch_reduce_transcripts(const ald_batch *cb, const int32_t *sid, int64_t tid_base, int32_t skip_single_exon, double single_exon_overlap, ald_tset_flat **out)
{
    if(!cb || !out) return ALD_ERR_INVALID;
    if(!cb->downloaded) return ald_set_err(ALD_ERR_STATE, "ald_batch_reduce_transcripts before ald_batch_download");
    ald_batch *b = const_cast<ald_batch*>(cb);
    HCHK(hipSetDevice(b->device));
    { int rc = device_path_table(b); if(rc != ALD_OK) return rc; }
    const int64_t np = b->total_paths;
    // the host's side: the record pool is in the batch's pinned landing area, the offsets in (graph, path) order and the coverages
    // (log(1 + weight), host libm) are the path table ald_batch_download built from the kernel's index
    const unsigned long long *h_roff = (const unsigned long long*)b->res.rec_off.data(); const double *h_cov = b->res.coverage.data();
    RedScratch S; S.red = b->red; S.pin = b->red_pin; S.st = b->stream;
    return reduce_core(S, (const uint32_t*)b->d_pool.p, (const unsigned long long*)b->d_ordoff.p, b->res.pool_data(), h_roff, h_cov, nullptr, np, b->hb.n(), sid, nullptr, tid_base, skip_single_exon, single_exon_overlap, out);
}

// The same reduction fed with a TRANSCRIPT STREAM (the format of ald_batch_transcript_stream; groups = runs of equal graph id, in
// ascending order) instead of a decomposed batch: the transcripts become records of a scratch pool, go to the device and through
// the very kernels a batch's records go through.  This is how transcripts that did not come out of the decomposition kernel -- the
// reference-generated cases of tests/golden/ref_tset.json, a stream received from another rank -- reach the device stage.
int ald_tset_reduce_stream(int32_t device, const uint32_t *words, int64_t n_words, const double *coverage, const int64_t *tid, int64_t tid_base, int32_t skip_single_exon, double single_exon_overlap, ald_tset_flat **out)
{
    if(!out || n_words < 0 || (n_words > 0 && !words)) return ALD_ERR_INVALID;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ald_set_err(ALD_ERR_NO_DEVICE, "no HIP device visible: the reduction has no CPU fallback");
    if(device < 0 || device >= ndev) return ald_set_err(ALD_ERR_INVALID, "device index out of range");
    HCHK(hipSetDevice(device));
    // records: header + two placeholder vertices + the exon words; groups re-numbered 0 .. G-1 in stream order
    std::vector<uint32_t> pool; std::vector<unsigned long long> roff; std::vector<double> cov; std::vector<int32_t> sid; std::vector<int64_t> label, tids;
    int64_t last = -1, ti = -1;                            // ti: ordinal of the transcript in the stream (index into `coverage`)
    for(int64_t o = 0; o < n_words; ) {
        ti++;
        if(o + ALD_TS_HDR > n_words) return ald_set_err(ALD_ERR_INVALID, "malformed transcript stream");
        const int64_t k = 2 * (int64_t)words[o + 5], len = ALD_TS_HDR + k;
        if((int32_t)words[o + 5] < 0 || o + len > n_words) return ald_set_err(ALD_ERR_INVALID, "malformed transcript stream");
        const int64_t g = (int64_t)words[o];
        if(g < last) return ald_set_err(ALD_ERR_INVALID, "transcript stream not in ascending graph order");
        if(g != last) { label.push_back(g); sid.push_back((int32_t)words[o + 2]); last = g; }
        if(skip_single_exon && k <= 2) { o += len; continue; }
        const size_t at = pool.size(); roff.push_back((unsigned long long)at);
        pool.resize(at + (size_t)rec_words(2, (unsigned)k), 0);
        uint32_t *r = pool.data() + at;
        r[0] = (uint32_t)(label.size() - 1); r[1] = words[o + 1]; r[2] = 2; r[3] = 0; r[4] = words[o + 4]; r[5] = words[o + 3] & 0xFF;
        r[6] = words[o + 6]; r[7] = words[o + 7]; r[8] = words[o + 10]; r[9] = words[o + 11]; r[10] = words[o + 8]; r[11] = words[o + 9]; r[12] = r[13] = 0; r[REC_NEXW] = (uint32_t)k; r[REC_NEXW + 1] = 0;
        memcpy(r + REC_HDR_WORDS + 2, words + o + ALD_TS_HDR, 4 * (size_t)k);
        double w; memcpy(&w, words + o + 6, 8); cov.push_back(coverage ? coverage[ti] : log(1.0 + w)); if(tid) tids.push_back(tid[ti]);
        o += len;
    }
    const int64_t np = (int64_t)roff.size();
    DevBuf red[20], d_pool, d_roff; PinBuf pin[8];
    struct Rel { DevBuf *r, *a, *c; PinBuf *p; ~Rel() { for(int i = 0; i < 20; i++) r[i].release(); a->release(); c->release(); for(int i = 0; i < 8; i++) p[i].release(); } } rel{red, &d_pool, &d_roff, pin};
    hipStream_t st = nullptr; HCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StRel { hipStream_t s; ~StRel() { hipStreamDestroy(s); } } strel{st};
    if(np > 0) {
        if(d_pool.ensure(4 * pool.size() + 64) || d_roff.ensure(8 * (size_t)np + 8)) return ald_set_err(ALD_ERR_NOMEM, "stream reduction buffers");
        HCHK(hipMemcpyAsync(d_pool.p, pool.data(), 4 * pool.size(), hipMemcpyHostToDevice, st));
        HCHK(hipMemcpyAsync(d_roff.p, roff.data(), 8 * (size_t)np, hipMemcpyHostToDevice, st));
        HCHK(hipStreamSynchronize(st));
    }
    RedScratch S; S.red = red; S.pin = pin; S.st = st;
    return reduce_core(S, (const uint32_t*)d_pool.p, (const unsigned long long*)d_roff.p, pool.data(), roff.data(), cov.data(), tid ? tids.data() : nullptr, np, (int)label.size(), sid.empty() ? nullptr : sid.data(), label.data(), tid_base, 0 /* filtered above */, single_exon_overlap, out);
}

// The stream of ald_batch_transcript_stream, word for word, built on the device and left there: a rank hands it to RCCL without the
// 200 MB making a round trip through host memory.  Nothing comes from the host: the record offsets in (graph, path) order are
// derived on the device from the index the decomposition kernel wrote (device_path_table), the exons are in the records; one
// number comes back, the length.
int ald_batch_device_transcript_stream(const ald_batch *cb, const int32_t *sid, int32_t skip_single_exon, void **dev_words, int64_t *n_words)
{
    if(!cb || !dev_words || !n_words) return ALD_ERR_INVALID;
    if(!cb->downloaded) return ald_set_err(ALD_ERR_STATE, "ald_batch_device_transcript_stream before ald_batch_download");
    ald_batch *b = const_cast<ald_batch*>(cb);
    HCHK(hipSetDevice(b->device));
    const int n = b->hb.n(); const int64_t np = b->total_paths;
    *dev_words = nullptr; *n_words = 0;
    if(np == 0) return ALD_OK;
    { int rc = device_path_table(b); if(rc != ALD_OK) return rc; }
    PinBuf &p_tot = b->red_pin[7];
    DevBuf &d_sid = b->red[10], &d_tmp = b->red[14];
    DevBuf &d_len = b->dts[0], &d_at = b->dts[1], &d_out = b->dts[2];
    if(p_tot.ensure(64)) return ald_set_err(ALD_ERR_NOMEM, "pinned counter");
    if(d_len.ensure(8 * (size_t)std::max<int64_t>(np, n) + 8) || d_at.ensure(8 * (size_t)np + 8) || (sid && d_sid.ensure(4 * (size_t)n + 4))) return ald_set_err(ALD_ERR_NOMEM, "transcript stream buffers");
    hipStream_t st = b->stream;
    if(sid) HCHK(hipMemcpyAsync(d_sid.p, sid, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    TxIn in; in.roff = (const unsigned long long*)b->d_ordoff.p; in.pool = (const uint32_t*)b->d_pool.p; in.np = np;
    hipLaunchKernelGGL(ts_len, dim3(grid_for(np + 1)), dim3(TX_BLOCK), 0, st, in, (int)(skip_single_exon != 0), (int64_t*)d_len.p);
    size_t scan_bytes = 0;
    HCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const int64_t*)d_len.p, (int64_t*)d_at.p, (int)(np + 1), st));
    if(d_tmp.ensure(scan_bytes + 256)) return ald_set_err(ALD_ERR_NOMEM, "scan scratch");
    HCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, scan_bytes, (const int64_t*)d_len.p, (int64_t*)d_at.p, (int)(np + 1), st));
    HCHK(hipMemcpyAsync(p_tot.p, (const int64_t*)d_at.p + np, 8, hipMemcpyDeviceToHost, st));
    HCHK(hipStreamSynchronize(st));
    const int64_t total = *(const int64_t*)p_tot.p;
    if(d_out.ensure(4 * (size_t)total + 64)) return ald_set_err(ALD_ERR_NOMEM, "transcript stream");
    hipLaunchKernelGGL(ts_emit, dim3(grid_for(16 * np)), dim3(TX_BLOCK), 0, st, in, (const int64_t*)d_at.p, sid ? (const int32_t*)d_sid.p : (const int32_t*)nullptr, (uint32_t*)d_out.p);
    HCHK(hipStreamSynchronize(st));
    if(hipGetLastError() != hipSuccess) return ald_set_err(ALD_ERR_HIP, "a transcript-stream kernel failed to launch");
    *dev_words = d_out.p; *n_words = total;
    return ALD_OK;
}

int ald_tset_flat_size(const ald_tset_flat *f, int64_t *n_items, int64_t *n_exons, int64_t *n_samples)
{
    if(!f) return ALD_ERR_INVALID;
    if(n_items) *n_items = (int64_t)f->hash.size(); if(n_exons) *n_exons = (int64_t)f->exon_lr.size() / 2; if(n_samples) *n_samples = (int64_t)f->sample_sid.size();
    return ALD_OK;
}
int ald_tset_flat_stats(const ald_tset_flat *f, double *device_ms, double *total_ms, int64_t *device_groups, int64_t *host_items)
{
    if(!f) return ALD_ERR_INVALID;
    if(device_ms) *device_ms = f->device_ms; if(total_ms) *total_ms = f->host_ms; if(device_groups) *device_groups = f->n_device_groups; if(host_items) *host_items = f->n_host_items;
    return ALD_OK;
}
int ald_tset_flat_export(const ald_tset_flat *f, uint64_t *hash, int32_t *count, char *strand, double *coverage, double *cov2, double *conf, double *abd,
                         int32_t *count1, int32_t *count2, int64_t *tid, int64_t *exon_offset, int32_t *exon_lr,
                         int64_t *sample_offset, int32_t *sample_sid, double *sample_cov2, double *sample_conf, double *sample_abd, int32_t *sample_count1)
{
    if(!f || !hash || !count || !strand || !coverage || !cov2 || !conf || !abd || !count1 || !count2 || !tid || !exon_offset || !exon_lr || !sample_offset || !sample_sid || !sample_cov2 || !sample_conf || !sample_abd || !sample_count1) return ALD_ERR_INVALID;
    const size_t n = f->hash.size();
#define ALD_CP(dst, src) if(!(src).empty()) memcpy(dst, (src).data(), sizeof((src)[0]) * (src).size())
    ALD_CP(hash, f->hash); ALD_CP(count, f->count); ALD_CP(strand, f->strand); ALD_CP(coverage, f->coverage); ALD_CP(cov2, f->cov2); ALD_CP(conf, f->conf); ALD_CP(abd, f->abd);
    ALD_CP(count1, f->count1); ALD_CP(count2, f->count2); ALD_CP(tid, f->tid); ALD_CP(exon_offset, f->exon_offset); ALD_CP(exon_lr, f->exon_lr); ALD_CP(sample_offset, f->sample_offset);
    ALD_CP(sample_sid, f->sample_sid); ALD_CP(sample_cov2, f->sample_cov2); ALD_CP(sample_conf, f->sample_conf); ALD_CP(sample_abd, f->sample_abd); ALD_CP(sample_count1, f->sample_count1);
#undef ALD_CP
    (void)n;
    return ALD_OK;
}
int ald_tset_flat_free(ald_tset_flat *f) { delete f; return ALD_OK; }

// transcript_set::add(transcript_set &) (transcript_set.cc:156-175): the reduced set of a batch merged into a persistent one, bucket by bucket
int ald_tset_add_flat(ald_tset *t, const ald_tset_flat *f)
{
    if(!t || !f) return ALD_ERR_INVALID;
    const size_t n = f->hash.size();
    // buckets never interact and bucket h lives in table h % ALD_TSET_SHARDS: thread th zips the buckets of ITS tables (every thread
    // scans the hashes, which are ascending; building the items is the work)
    const unsigned nthr = ald_sink_threads((int64_t)n);
    HostBatch::run_threads(nthr, [&](unsigned th) {
        for(size_t i = 0; i < n; ) {
            size_t j = i + 1; while(j < n && f->hash[j] == f->hash[i]) j++;
            if((f->hash[i] % ALD_TSET_SHARDS) % nthr != th) { i = j; continue; }
            aletsch::transcript_sink::bucket vec;
            for(size_t k = i; k < j; k++) {
                aletsch::sink_item z; aletsch::sink_transcript &r = z.trst;
                r.strand = f->strand[k]; r.coverage = f->coverage[k]; r.top.cov2 = f->cov2[k]; r.top.conf = f->conf[k]; r.top.abd = f->abd[k]; r.top.count1 = f->count1[k]; r.count2 = f->count2[k]; r.tid = f->tid[k];
                r.xs.assign(f->exon_lr.begin() + 2 * f->exon_offset[k], f->exon_lr.begin() + 2 * f->exon_offset[k + 1]);
                z.count = f->count[k];
                for(int64_t s = f->sample_offset[k]; s < f->sample_offset[k + 1]; s++) {
                    aletsch::sink_sample x; x.coverage = r.coverage; x.top.cov2 = f->sample_cov2[(size_t)s]; x.top.conf = f->sample_conf[(size_t)s]; x.top.abd = f->sample_abd[(size_t)s]; x.top.count1 = f->sample_count1[(size_t)s]; x.count2 = r.count2;
                    bool fresh; z.samples.slot(f->sample_sid[(size_t)s], x, fresh);
                }
                vec.push_back(std::move(z));
            }
            t->shard[f->hash[i] % ALD_TSET_SHARDS].add_bucket((size_t)f->hash[i], vec);
            i = j;
        }
    });
    return ALD_OK;
}

} // extern "C"
