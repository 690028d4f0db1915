// ald_abi.cpp -- implementation of the C ABI (include/aletsch_decomp.h) on the HIP runtime.
//
// Host side of the drop-in boundary: stages graphs into ONE pinned wire buffer, moves it to HBM with a single
// copy, launches the per-size-class persistent kernels (decomp_class.hip) on the batch's stream, and parses the
// packed path records that come back.  There is NO CPU compute path here: without a HIP device every compute
// entry point fails with ALD_ERR_NO_DEVICE.
#include "ald_internal.h"
#include <thread>
#include <chrono>
#include <iterator>
#include <mutex>
#include <atomic>
#include <map>
#include <unordered_map>
#include <string>
#include <cstdlib>
#include <cstdio>
#include <cmath>

extern "C" {
#define ALD_DECL(ID) int ald_launch_c##ID(const KernelArgs *, int, hipStream_t); int ald_occupancy_c##ID(); unsigned long long ald_hot_slab_bytes_c##ID(); \
                     int ald_launch_raw_c##ID(const KernelArgs *, int, hipStream_t); int ald_occupancy_raw_c##ID(); unsigned long long ald_hot_slab_bytes_raw_c##ID();
ALD_FOR_EACH_CLASS(ALD_DECL)
#undef ALD_DECL
}

/* how many graphs of LDS class 6 (q = 0) / 5 (q = 1) go to free slots of the slab twins: all the free slots for class 6, none of class 5
   (profiles/r04/s_cfg3_twin_spill_sweep.txt: 96.5 -> 94 ms for cfg3; moving class 5 as well, or only the part of class 6 beyond one round, loses) */
#define ALD_TWIN_SPILL_DEFAULT(q, sz, cap) ((q) == 0 ? (int64_t)1 << 40 : (int64_t)0)
namespace {

// A batch owns seven HIP streams and a pipelined caller keeps several batches in flight; the ROCm runtime maps all streams of a process
// onto FOUR hardware queues by default, so the D2H copy of batch k regularly sat in the same queue as the kernel of batch k + 1 and
// waited for it (download 24-38 ms instead of 3 ms per step).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, so the library
// sets it -- unless the caller did -- when it is LOADED: before any HIP call of its own, and before the first HIP call of a program
// that links it.  (A process that initialised HIP before it dlopen()ed the library keeps what it had)
__attribute__((constructor)) void ald_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

thread_local std::string g_err;
int set_err(int code, const std::string &s) { g_err = s; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) return set_err(ALD_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while(0)

typedef int (*launch_fn)(const KernelArgs *, int, hipStream_t);
typedef int (*occ_fn)();
typedef unsigned long long (*hot_fn)();
#define ALD_L(ID) ald_launch_c##ID,
#define ALD_O(ID) ald_occupancy_c##ID,
#define ALD_H(ID) ald_hot_slab_bytes_c##ID,
#define ALD_LR(ID) ald_launch_raw_c##ID,
#define ALD_OR(ID) ald_occupancy_raw_c##ID,
#define ALD_HR(ID) ald_hot_slab_bytes_raw_c##ID,
// indexed by kernel slot: the plain builds of the classes, then their raw builds (ald_internal.h: ALD_NUM_SLOTS)
const launch_fn k_launch[ALD_NUM_SLOTS] = { ALD_FOR_EACH_CLASS(ALD_L) ALD_FOR_EACH_CLASS(ALD_LR) };
const occ_fn k_occ[ALD_NUM_SLOTS] = { ALD_FOR_EACH_CLASS(ALD_O) ALD_FOR_EACH_CLASS(ALD_OR) };
const hot_fn k_hot[ALD_NUM_SLOTS] = { ALD_FOR_EACH_CLASS(ALD_H) ALD_FOR_EACH_CLASS(ALD_HR) };
#undef ALD_L
#undef ALD_O
#undef ALD_H
#undef ALD_LR
#undef ALD_OR
#undef ALD_HR

} // namespace

namespace {

// the decoded path table of a downloaded batch.  ald_batch_download builds it from the index the kernel wrote (no parse of the record
// stream), so there is nothing left to do here but to refuse a batch that has not been downloaded.
int ensure_index(const ald_batch *cb)
{
    if(!cb->downloaded) return set_err(ALD_ERR_STATE, "results requested before ald_batch_download");
    return ALD_OK;
}

} // namespace
int ald_ensure_index(const ald_batch *b) { return ensure_index(b); }
int ald_set_err(int code, const std::string &msg) { return set_err(code, msg); }
namespace {

int occupancy_for(ald_batch *b, int c)
{
    if(b->occ[c] < 0) { int o = k_occ[c](); if(o < 1) o = 1; if(o > 32) o = 32; b->occ[c] = o; }
    return b->occ[c];
}

// launch one pass: every class that has work gets its own persistent grid on the batch stream
// One pass = one persistent grid per size class that has work.  Staging (work lists and kernel arguments in HBM, slabs sized,
// classes dealt to the side streams) is separate from firing, so that the first pass of a batch is staged once, at upload time:
// a run is then three memsets and the launches.
int push_pass(ald_batch *b, const StagedPass &P, hipStream_t s = nullptr)         // work lists + arguments -> HBM (s: the stream the copies go through; default the batch's)
{
    if(P.tot == 0) return ALD_OK;
    if(!s) s = b->stream;
    HIPCHK(hipMemcpyAsync(b->d_work.p, P.flat.data(), 4 * P.tot, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b->d_args.p, P.args.data(), sizeof(KernelArgs) * ALD_NUM_SLOTS, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return ALD_OK;
}

int stage_pass(ald_batch *b, const std::vector<int32_t> work[ALD_NUM_CLASSES], int pass, StagedPass &P)
{
    const int n = b->hb.n();
    P.tot = 0; for(int c = 0; c < ALD_NUM_CLASSES; c++) P.tot += work[c].size();
    P.nord = 0;
    if(P.tot == 0) return ALD_OK;
    if(b->d_work.ensure(4 * (size_t)n + 64)) return set_err(ALD_ERR_NOMEM, "work list");
    if(b->d_counter.ensure(4 * ALD_NUM_SLOTS * 64)) return set_err(ALD_ERR_NOMEM, "counters");
    if(b->d_args.ensure(sizeof(KernelArgs) * ALD_NUM_SLOTS)) return set_err(ALD_ERR_NOMEM, "kernel args");
    size_t woff = 0;
    P.args.assign(ALD_NUM_SLOTS, KernelArgs()); P.flat.resize(P.tot);
    // the work list of a class splits into its staged graphs (plain build of the class) and its raw graphs (raw build: the pre-steps of
    // assembler::assemble run in the loading wave), each in the order it was given
    std::vector<int32_t> part[ALD_NUM_SLOTS];
    for(int c = 0; c < ALD_NUM_CLASSES; c++) {
        if(!b->hb.has_raw) { part[c] = work[c]; continue; }
        for(int32_t g : work[c]) part[b->hb.g_rawdist[(size_t)g] >= 0 ? ALD_NUM_CLASSES + c : c].push_back(g);
    }
    for(int k = 0; k < ALD_NUM_SLOTS; k++) {
        P.nblk[k] = 0;
        if(part[k].empty()) continue;
        const int c = k % ALD_NUM_CLASSES;
        ClassInfo ci = class_info(c);
        int per_cu = occupancy_for(b, k);
        if(const char *ov = getenv("ALD_WG_PER_CU")) { int q = atoi(ov); if(q >= 1 && q < per_cu) per_cu = q; }     // tuning knob: cap the persistent grid
        int want = b->n_cus * per_cu;
        const uint64_t stride = ci.slab_bytes + k_hot[k]();           // catch-all class: hot state + cold state per wave
        if(c == ALD_CATCH_ALL_CLASS && want > b->n_cus) want = b->n_cus;      // ~13 MB per wave: one wave per CU is plenty
        if(c == ALD_HUGE_CLASS && want > ALD_HUGE_CLASS_WAVES) want = ALD_HUGE_CLASS_WAVES;      // ~105 MB per wave
        if((size_t)want > part[k].size()) want = (int)part[k].size();
        if(want < 1) want = 1;
        if(b->d_slabs[k].ensure((size_t)want * stride)) return set_err(ALD_ERR_NOMEM, "class slab");
        P.nblk[k] = want;
        memcpy(P.flat.data() + woff, part[k].data(), 4 * part[k].size());
        KernelArgs &A = P.args[k]; memset(&A, 0, sizeof(A));
        A.in = b->hb.make_batch_in((uint8_t*)b->d_in.p, b->sec);
        A.out.status = (int32_t*)b->d_status.p; A.out.n_paths = (int32_t*)b->d_npaths.p; A.out.n_iters = (int32_t*)b->d_niters.p;
        A.out.pool_used = (unsigned long long*)b->d_poolused.p; A.out.pool = (uint32_t*)b->d_pool.p; A.out.pool_cap = b->pool_cap_words;
        A.out.index_used = (unsigned long long*)b->d_poolused.p + 1; A.out.index = (unsigned long long*)b->d_index.p; A.out.index_cap = b->index_cap; A.out.graph_first = (long long*)b->d_gfirst.p;
        A.out.trace_cap = b->trace_cap; A.out.trace_n = (int32_t*)b->d_trace_n.p; A.out.trace_codes = (int32_t*)b->d_trace_codes.p; A.out.trace_vals = (double*)b->d_trace_vals.p;
        A.prm = b->prm;
        A.work = (const int32_t*)b->d_work.p + woff; A.n_work = (int32_t)part[k].size(); A.attempt = pass;
        A.counter = (int32_t*)b->d_counter.p + 64 * k;
        A.slabs = (uint8_t*)b->d_slabs[k].p; A.slab_stride = stride;
        woff += part[k].size();
    }
    // kernels with work are dealt to the side streams, heaviest first onto the least loaded stream (cost ~ sum of V * E: the rule
    // cascade is superlinear in the graph size)
    double cost[ALD_NUM_SLOTS];
    for(int k = 0; k < ALD_NUM_SLOTS; k++) {
        cost[k] = 0;
        if(P.nblk[k] == 0) continue;
        for(int32_t g : part[k]) cost[k] += (double)b->hb.g_nv[g] * (double)b->hb.g_ne[g];
        cost[k] /= (double)P.nblk[k];                     // per resident wave
        P.order[P.nord++] = k;
    }
    std::sort(P.order, P.order + P.nord, [&](int x, int y) { return cost[x] > cost[y]; });
    if(const char *ev = getenv("ALD_LDS_FIRST")) {       // experiment: the LDS classes are launched before the slab-resident ones (1), or only the large LDS classes 5..9 (2)
        const int mode = atoi(ev);
        if(mode > 0) std::stable_sort(P.order, P.order + P.nord, [&](int x, int y) {
            auto first = [&](int k) { const int c = k % ALD_NUM_CLASSES; return mode == 1 ? c < ALD_FIRST_GLOBAL_CLASS : (c >= 5 && c < ALD_FIRST_GLOBAL_CLASS); };
            return first(x) && !first(y); });
    }
    // the last stream belongs to the small LDS classes (0..2: 16 KB of LDS per workgroup and less -- they fit beside anything and take a few
    // milliseconds: queued behind a large class they ran at the very end, cfg3's last 8 ms; profiles/r04/aa_cfg3_timeline_5_streams.txt),
    // the others are dealt to the remaining streams
    double load[ALD_SIDE_STREAMS_MAX] = {0};
    const int n_big = b->n_cstream > 1 ? b->n_cstream - 1 : 1;
    for(int q = 0; q < P.nord; q++) {
        const int k = P.order[q]; const int cls = k % ALD_NUM_CLASSES;
        if(b->n_cstream > 1 && cls <= 2) { P.stream_of[k] = b->n_cstream - 1; continue; }
        int st = 0; for(int z = 1; z < n_big; z++) if(load[z] < load[st]) st = z;
        load[st] += cost[k]; P.stream_of[k] = st;
    }
    return push_pass(b, P, pass == 0 ? b->up_stream : nullptr);      // (pass 0 is staged by ald_batch_upload: its copies keep to the upload stream)
}

// Buffers that a retry pass or a pool growth may have moved since pass 0 was staged (DevBuf::ensure frees and reallocates): the
// class slabs and the record pool.  Pass 0's arguments are brought up to date before they are used again.
bool refresh_pass_args(ald_batch *b, StagedPass &P)
{
    bool changed = false;
    for(int c = 0; c < ALD_NUM_SLOTS; c++) {
        if(P.nblk[c] == 0 || P.args.empty()) continue;
        KernelArgs &A = P.args[c];
        if(A.slabs != (uint8_t*)b->d_slabs[c].p) { A.slabs = (uint8_t*)b->d_slabs[c].p; changed = true; }
        if(A.out.pool != (uint32_t*)b->d_pool.p || A.out.pool_cap != b->pool_cap_words) { A.out.pool = (uint32_t*)b->d_pool.p; A.out.pool_cap = b->pool_cap_words; changed = true; }
        if(A.out.index != (unsigned long long*)b->d_index.p || A.out.index_cap != b->index_cap) { A.out.index = (unsigned long long*)b->d_index.p; A.out.index_cap = b->index_cap; changed = true; }
    }
    return changed;
}

int fire_pass(ald_batch *b, const StagedPass &P)          // fork on the side streams behind ev0, join on the batch stream before ev1
{
    if(P.tot == 0) return ALD_OK;
    HIPCHK(hipMemsetAsync(b->d_counter.p, 0, 4 * ALD_NUM_SLOTS * 64, b->stream));
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    for(int k = 0; k < P.nord; k++) {
        const int c = P.order[k], st = P.stream_of[c];
        b->blocks[c] = P.nblk[c]; b->launched_slab[c] = P.args[c].slabs;
        HIPCHK(hipStreamWaitEvent(b->cstream[st], b->ev0, 0));
        if(k_launch[c]((const KernelArgs*)b->d_args.p + c, P.nblk[c], b->cstream[st]) != 0) return set_err(ALD_ERR_HIP, "kernel launch failed");
        HIPCHK(hipEventRecord(b->cdone[c], b->cstream[st]));
        HIPCHK(hipStreamWaitEvent(b->stream, b->cdone[c], 0));
    }
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    return ALD_OK;
}

int launch_pass(ald_batch *b, const std::vector<int32_t> work[ALD_NUM_CLASSES], int pass)      // a retry pass: staged and fired at once
{
    StagedPass P;
    int rc = stage_pass(b, work, pass, P);
    if(rc != ALD_OK) return rc;
    b->pass0_on_device = false;                          // the work-list / argument buffers now hold this pass
    return fire_pass(b, P);
}

} // namespace

extern "C" {

const char *ald_last_error(void) { return g_err.c_str(); }
void ald_internal_set_error(const char *s) { g_err = s ? s : ""; }   /* used by the other translation units of the library */
const char *ald_version(void) { return "aletsch_amd-decomp 0.1 (gfx950)"; }

int ald_default_params(ald_params *p)
{
    if(!p) return ALD_ERR_INVALID;
    const double r[8] = {0.30, 0.00, 1.10, 1.10, 0.75, 0.30, 0.00, 1.00};     // util/parameters.cc:85-92
    for(int i = 0; i < 8; i++) p->max_decompose_error_ratio[i] = r[i];
    p->min_guaranteed_edge_weight = 0.01; p->min_transcript_coverage = 2.0; p->max_num_exons = 10000; p->reserved = 0;
    return ALD_OK;
}

namespace {
// pinned host memory for the arrays of a batch (host_pack.h: wire_alloc): portable, so that any device of the process can read it.
// The arrays are std::vectors: every growth is an allocation, a copy and a release, and hipHostFree SYNCHRONISES with the device -- a pack
// thread growing an array would stall the kernel of another slot.  So released blocks are not freed but kept, by size class (four
// classes per power of two: a request is rounded up by at most a quarter), and handed out again; what is kept is capped (ALD_PINNED_CACHE_MB,
// default 4096 -- the footprint of four 100 000-graph slots is 5.4 GB of live arrays; beyond the cap a release is a real hipHostFree).
// ald_batch_destroy of the last batch of the process frees the cache.
struct PinnedCache {
    std::mutex m;
    std::unordered_map<void*, size_t> size_of;                 // live and cached blocks -> rounded size
    std::map<size_t, std::vector<void*>> free_by_size;
    size_t cached_bytes = 0, cap_bytes = (size_t)4096 << 20;
    PinnedCache() { if(const char *ev = getenv("ALD_PINNED_CACHE_MB")) { const long long k = atoll(ev); if(k >= 0) cap_bytes = (size_t)k << 20; } }
    static size_t round_up(size_t b) { if(b < 4096) return 4096; size_t p2 = 4096; while(p2 < b) p2 <<= 1; const size_t q = p2 >> 3; return (b + q - 1) / q * q; }     // (p2/2, p2] in four steps
    void *get(size_t bytes) {
        const size_t r = round_up(bytes);
        { std::lock_guard<std::mutex> lk(m);
          auto it = free_by_size.find(r);
          if(it != free_by_size.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); cached_bytes -= r; return p; } }
        void *p = nullptr;
        if(hipHostMalloc(&p, r, hipHostMallocPortable) != hipSuccess) {       // out of pinned memory: give the cache back and try once more
            drop_all(); p = nullptr;
            if(hipHostMalloc(&p, r, hipHostMallocPortable) != hipSuccess) return nullptr;
        }
        std::lock_guard<std::mutex> lk(m); size_of[p] = r; return p;
    }
    void put(void *p) {
        if(!p) return;
        { std::lock_guard<std::mutex> lk(m);
          auto it = size_of.find(p);
          if(it != size_of.end() && cached_bytes + it->second <= cap_bytes) { free_by_size[it->second].push_back(p); cached_bytes += it->second; return; }
          if(it != size_of.end()) size_of.erase(it); }
        hipHostFree(p);
    }
    void drop_all() {
        std::vector<void*> v;
        { std::lock_guard<std::mutex> lk(m); for(auto &kv : free_by_size) { for(void *p : kv.second) { v.push_back(p); size_of.erase(p); } kv.second.clear(); } cached_bytes = 0; }
        for(void *p : v) hipHostFree(p);
    }
};
PinnedCache &pinned_cache() { static PinnedCache *c = new PinnedCache(); return *c; }      // (never destroyed: batches may outlive static destruction order)
std::atomic<int> g_live_batches{0};
void *wire_pinned_alloc(size_t bytes) { return pinned_cache().get(bytes ? bytes : 1); }
void wire_pinned_release(void *p) { pinned_cache().put(p); }
struct WireHooksInstaller { WireHooksInstaller() { wire_hooks().alloc = wire_pinned_alloc; wire_hooks().release = wire_pinned_release; } } g_wire_hooks_installer;
}

int ald_batch_create(const ald_params *p, int device, ald_batch **out)
{
    if(!out) return ALD_ERR_INVALID;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(ALD_ERR_NO_DEVICE, "no HIP device visible: the decomposition path has no CPU fallback");
    if(device < 0 || device >= ndev) return set_err(ALD_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ald_batch *b = new ald_batch();
    g_live_batches.fetch_add(1);
    b->device = device; b->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    params_from_abi(p, b->prm);
    bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&b->ev0) == hipSuccess && hipEventCreate(&b->ev1) == hipSuccess;
    if(const char *ev = getenv("ALD_SIDE_STREAMS")) { const int k = atoi(ev); if(k >= 1 && k <= ALD_SIDE_STREAMS_MAX) b->n_cstream = k; }      // tuning knob
    for(int q = 0; q < b->n_cstream && ok; q++) ok = hipStreamCreateWithFlags(&b->cstream[q], hipStreamNonBlocking) == hipSuccess;
    const int up_mode = getenv("ALD_UPLOAD_STREAM") ? atoi(getenv("ALD_UPLOAD_STREAM")) : 2;      // 0: uploads through the batch's own stream (A/B); 1: a stream of the lowest priority; 2: of the highest (default: DMA copies take nothing from a kernel, and they finish sooner: 24 against 31 ms per 1.35 GB batch)
    if(ok && up_mode != 0) {
        int least = 0, greatest = 0;
        if(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            if(hipStreamCreateWithPriority(&b->up_stream, hipStreamNonBlocking, up_mode == 2 ? greatest : least) != hipSuccess) b->up_stream = nullptr;      // (without it: the batch's stream, as before)
        }
    }
    for(int c = 0; c < ALD_NUM_SLOTS && ok; c++) ok = hipEventCreateWithFlags(&b->cdone[c], hipEventDisableTiming) == hipSuccess;
    if(!ok) { ald_batch_destroy(b); return set_err(ALD_ERR_HIP, "stream/event creation failed"); }
    *out = b;
    return ALD_OK;
}

int ald_batch_destroy(ald_batch *b)
{
    if(!b) return ALD_OK;
    hipSetDevice(b->device);
    if(b->stream) hipStreamSynchronize(b->stream);
    b->pin_in.release(); b->pin_out.release(); b->pin_small.release(); b->pin_index.release(); delete b->pass0; b->pass0 = nullptr;
    DevBuf *bufs[] = {&b->d_in, &b->d_status, &b->d_npaths, &b->d_niters, &b->d_pool, &b->d_poolused, &b->d_index, &b->d_gfirst, &b->d_pbegin, &b->d_ordoff, &b->d_trace_n, &b->d_trace_codes, &b->d_trace_vals, &b->d_work, &b->d_counter, &b->d_args};
    for(DevBuf *d : bufs) d->release();
    for(int c = 0; c < ALD_NUM_SLOTS; c++) b->d_slabs[c].release();
    for(DevBuf &d : b->red) d.release();
    for(DevBuf &d : b->dts) d.release();
    for(PinBuf &d : b->red_pin) d.release();
    for(int q = 0; q < ALD_SIDE_STREAMS_MAX; q++) if(b->cstream[q]) { hipStreamSynchronize(b->cstream[q]); hipStreamDestroy(b->cstream[q]); }
    for(int c = 0; c < ALD_NUM_SLOTS; c++) if(b->cdone[c]) hipEventDestroy(b->cdone[c]);
    if(b->ev0) hipEventDestroy(b->ev0);
    if(b->ev1) hipEventDestroy(b->ev1);
    if(b->up_stream) { hipStreamSynchronize(b->up_stream); hipStreamDestroy(b->up_stream); }
    if(b->stream) hipStreamDestroy(b->stream);
    delete b;
    if(g_live_batches.fetch_sub(1) == 1) pinned_cache().drop_all();      // the last batch of the process: the kept pinned blocks go back
    return ALD_OK;
}

int ald_batch_clear(ald_batch *b)
{
    if(!b) return ALD_ERR_INVALID;
    b->hb.clear(); b->res.clear(); b->uploaded = b->ran = b->downloaded = false; b->kernel_ms = -1;
    return ALD_OK;
}

// the batch's host arrays are pinned memory: an allocation the driver refuses must not leave through the C ABI as an exception
static int staging_out_of_memory(ald_batch *b)
{
    try { b->hb.clear(); } catch(...) {}
    return set_err(ALD_ERR_NOMEM, "out of (pinned) host memory while staging: the batch was cleared");
}

int ald_batch_add_graph(ald_batch *b, const ald_graph_view *g)
{
    if(!b || !g) return ALD_ERR_INVALID;
    b->uploaded = b->ran = b->downloaded = false;
    int rc;
    try { rc = b->hb.add_graph(*g); } catch(const std::bad_alloc &) { return staging_out_of_memory(b); }
    if(rc != ALD_OK) return set_err(rc, b->hb.err);
    return ALD_OK;
}

int ald_batch_add_packed(ald_batch *b, int32_t n, const int32_t *g_nv, const int32_t *g_ne, const int32_t *g_np,
                         const int32_t *vertex_offset, const int32_t *edge_target, const double *edge_weight, const uint8_t *edge_strand, const double *edge_abd,
                         const int32_t *edge_sample_offset, const int32_t *sample_id, const double *sample_abd,
                         const double *vertex_weight, const int32_t *vertex_lpos, const int32_t *vertex_rpos, const int32_t *vertex_type,
                         const int32_t *phasing_offset, const int32_t *phasing_vertex, const int32_t *phasing_count, const char *graph_strand, const int32_t *edge_count,
                         const int32_t *edge_creation_rank)
{
    if(!b || n < 0 || !g_nv || !g_ne) return ALD_ERR_INVALID;
    b->uploaded = b->ran = b->downloaded = false;
    int rc;
    try { rc = b->hb.add_packed(n, g_nv, g_ne, g_np, vertex_offset, edge_target, edge_weight, edge_strand, edge_abd, edge_sample_offset, sample_id, sample_abd,
                              vertex_weight, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, graph_strand, edge_count, edge_creation_rank); }
    catch(const std::bad_alloc &) { return staging_out_of_memory(b); }
    if(rc != ALD_OK) return set_err(rc, b->hb.err);
    return ALD_OK;
}

int ald_batch_add_packed_raw(ald_batch *b, int32_t n, const int32_t *g_nv, const int32_t *g_ne, const int32_t *g_np,
                             const int32_t *vertex_offset, const int32_t *edge_target, const double *edge_weight, const uint8_t *edge_strand, const double *edge_abd,
                             const int32_t *edge_sample_offset, const int32_t *sample_id, const double *sample_abd,
                             const double *vertex_weight, const int32_t *vertex_lpos, const int32_t *vertex_rpos, const int32_t *vertex_type,
                             const int32_t *phasing_offset, const int32_t *phasing_vertex, const int32_t *phasing_count, const char *graph_strand, const int32_t *edge_count,
                             const int32_t *edge_creation_rank,
                             const int32_t *raw_max_group_boundary_distance, const int32_t *g_nphase, const int32_t *phase_offset, const int32_t *phase_coord, const int32_t *phase_count)
{
    if(!b || n < 0 || !g_nv || !g_ne) return ALD_ERR_INVALID;
    b->uploaded = b->ran = b->downloaded = false;
    int rc;
    try { rc = b->hb.add_packed_raw(n, g_nv, g_ne, g_np, vertex_offset, edge_target, edge_weight, edge_strand, edge_abd, edge_sample_offset, sample_id, sample_abd,
                                  vertex_weight, vertex_lpos, vertex_rpos, vertex_type, phasing_offset, phasing_vertex, phasing_count, graph_strand, edge_count, edge_creation_rank,
                                  raw_max_group_boundary_distance, g_nphase, phase_offset, phase_coord, phase_count); }
    catch(const std::bad_alloc &) { return staging_out_of_memory(b); }
    if(rc != ALD_OK) return set_err(rc, b->hb.err);
    return ALD_OK;
}

int ald_batch_num_graphs(const ald_batch *b) { return b ? b->hb.n() : 0; }

int ald_batch_enable_trace(ald_batch *b, int32_t max_events_per_graph)
{
    if(!b || max_events_per_graph < 0) return ALD_ERR_INVALID;
    b->trace_cap = max_events_per_graph; b->uploaded = false;
    return ALD_OK;
}

int ald_batch_upload(ald_batch *b)
{
    if(!b) return ALD_ERR_INVALID;
    HIPCHK(hipSetDevice(b->device));
    const int n = b->hb.n();
    hipStream_t us = b->up_stream ? b->up_stream : b->stream;
    if(us != b->stream && b->ran && !b->downloaded) HIPCHK(hipStreamSynchronize(b->stream));      // a run whose results were never fetched may still read the input buffer
    b->in_bytes = b->hb.layout(b->sec);
    if(b->d_in.ensure(b->in_bytes)) return set_err(ALD_ERR_NOMEM, "device input buffer");
    // ONE buffer on the device, its sections filled straight from the batch's host arrays: they live in pinned memory (wire_alloc, hooks
    // installed by ald_batch_create), so every section is one asynchronous copy of the DMA engine and no host thread packs anything
    // (the pack into a second, pinned buffer -- 1.3 GB read and written, 20 ms on sixteen threads -- was as long as the PCIe transfer itself)
    for(int i = 0; i < HostBatch::S_COUNT; i++)
        if(b->sec[i].bytes) HIPCHK(hipMemcpyAsync((uint8_t*)b->d_in.p + b->sec[i].off, b->sec[i].src, b->sec[i].bytes, hipMemcpyHostToDevice, us));
    // outputs
    // a heuristic, not a bound (one graph can need about (E - V + 2) * (V + 14) words): a graph that finds the pool full reports
    // ALD_ST_POOL_FULL and ald_batch_download grows the pool.  ALD_DEBUG_POOL_WORDS starts it small so that tests reach that path.
    uint64_t pool = 0; for(int g = 0; g < n; g++) pool += 16ull * b->hb.g_ne[g] + 256;
    if(const char *ev = getenv("ALD_DEBUG_POOL_WORDS")) { const long long k = atoll(ev); if(k > 0 && (uint64_t)k < pool) pool = (uint64_t)k; }
    b->pool_cap_words = pool;
    b->index_cap = pool / (REC_HDR_WORDS + 2) + 1;         // a record is at least a header and two vertices long
    if(b->d_status.ensure(4 * (size_t)n + 4) || b->d_npaths.ensure(4 * (size_t)n + 4) || b->d_niters.ensure(4 * (size_t)n + 4) || b->d_pool.ensure(4 * pool + 64) || b->d_poolused.ensure(64)
       || b->d_index.ensure(8 * (size_t)b->index_cap + 64) || b->d_gfirst.ensure(8 * (size_t)n + 8))
        { hipStreamSynchronize(us); return set_err(ALD_ERR_NOMEM, "device output buffers"); }      // (the input copies are in flight: they read the batch's host arrays)
    if(b->trace_cap > 0) {
        if(b->d_trace_n.ensure(4 * (size_t)n + 4) || b->d_trace_codes.ensure(12ull * n * b->trace_cap + 4) || b->d_trace_vals.ensure(8ull * n * b->trace_cap + 8)) { hipStreamSynchronize(us); return set_err(ALD_ERR_NOMEM, "trace buffers"); }
        HIPCHK(hipMemsetAsync(b->d_trace_n.p, 0, 4 * (size_t)n + 4, us));
    }
    // (no wait here: the host work below needs the batch's host arrays only and runs while the DMA engines move them; push_pass, at the end of
    // stage_pass, waits for the stream -- the copies above included)
    // the first pass is part of what "resident" means: size class of every graph, longest-processing-time-first order inside a class
    // (the persistent waves pull graphs in list order, so the big graphs of a class start early and the tail is made of small ones),
    // work lists and kernel arguments in HBM
    {
        b->cls0.assign(n, -1);
        std::vector<int32_t> work[ALD_NUM_CLASSES];
        for(int g = 0; g < n; g++) {
            int64_t ns = b->hb.off_s[g + 1] - b->hb.off_s[g], npv = b->hb.off_pv[g + 1] - b->hb.off_pv[g];
            if(b->hb.g_rawdist[g] >= 0) npv += 2 * (b->hb.off_rc[g + 1] - b->hb.off_rc[g]);      // a raw graph's lists do not exist yet: an estimate (too small a class is retried one up)
            int c = debug_underclass(pick_class(b->hb.g_nv[g], b->hb.g_ne[g], ns, npv));
            b->cls0[g] = c;
            if(c >= 0) work[c].push_back(g);
        }
        // LDS form or slab-resident twin for the large classes?  The LDS-hungry classes of a batch (16 KB per workgroup and up) serialise --
        // each holds its CU's LDS for as long as it runs -- so the batch takes about the SUM of their times; the twins take no LDS and run beside them, 12 workgroups per CU
        // instead of 3, but a graph takes 2-3 times longer there and the first result comes after ~90 ms.  Estimate both (constants
        // from tools/_gpu_twin.py and the per-class times of cfg3, DESIGN.md section 5: one round of a class takes about MAXE / 60 ms,
        // the twins 90 ms + 19 us per graph) and move the graphs when the twins win.  ALD_DEBUG_TWIN=1 / 0 forces / forbids the move.
        {
            double others = 0, mine = 0; int64_t n_tw = 0;
            for(int c = 0; c < ALD_NUM_PICK_CLASSES; c++) {
                if(work[c].empty() || c == ALD_CATCH_ALL_CLASS) continue;
                const double cap = (double)b->n_cus * std::max(1, occupancy_for(b, c));
                const double t = std::max(1.0, (double)work[c].size() / cap) * class_info(c).maxe / 60.0;
                if(class_twin(c) >= 0) { mine += t; n_tw += (int64_t)work[c].size(); }
                else if(class_info(c).maxe >= 640) others += t;      // the small classes (8 KB of LDS and less per workgroup) fit beside anything and do not serialise
            }
            bool move = n_tw > 0 && std::max(1.1 * others, 90.0 + 0.019 * (double)n_tw) < others + mine;
            if(const char *ev = getenv("ALD_DEBUG_TWIN")) move = atoi(ev) != 0;
            if(move) for(int c = 0; c < ALD_NUM_PICK_CLASSES; c++) {
                const int tw = class_twin(c);
                if(tw < 0 || work[c].empty()) continue;
                for(int32_t g : work[c]) b->cls0[g] = tw;
                work[tw].swap(work[c]);
            }
        }
        // Round 4: the slab twins keep their sweep records between sweeps and run a 385..512-vertex graph in 67 ms where the LDS form of the
        // band takes 86; when they run anyway, their kernel has workgroup slots to spare (12 per CU, no LDS) for as long as its longest graph
        // takes, while the LDS classes beside them queue for LDS (profiles/r04/h_cfg3_timeline.txt: the LDS chain ends 20 ms after the
        // twins).  So the smallest graphs of classes 6 and 5 -- the two that hold the most LDS for the longest -- go to twin 11 (its
        // capacities hold any graph of theirs) while free twin slots last.  ALD_TWIN_SPILL="n6,n5" fixes the two counts (0,0: off).
        if(!work[11].empty() || !work[12].empty()) {
            int64_t free_tw = (int64_t)b->n_cus * std::max(1, occupancy_for(b, 11)) - (int64_t)work[11].size() - (int64_t)work[12].size();
            long want[2] = {-1, -1};
            if(const char *ev = getenv("ALD_TWIN_SPILL")) sscanf(ev, "%ld,%ld", &want[0], &want[1]);
            const ClassInfo k11 = class_info(11);
            for(int q = 0; q < 2 && free_tw > 0; q++) {
                const int c = 6 - q;
                if(work[c].empty()) continue;
                const int64_t cap = (int64_t)b->n_cus * std::max(1, occupancy_for(b, c)), sz = (int64_t)work[c].size();
                int64_t k = want[q] >= 0 ? want[q] : ALD_TWIN_SPILL_DEFAULT(q, sz, cap);
                k = std::min(k, std::min(free_tw, sz));
                if(k <= 0) continue;
                std::stable_sort(work[c].begin(), work[c].end(), [&](int32_t x, int32_t y) { return b->hb.g_ne[x] > b->hb.g_ne[y]; });
                int64_t moved = 0;
                while(moved < k && !work[c].empty()) {
                    const int32_t g = work[c].back();
                    if(!(b->hb.g_nv[g] <= k11.nw * 64 && 2 * b->hb.g_nv[g] <= k11.maxv && b->hb.g_ne[g] + k11.maxe / 10 <= k11.maxe)) break;
                    work[c].pop_back(); work[11].push_back(g); b->cls0[g] = 11; moved++;
                }
                free_tw -= moved;
            }
        }
        for(int c = 0; c < ALD_NUM_CLASSES; c++)
            std::stable_sort(work[c].begin(), work[c].end(), [&](int32_t x, int32_t y) { return b->hb.g_ne[x] > b->hb.g_ne[y]; });
        if(!b->pass0) b->pass0 = new StagedPass();
        int rc = stage_pass(b, work, 0, *b->pass0);
        if(rc != ALD_OK) { hipStreamSynchronize(us); return rc; }
        b->pass0_on_device = true;
    }
    HIPCHK(hipStreamSynchronize(us));             // (a batch without work never reached push_pass)
    b->uploaded = true; b->ran = false; b->downloaded = false;
    return ALD_OK;
}

static int start_run(ald_batch *b);
int ald_batch_run(ald_batch *b)
{
    if(!b) return ALD_ERR_INVALID;
    if(!b->uploaded || !b->pass0) return set_err(ALD_ERR_STATE, "ald_batch_run before ald_batch_upload");
    b->kernel_ms = 0;
    return start_run(b);
}
static int start_run(ald_batch *b)
{
    HIPCHK(hipSetDevice(b->device));
    const int n = b->hb.n();
    HIPCHK(hipMemsetAsync(b->d_poolused.p, 0, 64, b->stream));
    HIPCHK(hipMemsetAsync(b->d_status.p, 0, 4 * (size_t)n + 4, b->stream));
    HIPCHK(hipMemsetAsync(b->d_npaths.p, 0, 4 * (size_t)n + 4, b->stream));
    HIPCHK(hipMemsetAsync(b->d_gfirst.p, 0xFF, 8 * (size_t)n + 8, b->stream));      // graph_first = -1 for every graph no wave ever finishes (ALD_ST_TOO_LARGE: never queued)
    b->cls = b->cls0; b->attempt.assign(n, 0); b->status.assign(n, 0);
    b->passes = 0;
    if(refresh_pass_args(b, *b->pass0)) b->pass0_on_device = false;      // a slab or the pool moved since pass 0 was staged (stale slab pointers)
    if(!b->pass0_on_device) { int rc = push_pass(b, *b->pass0); if(rc != ALD_OK) return rc; b->pass0_on_device = true; }      // a retry pass of the last run used the buffers
    int rc = fire_pass(b, *b->pass0);
    if(rc != ALD_OK) return rc;
    b->ran = true; b->downloaded = false;
    return ALD_OK;
}

int ald_batch_sync(ald_batch *b)
{
    if(!b) return ALD_ERR_INVALID;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    return ALD_OK;
}

int ald_batch_download(ald_batch *b)
{
    if(!b) return ALD_ERR_INVALID;
    if(!b->ran) return set_err(ALD_ERR_STATE, "ald_batch_download before ald_batch_run");
    HIPCHK(hipSetDevice(b->device));
    const int n = b->hb.n();
    const bool prof = getenv("ALD_DOWNLOAD_PROF") != nullptr; const auto P0 = std::chrono::steady_clock::now(); auto P1 = P0, P2 = P0, P3 = P0;
    b->n_paths.assign(n, 0); b->n_iters.assign(n, 0);
    // Everything comes back through async copies on the batch's OWN stream into pinned memory.  (A synchronous hipMemcpy runs on the
    // null stream, which waits for every blocking stream of the device -- i.e. for the kernel of the NEXT batch, already in flight in a
    // pipelined caller: the download of batch k took as long as the kernel of batch k+1, and its record copy ran between two kernels.)
    if(b->pin_small.ensure(64 + 20 * (size_t)n + 64, true)) return set_err(ALD_ERR_NOMEM, "pinned status buffer");
    unsigned long long *h_used = (unsigned long long*)b->pin_small.p;
    long long *h_gf = (long long*)((uint8_t*)b->pin_small.p + 64);
    int32_t *st = (int32_t*)(h_gf + n), *h_np = st + n, *h_ni = h_np + n;
    unsigned long long used = 0, iused = 0;
  for(int regrow = 0; ; regrow++) {
    bool pool_full = false;
    for(int pass = 0; pass <= ALD_NUM_CLASSES; pass++) {
        HIPCHK(hipStreamSynchronize(b->stream));
        if(pass == 0 && regrow == 0) P1 = std::chrono::steady_clock::now();
        float ms = 0; if(hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->kernel_ms += ms;
        b->passes++;
        if(n > 0) { HIPCHK(hipMemcpyAsync(st, b->d_status.p, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream)); HIPCHK(hipStreamSynchronize(b->stream)); }
        if(pass == 0 && regrow == 0) P2 = std::chrono::steady_clock::now();
        // graphs whose working set overflowed their class are retried one class up (records carry the pass number)
        std::vector<int32_t> work[ALD_NUM_CLASSES]; bool any = false;
        for(int g = 0; g < n; g++) {
            if(b->cls[g] < 0) { b->status[g] = ALD_ST_TOO_LARGE; continue; }
            if(b->attempt[g] != pass) continue;                 // not part of this pass
            b->status[g] = st[g];
            if(st[g] == ALD_ST_POOL_FULL) pool_full = true;
            if(st[g] == ALD_ST_CAPACITY && getenv("ALD_DEBUG_RETRY")) fprintf(stderr, "[ald] graph %d (V=%d E=%d) overflowed class %d in pass %d\n", g, b->hb.g_nv[g], b->hb.g_ne[g], b->cls[g], pass);
            if(st[g] == ALD_ST_CAPACITY && class_retry_up(b->cls[g]) >= 0) { b->cls[g] = class_retry_up(b->cls[g]); b->attempt[g] = pass + 1; work[b->cls[g]].push_back(g); any = true; }
        }
        if(!any || pool_full) break;
        int rc = launch_pass(b, work, pass + 1);
        if(rc != ALD_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(h_used, b->d_poolused.p, 16, hipMemcpyDeviceToHost, b->stream)); HIPCHK(hipStreamSynchronize(b->stream)); used = h_used[0]; iused = h_used[1];
    if(!pool_full) break;
    // Some graph found the record pool full.  Records behind the first refused one may be missing (the bump pointer moved, nothing was
    // written), so the stream of this run is unusable as a whole: the pool grows -- `used` counts every request made, a lower bound
    // of the need -- and the batch is decomposed again from pass 0.  Rare by construction (single-graph batches of long sparse graphs
    // with many isoforms), so the cost of the second run does not matter.
    if(regrow >= 8) break;                                 // gives up: the graphs keep ALD_ST_POOL_FULL
    uint64_t want = std::max<uint64_t>(2 * b->pool_cap_words, used + used / 4 + 4096);
    if(b->d_pool.ensure(4 * want + 64)) return set_err(ALD_ERR_NOMEM, "record pool");
    b->pool_cap_words = want;
    b->index_cap = std::max<uint64_t>(want / (REC_HDR_WORDS + 2) + 1, iused + iused / 4 + 64);
    if(b->d_index.ensure(8 * (size_t)b->index_cap + 64)) return set_err(ALD_ERR_NOMEM, "result index");
    int rc = start_run(b);
    if(rc != ALD_OK) return rc;
  }
    if(used > b->pool_cap_words) used = b->pool_cap_words;
    if(iused > b->index_cap) iused = b->index_cap;
    b->res.clear();
    // the records land in a pinned buffer (kept across runs) through an async copy on the batch stream: the copy engine moves them
    // while another batch's kernel may be running, and the host thread only waits
    if(b->pin_out.ensure(4 * (size_t)used + 64, true) || b->pin_index.ensure(8 * (size_t)iused + 64, true)) return set_err(ALD_ERR_NOMEM, "pinned result buffer");
    P3 = std::chrono::steady_clock::now();
    if(n > 0) {
        HIPCHK(hipMemcpyAsync(h_np, b->d_npaths.p, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipMemcpyAsync(h_ni, b->d_niters.p, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipMemcpyAsync(h_gf, b->d_gfirst.p, 8 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
    }
    if(iused) HIPCHK(hipMemcpyAsync(b->pin_index.p, b->d_index.p, 8 * iused, hipMemcpyDeviceToHost, b->stream));
    if(used) HIPCHK(hipMemcpyAsync(b->pin_out.p, b->d_pool.p, 4 * used, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    const auto P3b = std::chrono::steady_clock::now();
    if(n > 0) { memcpy(b->n_paths.data(), h_np, 4 * (size_t)n); memcpy(b->n_iters.data(), h_ni, 4 * (size_t)n); }
    b->res.ext_pool = (const uint32_t*)b->pin_out.p; b->res.ext_words = used;
    b->res.status = b->status; b->res.n_iters = b->n_iters;
    // paths AND transcripts of the batch, decoded, in host memory: every record is reached through the index the kernel wrote
    // (index[graph_first[g] + p]); exons are already joined inside the records; coverage = log(1 + weight) with the host's libm
    b->total_paths = 0; b->paths_on_device = false;
    {
        const int rc = b->res.build(n, b->n_paths.data(), (const unsigned long long*)b->pin_index.p, iused, h_gf);
        if(rc != 0) return set_err(ALD_ERR_STATE, "result index is inconsistent with the record pool (rc=" + std::to_string(rc) + ")");
        b->total_paths = b->res.n_paths();
    }
    { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
      b->dl_ms[0] = ms(P0, P1); b->dl_ms[1] = ms(P1, P3); b->dl_ms[2] = ms(P3, P3b); b->dl_ms[3] = ms(P3b, std::chrono::steady_clock::now()); b->dl_bytes = 4 * (int64_t)used + 8 * (int64_t)iused + 20 * (int64_t)n; }
    if(prof) { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        fprintf(stderr, "[download] wait for the kernel %.2f ms, status copy %.2f ms, retries + counters %.2f ms, records + index (%.0f MB) + decode of %lld paths %.2f ms\n", ms(P0, P1), ms(P1, P2), ms(P2, P3), 4e-6 * (double)used + 8e-6 * (double)iused, (long long)b->total_paths, ms(P3, std::chrono::steady_clock::now())); }
    if(b->trace_cap > 0 && n > 0) {
        b->trace_n.resize(n); b->trace_codes.resize(3ull * n * b->trace_cap); b->trace_vals.resize((size_t)n * b->trace_cap);
        HIPCHK(hipMemcpy(b->trace_n.data(), b->d_trace_n.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b->trace_codes.data(), b->d_trace_codes.p, 12ull * n * b->trace_cap, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b->trace_vals.data(), b->d_trace_vals.p, 8ull * n * b->trace_cap, hipMemcpyDeviceToHost));
    }
    b->downloaded = true;
    return ALD_OK;
}

double ald_batch_last_kernel_ms(const ald_batch *b) { return b ? b->kernel_ms : -1; }

int ald_batch_last_download_ms(const ald_batch *b, double *wait_kernel_ms, double *status_retries_ms, double *copy_ms, double *decode_ms, int64_t *bytes_to_host)
{
    if(!b) return ALD_ERR_INVALID;
    if(wait_kernel_ms) *wait_kernel_ms = b->dl_ms[0]; if(status_retries_ms) *status_retries_ms = b->dl_ms[1]; if(copy_ms) *copy_ms = b->dl_ms[2]; if(decode_ms) *decode_ms = b->dl_ms[3];
    if(bytes_to_host) *bytes_to_host = b->dl_bytes;
    return ALD_OK;
}

/* the result index as the kernel wrote it (host copies of the last download): index[graph_first[g] + p] = word offset of record (g, p)
 * in the raw record stream; graph_first[g] = -1 for a graph without paths or one that did not end well */
int ald_batch_result_index(const ald_batch *b, const uint64_t **index, int64_t *n_entries, const int64_t **graph_first)
{
    if(!b || !b->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_result_index before ald_batch_download");
    if(index) *index = (const uint64_t*)b->pin_index.p;
    if(n_entries) *n_entries = b->total_paths;
    if(graph_first) *graph_first = (const int64_t*)((const uint8_t*)b->pin_small.p + 64);
    return ALD_OK;
}

int ald_batch_algorithmic_bytes(const ald_batch *b, int64_t *in_bytes, int64_t *out_bytes)
{
    if(!b) return ALD_ERR_INVALID;
    if(in_bytes) *in_bytes = b->hb.algorithmic_in_bytes();
    if(out_bytes) { *out_bytes = 0; if(b->downloaded && ensure_index(b) == ALD_OK) *out_bytes = b->res.out_bytes; }
    return ALD_OK;
}

int ald_batch_get_result(const ald_batch *b, int32_t graph, ald_result_view *out)
{
    if(!b || !out || !b->downloaded || graph < 0 || graph >= b->hb.n()) return ALD_ERR_INVALID;
    { int rc = ensure_index(b); if(rc != ALD_OK) return rc; }
    out->status = b->res.status[graph]; out->num_paths = (int32_t)(b->res.path_begin[graph + 1] - b->res.path_begin[graph]);
    out->num_iterations = b->res.n_iters[graph]; out->reserved = 0;
    return ALD_OK;
}

int ald_batch_export_iterations(const ald_batch *b, int32_t *num_iterations)
{
    if(!b || !num_iterations) return ALD_ERR_INVALID;
    if(!b->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_export_iterations before ald_batch_download");
    for(int g = 0; g < b->hb.n(); g++) num_iterations[g] = b->n_iters[g];
    return ALD_OK;
}

int ald_batch_get_path(const ald_batch *b, int32_t graph, int32_t path, ald_path_view *out)
{
    if(!b || !out || !b->downloaded || graph < 0 || graph >= b->hb.n()) return ALD_ERR_INVALID;
    { int rc = ensure_index(b); if(rc != ALD_OK) return rc; }
    int64_t i = b->res.path_begin[graph] + path;
    if(path < 0 || i >= b->res.path_begin[graph + 1]) return ALD_ERR_INVALID;
    const PathRec p = b->res.path((int64_t)(i));
    out->num_vertices = p.nv; out->vertices = (const int32_t*)b->res.vertices(p);
    out->weight = p.weight; out->abd = p.abd; out->conf = p.conf; out->reads = p.reads; out->length = p.length; out->count = p.count; out->strand = p.strand;
    return ALD_OK;
}

int ald_batch_export(const ald_batch *b, int64_t *total_paths, int64_t *total_path_vertices, int32_t *status, int32_t *path_offset,
                     double *weight, double *abd, double *conf, double *reads, int32_t *length, int32_t *count, char *strand,
                     int64_t *pv_offset, int32_t *path_vertices)
{
    if(!b || !b->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_export before ald_batch_download");
    { int rc = ensure_index(b); if(rc != ALD_OK) return rc; }
    return export_results(b->res, b->hb.n(), total_paths, total_path_vertices, status, path_offset, weight, abd, conf, reads, length, count, strand, pv_offset, path_vertices);
}

int ald_batch_get_trace(const ald_batch *b, int32_t graph, int32_t *n_events, const int32_t **codes, const double **values)
{
    if(!b || !n_events || !b->downloaded || graph < 0 || graph >= b->hb.n()) return ALD_ERR_INVALID;
    if(b->trace_cap <= 0) { *n_events = 0; return ALD_OK; }
    *n_events = b->trace_n[graph] < b->trace_cap ? b->trace_n[graph] : b->trace_cap;
    if(codes) *codes = b->trace_codes.data() + 3ull * graph * b->trace_cap;
    if(values) *values = b->trace_vals.data() + (size_t)graph * b->trace_cap;
    return ALD_OK;
}

/* scallop::build_transcripts / build_transcript (scallop.cc:3250-3266, essential.cc:719-748): the exon join is done by the kernel
 * (the record of a path carries the exons of its transcript), coverage = log(1 + weight) by ald_batch_download */
int ald_batch_get_transcript(const ald_batch *b, int32_t graph, int32_t path, ald_transcript_view *out)
{
    if(!b || !out || !b->downloaded || graph < 0 || graph >= b->hb.n()) return ALD_ERR_INVALID;
    int64_t i = b->res.path_begin[graph] + path;
    if(path < 0 || i >= b->res.path_begin[graph + 1]) return ALD_ERR_INVALID;
    const PathRec p = b->res.path((int64_t)(i));
    out->num_exons = p.nexw / 2; out->exons = b->res.exons(p);
    out->coverage = p.coverage; out->conf = p.conf; out->abd = p.abd; out->count1 = p.count; out->strand = p.strand;
    return ALD_OK;
}

int ald_batch_export_transcripts(const ald_batch *b, int64_t *total_exons, double *coverage, int64_t *exon_offset, int32_t *exon_lr)
{
    if(!b || !b->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_export_transcripts before ald_batch_download");
    int64_t te = 0; const int64_t np = b->res.n_paths();
    for(int64_t i = 0; i < np; i++) {
        const PathRec p = b->res.path((int64_t)((size_t)i));
        if(coverage) { coverage[i] = p.coverage; exon_offset[i] = te; if(p.nexw) memcpy(exon_lr + 2 * te, b->res.exons(p), 4 * (size_t)p.nexw); }
        te += p.nexw / 2;
    }
    if(coverage) exon_offset[np] = te;
    if(total_exons) *total_exons = te;
    return ALD_OK;
}

/* ---- result sink: transcript_set restated (aletsch_amd/host/transcript_sink.hpp) ---- */

int ald_tset_create(double single_exon_overlap, ald_tset **out) { if(!out) return ALD_ERR_INVALID; *out = new ald_tset(single_exon_overlap); return ALD_OK; }
int ald_tset_destroy(ald_tset *t) { delete t; return ALD_OK; }

int ald_tset_add(ald_tset *t, int32_t n_groups, const int64_t *group_offset, const int32_t *group_sid,
                 const char *strand, const double *coverage, const double *conf, const double *abd, const int32_t *count1,
                 const int64_t *tid, const int64_t *exon_offset, const int32_t *exon_lr, int32_t skip_single_exon)
{
    if(!t || n_groups < 0 || (n_groups > 0 && (!group_offset || !group_sid || !strand || !coverage || !conf || !abd || !count1 || !tid || !exon_offset || !exon_lr))) return ALD_ERR_INVALID;
    for(int g = 0; g < n_groups; g++) {
        aletsch::transcript_sink ts(t->overlap);
        for(int64_t i = group_offset[g]; i < group_offset[g + 1]; i++) {
            aletsch::sink_transcript x;
            x.strand = strand[i]; x.coverage = coverage[i]; x.top.cov2 = coverage[i]; x.top.conf = conf[i]; x.top.abd = abd[i]; x.top.count1 = count1[i]; x.count2 = 1; x.tid = tid[i];
            x.xs.assign(exon_lr + 2 * exon_offset[i], exon_lr + 2 * exon_offset[i + 1]);
            if(x.n_exons() <= 1 && skip_single_exon) continue;               // assembler.cc:1117
            ts.add(x, 1, group_sid[g]);                                      // assembler.cc:1120
        }
        t->add(ts);                                                          // assembler.cc:1130
    }
    return ALD_OK;
}

} // extern "C"  (the helpers below are templates / C++ types)
namespace {
unsigned sink_threads(int64_t n_transcripts)
{
    unsigned nthr = std::thread::hardware_concurrency(); if(nthr == 0) nthr = 1; if(nthr > 16) nthr = 16;
    if(const char *ev = getenv("ALD_SINK_THREADS")) { int k = atoi(ev); if(k >= 1 && k <= 64) nthr = (unsigned)k; }
    else if(n_transcripts < 20000) nthr = 1;
    if(nthr > ALD_TSET_SHARDS) nthr = ALD_TSET_SHARDS;
    return nthr;
}
} unsigned ald_sink_threads(int64_t n_items) { return sink_threads(n_items); } namespace {
const uint32_t ALD_NO_BUCKET = 0xFFFFFFFFu;       // hashes are below 2^31 + 1

// The merge of many graphs' transcripts.  Buckets never interact, so they are dealt to the host threads by hash: every thread walks
// the groups (graphs) in ascending order, builds the per-graph set of ITS buckets and merges it -- the same sequence of
// trans_item::merge calls per bucket as the serial loop of assembler.cc:1105-1133, hence the same result, without `mylock`.
// `make(i, x)` fills x from transcript i; grp[k] .. grp[k+1] are the transcripts of group k; bucket[i] = its chain key or ALD_NO_BUCKET.
struct no_prefetch { void operator()(int64_t) const {} };
template<class Make, class Pre = no_prefetch> void merge_groups(ald_tset *t, unsigned nthr, int64_t n_groups, const int64_t *grp, const int32_t *grp_sid, const uint32_t *bucket, Make make, Pre pre = Pre())
{
    // the transcripts are dealt to their owners first (a counting sort by owner over contiguous ranges, so that every owner's list is in
    // ascending (graph, path) order): an owner then walks its own sixteenth instead of testing every transcript of the batch
    const int64_t nt = n_groups > 0 ? grp[n_groups] : 0;
    auto owner = [&](uint32_t h) { return (unsigned)((h % ALD_TSET_SHARDS) % nthr); };
    std::vector<int64_t> base((size_t)nthr * nthr, 0), first((size_t)nthr + 1, 0);
    HostBatch::run_threads(nthr, [&](unsigned r) {
        int64_t *c = &base[(size_t)r * nthr];
        for(int64_t i = nt * r / nthr; i < nt * (r + 1) / nthr; i++) if(bucket[(size_t)i] != ALD_NO_BUCKET) c[owner(bucket[(size_t)i])]++;
    });
    for(unsigned o = 0; o < nthr; o++) { int64_t run = first[o]; for(unsigned r = 0; r < nthr; r++) { const int64_t c = base[(size_t)r * nthr + o]; base[(size_t)r * nthr + o] = run; run += c; } first[(size_t)o + 1] = run; }
    std::vector<int64_t> order((size_t)first[nthr]);
    HostBatch::run_threads(nthr, [&](unsigned r) {
        int64_t *c = &base[(size_t)r * nthr];
        for(int64_t i = nt * r / nthr; i < nt * (r + 1) / nthr; i++) if(bucket[(size_t)i] != ALD_NO_BUCKET) order[(size_t)c[owner(bucket[(size_t)i])]++] = i;
    });
    HostBatch::run_threads(nthr, [&](unsigned th) {
        aletsch::sink_transcript x;
        int64_t g = 0;
        // the loop is bound by cache misses (source record -> index slot -> table entry): what transcript k + 12 / k + 8 / k + 4 will touch is
        // asked for while transcript k is merged
        const int64_t end = first[(size_t)th + 1];
        static const int dist[3] = { getenv("ALD_SINK_AHEAD_REC") ? atoi(getenv("ALD_SINK_AHEAD_REC")) : 12, getenv("ALD_SINK_AHEAD_SLOT") ? atoi(getenv("ALD_SINK_AHEAD_SLOT")) : 8,
                                     getenv("ALD_SINK_AHEAD_ENTRY") ? atoi(getenv("ALD_SINK_AHEAD_ENTRY")) : 4 };      // tuning knobs (0 = off)
        auto ahead = [&](int64_t k) {
            if(dist[0] > 0 && k + dist[0] < end) pre(order[(size_t)(k + dist[0])]);
            if(dist[1] > 0 && k + dist[1] < end) { const uint32_t h = bucket[(size_t)order[(size_t)(k + dist[1])]]; t->shard[h % ALD_TSET_SHARDS].mt.prefetch_slot(h); }
            if(dist[2] > 0 && k + dist[2] < end) { const uint32_t h = bucket[(size_t)order[(size_t)(k + dist[2])]]; t->shard[h % ALD_TSET_SHARDS].mt.prefetch_entry(h); }
        };
        for(int64_t k = first[th]; k < end; ) {
            while(grp[g + 1] <= order[(size_t)k]) g++;                           // the graph of this owner's next transcript
            int64_t k1 = k; while(k1 < end && order[(size_t)k1] < grp[g + 1]) k1++;
            const int64_t *idx = &order[(size_t)k]; const int cnt = (int)(k1 - k);
            for(int64_t q = k; q < k1; q++) ahead(q);
            k = k1;
            const int s_id = grp_sid ? grp_sid[g] : -1;
            // a graph that puts a single transcript into this thread's tables needs no per-graph set: merging a one-item set is the same
            // as adding the item (transcript_set.cc:149-175)
            if(cnt == 1) { make(idx[0], x); const uint32_t h = bucket[(size_t)idx[0]]; t->shard[h % ALD_TSET_SHARDS].add_hashed(x, h, 1, s_id); continue; }
            // ... and so is merging a set whose items all sit in DIFFERENT buckets: transcript_set::add(set) goes bucket by bucket
            // (transcript_set.cc:156-175), and a bucket that receives one item is zipped exactly as add() would place that item.  Only
            // transcripts of one graph that share a bucket (equal intron chains, single-exon clusters) need the graph's own set first.
            if(cnt <= 16) {
                bool distinct = true;
                for(int a = 1; a < cnt && distinct; a++) for(int q = 0; q < a && distinct; q++) distinct = bucket[(size_t)idx[q]] != bucket[(size_t)idx[a]];
                if(distinct) { for(int q = 0; q < cnt; q++) { make(idx[q], x); const uint32_t h = bucket[(size_t)idx[q]]; t->shard[h % ALD_TSET_SHARDS].add_hashed(x, h, 1, s_id); } continue; }
            }
            aletsch::transcript_sink ts(t->overlap);
            for(int q = 0; q < cnt; q++) {
                make(idx[q], x);
                ts.add_hashed(x, bucket[(size_t)idx[q]], 1, s_id);              // assembler.cc:1120
            }
            t->add(ts);                                                        // assembler.cc:1130 (only tables this thread owns are touched)
        }
    });
}

} // namespace

extern "C" {
int ald_tset_add_batch(ald_tset *t, const ald_batch *b, const int32_t *sid, int64_t tid_base, int32_t skip_single_exon)
{
    if(!t || !b || !b->downloaded) return ALD_ERR_INVALID;
    auto T0 = std::chrono::steady_clock::now();
    const int n = b->hb.n();
    const int64_t np = b->res.n_paths();
    const unsigned nthr = sink_threads(np);
    // pass 1: bucket of every transcript -- the key was taken by the download while it decoded the record (HostResults::build)
    std::vector<uint32_t> bucket((size_t)np, ALD_NO_BUCKET);
    HostBatch::run_threads(nthr, [&](unsigned th) {
        for(int64_t i = np * th / nthr; i < np * (th + 1) / nthr; i++)
            if(!(b->res.n_exon_words[(size_t)i] <= 2 && skip_single_exon)) bucket[(size_t)i] = b->res.chain_key[(size_t)i];      // assembler.cc:1117
    });
    auto T2 = std::chrono::steady_clock::now();
    // pass 2: thread th owns the tables th, th + nthr, ...
    merge_groups(t, nthr, n, b->res.path_begin.data(), sid, bucket.data(), [&](int64_t i, aletsch::sink_transcript &x) {
        const PathRec p = b->res.path((int64_t)((size_t)i));
        x.strand = p.strand; x.coverage = p.coverage; x.top.cov2 = x.coverage; x.top.conf = p.conf; x.top.abd = p.abd; x.top.count1 = p.count; x.count2 = 1;
        x.tid = tid_base + (((int64_t)p.graph << 20) | (int64_t)p.index);
        const int32_t *ex = b->res.exons(p);
        x.xs.assign(ex, ex + p.nexw);
    }, [&](int64_t i) { __builtin_prefetch(b->res.rec(i)); __builtin_prefetch(b->res.rec(i) + 16); });
    if(getenv("ALD_SINK_PROF")) { auto T3 = std::chrono::steady_clock::now(); auto ms = [](auto a, auto b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); }; fprintf(stderr, "[sink] hash pass %.1f ms, merge pass %.1f ms (%u threads)\n", ms(T0, T2), ms(T2, T3), nthr); }
    return ALD_OK;
}

/* ---- finished transcripts as ONE self-contained stream: what ranks exchange in the multi-GPU gather and what a device list merges ----
 * 4-byte words, transcripts in ascending (graph, path index) order:
 *   [graph, path, sid, strand, count1, n_exons, weight f64, conf f64, abd f64, (l, r) * n_exons]        TS_HDR + 2 * n_exons words
 * Records of abandoned attempts and of graphs that did not end well are already gone (HostResults::build), exons are joined. */
enum { TS_HDR = ALD_TS_HDR };
int ald_batch_transcript_stream(const ald_batch *cb, const int32_t *sid, int32_t skip_single_exon, const uint32_t **words, int64_t *n_words)
{
    if(!cb || !words || !n_words) return ALD_ERR_INVALID;
    if(!cb->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_transcript_stream before ald_batch_download");
    ald_batch *b = const_cast<ald_batch*>(cb);
    const int64_t np = b->res.n_paths();
    const unsigned nthr = sink_threads(np);
    std::vector<int64_t> at((size_t)np + 1, 0);
    for(int64_t i = 0; i < np; i++) { const int k = (int)b->res.rec(i)[REC_NEXW]; at[(size_t)i + 1] = at[(size_t)i] + ((k <= 2 && skip_single_exon) ? 0 : TS_HDR + k); }
    b->tstream.resize((size_t)at[(size_t)np] + 2);
    uint32_t *out = b->tstream.data();
    HostBatch::run_threads(nthr, [&](unsigned th) {
        for(int64_t i = np * th / nthr; i < np * (th + 1) / nthr; i++) {
            if(at[(size_t)i + 1] == at[(size_t)i]) continue;
            const PathRec p = b->res.path((int64_t)((size_t)i)); uint32_t *w = out + at[(size_t)i];
            w[0] = (uint32_t)p.graph; w[1] = (uint32_t)p.index; w[2] = (uint32_t)(sid ? sid[p.graph] : -1); w[3] = (uint32_t)(unsigned char)p.strand;
            w[4] = (uint32_t)p.count; w[5] = (uint32_t)(p.nexw / 2);
            memcpy(w + 6, &p.weight, 8); memcpy(w + 8, &p.conf, 8); memcpy(w + 10, &p.abd, 8);
            if(p.nexw) memcpy(w + TS_HDR, b->res.exons(p), 4 * (size_t)p.nexw);
        }
    });
    *words = out; *n_words = at[(size_t)np];
    return ALD_OK;
}

int ald_tset_add_stream(ald_tset *t, const uint32_t *words, int64_t n_words, int32_t graph_offset, int64_t tid_base)
{
    if(!t || n_words < 0 || (n_words > 0 && !words)) return ALD_ERR_INVALID;
    const auto T0 = std::chrono::steady_clock::now();
    // record boundaries and groups (one serial walk: the lengths are in the records)
    std::vector<int64_t> offs, grp; std::vector<int32_t> grp_sid;
    int64_t last_graph = -1;
    for(int64_t o = 0; o < n_words; ) {
        if(o + TS_HDR > n_words) return set_err(ALD_ERR_INVALID, "malformed transcript stream");
        const int64_t len = TS_HDR + 2 * (int64_t)words[o + 5];
        if(o + len > n_words || (int32_t)words[o + 5] < 0) return set_err(ALD_ERR_INVALID, "malformed transcript stream");
        const int64_t g = (int64_t)words[o];
        if(g < last_graph) return set_err(ALD_ERR_INVALID, "transcript stream not in ascending graph order");
        if(g != last_graph) { grp.push_back((int64_t)offs.size()); grp_sid.push_back((int32_t)words[o + 2]); last_graph = g; }
        offs.push_back(o); o += len;
    }
    const int64_t nt = (int64_t)offs.size(); grp.push_back(nt);
    const unsigned nthr = sink_threads(nt);
    const auto T1 = std::chrono::steady_clock::now();
    std::vector<uint32_t> bucket((size_t)nt);
    HostBatch::run_threads(nthr, [&](unsigned th) {
        for(int64_t i = nt * th / nthr; i < nt * (th + 1) / nthr; i++) { const uint32_t *w = words + offs[(size_t)i]; bucket[(size_t)i] = (uint32_t)aletsch::sink_transcript::chain_key((const int32_t*)(w + TS_HDR), 2 * (size_t)w[5]); }
    });
    const auto T2 = std::chrono::steady_clock::now();
    merge_groups(t, nthr, (int64_t)grp_sid.size(), grp.data(), grp_sid.data(), bucket.data(), [&](int64_t i, aletsch::sink_transcript &x) {
        const uint32_t *w = words + offs[(size_t)i];
        double weight, conf, abd; memcpy(&weight, w + 6, 8); memcpy(&conf, w + 8, 8); memcpy(&abd, w + 10, 8);
        x.strand = (char)w[3]; x.coverage = log(1.0 + weight); x.top.cov2 = x.coverage; x.top.conf = conf; x.top.abd = abd; x.top.count1 = (int32_t)w[4]; x.count2 = 1;
        x.tid = tid_base + ((((int64_t)w[0] + graph_offset) << 20) | (int64_t)w[1]);
        x.xs.assign((const int32_t*)(w + TS_HDR), (const int32_t*)(w + TS_HDR) + 2 * (size_t)w[5]);
    }, [&](int64_t i) { __builtin_prefetch(words + offs[(size_t)i]); __builtin_prefetch(words + offs[(size_t)i] + 16); });
    if(getenv("ALD_SINK_PROF")) { const auto T3 = std::chrono::steady_clock::now(); auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        fprintf(stderr, "[sink] stream of %lld transcripts: boundaries %.1f ms, hash pass %.1f ms, merge pass %.1f ms (%u threads)\n", (long long)nt, ms(T0, T1), ms(T1, T2), ms(T2, T3), nthr); }
    return ALD_OK;
}

int ald_tset_merge(ald_tset *dst, ald_tset *src)
{
    if(!dst || !src || dst == src) return ALD_ERR_INVALID;
    for(auto &sh : src->shard) { dst->add(sh); sh.clear(); }
    return ALD_OK;
}

int ald_tset_size(const ald_tset *t, int64_t *n_items, int64_t *n_exons, int64_t *n_samples)
{
    if(!t) return ALD_ERR_INVALID;
    int64_t a = 0, e = 0, s = 0;
    for(auto &sh : t->shard) for(auto &x : sh.mt) for(auto &z : x.second) { a++; e += (int64_t)z.trst.n_exons(); s += (int64_t)z.samples.size(); }
    if(n_items) *n_items = a; if(n_exons) *n_exons = e; if(n_samples) *n_samples = s;
    return ALD_OK;
}

int ald_tset_export(const ald_tset *t, uint64_t *hash, int32_t *count, char *strand, double *coverage, double *cov2, double *conf, double *abd,
                    int32_t *count1, int32_t *count2, int64_t *tid, int64_t *exon_offset, int32_t *exon_lr,
                    int64_t *sample_offset, int32_t *sample_sid, double *sample_cov2, double *sample_conf, double *sample_abd, int32_t *sample_count1)
{
    if(!t || !hash || !count || !strand || !coverage || !cov2 || !conf || !abd || !count1 || !count2 || !tid || !exon_offset || !exon_lr || !sample_offset || !sample_sid || !sample_cov2 || !sample_conf || !sample_abd || !sample_count1) return ALD_ERR_INVALID;
    int64_t i = 0, e = 0, s = 0;
    std::vector<size_t> keys; for(auto &sh : t->shard) for(auto &x : sh.mt) keys.push_back(x.first);
    std::sort(keys.begin(), keys.end());
    for(size_t key : keys) for(auto &z : t->shard[key % ALD_TSET_SHARDS].mt.find(key)->second) {        // the reference's iteration order: ascending hash
        const aletsch::sink_transcript &r = z.trst;
        hash[i] = (uint64_t)key; count[i] = z.count; strand[i] = r.strand; coverage[i] = r.coverage; cov2[i] = r.top.cov2; conf[i] = r.top.conf; abd[i] = r.top.abd;
        count1[i] = r.top.count1; count2[i] = r.count2; tid[i] = r.tid; exon_offset[i] = e; sample_offset[i] = s;
        memcpy(exon_lr + 2 * e, r.xs.data(), 4 * r.xs.size()); e += (int64_t)r.n_exons();
        for(auto &q : z.samples) { sample_sid[s] = q.first; sample_cov2[s] = q.second.top.cov2; sample_conf[s] = q.second.top.conf; sample_abd[s] = q.second.top.abd; sample_count1[s] = q.second.top.count1; s++; }
        i++;
    }
    exon_offset[i] = e; sample_offset[i] = s;
    return ALD_OK;
}

int ald_batch_device_records(const ald_batch *b, const void **device_words, int64_t *n_words)
{
    if(!b || !device_words || !n_words) return ALD_ERR_INVALID;
    if(!b->downloaded) return set_err(ALD_ERR_STATE, "ald_batch_device_records before ald_batch_download");
    *device_words = b->d_pool.p; *n_words = (int64_t)b->res.pool_size();
    return ALD_OK;
}

int ald_records_add_graph_offset(uint32_t *words, int64_t n_words, int32_t graph_offset)
{
    if((!words && n_words > 0) || n_words < 0) return ALD_ERR_INVALID;
    int64_t o = 0;
    while(o + REC_HDR_WORDS <= n_words) {
        words[o] += (uint32_t)graph_offset;
        const int64_t w = (int64_t)rec_words(words[o + 2], words[o + REC_NEXW]);
        if(words[o + 2] < 2 || o + w > n_words) return set_err(ALD_ERR_INVALID, "malformed record stream");
        o += w;
    }
    return ALD_OK;
}

/* raw packed record stream of the last download (what ranks exchange over RCCL in the multi-GPU gather) */
int ald_batch_raw_records(const ald_batch *b, const uint32_t **words, int64_t *n_words)
{
    if(!b || !b->downloaded || !words || !n_words) return ALD_ERR_INVALID;
    *words = b->res.pool_data(); *n_words = (int64_t)b->res.pool_size();
    return ALD_OK;
}

int ald_batch_debug_slab(const ald_batch *b, int32_t cls, const void **launched_with, const void **owned)
{
    if(!b || cls < 0 || cls >= ALD_NUM_CLASSES) return ALD_ERR_INVALID;
    if(launched_with) *launched_with = b->launched_slab[cls];
    if(owned) *owned = b->d_slabs[cls].p;
    return ALD_OK;
}

/* diagnostics used by bench.py / DESIGN.md: class sizes, occupancy and the grid of the last run */
int ald_batch_class_info(ald_batch *b, int32_t cls, int32_t *maxv, int32_t *maxe, int32_t *blocks_per_cu, int32_t *blocks_last_run, int64_t *slab_bytes, int32_t *n_graphs)
{
    if(!b || cls < 0 || cls >= ALD_NUM_CLASSES) return ALD_ERR_INVALID;
    ClassInfo ci = class_info(cls);
    if(maxv) *maxv = ci.maxv; if(maxe) *maxe = ci.maxe; if(slab_bytes) *slab_bytes = (int64_t)ci.slab_bytes;
    if(blocks_per_cu) { hipSetDevice(b->device); *blocks_per_cu = occupancy_for(b, cls); }
    if(blocks_last_run) *blocks_last_run = b->blocks[cls];
    if(n_graphs) { int k = 0; for(int c : b->cls) if(c == cls) k++; *n_graphs = k; }
    return ALD_OK;
}

} // extern "C"
