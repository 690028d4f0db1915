// ald_internal.h -- what the translation units of the host library share: the batch object behind the opaque `ald_batch` handle.
// Not installed, not part of the ABI (include/aletsch_decomp.h is).
#pragma once
#include <hip/hip_runtime.h>
#include "host_pack.h"
#include "../host/transcript_sink.hpp"
#include <string>
#include <vector>

using namespace ald;

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) { if(bytes <= cap) return 0; if(p) hipFree(p); p = nullptr; cap = 0; size_t want = bytes + bytes / 4 + 256; if(hipMalloc(&p, want) != hipSuccess) return -1; cap = want; return 0; }
    void release() { if(p) hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    // landing: a buffer the copy engine fills and the CPU then READS a lot (record parsing, exon gathers): non-coherent pinned memory is
    // cacheable on the CPU side; visibility is given by the stream synchronisation every reader does first
    int ensure(size_t bytes, bool landing = false) { if(bytes <= cap) return 0; if(p) hipHostFree(p); p = nullptr; cap = 0; size_t want = bytes + bytes / 4 + 256;
        const unsigned flags = landing && !getenv("ALD_PIN_COHERENT") ? hipHostMallocNonCoherent : hipHostMallocDefault;
        if(hipHostMalloc(&p, want, flags) != hipSuccess) return -1; cap = want; return 0; }
    void release() { if(p) hipHostFree(p); p = nullptr; cap = 0; }
};

enum { ALD_SIDE_STREAMS = 6, ALD_SIDE_STREAMS_MAX = 8 };      // five for the large classes + one for classes 0..2 (stage_pass): cfg3 88-89 ms against 93 with three (profiles/r04/x_cfg3_side_streams.txt); a batch of one class (the bench) uses one
// kernel slots of a pass: slot c = the plain build of size class c (staged graphs), slot ALD_NUM_CLASSES + c = its raw build (graphs whose
// pre-steps run on the device); a batch without raw graphs only ever uses the first half
enum { ALD_NUM_SLOTS = 2 * ALD_NUM_CLASSES };     // default / upper bound of the side streams the classes of a pass are dealt to
// one pass of a batch, staged: work lists, kernel arguments, grid sizes, stream assignment (see stage_pass)
struct StagedPass {
    std::vector<int32_t> flat; std::vector<KernelArgs> args;
    int nblk[ALD_NUM_SLOTS]; int order[ALD_NUM_SLOTS]; int stream_of[ALD_NUM_SLOTS]; int nord = 0; size_t tot = 0;
};

struct ald_batch {
    int device = 0; int n_cus = 0;
    Params prm;
    HostBatch hb{true};                             // its arrays in pinned memory: ald_batch_upload copies them to the device as they are
    HostBatch::Section sec[HostBatch::S_COUNT];
    uint64_t in_bytes = 0;
    hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // ald_batch_upload's copies go through a stream of another PRIORITY class: the runtime maps the streams of one priority onto a small
    // pool of hardware queues (GPU_MAX_HW_QUEUES), a queue runs its packets in order, and a batch's 34 input copies (25 ms of DMA) in front
    // of ANOTHER batch's kernel launch on the same queue delayed that kernel by 3-4 ms per step of a pipelined caller
    // (profiles/r04/zc_h2d_inclusive_step.txt).  Streams of a different priority have queues of their own.
    hipStream_t up_stream = nullptr;
    // the size classes run concurrently on a few side streams.  Not one per class: a process only gets a handful of hardware queues
    // (4 by default) and streams beyond that share them in creation order, which can put the two heaviest classes behind each other
    hipStream_t cstream[ALD_SIDE_STREAMS_MAX] = {}; int n_cstream = ALD_SIDE_STREAMS;
    hipEvent_t cdone[ALD_NUM_SLOTS] = {};
    PinBuf pin_in, pin_out, pin_small, pin_index;          // wire buffer / record landing area / status + counters landing area / the result index
    DevBuf d_in, d_status, d_npaths, d_niters, d_pool, d_poolused, d_trace_n, d_trace_codes, d_trace_vals, d_work, d_counter, d_args;
    DevBuf d_index, d_gfirst;                              // result index written by the kernel: index[graph_first[g] + p] = pool offset of record (g, p)
    DevBuf d_pbegin, d_ordoff;                             // the same in (graph, path) order, built on the device on demand (tset_reduce.hip: device_path_table)
    uint64_t index_cap = 0; int64_t total_paths = 0; bool paths_on_device = false;
    double dl_ms[4] = {0, 0, 0, 0}; int64_t dl_bytes = 0;   // last download: waiting for the kernel / status + retries / D2H copies / decode (diagnostics)
    DevBuf d_slabs[ALD_NUM_SLOTS];
    int blocks[ALD_NUM_SLOTS] = {};
    int occ[ALD_NUM_SLOTS]; ald_batch() { for(int c = 0; c < ALD_NUM_SLOTS; c++) occ[c] = -1; }
    StagedPass *pass0 = nullptr; bool pass0_on_device = false; std::vector<int32_t> cls0;      // first pass of the uploaded batch, staged at upload time
    uint64_t pool_cap_words = 0;
    int trace_cap = 0;
    bool uploaded = false, ran = false, downloaded = false;
    double kernel_ms = -1;
    // per-graph scheduling state
    std::vector<int32_t> cls, attempt, status, n_paths, n_iters;
    std::vector<int32_t> trace_n, trace_codes; std::vector<double> trace_vals;
    HostResults res;
    int passes = 0;
    const void *launched_slab[ALD_NUM_SLOTS] = {};      // test hook (ald_batch_debug_slab)
    rvec<uint32_t> tstream;                                // last transcript stream built from this batch (ald_batch_transcript_stream)
    DevBuf red[20]; PinBuf red_pin[8];                     // scratch of ald_batch_reduce_transcripts, kept across calls (tset_reduce.hip)
    DevBuf dts[3];                                         // ald_batch_device_transcript_stream: lengths / offsets / the stream itself
};


// The result sink behind ald_tset_*.  Buckets (intron-chain hashes) never interact: the set is kept as NSHARD independent tables, bucket h
// in table h % NSHARD, so that a whole batch can be merged by NSHARD host threads without a lock; the export walks all keys in ascending order.
enum { ALD_TS_HDR = 12 };          // words in front of a transcript's exons in a transcript stream (ald_batch_transcript_stream)
enum { ALD_TSET_SHARDS = 16 };
struct ald_tset {
    std::vector<aletsch::transcript_sink> shard; double overlap;
    explicit ald_tset(double ov) : shard(ALD_TSET_SHARDS, aletsch::transcript_sink(ov)), overlap(ov) {}
    void add(aletsch::transcript_sink &ts) { for(auto &x : ts.mt) shard[x.first % ALD_TSET_SHARDS].add_bucket(x.first, x.second); }
};

unsigned ald_sink_threads(int64_t n_items);          // host threads for a merge of n items (ALD_SINK_THREADS overrides; at most one per table)
int ald_ensure_index(const ald_batch *b);                 // per-graph index over the record stream of the last download (built on first use)
int ald_set_err(int code, const std::string &msg);        // sets the calling thread's ald_last_error() text, returns `code`
