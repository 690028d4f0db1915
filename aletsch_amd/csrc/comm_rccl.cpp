// comm_rccl.cpp -- the path's one exchange step for a MULTI-PROCESS host (one process per MI355X), behind the C ABI: every rank's
// finished-transcript stream (ald_batch_transcript_stream) travels to rank 0 over RCCL / xGMI, where ald_tset_add_stream merges the
// streams in rank order == ascending global graph id.  The analogue of the reference's `tm.add(ts)` under `mylock`
// (meta/assembler.cc:1127-1132), once per batch instead of once per graph.
//
// RCCL is loaded on first use (dlopen of librccl.so, RTLD_LOCAL): a host that runs all its devices in ONE process
// (aletsch::gpu_assembly_queue over a device list) never needs it, and a Python test process that already carries torch's own RCCL is
// not handed a second copy at load time.  Bootstrap is the caller's: rank 0 makes the 128-byte id (ald_comm_unique_id) and ships it
// to the other ranks by whatever it has (a file, a pipe, MPI, the reference's own thread pool has no such thing).
//
// Exchange = sizes by ncclAllGather (one int64 per rank), payloads by grouped ncclSend / ncclRecv to rank 0 -- point-to-point, which
// is what xGMI is; payload per rank is ~0.2 GB at 125 k graphs, i.e. ~1.4 ms per link.
#include "ald_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>

namespace {

struct Rccl {
    void *h = nullptr; bool ok = false; std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl &rccl()
{
    static Rccl R; static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("ALD_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for(const char *n : names) { if(!n) continue; R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if(R.h) break; }
        if(!R.h) { R.err = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
#define ALD_SYM(field, name) R.field = (decltype(R.field))dlsym(R.h, name); if(!R.field) { R.err = std::string("librccl.so lacks ") + name; return; }
        ALD_SYM(GetUniqueId, "ncclGetUniqueId") ALD_SYM(CommInitRank, "ncclCommInitRank") ALD_SYM(CommDestroy, "ncclCommDestroy") ALD_SYM(AllGather, "ncclAllGather")
        ALD_SYM(Send, "ncclSend") ALD_SYM(Recv, "ncclRecv") ALD_SYM(GroupStart, "ncclGroupStart") ALD_SYM(GroupEnd, "ncclGroupEnd") ALD_SYM(GetErrorString, "ncclGetErrorString")
#undef ALD_SYM
        R.ok = true;
    });
    return R;
}
#define NCHK(x) do { ncclResult_t r_ = (x); if(r_ != ncclSuccess) return ald_set_err(ALD_ERR_HIP, std::string(#x) + ": " + rccl().GetErrorString(r_)); } while(0)
#define HCHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) return ald_set_err(ALD_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while(0)

} // namespace

struct ald_comm {
    ncclComm_t comm = nullptr; int world = 1, rank = 0, device = 0; hipStream_t stream = nullptr, copy_stream = nullptr;
    DevBuf d_send, d_sizes;
    // two sets of receive buffers (rank 0): while the host merges what gather k brought, gather k + 1 receives and copies into the other set
    struct Set { DevBuf d_recv; PinBuf h_recv; std::vector<int64_t> offsets; std::vector<int32_t> goffs; std::vector<hipEvent_t> landed; hipEvent_t received = nullptr; bool open = false; } set[2];
    int cur = 1;                                   // the set of the gather begun last
};

extern "C" {

int ald_comm_unique_id(uint8_t id[128])
{
    if(!id) return ALD_ERR_INVALID;
    Rccl &R = rccl(); if(!R.ok) return ald_set_err(ALD_ERR_NO_DEVICE, R.err);
    ncclUniqueId u; NCHK(R.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return ALD_OK;
}

int ald_comm_create(const uint8_t id[128], int32_t world, int32_t rank, int32_t device, ald_comm **out)
{
    if(!id || !out || world < 1 || rank < 0 || rank >= world) return ALD_ERR_INVALID;
    Rccl &R = rccl(); if(!R.ok) return ald_set_err(ALD_ERR_NO_DEVICE, R.err);
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return ald_set_err(ALD_ERR_NO_DEVICE, "no such HIP device");
    HCHK(hipSetDevice(device));
    ald_comm *c = new ald_comm(); c->world = world; c->rank = rank; c->device = device;
    ncclUniqueId u; memcpy(&u, id, 128);
    ncclResult_t r = R.CommInitRank(&c->comm, world, u, rank);
    if(r != ncclSuccess) { delete c; return ald_set_err(ALD_ERR_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(r)); }
    if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { R.CommDestroy(c->comm); delete c; return ald_set_err(ALD_ERR_HIP, "stream creation failed"); }
    for(auto &st : c->set) { if(hipEventCreateWithFlags(&st.received, hipEventDisableTiming) != hipSuccess) { R.CommDestroy(c->comm); delete c; return ald_set_err(ALD_ERR_HIP, "event creation failed"); } }
    *out = c;
    return ALD_OK;
}

int ald_comm_destroy(ald_comm *c)
{
    if(!c) return ALD_OK;
    hipSetDevice(c->device);
    if(c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if(c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    if(c->comm) rccl().CommDestroy(c->comm);
    c->d_send.release(); c->d_sizes.release();
    for(auto &st : c->set) { st.d_recv.release(); st.h_recv.release(); for(hipEvent_t e : st.landed) hipEventDestroy(e); if(st.received) hipEventDestroy(st.received); }
    delete c;
    return ALD_OK;
}

/* The collective in two halves.  ald_comm_gather_begin: every rank passes its stream + the global id of its first graph; the sizes are
 * exchanged (one small synchronous round), the payloads are ENQUEUED -- grouped Send / Recv to rank 0 on the communicator's stream -- and
 * on rank 0 the copy of every received stream to pinned host memory is enqueued behind them on a second stream, one copy per rank with an
 * event of its own: the call returns while the data is still on its way.  ald_comm_gather_wait(upto): blocks until the streams of ranks
 * 0..upto (-1: all; a rank other than 0: until its own send has left) have landed, then hands out the pointers.  Rank 0 can so merge
 * rank r's stream while rank r + 1's is still being copied, and -- the buffers exist twice -- begin gather k + 1 before it has merged
 * gather k: what a wait handed out stays valid until the SECOND next begin on this communicator. */
int ald_comm_gather_begin(ald_comm *c, const uint32_t *words, int64_t n_words, int32_t graph_offset)
{
    if(!c || n_words < 0 || (n_words > 0 && !words)) return ALD_ERR_INVALID;
    Rccl &R = rccl();
    HCHK(hipSetDevice(c->device));
    const int W = c->world;
    c->cur ^= 1; ald_comm::Set &S = c->set[c->cur];
    // sizes and graph offsets of every rank: one (n_words, graph_offset) pair each
    if(c->d_sizes.ensure(16 * (size_t)(W + 1))) return ald_set_err(ALD_ERR_NOMEM, "size exchange buffer");
    int64_t mine[2] = {n_words, (int64_t)graph_offset};
    HCHK(hipMemcpyAsync(c->d_sizes.p, mine, 16, hipMemcpyHostToDevice, c->stream));
    NCHK(R.AllGather(c->d_sizes.p, (char*)c->d_sizes.p + 16, 2, ncclInt64, c->comm, c->stream));
    std::vector<int64_t> all(2 * (size_t)W);
    HCHK(hipMemcpyAsync(all.data(), (char*)c->d_sizes.p + 16, 16 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    S.offsets.assign((size_t)W + 1, 0); S.goffs.assign((size_t)W, 0);
    for(int r = 0; r < W; r++) { S.offsets[(size_t)r + 1] = S.offsets[(size_t)r] + all[2 * (size_t)r]; S.goffs[(size_t)r] = (int32_t)all[2 * (size_t)r + 1]; }
    // payloads: every rank sends, rank 0 receives each stream at its offset
    // a stream that already lives in HBM (ald_batch_device_transcript_stream) is sent from where it is; a host stream is staged first
    const void *src = words;
    {
        hipPointerAttribute_t at; bool on_device = false;
        if(n_words && hipPointerGetAttributes(&at, words) == hipSuccess) on_device = (at.type == hipMemoryTypeDevice);
        (void)hipGetLastError();                                   // (a plain host pointer makes the query fail: not an error here)
        if(!on_device) {
            if(c->d_send.ensure(4 * (size_t)n_words + 64)) return ald_set_err(ALD_ERR_NOMEM, "send buffer");
            if(n_words) HCHK(hipMemcpyAsync(c->d_send.p, words, 4 * (size_t)n_words, hipMemcpyHostToDevice, c->stream));
            src = c->d_send.p;
        }
    }
    const int64_t total = S.offsets[(size_t)W];
    if(c->rank == 0) {
        if(S.d_recv.ensure(4 * (size_t)total + 64) || S.h_recv.ensure(4 * (size_t)total + 64)) return ald_set_err(ALD_ERR_NOMEM, "receive buffers");
        while((int)S.landed.size() < W) { hipEvent_t e; HCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); S.landed.push_back(e); }
    }
    // the grouped section never returns between GroupStart and GroupEnd: a failed Send / Recv is remembered, the group is closed (the
    // calling thread must not stay in group mode: every later RCCL call on it would be deferred) and the stream drained -- it still
    // holds the staging copy from the caller's buffer -- before the error goes back
    {
        NCHK(R.GroupStart());
        ncclResult_t bad = ncclSuccess; const char *what = "";
        if(n_words) { const ncclResult_t r_ = R.Send(src, (size_t)n_words, ncclUint32, 0, c->comm, c->stream); if(r_ != ncclSuccess) { bad = r_; what = "ncclSend"; } }
        if(c->rank == 0) for(int r = 0; r < W && bad == ncclSuccess; r++) {
            const int64_t k = all[2 * (size_t)r]; if(!k) continue;
            const ncclResult_t r_ = R.Recv((uint32_t*)S.d_recv.p + S.offsets[(size_t)r], (size_t)k, ncclUint32, r, c->comm, c->stream);
            if(r_ != ncclSuccess) { bad = r_; what = "ncclRecv"; }
        }
        const ncclResult_t ge = R.GroupEnd();
        if(bad != ncclSuccess || ge != ncclSuccess) {
            (void)hipStreamSynchronize(c->stream);
            return ald_set_err(ALD_ERR_HIP, std::string(bad != ncclSuccess ? what : "ncclGroupEnd") + ": " + R.GetErrorString(bad != ncclSuccess ? bad : ge));
        }
    }
    HCHK(hipEventRecord(S.received, c->stream));
    if(c->rank == 0) {
        // behind the receives, on the copy stream: one D2H per rank, in rank order, an event after each -- the exchange stream is free for
        // the next gather's size round at once
        HCHK(hipStreamWaitEvent(c->copy_stream, S.received, 0));
        for(int r = 0; r < W; r++) {
            const int64_t k = all[2 * (size_t)r];
            if(k) HCHK(hipMemcpyAsync((uint32_t*)S.h_recv.p + S.offsets[(size_t)r], (const uint32_t*)S.d_recv.p + S.offsets[(size_t)r], 4 * (size_t)k, hipMemcpyDeviceToHost, c->copy_stream));
            HCHK(hipEventRecord(S.landed[(size_t)r], c->copy_stream));
        }
    }
    S.open = true;
    return ALD_OK;
}

int ald_comm_gather_wait(ald_comm *c, int32_t upto, const uint32_t **all_words, const int64_t **offsets, const int32_t **graph_offsets)
{
    if(!c) return ALD_ERR_INVALID;
    ald_comm::Set &S = c->set[c->cur];
    if(!S.open) return ald_set_err(ALD_ERR_STATE, "ald_comm_gather_wait without a gather in flight");
    HCHK(hipSetDevice(c->device));
    const int W = c->world;
    if(c->rank == 0) { const int last = (upto < 0 || upto >= W) ? W - 1 : upto; HCHK(hipEventSynchronize(S.landed[(size_t)last])); }
    else HCHK(hipEventSynchronize(S.received));
    if(all_words) *all_words = c->rank == 0 ? (const uint32_t*)S.h_recv.p : nullptr;
    if(offsets) *offsets = S.offsets.data();
    if(graph_offsets) *graph_offsets = S.goffs.data();
    return ALD_OK;
}

/* both halves in one call: rank 0 gets all streams back to back in rank order, offsets[world + 1] into them, and every rank's graph offset */
int ald_comm_gather_streams(ald_comm *c, const uint32_t *words, int64_t n_words, int32_t graph_offset,
                            const uint32_t **all_words, const int64_t **offsets, const int32_t **graph_offsets)
{
    const int rc = ald_comm_gather_begin(c, words, n_words, graph_offset);
    if(rc != ALD_OK) return rc;
    return ald_comm_gather_wait(c, -1, all_words, offsets, graph_offsets);
}

} // extern "C"
