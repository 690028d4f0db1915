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
    ncclComm_t comm = nullptr; int world = 1, rank = 0, device = 0; hipStream_t stream = nullptr;
    DevBuf d_send, d_recv, d_sizes; PinBuf h_recv;
    std::vector<int64_t> offsets; std::vector<int32_t> goffs;
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
    if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { R.CommDestroy(c->comm); delete c; return ald_set_err(ALD_ERR_HIP, "stream creation failed"); }
    *out = c;
    return ALD_OK;
}

int ald_comm_destroy(ald_comm *c)
{
    if(!c) return ALD_OK;
    hipSetDevice(c->device);
    if(c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if(c->comm) rccl().CommDestroy(c->comm);
    c->d_send.release(); c->d_recv.release(); c->d_sizes.release(); c->h_recv.release();
    delete c;
    return ALD_OK;
}

/* every rank: its stream + the global id of its first graph; rank 0 gets all streams back to back in rank order (valid until the next
 * call on this communicator), offsets[world + 1] into them, and every rank's graph offset */
int ald_comm_gather_streams(ald_comm *c, const uint32_t *words, int64_t n_words, int32_t graph_offset,
                            const uint32_t **all_words, const int64_t **offsets, const int32_t **graph_offsets)
{
    if(!c || n_words < 0 || (n_words > 0 && !words)) return ALD_ERR_INVALID;
    Rccl &R = rccl();
    HCHK(hipSetDevice(c->device));
    const int W = c->world;
    // sizes and graph offsets of every rank: one (n_words, graph_offset) pair each
    if(c->d_sizes.ensure(16 * (size_t)(W + 1))) return ald_set_err(ALD_ERR_NOMEM, "size exchange buffer");
    int64_t mine[2] = {n_words, (int64_t)graph_offset};
    HCHK(hipMemcpyAsync(c->d_sizes.p, mine, 16, hipMemcpyHostToDevice, c->stream));
    NCHK(R.AllGather(c->d_sizes.p, (char*)c->d_sizes.p + 16, 2, ncclInt64, c->comm, c->stream));
    std::vector<int64_t> all(2 * (size_t)W);
    HCHK(hipMemcpyAsync(all.data(), (char*)c->d_sizes.p + 16, 16 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    c->offsets.assign((size_t)W + 1, 0); c->goffs.assign((size_t)W, 0);
    for(int r = 0; r < W; r++) { c->offsets[(size_t)r + 1] = c->offsets[(size_t)r] + all[2 * (size_t)r]; c->goffs[(size_t)r] = (int32_t)all[2 * (size_t)r + 1]; }
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
    const int64_t total = c->offsets[(size_t)W];
    if(c->rank == 0) { if(c->d_recv.ensure(4 * (size_t)total + 64) || c->h_recv.ensure(4 * (size_t)total + 64)) return ald_set_err(ALD_ERR_NOMEM, "receive buffers"); }
    // the grouped section never returns between GroupStart and GroupEnd: a failed Send / Recv is remembered, the group is closed (the
    // calling thread must not stay in group mode: every later RCCL call on it would be deferred) and the stream drained -- it still
    // holds the staging copy from the caller's buffer -- before the error goes back
    {
        NCHK(R.GroupStart());
        ncclResult_t bad = ncclSuccess; const char *what = "";
        if(n_words) { const ncclResult_t r_ = R.Send(src, (size_t)n_words, ncclUint32, 0, c->comm, c->stream); if(r_ != ncclSuccess) { bad = r_; what = "ncclSend"; } }
        if(c->rank == 0) for(int r = 0; r < W && bad == ncclSuccess; r++) {
            const int64_t k = all[2 * (size_t)r]; if(!k) continue;
            const ncclResult_t r_ = R.Recv((uint32_t*)c->d_recv.p + c->offsets[(size_t)r], (size_t)k, ncclUint32, r, c->comm, c->stream);
            if(r_ != ncclSuccess) { bad = r_; what = "ncclRecv"; }
        }
        const ncclResult_t ge = R.GroupEnd();
        if(bad != ncclSuccess || ge != ncclSuccess) {
            (void)hipStreamSynchronize(c->stream);
            return ald_set_err(ALD_ERR_HIP, std::string(bad != ncclSuccess ? what : "ncclGroupEnd") + ": " + R.GetErrorString(bad != ncclSuccess ? bad : ge));
        }
    }
    if(c->rank == 0 && total) HCHK(hipMemcpyAsync(c->h_recv.p, c->d_recv.p, 4 * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    if(all_words) *all_words = c->rank == 0 ? (const uint32_t*)c->h_recv.p : nullptr;
    if(offsets) *offsets = c->offsets.data();
    if(graph_offsets) *graph_offsets = c->goffs.data();
    return ALD_OK;
}

} // extern "C"
