// mock_rccl.cc -- TEST-ONLY stand-in for librccl.so, loaded through ALD_RCCL_LIB by tests/host_adapter/comm_ranks_test.cc.
//
// RCCL refuses two ranks on one device and the GPU box has one, so the multi-rank branch of ald_comm_gather_streams (sizes of W ranks,
// rank 0's receive offsets, the grouped Send / Recv, graph offsets) could never execute there.  This library implements the nine
// nccl* entry points comm_rccl.cpp binds, for ranks that are THREADS of one process sharing one GPU: a communicator is a slot in a
// process-wide table keyed by the unique id; AllGather and the grouped Send / Recv move bytes with hipMemcpyAsync between the ranks'
// device buffers once every rank has arrived (a barrier per collective).  It also lets a test inject a failing ncclSend (environment
// ALD_MOCK_RCCL_FAIL_SEND=<rank>) to drive the error path that must close the group.  Never linked into the product library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; hipStream_t st; };
struct World {
    int n = 0, arrived = 0, gen = 0; std::mutex m; std::condition_variable cv;
    std::vector<const void*> ag_src; std::vector<hipStream_t> ag_st;
    std::vector<std::vector<Op>> ops;                 // per rank: the ops of the group being closed
    int joined = 0;
    void barrier() { std::unique_lock<std::mutex> l(m); const int g = gen; if(++arrived == n) { arrived = 0; gen++; cv.notify_all(); } else cv.wait(l, [&] { return gen != g; }); }
};
struct Comm { World *w; int rank; };
std::mutex g_m; std::map<unsigned long long, World*> g_worlds; unsigned long long g_next = 1;
thread_local int t_group = 0; thread_local std::vector<std::pair<Comm*, Op>> t_pending; thread_local Comm *t_failed = nullptr;      // t_failed: a Send of this group was refused
thread_local Comm *t_comm = nullptr;        // the communicator this thread (= this rank) works with: a rank with nothing to send still joins the rendezvous
size_t elt(ncclDataType_t t) { return (t == ncclInt64 || t == ncclUint64 || t == ncclFloat64) ? 8 : (t == ncclInt8 || t == ncclUint8) ? 1 : 4; }
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { std::lock_guard<std::mutex> l(g_m); memset(id, 0, sizeof(*id)); const unsigned long long k = g_next++; memcpy(id, &k, 8); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank)
{
    unsigned long long k; memcpy(&k, &id, 8);
    World *w;
    { std::lock_guard<std::mutex> l(g_m); World *&slot = g_worlds[k]; if(!slot) { slot = new World(); slot->n = n; slot->ag_src.resize(n); slot->ag_st.resize(n); slot->ops.resize(n); } w = slot; }
    if(w->n != n || rank < 0 || rank >= n) return ncclInvalidArgument;
    *comm = (ncclComm_t)new Comm{w, rank}; t_comm = (Comm*)*comm;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (Comm*)c; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInternalError ? "mock: injected failure" : "mock: error"); }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t cc, hipStream_t st)
{
    Comm *c = (Comm*)cc; World *w = c->w; const size_t bytes = count * elt(t);
    if(hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;       // the caller's H2D of its contribution is on this stream
    w->ag_src[c->rank] = send; w->ag_st[c->rank] = st;
    w->barrier();
    for(int r = 0; r < w->n; r++) if(hipMemcpyAsync((char*)recv + (size_t)r * bytes, w->ag_src[r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    if(hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    w->barrier();                                                                    // nobody reuses its source before every rank has read it
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { t_group++; return ncclSuccess; }
ncclResult_t ncclSend(const void *src, size_t count, ncclDataType_t t, int peer, ncclComm_t cc, hipStream_t st)
{
    Comm *c = (Comm*)cc;
    if(const char *e = getenv("ALD_MOCK_RCCL_FAIL_SEND")) if(atoi(e) == c->rank) { t_failed = c; return ncclInternalError; }
    if(t_group <= 0) return ncclInvalidUsage;
    t_pending.push_back({c, Op{true, src, nullptr, count * elt(t), peer, st}});
    return ncclSuccess;
}
ncclResult_t ncclRecv(void *dst, size_t count, ncclDataType_t t, int peer, ncclComm_t cc, hipStream_t st)
{
    if(t_group <= 0) return ncclInvalidUsage;
    t_pending.push_back({(Comm*)cc, Op{false, nullptr, dst, count * elt(t), peer, st}});
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if(t_group <= 0) return ncclInvalidUsage;
    if(--t_group > 0) return ncclSuccess;
    // every rank closes one group per gather, so every rank meets the others here -- also one with nothing to send (an empty stream)
    // and one whose Send was refused (nothing to offer: its peers then find no matching send and fail as well instead of waiting for ever)
    Comm *c = t_failed ? t_failed : (t_pending.empty() ? t_comm : t_pending.front().first);
    if(!c) return ncclSuccess;
    World *w = c->w;
    if(t_failed) { t_pending.clear(); t_failed = nullptr; }
    for(auto &p : t_pending) if(p.second.send && hipStreamSynchronize(p.second.st) != hipSuccess) return ncclUnhandledCudaError;    // staged payload is complete
    { std::lock_guard<std::mutex> l(w->m); w->ops[c->rank].clear(); for(auto &p : t_pending) w->ops[c->rank].push_back(p.second); }
    t_pending.clear();
    w->barrier();
    ncclResult_t rc = ncclSuccess;
    for(const Op &o : w->ops[c->rank]) {
        if(o.send) continue;
        const Op *match = nullptr;
        for(const Op &s : w->ops[o.peer]) if(s.send && s.peer == c->rank && s.bytes == o.bytes) { match = &s; break; }
        if(!match) { rc = ncclInvalidUsage; continue; }                              // a receive nobody sends to, or of another size: the offsets are wrong
        if(hipMemcpyAsync(o.dst, match->src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    for(const Op &o : w->ops[c->rank]) if(!o.send && hipStreamSynchronize(o.st) != hipSuccess) rc = ncclUnhandledCudaError;
    w->barrier();
    return rc;
}
int mock_rccl_thread_in_group() { return t_group; }       // test hook: is the calling thread still inside an open group?
}
