// comm_ranks_test.cc -- GPU tier: W ranks of ald_comm_gather_streams as THREADS of one process on one GPU, RCCL replaced by
// tests/host_adapter/mock_rccl.cc (ALD_RCCL_LIB).  Checks what only a multi-rank run can: the size exchange of W ranks, rank 0's
// receive offsets, every payload at its place (host-staged and device-resident sources, empty streams), the graph offsets -- and that
// a failing ncclSend leaves the calling thread OUT of group mode with the error reported (comm_rccl.cpp error path).
//   usage: comm_ranks_test <world> [fail_rank]
#include "aletsch_decomp.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static std::vector<uint32_t> stream_of(int rank, int step)      // a recognisable payload of rank-dependent size (rank 2 of step 1 sends nothing)
{
    size_t n = (step == 1 && rank == 2) ? 0 : 1000 + 137 * (size_t)rank + 11 * (size_t)step;
    std::vector<uint32_t> v(n);
    for(size_t i = 0; i < n; i++) v[i] = (uint32_t)(rank * 1000003u + step * 7919u + i);
    return v;
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 2; const int fail_rank = argc > 2 ? atoi(argv[2]) : -1;
    uint8_t id[128];
    if(ald_comm_unique_id(id) != ALD_OK) { fprintf(stderr, "unique id: %s\n", ald_last_error()); return 2; }
    std::vector<int> rc(W, 0);
    std::vector<std::thread> th;
    for(int r = 0; r < W; r++) th.emplace_back([&, r]() {
        ald_comm *c = nullptr;
        if(ald_comm_create(id, W, r, 0, &c) != ALD_OK) { fprintf(stderr, "rank %d create: %s\n", r, ald_last_error()); rc[r] = 3; return; }
        for(int step = 0; step < 3 && rc[r] == 0; step++) {
            std::vector<uint32_t> mine = stream_of(r, step);
            const uint32_t *src = mine.data(); void *dsrc = nullptr;
            if(step == 2 && !mine.empty()) {                      // last step: the stream already lives in HBM and is sent from there
                if(hipMalloc(&dsrc, 4 * mine.size()) != hipSuccess || hipMemcpy(dsrc, mine.data(), 4 * mine.size(), hipMemcpyHostToDevice) != hipSuccess) { rc[r] = 4; break; }
                src = (const uint32_t*)dsrc;
            }
            const uint32_t *all = nullptr; const int64_t *offs = nullptr; const int32_t *goffs = nullptr;
            // steps 0 / 2: the one-call form; step 1: the two halves (begin returns with the data still on its way, rank 0 waits stream by stream)
            int e;
            if(step != 1) e = ald_comm_gather_streams(c, src, (int64_t)mine.size(), 1000 * r + step, &all, &offs, &goffs);
            else {
                e = ald_comm_gather_begin(c, src, (int64_t)mine.size(), 1000 * r + step);
                if(e == ALD_OK && r == 0) for(int q = 0; q < W && e == ALD_OK; q++) e = ald_comm_gather_wait(c, q, &all, &offs, &goffs);
                else if(e == ALD_OK) e = ald_comm_gather_wait(c, -1, &all, &offs, &goffs);
            }
            if(fail_rank >= 0) {
                // the injected failure: the failing rank must get an error AND be out of group mode; the others must not hang
                void *h = dlopen(getenv("ALD_RCCL_LIB"), RTLD_NOW | RTLD_NOLOAD);
                int (*in_group)() = h ? (int (*)())dlsym(h, "mock_rccl_thread_in_group") : nullptr;
                if(r == fail_rank && (e == ALD_OK || !in_group || in_group() != 0)) { fprintf(stderr, "rank %d: failed send not handled (rc=%d, in_group=%d)\n", r, e, in_group ? in_group() : -1); rc[r] = 5; }
                if(dsrc) (void)hipFree(dsrc);
                break;
            }
            if(e != ALD_OK) { fprintf(stderr, "rank %d step %d gather: %s\n", r, step, ald_last_error()); rc[r] = 6; if(dsrc) (void)hipFree(dsrc); break; }
            int64_t at = 0;
            for(int q = 0; q < W && rc[r] == 0; q++) {
                std::vector<uint32_t> want = stream_of(q, step);
                if(offs[q] != at || goffs[q] != 1000 * q + step) { fprintf(stderr, "rank %d step %d: offsets[%d] = %lld (want %lld), graph offset %d\n", r, step, q, (long long)offs[q], (long long)at, goffs[q]); rc[r] = 7; }
                if(r == 0 && rc[r] == 0 && memcmp(all + at, want.data(), 4 * want.size()) != 0) { fprintf(stderr, "rank 0 step %d: payload of rank %d is wrong\n", step, q); rc[r] = 8; }
                at += (int64_t)want.size();
            }
            if(rc[r] == 0 && offs[W] != at) rc[r] = 9;
            if(r != 0 && all != nullptr) rc[r] = 10;
            if(dsrc) (void)hipFree(dsrc);
        }
        ald_comm_destroy(c);
    });
    for(auto &t : th) t.join();
    int bad = 0; for(int r = 0; r < W; r++) if(rc[r]) { fprintf(stderr, "rank %d: failure code %d\n", r, rc[r]); bad = 1; }
    if(!bad) printf("COMM_RANKS_OK world=%d%s\n", W, fail_rank >= 0 ? " (injected send failure handled)" : "");
    return bad;
}
