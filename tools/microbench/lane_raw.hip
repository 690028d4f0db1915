// lane_raw.hip -- is a global-memory store by one lane of a wave visible to a load by ANOTHER lane of the same wave that is issued after it,
// without a wait on the memory counters in between?  (diagnostic of round 4: it ruled the memory system out as the cause of a fault that turned out to be spill placement -- decomp_common.h, wsync(); profiles/r04/zb_*)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lane_raw.hip -o build/lane_raw && build/lane_raw
// One wave per workgroup, as the decomposition kernels.  Per iteration k: (optionally) every lane first READS the word, so that its line sits
// in the CU's vector L1; lane (k & 63) stores k to the word -- as a 16-bit or a 32-bit store into a 64-bit record --; a wavefront-scope fence
// (no instruction) or an explicit s_waitcnt; every lane loads the record (64-bit load) and counts a STALE read (value != k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while(0)

template<int WAIT, int PRELOAD, int NARROW, int SPREAD>
__global__ void __launch_bounds__(64) raw_kernel(unsigned long long *buf, unsigned long long *stale, int iters, int words_per_wave)
{
    unsigned long long *w = buf + (size_t)blockIdx.x * words_per_wave;
    unsigned long long bad = 0, sink = 0;
    const int lane = threadIdx.x;
    for(int k = 1; k <= iters; k++) {
        const int slot = SPREAD ? (int)((unsigned)(k * 2654435761u) % (unsigned)words_per_wave) : 0;
        unsigned long long *rec = w + slot;                  // plain accesses, as the engine's: volatile ones carry the sc0 sc1 bits and bypass the vector L1
        if(PRELOAD) sink += *rec;                                                    // the line is in L1 now
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if(lane == (k & 63)) {
            if(NARROW) ((uint16_t*)rec)[3] = (uint16_t)k;                     // one 16-bit field of the record (the list links are such fields)
            else       ((uint32_t*)rec)[1] = (uint32_t)k;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if(WAIT) __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long v = *rec;                                           // every lane, 64-bit load of the record
        const unsigned got = NARROW ? (unsigned)(v >> 48) : (unsigned)(v >> 32);
        const unsigned want = NARROW ? ((unsigned)k & 0xFFFFu) : (unsigned)k;
        if(got != want) bad++;
    }
    atomicAdd(stale + 1, sink & 1);                                                  // (keeps the preloads alive)
    if(bad) atomicAdd(stale, bad);
}

template<int WAIT, int PRELOAD, int NARROW, int SPREAD> static int run(const char *name, int blocks, int iters)
{
    const int wpw = 4096;
    unsigned long long *buf, *stale, h[2] = {0, 0};
    CHK(hipMalloc(&buf, (size_t)blocks * wpw * 8)); CHK(hipMemset(buf, 0, (size_t)blocks * wpw * 8));
    CHK(hipMalloc(&stale, 16)); CHK(hipMemset(stale, 0, 16));
    hipLaunchKernelGGL((raw_kernel<WAIT, PRELOAD, NARROW, SPREAD>), dim3(blocks), dim3(64), 0, 0, buf, stale, iters, wpw);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h, stale, 16, hipMemcpyDeviceToHost));
    printf("  %-72s waves %5d x %6d iterations x 64 lanes: stale reads %llu\n", name, blocks, iters, h[0]);
    CHK(hipFree(buf)); CHK(hipFree(stale));
    return 0;
}

int main()
{
    for(int blocks : {1, 256, 3072}) {
        const int it = blocks == 1 ? 200000 : 20000;
        if(run<0, 0, 0, 0>("fence only, 32-bit store, one word", blocks, it)) return 1;
        if(run<0, 1, 0, 0>("fence only, 32-bit store, one word, line read first", blocks, it)) return 1;
        if(run<0, 1, 1, 0>("fence only, 16-bit store, one word, line read first", blocks, it)) return 1;
        if(run<0, 1, 1, 1>("fence only, 16-bit store, 4096 words per wave, line read first", blocks, it)) return 1;
        if(run<0, 0, 1, 1>("fence only, 16-bit store, 4096 words per wave", blocks, it)) return 1;
        if(run<1, 1, 1, 1>("s_waitcnt,  16-bit store, 4096 words per wave, line read first", blocks, it)) return 1;
    }
    return 0;
}
