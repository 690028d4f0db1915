// lds_issue.hip -- what an LDS instruction costs on gfx950 when one lane of the wave is active and when all 64 are (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_issue.hip -o build/lds_issue && build/lds_issue
// Two questions behind the decomposition kernel's design (one graph per wave, most list work on lane 0):
//   (a) latency of a chain of DEPENDENT ds_read_b64 (a list walk) with W one-wave workgroups resident per CU;
//   (b) throughput of INDEPENDENT ds_read_b64 / ds_read_b128 per CU with 1 and with 64 active lanes: does an LDS instruction of a wave
//       with a single active lane occupy the LDS pipeline for as long as a full one?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while(0)

template<int LANES> __global__ void __launch_bounds__(64) chain_kernel(unsigned long long *out, int hops)
{
    extern __shared__ unsigned long long lds[];          // 1024 entries: a ring of "links"
    for(int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned long long)((i * 37 + 11) & 1023);
    __syncthreads();
    if((int)threadIdx.x < LANES) {
        unsigned long long acc = 0; unsigned cur = threadIdx.x;
        for(int h = 0; h < hops; h++) { const unsigned long long v = lds[cur]; acc += v; cur = (unsigned)v & 1023u; }   // dependent
        out[blockIdx.x * 64 + threadIdx.x] = acc + cur;
    }
}
template<int LANES, int WIDTH> __global__ void __launch_bounds__(64) stream_kernel(unsigned long long *out, int reps)
{
    extern __shared__ unsigned long long lds[];
    for(int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned long long)i;
    __syncthreads();
    if((int)threadIdx.x < LANES) {
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        const unsigned base = threadIdx.x * (WIDTH / 8);
        for(int r = 0; r < reps; r++) {                   // 8 independent reads per iteration
            const unsigned o = (unsigned)(r * 8) & 255u;
            if(WIDTH == 8) {
                a0 += lds[(base + o) & 1023]; a1 += lds[(base + o + 64) & 1023]; a2 += lds[(base + o + 128) & 1023]; a3 += lds[(base + o + 192) & 1023];
                a0 += lds[(base + o + 256) & 1023]; a1 += lds[(base + o + 320) & 1023]; a2 += lds[(base + o + 384) & 1023]; a3 += lds[(base + o + 448) & 1023];
            } else {
                const ulonglong2 *p = (const ulonglong2*)lds; ulonglong2 v;
                v = p[((base + o) & 1023) / 2]; a0 += v.x + v.y; v = p[((base + o + 64) & 1023) / 2]; a1 += v.x + v.y;
                v = p[((base + o + 128) & 1023) / 2]; a2 += v.x + v.y; v = p[((base + o + 192) & 1023) / 2]; a3 += v.x + v.y;
                v = p[((base + o + 256) & 1023) / 2]; a0 += v.x + v.y; v = p[((base + o + 320) & 1023) / 2]; a1 += v.x + v.y;
                v = p[((base + o + 384) & 1023) / 2]; a2 += v.x + v.y; v = p[((base + o + 448) & 1023) / 2]; a3 += v.x + v.y;
            }
        }
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    }
}
typedef void (*kern_t)(unsigned long long*, int);
static int timed(kern_t k, int wg_per_cu, int n, unsigned long long *d_out, double &ms_out)
{
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for(int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 8192, 0, d_out, n);       // 8 KB of LDS per workgroup: 20 fit a CU, as in class 1 of the decomposition kernel
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    ms_out = ms; CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return 0;
}
int main()
{
    unsigned long long *d_out; CHK(hipMalloc(&d_out, 8ull * 64 * 256 * 32));
    const double GHZ = 2.4; const int hops = 20000, reps = 4000;
    const int W[] = {1, 4, 8, 12, 16, 20};
    printf("(a) dependent chain of ds_read_b64: cycles per hop seen by a wave (engine clock taken as %.1f GHz)\n", GHZ);
    for(int w : W) {
        double m1, m64;
        if(timed(chain_kernel<1>, w, hops, d_out, m1) || timed(chain_kernel<64>, w, hops, d_out, m64)) return 1;
        printf("    %2d wg/CU: 1 active lane %7.1f cycles/hop   64 active lanes %7.1f cycles/hop\n", w, m1 * 1e-3 * GHZ * 1e9 / hops, m64 * 1e-3 * GHZ * 1e9 / hops);
    }
    printf("(b) independent reads, 8 in flight per wave: cycles of CU time per LDS instruction (= ms x clock / (instructions per CU))\n");
    for(int w : W) {
        double a, b, c, d;
        if(timed(stream_kernel<1, 8>, w, reps, d_out, a) || timed(stream_kernel<64, 8>, w, reps, d_out, b) || timed(stream_kernel<1, 16>, w, reps, d_out, c) || timed(stream_kernel<64, 16>, w, reps, d_out, d)) return 1;
        const double per = 1e-3 * GHZ * 1e9 / ((double)reps * 8 * w);
        printf("    %2d wg/CU: b64 1 lane %6.2f   b64 64 lanes %6.2f   b128 1 lane %6.2f   b128 64 lanes %6.2f\n", w, a * per, b * per, c * per, d * per);
    }
    CHK(hipFree(d_out));
    return 0;
}
