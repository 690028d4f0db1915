// decomp_class.hip -- one size class of the decomposition kernel (compile with -DALD_CLASS_ID=0..4).
//
// gfx950 kernel: one 64-lane workgroup (= one wavefront) per splice graph, persistent over an atomic
// work counter.  Every wave reaches the `break` in wave_main once the class is drained, so the grid
// always drains.  No MFMA: integer / pointer-chasing work with FP64 compares.
#include <hip/hip_runtime.h>
#ifdef ALD_ROWS
#include "decomp_device_rows.h"     /* make ROWS=1: the adjacency-row form of the engine (A/B build: profiles/r04/) */
#else
#include "decomp_device.h"
#endif

#ifdef ALD_RAW_VARIANT
#define ALD_KERNEL_NAME ALD_CAT(ald_decomp_kernel_raw_c, ALD_CLASS_ID)
#define ALD_ENTRY(x) ALD_CAT(ALD_CAT(x, _raw_c), ALD_CLASS_ID)
#else
#define ALD_KERNEL_NAME ALD_CAT(ald_decomp_kernel_c, ALD_CLASS_ID)
#define ALD_ENTRY(x) ALD_CAT(ALD_CAT(x, _c), ALD_CLASS_ID)
#endif
#ifndef ALD_WAVES_PER_EU
#define ALD_WAVES_PER_EU 4      /* register budget: 512 / 4 = 128 VGPRs per lane (MI355X_MICROARCH.md, Register files) */
#endif

extern "C" __global__ void __launch_bounds__(64, ALD_WAVES_PER_EU) ALD_KERNEL_NAME(const ald::KernelArgs *A)
{
    ALD_CLASS_NS::wave_main((ALD_GLOBAL const ald::KernelArgs*)A, (int)blockIdx.x);
}

extern "C" int ALD_ENTRY(ald_launch)(const ald::KernelArgs *dA, int blocks, hipStream_t stream)
{
    (void)hipGetLastError();                       // drop any stale sticky error of this thread before judging the launch
    hipLaunchKernelGGL(ALD_KERNEL_NAME, dim3(blocks), dim3(64), 0, stream, dA);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// bytes of hot state that live in the wave's slab (0 for the LDS classes)
extern "C" unsigned long long ALD_ENTRY(ald_hot_slab_bytes)()
{
#if ALD_CLASS_ID >= ALD_FIRST_GLOBAL_CLASS
    return (sizeof(ALD_CLASS_NS::Hot) + 255) / 256 * 256;
#else
    return 0;
#endif
}

extern "C" int ALD_ENTRY(ald_occupancy)()
{
    int nb = 0;
    if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ALD_KERNEL_NAME, 64, 0) != hipSuccess) return 0;
    return nb;
}
