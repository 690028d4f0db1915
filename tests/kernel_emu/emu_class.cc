// emu_class.cc -- TEST-ONLY: single-lane CPU emulation of one size class of the HIP engine
// (aletsch_amd/csrc/decomp_device.h compiled with -DALD_EMU -DALD_CLASS_ID=k).  Lets the CPU-only test tier
// check the array-based algorithm against the oracle before it ever runs on a GPU.  It exercises none of the
// wave-level synchronisation, and it is never linked into the product library.
#ifndef ALD_EMU
#error "emulation build only"
#endif
#ifdef ALD_ROWS
#include "../../aletsch_amd/csrc/decomp_device_rows.h"      /* make ROWS=1 -> libkernel_emu_rows.so: the adjacency-row form of the engine */
#else
#include "../../aletsch_amd/csrc/decomp_device.h"
#endif

extern "C" void ALD_CAT(emu_run_class_, ALD_CLASS_ID)(const ald::KernelArgs *A)
{
    ALD_CLASS_NS::wave_main(A, 0);
}
