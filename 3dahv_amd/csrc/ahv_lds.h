// ahv_lds.h -- 16-byte LDS stores that are safe against the store-data hazard of gfx950.
//
// Found in round 3 (tools/dbg): a VALU instruction that overwrites a data register of a
// ds_write_b128 within two wait states of the store can reach the LDS instead of the value the store was given.
// hipcc (ROCm 7.2) pads this ">64-bit store data" hazard for VMEM / FLAT stores (GCNHazardRecognizer::
// createsVALUHazard) but not for DS stores.  It needs the overwriting instruction to issue right behind the store,
// so it showed only where a wave ran at s_setprio 1 beside a partner wave (the split-f16 gather: sixteen conversion
// temporaries stored and recomputed back to back), in the first launches of a process (low clock), on the younger
// wave of each SIMD: ~1 % of the hypotheses came out 1e-3 off.  One `s_nop 1` that READS the stored registers (so
// that no redefinition of them can be scheduled in front of it) closes the window; cost: two cycles per store.
#pragma once
#include <hip/hip_runtime.h>

namespace ahv {

template <typename V>
__device__ __forceinline__ void lds_store128(void* p, V v)
{
    static_assert(sizeof(V) == 16, "16-byte vectors only");
    *reinterpret_cast<V*>(p) = v;
    asm volatile("s_nop 1" ::"v"(v) : "memory");
}

}  // namespace ahv
