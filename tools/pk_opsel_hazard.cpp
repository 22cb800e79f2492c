// pk_opsel_hazard.cpp -- stand-alone reproduction of the gfx950 hazard documented at low_half() (3dahv_amd/csrc/ahv_dual.h):
// a packed-fp32 instruction whose LOW lane reads the HIGH half of a source (op_sel:[0,1]) next to XDL MFMAs that start on
// an idle matrix pipe.  512-thread workgroups, two waves per SIMD; every wave alternates a VALU phase (v_pk_mul_f32 on known
// operands, results checked on the spot) with an MFMA phase (v_mfma_f32_16x16x32_f16 separated by GAP); waves 4-7 start
// with the other phase, so that the partner of a wave in its VALU phase is issuing MFMAs.
//   build: hipcc --offload-arch=gfx950 -O3 tools/pk_opsel_hazard.cpp -o tools/pk_opsel_hazard      run: tools/pk_opsel_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: op_sel:[0,1] (low lane <- high half of src1; high lane <- high half)   MODE 1: op_sel_hi:[1,0] (both lanes <- low half)
// MF 0: v_mfma_f32_16x16x32_f16 (XDL)   MF 1: v_mfma_f32_16x16x4_f32 (the fp32 scorer's MFMA)
template <int MODE, int GAP, int MF = 0>
__global__ __launch_bounds__(512, 2) void probe(unsigned* bad, int rounds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x2 x = {1.0f + lane, 2.0f + lane}, y = {3.0f + lane, 5.0f + lane};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.125f * (lane & 7)); b[i] = (_Float16)(0.25f * (i + 1)); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned wrong_lo = 0, wrong_hi = 0;
    for (int r = 0; r < rounds; ++r) {
        if (((r + (wave >> 2)) & 1) == 0) {
            for (int k = 0; k < 64; ++k) {
                f32x2 p;
                if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(x), "v"(y));
                else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(x), "v"(y));
                const float want_lo = MODE == 0 ? x[0] * y[1] : x[0] * y[0], want_hi = MODE == 0 ? x[1] * y[1] : x[1] * y[0];
                wrong_lo += p[0] != want_lo;
                wrong_hi += p[1] != want_hi;
            }
        } else {
            for (int k = 0; k < 16; ++k) {
                if (MF == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(a), "v"(b), "n"(GAP));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(x[0]), "v"(y[1]), "n"(GAP));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (acc[0] == 12345.0f) wrong_lo += 1;  // keeps the MFMAs
    atomicAdd(&bad[(lane >> 4)], wrong_lo);
    atomicAdd(&bad[4 + (lane >> 4)], wrong_hi);
}

template <int MODE, int GAP, int MF = 0>
static void run(unsigned* d, const char* what)
{
    (void)hipMemset(d, 0, 32);
    hipLaunchKernelGGL((probe<MODE, GAP, MF>), dim3(1024), dim3(512), 0, 0, d, 64);
    (void)hipDeviceSynchronize();
    unsigned h[8];
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%-34s gap %2d wait states: wrong LOW results by 16-lane group %u %u %u %u, wrong HIGH results %u %u %u %u  (of %u per group)\n",
           what, GAP, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1024u * 8 * 16 * 32 * 64);
}

int main()
{
    unsigned* d;
    (void)hipMalloc(&d, 32);
    run<0, 0>(d, "op_sel:[0,1] (low <- high half)");
    run<0, 4>(d, "op_sel:[0,1] (low <- high half)");
    run<0, 8>(d, "op_sel:[0,1] (low <- high half)");
    run<0, 16>(d, "op_sel:[0,1] (low <- high half)");
    run<1, 0>(d, "op_sel_hi:[1,0] (high <- low half)");
    run<1, 8>(d, "op_sel_hi:[1,0] (high <- low half)");
    run<1, 16>(d, "op_sel_hi:[1,0] (high <- low half)");
    run<0, 0, 1>(d, "op_sel:[0,1] next to fp32 MFMAs");
    run<0, 8, 1>(d, "op_sel:[0,1] next to fp32 MFMAs");
    run<0, 16, 1>(d, "op_sel:[0,1] next to fp32 MFMAs");
    return 0;
}
