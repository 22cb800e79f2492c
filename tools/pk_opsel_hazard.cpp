// pk_opsel_hazard.cpp -- stand-alone reproduction of the gfx950 hazard documented at low_half() (3dahv_amd/csrc/ahv_dual.h):
// a packed-fp32 instruction whose LOW lane reads the HIGH half of a source (op_sel bit set) next to XDL MFMAs that start on
// an idle matrix pipe.  Workgroups of 512 threads (two waves per SIMD) or 256 (one per SIMD); every wave alternates a VALU
// phase (64 packed instructions on known operands, each result checked on the spot) with an MFMA phase (16 MFMAs separated
// by GAP wait states); the second half of the waves starts with the other phase, so that with 512 threads the SIMD partner
// of a wave in its VALU phase is issuing MFMAs, and with 256 threads only waves of OTHER SIMDs are.
//   build: hipcc --offload-arch=gfx950 -O3 tools/pk_opsel_hazard.cpp -o tools/pk_opsel_hazard      run: tools/pk_opsel_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op { MUL_SRC1_HI, MUL_SRC0_HI, MUL_LO_BCAST, FMA_SRC1_HI, FMA_SRC2_HI, ADD_SRC1_HI, MUL_STRAIGHT };
enum Mf { F16_16x16x32, F32_16x16x4, BF16_16x16x32, F16_32x32x16 };

template <int OP, int GAP, int MF, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void probe(unsigned* bad, int rounds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float occupy[THREADS == 256 ? 36 * 1024 : 1];  // 144 KB: the 256-thread form gets a CU (and its SIMDs) to itself
    if (THREADS == 256 && rounds < 0) occupy[threadIdx.x] = 1.0f;
    f32x2 x = {1.0f + lane, 2.0f + lane}, y = {3.0f + lane, 5.0f + lane}, z = {7.0f + lane, 11.0f + lane};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.125f * (lane & 7)); b[i] = (_Float16)(0.25f * (i + 1)); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16 = {};
    unsigned wrong_lo = 0, wrong_hi = 0;
    for (int r = 0; r < rounds; ++r) {
        if (((r + (wave >= THREADS / 128)) & 1) == 0) {
            for (int k = 0; k < 64; ++k) {
                f32x2 p;
                float lo, hi;
                if (OP == MUL_SRC1_HI) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(x), "v"(y)); lo = x[0] * y[1]; hi = x[1] * y[1]; }
                else if (OP == MUL_SRC0_HI) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(p) : "v"(x), "v"(y)); lo = x[1] * y[0]; hi = x[1] * y[1]; }
                else if (OP == MUL_LO_BCAST) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(x), "v"(y)); lo = x[0] * y[0]; hi = x[1] * y[0]; }
                else if (OP == FMA_SRC1_HI) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(x), "v"(y), "v"(z)); lo = fmaf(x[0], y[1], z[0]); hi = fmaf(x[1], y[1], z[1]); }
                else if (OP == FMA_SRC2_HI) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(p) : "v"(x), "v"(y), "v"(z)); lo = fmaf(x[0], y[0], z[1]); hi = fmaf(x[1], y[1], z[1]); }
                else if (OP == ADD_SRC1_HI) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(x), "v"(y)); lo = x[0] + y[1]; hi = x[1] + y[1]; }
                else { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y)); lo = x[0] * y[0]; hi = x[1] * y[1]; }
                wrong_lo += p[0] != lo;
                wrong_hi += p[1] != hi;
            }
        } else {
            for (int k = 0; k < 16; ++k) {
                if (MF == F16_16x16x32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(a), "v"(b), "n"(GAP));
                else if (MF == F32_16x16x4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(x[0]), "v"(y[1]), "n"(GAP));
                else if (MF == BF16_16x16x32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc) : "v"(a), "v"(b), "n"(GAP));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t.rept %3\n\ts_nop 0\n\t.endr" : "+v"(acc16) : "v"(a), "v"(b), "n"(GAP));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (acc[0] == 12345.0f || acc16[3] == 12345.0f || (THREADS == 256 && rounds < 0 && occupy[lane] == 2.0f)) wrong_lo += 1;  // keeps the MFMAs
    atomicAdd(&bad[(lane >> 4)], wrong_lo);
    atomicAdd(&bad[4 + (lane >> 4)], wrong_hi);
}

template <int OP, int GAP, int MF = F16_16x16x32, int THREADS = 512>
static void run(unsigned* d, const char* what)
{
    (void)hipMemset(d, 0, 32);
    hipLaunchKernelGGL((probe<OP, GAP, MF, THREADS>), dim3(1024), dim3(THREADS), 0, 0, d, 64);
    (void)hipDeviceSynchronize();
    unsigned h[8];
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%-58s gap %2d: wrong LOW by 16-lane group %8u %8u %8u %8u | wrong HIGH %u %u %u %u\n", what, GAP, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main()
{
    unsigned* d;
    (void)hipMalloc(&d, 32);
    printf("results checked per 16-lane group: %u (512-thread rows), %u (256-thread rows)\n", 1024u * 8 * 16 * 32 * 64, 1024u * 4 * 16 * 32 * 64);
    run<MUL_SRC1_HI, 0>(d, "v_pk_mul op_sel:[0,1] (low <- src1 high) | f16 MFMA");
    run<MUL_SRC1_HI, 4>(d, "v_pk_mul op_sel:[0,1] (low <- src1 high) | f16 MFMA");
    run<MUL_SRC1_HI, 8>(d, "v_pk_mul op_sel:[0,1] (low <- src1 high) | f16 MFMA");
    run<MUL_SRC1_HI, 16>(d, "v_pk_mul op_sel:[0,1] (low <- src1 high) | f16 MFMA");
    run<MUL_SRC0_HI, 8>(d, "v_pk_mul op_sel:[1,0] (low <- src0 high) | f16 MFMA");
    run<FMA_SRC1_HI, 8>(d, "v_pk_fma op_sel:[0,1,0] (low <- src1 high) | f16 MFMA");
    run<FMA_SRC2_HI, 8>(d, "v_pk_fma op_sel:[0,0,1] (low <- src2 high) | f16 MFMA");
    run<ADD_SRC1_HI, 8>(d, "v_pk_add op_sel:[0,1] (low <- src1 high) | f16 MFMA");
    run<MUL_LO_BCAST, 0>(d, "v_pk_mul op_sel_hi:[1,0] (high <- src1 low) | f16 MFMA");
    run<MUL_LO_BCAST, 8>(d, "v_pk_mul op_sel_hi:[1,0] (high <- src1 low) | f16 MFMA");
    run<MUL_LO_BCAST, 16>(d, "v_pk_mul op_sel_hi:[1,0] (high <- src1 low) | f16 MFMA");
    run<MUL_STRAIGHT, 8>(d, "v_pk_mul, no op_sel | f16 MFMA");
    run<MUL_SRC1_HI, 8, BF16_16x16x32>(d, "v_pk_mul op_sel:[0,1] | bf16 16x16x32 MFMA");
    run<MUL_SRC1_HI, 8, F16_32x32x16>(d, "v_pk_mul op_sel:[0,1] | f16 32x32x16 MFMA");
    run<MUL_SRC1_HI, 0, F32_16x16x4>(d, "v_pk_mul op_sel:[0,1] | fp32 16x16x4 MFMA");
    run<MUL_SRC1_HI, 8, F32_16x16x4>(d, "v_pk_mul op_sel:[0,1] | fp32 16x16x4 MFMA");
    run<MUL_SRC1_HI, 16, F32_16x16x4>(d, "v_pk_mul op_sel:[0,1] | fp32 16x16x4 MFMA");
    run<MUL_SRC1_HI, 8, F16_16x16x32, 256>(d, "v_pk_mul op_sel:[0,1] | f16 MFMA on OTHER SIMDs only");
    run<MUL_SRC1_HI, 16, F16_16x16x32, 256>(d, "v_pk_mul op_sel:[0,1] | f16 MFMA on OTHER SIMDs only");
    return 0;
}
