// pk_probe.cpp -- developer probe: issue rate of v_pk_fma_f32 (two fp32 FMAs per lane and instruction, weight
// broadcast through op_sel) against v_fma_f32 on gfx950, with 1 / 2 waves per SIMD, alone and interleaved with
// fp32 MFMAs (16x16x4) the way the fused scorer's gather and GEMM alternate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: 128 scalar v_fma per iteration; 1: 64 v_pk_fma (same FLOPs); 2: MFMA only (32 per iteration);
// 3: 128 v_fma then 32 MFMA; 4: 64 v_pk_fma then 32 MFMA
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const float* __restrict__ in, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    float w[8];
    for (int i = 0; i < 8; ++i) w[i] = in[i] * 1e-3f + 1.0f;
    f32x2 v[8], a[8];
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{in[lane + i], in[lane + 8 + i]}; a[i] = f32x2{0.f, 0.f}; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ma = in[lane], mb = in[lane + 64];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[i][0] = __builtin_fmaf(v[i][0], w[n], a[i][0]);
                    a[i][1] = __builtin_fmaf(v[i][1], w[n], a[i][1]);
                }
        }
        if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(v[i], f32x2{w[n], w[n]}, a[i]);
        }
        if (MODE >= 2 && MODE != 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[i], 0, 0, 0);
        }
        if (MODE == 5) {  // the dependency pattern of the scorer's x / y slabs: two accumulators, alternating
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[i], 0, 0, 0);
        }
        // keep the loop-carried values opaque so that nothing is hoisted or folded across iterations
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int THREADS>
int run(const char* name, const float* din, float* dout, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, din, dout, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-64s %.3f ms -> %.0f cycles@2.4GHz per wave-iteration\n", name, best, best * 1e-3 * 2.4e9 / iters);
    return 0;
}

int main()
{
    std::vector<float> in(256);
    for (size_t i = 0; i < in.size(); ++i) in[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    float *din, *dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, 256 * 512 * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    const int it = 20000;
    if (run<0, 256>("128 v_fma_f32, 1 wave/SIMD", din, dout, it)) return 1;
    if (run<1, 256>("64 v_pk_fma_f32 (same FLOPs), 1 wave/SIMD", din, dout, it)) return 1;
    if (run<0, 512>("128 v_fma_f32, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<1, 512>("64 v_pk_fma_f32, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<2, 256>("32 MFMA 16x16x4 only (8 independent accumulators), 1 wave/SIMD", din, dout, it)) return 1;
    if (run<5, 256>("32 MFMA 16x16x4, 2 alternating accumulators, 1 wave/SIMD", din, dout, it)) return 1;
    if (run<5, 512>("32 MFMA 16x16x4, 2 alternating accumulators, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<2, 512>("32 MFMA 16x16x4 only, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<3, 512>("128 v_fma_f32 + 32 MFMA, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<4, 512>("64 v_pk_fma_f32 + 32 MFMA, 2 waves/SIMD", din, dout, it)) return 1;
    return 0;
}
