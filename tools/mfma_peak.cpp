// mfma_peak.cpp -- developer probe: fp32 MFMA issue rate, and what VALU work beside it costs,
// on this box (one or two waves per SIMD, random operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// SHAPE 0: 16x16x4 (8 accumulators), 1: 32x32x2 (4 accumulators).  VK 0: none, 1: v_fma_f32 (independent),
// 2: v_pk_fma_f32.  NV = VALU instructions per MFMA.
template <int SHAPE, int VK, int NV, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out,
                                                                  int iters, unsigned long long* __restrict__ clk)
{
    const int lane = threadIdx.x & 255;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[lane * 16 + i]; b[i] = in[lane * 16 + 8 + i]; }
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
    float v[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a[i]; p[i] = f32x2{a[i], b[i]}; }
    const float c1 = in[0] * 1e-3f + 1.0f, c2 = in[1] * 1e-3f;
    const f32x2 pc1 = {c1, c1}, pc2 = {c2, c2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SHAPE == 0) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc4[i], 0, 0, 0);
            else if ((i & 1) == 0) acc16[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc16[i >> 1], 0, 0, 0);
            const int nv = (SHAPE == 0) ? NV : ((i & 1) == 0 ? 2 * NV : 0);
#pragma unroll
            for (int k = 0; k < nv; ++k) {
                if (VK == 1) v[(i + k) & 7] = __builtin_fmaf(v[(i + k) & 7], c1, c2);
                if (VK == 2) p[(i + k) & 7] = __builtin_elementwise_fma(p[(i + k) & 7], pc1, pc2);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc16[i][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int VK, int NV, int THREADS>
int run(const float* din, float* dout, unsigned long long* dclk, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9; std::vector<unsigned long long> clk(512);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma_loop<SHAPE, VK, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, din, dout, iters, dclk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(clk.data(), dclk, 512 * 8, hipMemcpyDeviceToHost)); }
    }
    const double waves = 256.0 * THREADS / 64;
    const double flops = waves * iters * 8 * 2048;  // 8 x 16x16x4 or 4 x 32x32x2 per iteration
    const double valu = (double)iters * 8 * NV;      // VALU instructions per wave (16x16x4-equivalent slots)
    printf("%s waves/SIMD=%d %-8s x%d per 16x16x4-slot: %.3f ms %.1f TFLOP/s | %.1f ticks/slot (%.1f ticks per VALU over base) clk %.2f GHz\n",
           SHAPE ? "32x32x2" : "16x16x4", THREADS / 256, VK == 0 ? "-" : VK == 1 ? "v_fma" : "v_pk_fma", NV, best,
           flops / (best * 1e-3) / 1e12, (double)clk[0] / (iters * 8.0), 0.0, (double)clk[0] / clk[1] * 0.1);
    (void)valu;
    return 0;
}

int main()
{
    std::vector<float> in(256 * 16);
    for (size_t i = 0; i < in.size(); ++i) in[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    float *din, *dout; unsigned long long* dclk;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dclk, 512 * 8));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    const int it = 10000;
#define R(S, VK, NV, T) if (run<S, VK, NV, T>(din, dout, dclk, it)) return 1;
    R(0, 0, 0, 256) R(1, 0, 0, 256) R(0, 0, 0, 512) R(1, 0, 0, 512)
    R(0, 1, 1, 256) R(0, 1, 2, 256) R(0, 1, 4, 256) R(0, 2, 1, 256) R(0, 2, 2, 256)
    R(1, 1, 1, 256) R(1, 1, 2, 256) R(1, 1, 4, 256) R(1, 2, 2, 256)
    R(0, 1, 2, 512) R(0, 1, 4, 512) R(1, 1, 2, 512) R(1, 1, 4, 512) R(1, 2, 2, 512)
    return 0;
}
