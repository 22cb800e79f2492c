// valu_probe.cpp -- developer probe: (1) fp32 VALU issue rate with 1 / 2 waves per SIMD,
// (2) do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap (fp32 MFMA 32x32x2)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// role 0: VALU only; 1: MFMA only; 2: waves 0-3 MFMA, waves 4-7 VALU (THREADS must be 512)
template <int ROLE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const float* __restrict__ in, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = in[lane], b = in[lane + 64];
    const float c1 = in[0] * 1e-3f + 1.0f, c2 = in[1] * 1e-3f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[lane + i];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const bool do_mfma = (ROLE == 1) || (ROLE == 2 && wave < 4);
    const bool do_valu = (ROLE == 0) || (ROLE == 2 && wave >= 4);
    if (do_mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int ROLE, int THREADS>
int run(const char* name, const float* din, float* dout, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<ROLE, THREADS>), dim3(256), dim3(THREADS), 0, 0, din, dout, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per SIMD: cycles at 2.4 GHz per loop iteration (64 VALU or 4 MFMA per wave-iteration)
    printf("%-44s %.3f ms -> %.1f cycles@2.4GHz per wave-iteration\n", name, best, best * 1e-3 * 2.4e9 / iters);
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
    if (run<0, 256>("VALU only, 1 wave/SIMD (64 v_fma / iter)", din, dout, it)) return 1;
    if (run<0, 512>("VALU only, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<0, 1024>("VALU only, 4 waves/SIMD", din, dout, it)) return 1;
    if (run<1, 256>("MFMA only, 1 wave/SIMD (4 x 32x32x2 / iter)", din, dout, it)) return 1;
    if (run<1, 512>("MFMA only, 2 waves/SIMD", din, dout, it)) return 1;
    if (run<2, 512>("wave A MFMA + wave B VALU on each SIMD", din, dout, it)) return 1;
    return 0;
}
