// ahv_score.hip -- fused rotation-hypothesis scorer (the north-star kernel).
//
// One launch scores N hypotheses for B volume pairs:
//   rotate_volume (utils.py:113-131) -> forward_3d2d (modules/modules.py:112-124)
//   -> (f_src*f_tgt).sum(ch).mean(pos) (test_co3d.py:143) -> running max (test_co3d.py:145)
// HBM traffic per hypothesis: 36 B of R in, 4 B of score out.  Everything else
// (source volume, head weights, target feature) is resident on chip.
//
// Launch shape: 256-thread workgroups (4 waves, one per SIMD, 512-register
// budget), a persistent grid of ~one workgroup per CU; blockIdx.y strides over
// the batch, (blockIdx.x, wave) strides over hypotheses.  LDS per workgroup:
// 40 KiB source image + 4 waves x 2 quarter buffers x 8 KiB = 104 KiB.
#include "ahv_device.h"

namespace ahv {

constexpr int kScoreThreads = 256;
constexpr int kScoreLdsFloats = kSrcFloats + 4 * 2 * kQuarterFloats;

__device__ __forceinline__ float hyp_score(const f32x4 (&v)[2][4], const f32x4 (&tg)[4][2])
{
    // F.normalize(dim=1) then dot with the unit-norm target, mean over 64 positions
    // (modules/modules.py:122, test_co3d.py:143).  A lane holds 8 of the 32 channels
    // of position (16t + lane&15); the other 24 sit in lanes l^16, l^32, l^48.
    float tot = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float ss = 0.0f, dt = 0.0f;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = v[m2][t][r];
                ss += x * x;
                dt += x * tg[t][m2][r];
            }
        ss += __shfl_xor(ss, 16, 64);
        dt += __shfl_xor(dt, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        dt += __shfl_xor(dt, 32, 64);
        tot += dt / fmaxf(sqrtf(ss), 1e-12f);
    }
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) tot += __shfl_xor(tot, s, 64);
    return tot * (1.0f / 64.0f);
}

__global__ __launch_bounds__(kScoreThreads, 1) void score_hypotheses_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ feat_tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, float* __restrict__ scores,
    unsigned long long* __restrict__ best_key)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* srcT = smem;
    float* buf0 = smem + kSrcFloats + wave * (2 * kQuarterFloats);
    float* buf1 = buf0 + kQuarterFloats;

    HeadFrags f;
    load_head_frags(f, W1, W2, b2, lane);

    const int n16 = lane & 15, kq = lane >> 4;
    const long hstep = (long)gridDim.x * 4;

    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();  // every wave is done with the previous sample's source image
        stage_src_volume(srcT, vol_src + (long)b * (16 * 512), tid, kScoreThreads);
        f32x4 tg[4][2];
        {
            const float* ft = feat_tgt + (long)b * (32 * 64);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tg[t][m2][r] = ft[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n16];
        }
        __syncthreads();

        unsigned long long best = 0ull;
        const float* Rb = R + (long)b * r_batch_stride;
        for (long h = (long)blockIdx.x * 4 + wave; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];  // wave-uniform -> scalar loads

            f32x4 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

            tri_quarter<0>(buf0, srcT, Rm, lane);
            wave_lds_fence();
            gemm1_quarter<0>(acc, f, buf0, lane);
            tri_quarter<1>(buf1, srcT, Rm, lane);
            wave_lds_fence();
            gemm1_quarter<1>(acc, f, buf1, lane);
            wave_lds_fence();
            tri_quarter<2>(buf0, srcT, Rm, lane);
            wave_lds_fence();
            gemm1_quarter<2>(acc, f, buf0, lane);
            wave_lds_fence();
            tri_quarter<3>(buf1, srcT, Rm, lane);
            wave_lds_fence();
            gemm1_quarter<3>(acc, f, buf1, lane);
            wave_lds_fence();

            f32x4 v[2][4];
            gemm2(v, acc, f);
            const float s = hyp_score(v, tg);
            if (scores != nullptr && lane == 0) scores[(long)b * N + h] = s;
            const unsigned long long key = pack_key(s, (unsigned)(n_offset + h));
            best = key > best ? key : best;
        }
        if (best_key != nullptr && lane == 0 && best != 0ull) atomicMax(best_key + b, best);
    }
}

__global__ void unpack_best_kernel(const unsigned long long* __restrict__ best_key, int B,
                                   float* __restrict__ best_score, long* __restrict__ best_idx)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = best_key[b];
    if (best_score) best_score[b] = (k == 0ull) ? -INFINITY : key_score(k);
    if (best_idx) best_idx[b] = (k == 0ull) ? -1l : (long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
}

}  // namespace ahv

// ---- host-side launchers (called by the C ABI in ahv_abi.hip) ----------------------
namespace ahv {

hipError_t launch_score_hypotheses(const float* vol_src, const float* feat_tgt, const float* R,
                                   int64_t r_batch_stride, int64_t n_offset, const float* W1,
                                   const float* W2, const float* b2, int B, int64_t N, float* scores,
                                   uint64_t* best_key, int num_cu, hipStream_t stream)
{
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "key width");
    const size_t lds = sizeof(float) * kScoreLdsFloats;
    static thread_local int attr_dev = -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_hypotheses_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    // Persistent grid: ~one workgroup per CU.  y strides over the batch, x over hypotheses.
    int gy = B < num_cu ? B : num_cu;
    int gx = num_cu / gy;
    const int64_t need = (N + 3) / 4;  // workgroups that can get at least one hypothesis per wave
    if (gx > need) gx = (int)need;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(score_hypotheses_kernel, dim3(gx, gy), dim3(kScoreThreads), lds, stream, vol_src,
                       feat_tgt, R, (long)r_batch_stride, (long)n_offset, W1, W2, b2, B, (long)N, scores,
                       reinterpret_cast<unsigned long long*>(best_key));
    return hipGetLastError();
}

hipError_t launch_unpack_best(const uint64_t* best_key, int B, float* best_score, int64_t* best_idx,
                              hipStream_t stream)
{
    const int threads = 64;
    hipLaunchKernelGGL(unpack_best_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, stream,
                       reinterpret_cast<const unsigned long long*>(best_key), B, best_score,
                       reinterpret_cast<long*>(best_idx));
    return hipGetLastError();
}

}  // namespace ahv
