// ahv_score_legacy.h -- the three earlier fp32 formulations of the fused scorer, kept ONLY for A/B runs in
// tools/kbench.cpp (developer micro-benchmark).  They are not compiled into libahv_hip.so: the product has one
// fp32 kernel (score_hypotheses_dual_kernel<false>, 3dahv_amd/csrc/ahv_score.hip) and its opt-in split-f16
// sibling.  All three pass the same parity tests when selected in kbench:
//   0 = 16x16x4 MFMA, W1 in registers, phase-sequential (first correct kernel, 0.98 ms per 50 000 hypotheses)
//   1 = same with the micro-step software pipeline (1.04 ms)
//   2 = 32x32x2 MFMA, W1 in registers, half-volume phases (0.954 ms)
// Launch shape: 256-thread workgroups (4 waves, one per SIMD, 512-register budget).
// Included by tools/kbench.cpp AFTER 3dahv_amd/csrc/ahv_score.hip (AHV_STAMP, hyp_score_rs, launch helpers).
#pragma once
#include "ahv_pipeline.h"
#include "ahv_wide.h"

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

// Same score with a reduce-scatter instead of four all-reduces: after two exchange steps lane (col, kq)
// owns the complete sums of ONE position (tile t = kq, column col), so the normalisation runs once per

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


// ---------------------------------------------------------------------------------------
// Pipelined variant (variant 1).  Same arithmetic as above, but the trilinear gather of the NEXT
// quarter is interleaved, instruction by instruction, with the MFMAs of the CURRENT quarter:
// one wave per SIMD cannot rely on another wave to fill the matrix pipe's shadow, so the
// overlap is built into the instruction stream (sched_group_barrier: 1 MFMA : a few VALU/DS).
// The pipeline also crosses hypotheses: quarter 0 of hypothesis h+1 is gathered under the
// last GEMM quarter of hypothesis h.
//
// The two quarter buffers and the source image are separate __shared__ arrays so that the
// compiler knows a stage's LDS writes (other buffer) cannot alias its LDS reads and may
// interleave them freely; wave_lds_fence() between stages keeps the cross-lane RAW/WAR order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScoreThreads, 1) void score_hypotheses_pipelined_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ feat_tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, float* __restrict__ scores,
    unsigned long long* __restrict__ best_key)
{
    __shared__ __attribute__((aligned(16))) float lds_src[kSrcFloats];
    __shared__ __attribute__((aligned(16))) float lds_even[4 * kQuarterFloats];
    __shared__ __attribute__((aligned(16))) float lds_odd[4 * kQuarterFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* srcT = lds_src;
    float* bufE = lds_even + wave * kQuarterFloats;
    float* bufO = lds_odd + wave * kQuarterFloats;

    HeadFrags f;
    load_head_frags(f, W1, W2, b2, lane);
    const LaneConst lc = make_lane_const(lane);
    TriState ts;

    const int n16 = lane & 15, kq = lane >> 4;
    const long hstep = (long)gridDim.x * 4;

    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kScoreThreads);
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
        long h = (long)blockIdx.x * 4 + wave;
        float Rm[9];
        if (h < N) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            tri_quarter<0>(bufE, srcT, Rm, lane);  // pipeline prologue
            wave_lds_fence();
        }
#ifdef AHV_STAMPS
        unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
        for (; h < N; h += hstep) {
#ifdef AHV_STAMPS
            unsigned long long t0, t1, t2, t3, t4, t5, t6;
#endif
            AHV_STAMP(t0)
            const long hn = (h + hstep < N) ? h + hstep : h;  // last round re-gathers its own q0 (unused)
            float Rn[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rn[i] = Rb[hn * 9 + i];

            f32x4 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

            pipelined_stage<0>(acc, f, bufE, bufO, srcT, Rm, lc, ts);
            AHV_STAMP(t1)
            pipelined_stage<1>(acc, f, bufO, bufE, srcT, Rm, lc, ts);
            AHV_STAMP(t2)
            pipelined_stage<2>(acc, f, bufE, bufO, srcT, Rm, lc, ts);
            AHV_STAMP(t3)
            pipelined_stage<3>(acc, f, bufO, bufE, srcT, Rn, lc, ts);
            AHV_STAMP(t4)

            f32x4 v[2][4];
            gemm2(v, acc, f);
            AHV_STAMP(t5)
            const float s = hyp_score(v, tg);
            if (scores != nullptr && lane == 0) scores[(long)b * N + h] = s;
            const unsigned long long key = pack_key(s, (unsigned)(n_offset + h));
            best = key > best ? key : best;
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rn[i];
            AHV_STAMP(t6)
#ifdef AHV_STAMPS
            tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3;
            tsum[4] += t5 - t4; tsum[5] += t6 - t5; tsum[6] += 1;
#endif
        }
#ifdef AHV_STAMPS
        if (lane == 0) {
            const int gw = (blockIdx.x * 4 + wave) & 1023;
            for (int i = 0; i < 7; ++i) g_stamps[gw * 16 + i] = tsum[i];
        }
#endif
        if (best_key != nullptr && lane == 0 && best != 0ull) atomicMax(best_key + b, best);
    }
}


// ---------------------------------------------------------------------------------------
// Wide variant (variant 2): 32x32x2 fp32 MFMA, half-volume phases (ahv_wide.h).
// LDS: 40 KiB source image + 4 waves x 16 KiB half-volume image = 104 KiB.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScoreThreads, 1) void score_hypotheses_wide_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ feat_tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, float* __restrict__ scores,
    unsigned long long* __restrict__ best_key)
{
    __shared__ __attribute__((aligned(16))) float lds_src[kSrcFloats];
    __shared__ __attribute__((aligned(16))) float lds_half[4 * kHalfFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* srcT = lds_src;
    float* buf = lds_half + wave * kHalfFloats;

    WideFrags f;
    load_wide_frags(f, W1, W2, b2, lane);
    const WideLane L = make_wide_lane(lane);
    const long hstep = (long)gridDim.x * 4;

    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();  // every wave is done with the previous sample's source image
        stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kScoreThreads);
        f32x16 tg[2];
        {
            const float* ft = feat_tgt + (long)b * (32 * 64);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tg[t][r] = ft[(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 64 + 32 * t + (lane & 31)];
        }
        __syncthreads();

        unsigned long long best = 0ull;
        const float* Rb = R + (long)b * r_batch_stride;
#ifdef AHV_STAMPS
        unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
        long h = (long)blockIdx.x * 4 + wave;
        float Rn[9];  // rotation of the NEXT hypothesis: its scalar loads fly during the current one
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = Rb[(h < N ? h : 0) * 9 + i];
        for (; h < N; h += hstep) {
#ifdef AHV_STAMPS
            unsigned long long t0, t1, t2, t3, t4, t5, t6;
#endif
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rn[i];
            {
                const long hn = (h + hstep < N) ? h + hstep : h;
#pragma unroll
                for (int i = 0; i < 9; ++i) Rn[i] = Rb[hn * 9 + i];  // wave-uniform -> scalar loads
            }

            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

            AHV_STAMP(t0)
            gather_half<0>(buf, srcT, Rm, L);
            wave_lds_fence();
            AHV_STAMP(t1)
            gemm1_half<0>(acc, f, buf, L);
            wave_lds_fence();
            AHV_STAMP(t2)
            gather_half<1>(buf, srcT, Rm, L);
            wave_lds_fence();
            AHV_STAMP(t3)
            gemm1_half<1>(acc, f, buf, L);
            wave_lds_fence();
            AHV_STAMP(t4)

            f32x16 v[2];
            gemm2_wide(v, acc, f);
            AHV_STAMP(t5)
            const float s = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, score_wide(v, tg)), 63));  // total is in lane 63
            if (scores != nullptr && lane == 0) scores[(long)b * N + h] = s;
            const unsigned long long key = pack_key(s, (unsigned)(n_offset + h));
            best = key > best ? key : best;
            AHV_STAMP(t6)
#ifdef AHV_STAMPS
            tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3;
            tsum[4] += t5 - t4; tsum[5] += t6 - t5; tsum[6] += 1;
#endif
        }
#ifdef AHV_STAMPS
        if (lane == 0) {
            const int gw = (blockIdx.x * 4 + wave) & 1023;
            for (int i = 0; i < 7; ++i) g_stamps[gw * 16 + i] = tsum[i];
        }
#endif
        if (best_key != nullptr && lane == 0 && best != 0ull) atomicMax(best_key + b, best);
    }
}

inline hipError_t launch_score_legacy(int variant, const float* vol_src, const float* feat_tgt, const float* R,
                                      int64_t r_batch_stride, int64_t n_offset, const float* W1, const float* W2,
                                      const float* b2, int B, int64_t N, float* scores, uint64_t* best_key, int num_cu,
                                      hipStream_t stream)
{
    const size_t lds = sizeof(float) * kScoreLdsFloats;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_hypotheses_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int gy = B < num_cu ? B : num_cu;
    int gx = num_cu / gy;
    const int64_t need = (N + 3) / 4;  // workgroups that can get at least one hypothesis per wave
    if (gx > need) gx = (int)need;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, gy);
#define AHV_LAUNCH(K, L) \
    launch_score_kernel(K, L, grid, kScoreThreads, stream, vol_src, feat_tgt, R, (long)r_batch_stride, (long)n_offset, W1, W2, b2, B, (long)N, scores, reinterpret_cast<unsigned long long*>(best_key))
    switch (variant) {
        case 0: return AHV_LAUNCH(score_hypotheses_kernel, lds);
        case 1: return AHV_LAUNCH(score_hypotheses_pipelined_kernel, 0);
        default: return AHV_LAUNCH(score_hypotheses_wide_kernel, 0);
    }
#undef AHV_LAUNCH
}

}  // namespace ahv
