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
#include "ahv_pipeline.h"
#include "ahv_wide.h"
#include "ahv_dual.h"
#include "ahv_split.h"

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
// lane instead of four times and 6 cross-lane moves replace 20.  Result uniform (read from lane 63).
__device__ __forceinline__ float hyp_score_rs(const f32x4 (&v)[2][4], const f32x4 (&tg)[4][2], int lane)
{
    float ss[4], dt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ss[t] = 0.0f;
        dt[t] = 0.0f;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = v[m2][t][r];
                ss[t] += x * x;
                dt[t] += x * tg[t][m2][r];
            }
    }
    const bool up = (lane & 32) != 0;  // upper half keeps tiles 2,3 and hands over 0,1
    float s2[2], d2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        s2[i] = (up ? ss[2 + i] : ss[i]) + __shfl_xor(up ? ss[i] : ss[2 + i], 32, 64);
        d2[i] = (up ? dt[2 + i] : dt[i]) + __shfl_xor(up ? dt[i] : dt[2 + i], 32, 64);
    }
    const bool odd = (lane & 16) != 0;  // odd rows keep the second tile of their pair
    const float s1 = (odd ? s2[1] : s2[0]) + __shfl_xor(odd ? s2[0] : s2[1], 16, 64);
    const float d1 = (odd ? d2[1] : d2[0]) + __shfl_xor(odd ? d2[0] : d2[1], 16, 64);
    const float c = d1 / fmaxf(sqrtf(s1), 1e-12f);
    const float tot = wave_sum_dpp(c);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63)) * (1.0f / 64.0f);
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


// ---------------------------------------------------------------------------------------
// Pipelined variant (default).  Same arithmetic as above, but the trilinear gather of the NEXT
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
#ifdef AHV_STAMPS
// Diagnostic build only (tools/kbench.cpp): per-wave cycle sums of the loop segments.  The stamps
// leave the kernel through this buffer alone; no output value depends on them.
__device__ unsigned long long g_stamps[2048 * 16];
#define AHV_STAMP(var)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#else
#define AHV_STAMP(var)
#endif

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
// Wide variant (default): 32x32x2 fp32 MFMA, half-volume phases (ahv_wide.h).
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


// ---------------------------------------------------------------------------------------
// Dual variant: 512 threads = two waves per SIMD, 16x16x4 MFMA, W1 fragments in LDS (ahv_dual.h).
// ---------------------------------------------------------------------------------------
constexpr int kDualThreads = 512;

// SPLIT = true: GEMM1 on the f16 matrix pipe with hi/lo split operands (ahv_split.h, score_variant 4).
template <bool SPLIT>
__global__ __launch_bounds__(kDualThreads, 2) void score_hypotheses_dual_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ feat_tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, float* __restrict__ scores,
    unsigned long long* __restrict__ best_key)
{
    __shared__ __attribute__((aligned(16))) float lds_src[kSrcFloats];
    __shared__ __attribute__((aligned(16))) float lds_w1[kW1TableFloats];
    __shared__ __attribute__((aligned(16))) float lds_q[8 * kQuarterFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* srcT = lds_src;
    float* buf = lds_q + wave * kQuarterFloats;

    static_assert(kSplitTableBytes == sizeof(float) * kW1TableFloats && kSplitImageBytes == sizeof(float) * kQuarterFloats, "LDS budget");
    int w1_exp = 0;
    if (SPLIT) {
        float m = 0.0f;
        for (int i = tid; i < 32 * 384; i += kDualThreads) m = fmaxf(m, fabsf(W1[i]));
        w1_exp = split_prescale_exp(block_absmax(m, lds_q, tid));
        stage_w1_split(reinterpret_cast<f16x8*>(lds_w1), W1, ldexpf(1.0f, w1_exp), tid, kDualThreads);
    } else {
        stage_w1_table(lds_w1, W1, tid, kDualThreads);
    }
    DualFrags f0;
    load_dual_frags(f0, W2, b2, lane);
    const long hstep = (long)gridDim.x * 8;

    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        DualFrags f = f0;
        if (SPLIT) {
            const int v_exp = stage_src_volume_scaled(lds_src, vol_src + (long)b * (16 * 512), lds_q, tid);
            const float unscale = ldexpf(1.0f, -(w1_exp + v_exp));  // exact; relu commutes with it
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f.a2[m][r][0] *= unscale;
                    f.a2[m][r][1] *= unscale;
                }
        } else {
            stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kDualThreads);
        }
        // Target features as the per-lane fragments the score needs, parked in the 16-byte pad of source
        // rows 0..511 (row (2t + m2)*64 + lane): 32 registers less per wave, and the 80-byte row stride
        // makes the eight ds_read_b128 of the epilogue conflict-free.
        {
            const float* ft = feat_tgt + (long)b * (32 * 64);
            const int l = tid & 63, t = tid >> 7, m2 = (tid >> 6) & 1;
            f32x4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ft[(16 * m2 + 4 * (l >> 4) + r) * 64 + 16 * t + (l & 15)];
            *reinterpret_cast<f32x4*>(lds_src + tid * kSrcStride + 16) = x;
        }
        __syncthreads();

#ifdef AHV_DUAL_STAGGER
        // Waves 4-7 (the second wave of each SIMD) start late, so that the two waves of a SIMD sit in
        // complementary phases (one gathers while the other contracts) instead of in lockstep.
        if (wave >= 4) {
#pragma unroll
            for (int i = 0; i < AHV_DUAL_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
        }
#endif
        unsigned long long best = 0ull;
        const float* Rb = R + (long)b * r_batch_stride;
        // Hypothesis h -> (workgroup h % gridDim.x, wave slot (h / gridDim.x) % 8): the last, partial round of
        // the persistent grid then spreads over ALL CUs with few waves each (a lone wave on a SIMD runs
        // ~1.6x faster than a pair) instead of filling some CUs completely and leaving the rest idle.
        long h = (long)wave * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        float Rn[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = Rb[(h < N ? h : 0) * 9 + i];
#ifdef AHV_STAMPS
        unsigned long long tsum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rn[i];
            {
                const long hn = (h + hstep < N) ? h + hstep : h;
#pragma unroll
                for (int i = 0; i < 9; ++i) Rn[i] = Rb[hn * 9 + i];
            }
            f32x4 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef AHV_STAMPS
            unsigned long long ts[11];
#define AHV_TS(i) AHV_STAMP(ts[i])
#else
#define AHV_TS(i)
#endif
            AHV_TS(0)
            if (SPLIT) tri_quarter_split<0>(reinterpret_cast<char*>(buf), srcT, Rm, lane);
            else tri_quarter<0>(buf, srcT, Rm, lane);
            wave_lds_fence();
            AHV_TS(1)
            if (SPLIT) gemm1_quarter_split<0>(acc, reinterpret_cast<const f16x8*>(lds_w1), reinterpret_cast<const char*>(buf), lane);
            else gemm1_quarter_lds<0>(acc, lds_w1, buf, lane);
            wave_lds_fence();
            AHV_TS(2)
            if (SPLIT) tri_quarter_split<1>(reinterpret_cast<char*>(buf), srcT, Rm, lane);
            else tri_quarter<1>(buf, srcT, Rm, lane);
            wave_lds_fence();
            AHV_TS(3)
            if (SPLIT) gemm1_quarter_split<1>(acc, reinterpret_cast<const f16x8*>(lds_w1), reinterpret_cast<const char*>(buf), lane);
            else gemm1_quarter_lds<1>(acc, lds_w1, buf, lane);
            wave_lds_fence();
            AHV_TS(4)
            if (SPLIT) tri_quarter_split<2>(reinterpret_cast<char*>(buf), srcT, Rm, lane);
            else tri_quarter<2>(buf, srcT, Rm, lane);
            wave_lds_fence();
            AHV_TS(5)
            if (SPLIT) gemm1_quarter_split<2>(acc, reinterpret_cast<const f16x8*>(lds_w1), reinterpret_cast<const char*>(buf), lane);
            else gemm1_quarter_lds<2>(acc, lds_w1, buf, lane);
            wave_lds_fence();
            AHV_TS(6)
            if (SPLIT) tri_quarter_split<3>(reinterpret_cast<char*>(buf), srcT, Rm, lane);
            else tri_quarter<3>(buf, srcT, Rm, lane);
            wave_lds_fence();
            AHV_TS(7)
            if (SPLIT) gemm1_quarter_split<3>(acc, reinterpret_cast<const f16x8*>(lds_w1), reinterpret_cast<const char*>(buf), lane);
            else gemm1_quarter_lds<3>(acc, lds_w1, buf, lane);
            wave_lds_fence();
            AHV_TS(8)

            f32x4 v[2][4];
            gemm2_dual(v, acc, f);
            AHV_TS(9)
            f32x4 tg[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
                    tg[t][m2] = *reinterpret_cast<const f32x4*>(lds_src + ((2 * t + m2) * 64 + lane) * kSrcStride + 16);
            const float s = hyp_score_rs(v, tg, lane);
            if (scores != nullptr && lane == 0) scores[(long)b * N + h] = s;
            const unsigned long long key = pack_key(s, (unsigned)(n_offset + h));
            best = key > best ? key : best;
            AHV_TS(10)
#ifdef AHV_STAMPS
            for (int i = 0; i < 10; ++i) tsum[i] += ts[i + 1] - ts[i];
            tsum[10] += 1;
#endif
        }
#ifdef AHV_STAMPS
        if (lane == 0) {
            const int gw = (blockIdx.x * 8 + wave) & 2047;
            for (int i = 0; i < 11; ++i) g_stamps[gw * 16 + i] = tsum[i];
        }
#endif
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

int g_score_variant = 3;  // 0 = 16x16x4 phase-sequential, 1 = 16x16x4 micro-step pipelined, 2 = 32x32x2 wide, 3 = dual: 2 waves/SIMD, W1 in LDS (default)

template <typename K>
static hipError_t launch_score_kernel(K kernel, size_t lds, dim3 grid, int threads, hipStream_t stream, const float* vol_src,
                                      const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                      int64_t n_offset, const float* W1, const float* W2, const float* b2, int B,
                                      int64_t N, float* scores, uint64_t* best_key)
{
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, stream, vol_src, feat_tgt, R, (long)r_batch_stride,
                       (long)n_offset, W1, W2, b2, B, (long)N, scores,
                       reinterpret_cast<unsigned long long*>(best_key));
    return hipGetLastError();
}

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
    const int waves_per_wg = (g_score_variant >= 3) ? 8 : 4;
    const int64_t need_w = (N + waves_per_wg - 1) / waves_per_wg;
    if (g_score_variant >= 3 && gx > need_w) gx = (int)(need_w < 1 ? 1 : need_w);
    const dim3 grid(gx, gy);
#define AHV_LAUNCH(K, L, T) \
    launch_score_kernel(K, L, grid, T, stream, vol_src, feat_tgt, R, r_batch_stride, n_offset, W1, W2, b2, B, N, scores, best_key)
    switch (g_score_variant) {
        case 0: return AHV_LAUNCH(score_hypotheses_kernel, lds, kScoreThreads);
        case 1: return AHV_LAUNCH(score_hypotheses_pipelined_kernel, 0, kScoreThreads);
        case 3: return AHV_LAUNCH(score_hypotheses_dual_kernel<false>, 0, kDualThreads);
        case 4: return AHV_LAUNCH(score_hypotheses_dual_kernel<true>, 0, kDualThreads);
        default: return AHV_LAUNCH(score_hypotheses_wide_kernel, 0, kScoreThreads);
    }
#undef AHV_LAUNCH
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
