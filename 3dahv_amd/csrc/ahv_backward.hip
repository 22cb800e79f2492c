// ahv_backward.hip -- backward of the fused scorer (what Estimator.infoNCE_loss,
// modules/model_co3d.py:41-61, needs from autograd): given dL/dscore[b][n], the gradients w.r.t. the source
// volume, the target feature and the head weights.  No gradient flows to the rotations (they are sampled).
//
//   scores[b][n] = 1/64 sum_pos <normalize(W2 relu(W1 slabs(rot(V_b, R_n))) + b2)[:, pos], tg_b[:, pos]>
//
// Three launches, nothing of the 32 KiB-per-hypothesis rotated volumes or 96 KiB slab tensors ever reaches HBM:
//   1.  score_backward_head_kernel   recomputes the forward of each hypothesis (same code as the forward
//       kernel), back-propagates through the score, the normalisation, GEMM2 and the ReLU and leaves
//       du = dL/du (32 x 64 floats per hypothesis) in the caller's workspace; accumulates d feat_tgt, d W2, d b2.
//   2a. score_backward_w1_kernel     re-gathers each quarter of the rotated volume, dW1 += du X^T in registers.
//   2b. score_backward_volume_kernel forms dX = W1^T du and scatters it through the trilinear weights into a
//       per-workgroup LDS image of dV that is flushed once per sample.
// All contractions are fp32 MFMA 16x16x4 (one float per lane and operand, so any LDS layout can feed them).
// Accumulation across waves/workgroups uses float atomics: gradients are reproducible to rounding, not bitwise.
// Accumulation targets are zeroed by launch_zero_fill (ahv_ops.hip), never by hipMemsetAsync (not graph-safe here).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ahv_device.h"
#ifndef AHV_DIAG_NO_FP32_LOW_HALF  // (tools/kbench_bwd A/B build of the unprotected kernels)
#define AHV_FP32_LOW_HALF  // these kernels leave registers free on their SIMDs: see low_half (ahv_dual.h)
#endif
#include "ahv_dual.h"

namespace ahv {

constexpr int kBwdThreads = 256;  // 4 waves, one per SIMD: 512 registers per wave

// max |du| per sample is what the LDS-atomic dV kernel of rounds 2-5 sizes its fixed-point scale with.  The shipped dV kernel
// (read-modify-write in fp32, round 6) does not need it, so the head kernels only compute it for the builds that launch the
// atomic kernel (-DAHV_BWD_VOLUME_ATOMICS, tools/kbench_bwd): ~30 vector instructions per position tile otherwise spent for nothing.
#if defined(AHV_BWD_VOLUME_ATOMICS) && !defined(AHV_BWD_DU_AMAX)
#define AHV_BWD_DU_AMAX 1
#endif

// u / du of one hypothesis in the workspace (2 048 floats): the MFMA accumulator layout of the scorer, tile by tile --
//   word(o, pos) = ((pos >> 4) * 2 + (o >> 4)) * 256 + (((o >> 2) & 3) * 16 + (pos & 15)) * 4 + (o & 3)
// i.e. [t][m][lane = 16 kq + n][r] for o = 16 m + 4 kq + r, pos = 16 t + n.  A wave writes and reads a (t, m) fragment with
// one 16-byte access per lane, the training forward's u and the head kernel's du share the words (du overwrites u tile
// by tile), and a lane that wants du[4 sp + kq][16 t + n] (kernels 2b) finds the 64 lanes' words in one 256-byte run.
__device__ __forceinline__ int du_word(int o, int pos)
{
    return ((pos >> 4) * 2 + (o >> 4)) * 256 + (((o >> 2) & 3) * 16 + (pos & 15)) * 4 + (o & 3);
}

// du image in LDS: du[o][pos] at o*64 + (pos ^ ((o & 7) << 2)).  The XOR keeps aligned groups of four
// positions together (float4 fills) and spreads rows over banks for the transposed reads (k = pos).
__device__ __forceinline__ int dimg(int o, int pos) { return o * 64 + (pos ^ ((o & 7) << 2)); }

__device__ __forceinline__ void lds_add_i64(long long* p, long long v)
{
#ifdef AHV_DIAG_NO_ATOMICS  // diagnostic build of tools/kbench_bwd only (wrong results): price of the LDS atomics
    asm volatile("" ::"v"(p), "v"(v));
#else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// |x| < 2^40 -> nearest integer as int64 (there is no f32 -> i64 convert): add 1.5 * 2^52 in fp64, where one ulp
// is exactly 1, and read the integer out of the mantissa.
__device__ __forceinline__ long long to_fixed(float x)
{
    const double magic = 6755399441055744.0;  // 2^52 + 2^51
    return __double_as_longlong((double)x + magic) - __double_as_longlong(magic);
}

// Corner weights and DENSE channel-last row offsets (16 words per voxel, voxel = z*64 + y*8 + x).
__device__ __forceinline__ void tri_coef_dense(float (&w)[8], int (&a)[8], const float* Rm, float x, float y, float z)
{
    const float gx = Rm[0] * x + Rm[1] * y + Rm[2] * z;
    const float gy = Rm[3] * x + Rm[4] * y + Rm[5] * z;
    const float gz = Rm[6] * x + Rm[7] * y + Rm[8] * z;
    float wx0, wx1, wy0, wy1, wz0, wz1;
    int ox0, ox1, oy0, oy1, oz0, oz1;
    axis_coef(gx, wx0, wx1, ox0, ox1, 16);
    axis_coef(gy, wy0, wy1, oy0, oy1, 8 * 16);
    axis_coef(gz, wz0, wz1, oz0, oz1, 64 * 16);
    const float w00 = wz0 * wy0, w01 = wz0 * wy1, w10 = wz1 * wy0, w11 = wz1 * wy1;
    w[0] = w00 * wx0; w[1] = w00 * wx1; w[2] = w01 * wx0; w[3] = w01 * wx1;
    w[4] = w10 * wx0; w[5] = w10 * wx1; w[6] = w11 * wx0; w[7] = w11 * wx1;
    const int a00 = oz0 + oy0, a01 = oz0 + oy1, a10 = oz1 + oy0, a11 = oz1 + oy1;
    a[0] = a00 + ox0; a[1] = a00 + ox1; a[2] = a01 + ox0; a[3] = a01 + ox1;
    a[4] = a10 + ox0; a[5] = a10 + ox1; a[6] = a11 + ox0; a[7] = a11 + ox1;
}

__device__ __forceinline__ void global_add(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// Kernel 1: forward recompute + backward through score / normalise / GEMM2 / ReLU.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBwdThreads, 1) void score_backward_head_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ feat_tgt, const float* __restrict__ R,
    long r_batch_stride, const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ b2,
    int B, long N, const float* __restrict__ grad_scores, float* __restrict__ du_ws,
    unsigned* __restrict__ du_max_bits, float* __restrict__ grad_feat_tgt, float* __restrict__ grad_W2, float* __restrict__ grad_b2)
{
    __shared__ __attribute__((aligned(1024))) float lds_src[kSrcFloats];  // at LDS address 0: see ahv_score.hip
    __shared__ __attribute__((aligned(16))) float lds_w1[kW1TableFloats];
    __shared__ __attribute__((aligned(16))) float lds_q[4 * kQuarterFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4, row = lane & 15;
    float* buf = lds_q + wave * kQuarterFloats;

    stage_w1_table(lds_w1, W1, tid, kBwdThreads);
    DualFrags f;
    load_dual_frags(f, W2, b2, lane);
    const GatherLane glane = gather_lane(lane);
    const GatherDst gdst = gather_dst_swizzled(lane);
    float a2t[2][4][2];  // dr = W2^T dv: A[row = o][k = o2]: [m2][r2][m] = W2[16 m2 + 4 kq + r2][16 m + row]
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
            for (int m = 0; m < 2; ++m) a2t[m2][r2][m] = W2[(16 * m2 + 4 * kq + r2) * 32 + 16 * m + row];

    f32x4 dW2[2][2];   // [mt][nt][r]: dW2[16 mt + 4 kq + r][16 nt + n]
    float db2p[2][4];  // [m2][r]: partial over this lane's columns of db2[16 m2 + 4 kq + r]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dW2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) db2p[i][r] = 0.0f;

    const long hstep = (long)gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kBwdThreads);
        __syncthreads();
        f32x4 tg[4][2], dtg[4][2];
        {
            const float* ft = feat_tgt + (long)b * (32 * 64);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tg[t][m2][r] = ft[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n];
                        dtg[t][m2][r] = 0.0f;
                    }
        }
        const float* Rb = R + (long)b * r_batch_stride;
        float du_amax = 0.0f;
        for (long h = (long)wave * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y); h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            f32x4 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            {   // forward recompute: the fused scorer's pipelined gather / GEMM1 (ahv_dual.h)
                GatherHyp gh;
                gather_hyp(gh, Rm, glane);
                HatState st;
                // quarters in the scorer's order 0, 3, 1, 2 (quarter 3 - Q is the point mirror of Q and reuses its set-up)
                hat_prologue<0>(st, lds_src, gh);
                hat_body(st, buf, gdst); wave_lds_fence();
                gemm1_quarter_pipe<0>(acc, lds_w1, buf, lane, [&] { hat_prologue_mirror(st, lds_src); }); wave_lds_fence();
                hat_body<true>(st, buf, gdst); wave_lds_fence();
                gemm1_quarter_pipe<3>(acc, lds_w1, buf, lane, [&] { hat_prologue<1>(st, lds_src, gh); }); wave_lds_fence();
                hat_body(st, buf, gdst); wave_lds_fence();
                gemm1_quarter_pipe<1>(acc, lds_w1, buf, lane, [&] { hat_prologue_mirror(st, lds_src); }); wave_lds_fence();
                hat_body<true>(st, buf, gdst); wave_lds_fence();
                gemm1_quarter_pipe<2>(acc, lds_w1, buf, lane, [] {}); wave_lds_fence();
            }
            f32x4 v[2][4];
            gemm2_dual(v, acc, f);

            // score = 1/64 sum_pos <v / max(|v|, eps), tg>; F.normalize's clamp passes no gradient to the norm
            // when it is below eps
            const float g = grad_scores[(long)b * N + h] * (1.0f / 64.0f);
            f32x4 dv[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float ss = 0.0f, dt = 0.0f;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ss += v[m2][t][r] * v[m2][t][r];
                        dt += v[m2][t][r] * tg[t][m2][r];
                    }
                ss += __shfl_xor(ss, 16, 64); dt += __shfl_xor(dt, 16, 64);
                ss += __shfl_xor(ss, 32, 64); dt += __shfl_xor(dt, 32, 64);
                const float nrm = sqrtf(ss);
                const bool clamped = nrm < 1e-12f;
                const float inv = 1.0f / fmaxf(nrm, 1e-12f);
                const float c3 = clamped ? 0.0f : dt * inv * inv * inv;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dv[m2][t][r] = g * (tg[t][m2][r] * inv - c3 * v[m2][t][r]);
                        dtg[t][m2][r] += g * inv * v[m2][t][r];
                    }
            }
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) db2p[m2][r] += dv[m2][0][r] + dv[m2][1][r] + dv[m2][2][r] + dv[m2][3][r];

            // dr = W2^T dv (accumulator registers of dv are the B operand), du = dr where u > 0
            f32x4 du[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) du[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        du[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[m2][r2][0], dv[m2][t][r2], du[0][t], 0, 0, 0);
                        du[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[m2][r2][1], dv[m2][t][r2], du[1][t], 0, 0, 0);
                    }
            float* dst = du_ws + ((long)b * N + h) * 2048;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = acc[m][t][r] > 0.0f ? du[m][t][r] : 0.0f;
                        dst[((t * 2 + m) * 64 + lane) * 4 + r] = x;
                        // max |du| of the sample (kernel 2b sizes its fixed-point scale with it); NaN / inf poison it
#ifdef AHV_BWD_DU_AMAX
                        du_amax = (x == x) ? fmaxf(du_amax, fabsf(x)) : __builtin_inff();
#endif
                    }

            // dW2 += dv relu(u)^T: the contraction runs over positions, so both operands go through the
            // wave's LDS image once (dv as A, relu(u) as B).
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[dimg(16 * m2 + 4 * kq + r, 16 * t + n)] = dv[m2][t][r];
            wave_lds_fence();
            float av[2][16];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 16; ++s) av[mt][s] = buf[dimg(16 * mt + row, 4 * s + kq)];
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[dimg(16 * m + 4 * kq + r, 16 * t + n)] = fmaxf(acc[m][t][r], 0.0f);
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float bv = buf[dimg(16 * nt + n, 4 * s + kq)];
                    dW2[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv, dW2[0][nt], 0, 0, 0);
                    dW2[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv, dW2[1][nt], 0, 0, 0);
                }
            wave_lds_fence();
        }
#ifdef AHV_BWD_DU_AMAX
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) du_amax = fmaxf(du_amax, __shfl_xor(du_amax, sft, 64));
        if (lane == 0) atomicMax(du_max_bits + b, __float_as_uint(du_amax));  // non-negative floats order like uints
#else
        (void)du_amax; (void)du_max_bits;
#endif
        float* gft = grad_feat_tgt + (long)b * (32 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) global_add(gft + (16 * m2 + 4 * kq + r) * 64 + 16 * t + n, dtg[t][m2][r]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) global_add(grad_W2 + (16 * mt + 4 * kq + r) * 32 + 16 * nt + n, dW2[mt][nt][r]);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = db2p[m2][r];
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
            if (n == 0) global_add(grad_b2 + 16 * m2 + 4 * kq + r, x);
        }
}

// ---------------------------------------------------------------------------------------------------
// Kernel 1', round 6: the same backward through score / normalise / GEMM2 / ReLU WITHOUT the forward recompute -- u comes from
// the workspace, where the training forward (ahv_score_hypotheses_train_f32) left it in the fragment layout
// [t][m][lane][r] (du_word).  That removes gather + GEMM1 (832 of 960 MFMAs and ~900 of 2 547 vector instructions per hypothesis) from
// the backward for 8 KB more HBM traffic per hypothesis in each direction.
// Everything here is independent per position tile t (16 of the 64 positions): a hypothesis is walked tile by tile with
// ~90 transient registers (the whole-hypothesis form of kernel 1 keeps 6 x 32 alive and ran one wave per SIMD at 506
// registers: 2.2 ms at B = 32 x 9 000 for 16 KB of traffic and 192 MFMAs per hypothesis).  The per-sample state that would
// need a dynamic register index moves to LDS: the target fragments (shared, read-only) and the wave's d feat_tgt accumulator
// (lane-linear 16-byte read-modify-writes).  512 threads: two waves per SIMD.
// ---------------------------------------------------------------------------------------------------
constexpr int kSavedThreads = 512;

__global__ __launch_bounds__(kSavedThreads, 2) void score_backward_head_saved_kernel(
    const float* __restrict__ feat_tgt, const float* __restrict__ W2, const float* __restrict__ b2,
    int B, long N, const float* __restrict__ grad_scores, float* __restrict__ du_ws,
    unsigned* __restrict__ du_max_bits, float* __restrict__ grad_feat_tgt, float* __restrict__ grad_W2, float* __restrict__ grad_b2)
{
    __shared__ __attribute__((aligned(16))) float lds_tg[4 * 2 * 64 * 4];        // target fragments [t][m2][lane][r]
    __shared__ __attribute__((aligned(16))) float lds_dtg[8 * 4 * 2 * 64 * 4];   // per wave: d feat_tgt, same layout
    __shared__ __attribute__((aligned(16))) float lds_tr[8 * 2 * 32 * 20];       // per wave: dv tile and relu(u) tile, [o][16 pos + 4 pad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4, row = lane & 15;
    float* dtg = lds_dtg + wave * (4 * 2 * 64 * 4) + lane * 4;
    float* trd = lds_tr + wave * (2 * 32 * 20);   // dv tile
    float* tru = trd + 32 * 20;                   // relu(u) tile

    DualFrags f;
    load_dual_frags(f, W2, b2, lane);
    float a2t[2][4][2];  // dr = W2^T dv: A[row = o][k = o2]: [m2][r2][m] = W2[16 m2 + 4 kq + r2][16 m + row]
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
            for (int m = 0; m < 2; ++m) a2t[m2][r2][m] = W2[(16 * m2 + 4 * kq + r2) * 32 + 16 * m + row];
    f32x4 dW2[2][2];   // [mt][nt][r]: dW2[16 mt + 4 kq + r][16 nt + n]
    float db2p[2][4];  // [m2][r]: partial over this lane's columns of db2[16 m2 + 4 kq + r]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dW2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) db2p[i][r] = 0.0f;

    const long hstep = (long)gridDim.x * 8;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        {   // target fragments of the sample: tg[t][m2][r] = ft[16 m2 + 4 kq + r][16 t + n], one (t, m2) per wave
            const float* ft = feat_tgt + (long)b * (32 * 64);
            const int t = wave >> 1, m2 = wave & 1;
            f32x4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ft[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n];
            *reinterpret_cast<f32x4*>(lds_tg + ((t * 2 + m2) * 64 + lane) * 4) = x;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dtg + i * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        float du_amax = 0.0f;
        long h = (long)wave * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        // u of (hypothesis, tile): two 16-byte loads per lane, requested a tile ahead (the last tile requests the next
        // hypothesis' first).  du overwrites u tile by tile, word for word (du_word): a tile's loads are consumed before its
        // stores are issued, and the other tiles live in other words.
        f32x4 un0 = f32x4{0.f, 0.f, 0.f, 0.f}, un1 = un0;
        if (h < N) {
            const float* up = du_ws + ((long)b * N + h) * 2048 + lane * 4;
            un0 = *reinterpret_cast<const f32x4*>(up);
            un1 = *reinterpret_cast<const f32x4*>(up + 256);
        }
        for (; h < N; h += hstep) {
            const float g = grad_scores[(long)b * N + h] * (1.0f / 64.0f);
            float* dst = du_ws + ((long)b * N + h) * 2048;
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                const f32x4 a0 = un0, a1 = un1;   // u[16 m + 4 kq + r][16 t + n], m = 0 / 1
                {
                    const bool last = t == 3;
                    const long hn = last ? (h + hstep < N ? h + hstep : h) : h;
                    const float* up = du_ws + ((long)b * N + hn) * 2048 + lane * 4 + (last ? 0 : t + 1) * 512;
                    if (!last || hn != h) {   // (never re-read this hypothesis' first tile: it already holds du)
                        un0 = *reinterpret_cast<const f32x4*>(up);
                        un1 = *reinterpret_cast<const f32x4*>(up + 256);
                    }
                }
                // v = W2 relu(u) + b2 for this tile
                f32x4 v0 = f.bias[0], v1 = f.bias[1];
                const f32x4 ru0 = relu4(a0), ru1 = relu4(a1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[0][r][0], ru0[r], v0, 0, 0, 0);
                    v1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[0][r][1], ru0[r], v1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[1][r][0], ru1[r], v0, 0, 0, 0);
                    v1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[1][r][1], ru1[r], v1, 0, 0, 0);
                }
                // score = 1/64 sum_pos <v / max(|v|, eps), tg>; F.normalize's clamp passes no gradient to the norm below eps
                const f32x4 tg0 = *reinterpret_cast<const f32x4*>(lds_tg + ((t * 2 + 0) * 64 + lane) * 4);
                const f32x4 tg1 = *reinterpret_cast<const f32x4*>(lds_tg + ((t * 2 + 1) * 64 + lane) * 4);
                float ss = 0.0f, dt = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ss += v0[r] * v0[r] + v1[r] * v1[r];
                    dt += v0[r] * tg0[r] + v1[r] * tg1[r];
                }
                ss += __shfl_xor(ss, 16, 64); dt += __shfl_xor(dt, 16, 64);
                ss += __shfl_xor(ss, 32, 64); dt += __shfl_xor(dt, 32, 64);
                const float nrm = sqrtf(ss);
                const bool clamped = nrm < 1e-12f;
                const float inv = 1.0f / fmaxf(nrm, 1e-12f);
                const float c3 = clamped ? 0.0f : dt * inv * inv * inv;
                f32x4 dv0, dv1;
                {
                    f32x4 d0 = *reinterpret_cast<const f32x4*>(dtg + (t * 2 + 0) * 256);
                    f32x4 d1 = *reinterpret_cast<const f32x4*>(dtg + (t * 2 + 1) * 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dv0[r] = g * (tg0[r] * inv - c3 * v0[r]);
                        dv1[r] = g * (tg1[r] * inv - c3 * v1[r]);
                        d0[r] += g * inv * v0[r];
                        d1[r] += g * inv * v1[r];
                        db2p[0][r] += dv0[r];
                        db2p[1][r] += dv1[r];
                    }
                    *reinterpret_cast<f32x4*>(dtg + (t * 2 + 0) * 256) = d0;
                    *reinterpret_cast<f32x4*>(dtg + (t * 2 + 1) * 256) = d1;
                }
                // dr = W2^T dv (the accumulator registers of dv are the B operand), du = dr where u > 0
                f32x4 du0 = f32x4{0.f, 0.f, 0.f, 0.f}, du1 = du0;
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) {
                    du0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[0][r2][0], dv0[r2], du0, 0, 0, 0);
                    du1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[0][r2][1], dv0[r2], du1, 0, 0, 0);
                }
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) {
                    du0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[1][r2][0], dv1[r2], du0, 0, 0, 0);
                    du1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[1][r2][1], dv1[r2], du1, 0, 0, 0);
                }
                f32x4 x0, x1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (u != u: the forward's exact path saved NaN for a sample with a non-finite voxel or weight)
                    x0[r] = a0[r] > 0.0f ? du0[r] : (a0[r] == a0[r] ? 0.0f : a0[r]);
                    x1[r] = a1[r] > 0.0f ? du1[r] : (a1[r] == a1[r] ? 0.0f : a1[r]);
                    // max |du| of the sample (kernel 2b's LDS-atomic form sizes its fixed-point scale with it); NaN / inf poison it
#ifdef AHV_BWD_DU_AMAX
                    du_amax = (x0[r] == x0[r]) ? fmaxf(du_amax, fabsf(x0[r])) : __builtin_inff();
                    du_amax = (x1[r] == x1[r]) ? fmaxf(du_amax, fabsf(x1[r])) : __builtin_inff();
#endif
                }
                *reinterpret_cast<f32x4*>(dst + t * 512 + lane * 4) = x0;        // du over the tile's u, same words
                *reinterpret_cast<f32x4*>(dst + t * 512 + 256 + lane * 4) = x1;
                // dW2 += dv relu(u)^T over the tile's 16 positions: both operands through the wave's LDS tiles [o][pos]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    trd[(4 * kq + r) * 20 + n] = dv0[r];
                    trd[(16 + 4 * kq + r) * 20 + n] = dv1[r];
                    tru[(4 * kq + r) * 20 + n] = ru0[r];
                    tru[(16 + 4 * kq + r) * 20 + n] = ru1[r];
                }
                wave_lds_fence();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {   // k-step: positions 4 sp + kq
                    const float av0 = trd[row * 20 + 4 * sp + kq], av1 = trd[(16 + row) * 20 + 4 * sp + kq];
                    const float bv0 = tru[n * 20 + 4 * sp + kq], bv1 = tru[(16 + n) * 20 + 4 * sp + kq];
                    dW2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0, dW2[0][0], 0, 0, 0);
                    dW2[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv1, dW2[0][1], 0, 0, 0);
                    dW2[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv0, dW2[1][0], 0, 0, 0);
                    dW2[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1, dW2[1][1], 0, 0, 0);
                }
                wave_lds_fence();
            }
        }
#ifdef AHV_BWD_DU_AMAX
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) du_amax = fmaxf(du_amax, __shfl_xor(du_amax, sft, 64));
        if (lane == 0) atomicMax(du_max_bits + b, __float_as_uint(du_amax));  // non-negative floats order like uints
#else
        (void)du_amax; (void)du_max_bits;
#endif
        float* gft = grad_feat_tgt + (long)b * (32 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dtg + (t * 2 + m2) * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r) global_add(gft + (16 * m2 + 4 * kq + r) * 64 + 16 * t + n, d[r]);
            }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) global_add(grad_W2 + (16 * mt + 4 * kq + r) * 32 + 16 * nt + n, dW2[mt][nt][r]);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = db2p[m2][r];
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
            if (n == 0) global_add(grad_b2 + 16 * m2 + 4 * kq + r, x);
        }
}

// ---------------------------------------------------------------------------------------------------
// Kernels 2a / 2b.  Both walk the hypotheses again with du from the workspace.  They are separate launches
// because each carries persistent state per wave (2a: the dW1 accumulators, 2b: the W1^T fragments; 96 registers
// each since the outputs are split over the two waves of a SIMD) next to a gather / scatter that wants ~100 more:
// together they spill, apart they do not.
//   2a  score_backward_w1_kernel      re-gathers each quarter X of the rotated volume, dW1 += du X^T
//   2b  score_backward_volume_kernel  dX = W1^T du per quarter, dV += trilinear^T dX in an LDS image of the
//                                     sample's volume gradient, flushed once per sample
// ---------------------------------------------------------------------------------------------------

// LDS images of 2a, both LINEAR with a small pad so that every access is "per-lane base + compile-time constant"
// (one address register, immediate offsets) and the MFMA operand reads are at worst 2-way bank conflicts:
//   X[c][voxel = a0*64 + b*8 + e], kXwStride = 129 floats per channel plane (odd: the z slab's B operand, 16 lanes =
//   16 channels of one voxel, lands on 16 different banks);
//   du[o][pos], kDuStride = 68 floats per row (rows 4 apart sit 16 banks apart: the fragment-wise fill is conflict-free).
constexpr int kXwStride = 129;
constexpr int kXwFloats = 16 * kXwStride;
constexpr int kDuStride = 68;
constexpr int kDuFloats = 32 * kDuStride;

// Kernel 2a runs TWO waves per SIMD.  The 192 accumulator registers of dW1 are what kept it at one wave per
// SIMD, where every LDS round trip and every VALU burst is exposed (the no-MFMA build of the one-wave kernel still
// took 0.75 of its 1.0 ms).  The OUTPUT is split instead: wave role r = wave / 4 owns the dW1 rows of m-tile r
// (96 accumulator registers) and needs only du rows 16 r .. 16 r + 15; waves w and w + 4 (same SIMD) walk the
// same hypotheses independently -- no barrier, each with its own X image -- so each hypothesis is gathered twice
// (the blend is ~1/6 of the per-hypothesis work) and in exchange the two waves of a SIMD drift apart and cover
// each other's latencies exactly as in the forward kernel.  (Sharing one gather per pair behind workgroup
// barriers was tried first: the barriers put both waves of a SIMD in the same phase and it ran 10 % SLOWER than
// one wave per SIMD.)
template <int DEPTH>
struct HatRing {
    f32x4 slot[DEPTH][4];
};

template <int S, int DEPTH>
__device__ __forceinline__ void hat_pass_request(HatRing<DEPTH>& rg, const HatVoxel& vx)
{
    const f32x4* row = reinterpret_cast<const f32x4*>(vx.base + hat_off(S & 7));
#pragma unroll
    for (int j = 0; j < 4; ++j) rg.slot[S % DEPTH][j] = row[j];
}

template <int S, int DEPTH>
struct HatPassSteps {  // the 8 corner steps of ONE voxel, source rows requested DEPTH steps ahead (cf. HatSteps, ahv_dual.h)
    static __device__ __forceinline__ void run(HatRing<DEPTH>& rg, const HatVoxel& vx, f32x2 (&o)[8])
    {
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 wn = {vx.w[S], vx.w[S]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = rg.slot[S % DEPTH][j];
            const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
            if (S == 0) {
                o[2 * j] = lo * wn;
                o[2 * j + 1] = hi * wn;
            } else {
                o[2 * j] = __builtin_elementwise_fma(lo, wn, o[2 * j]);
                o[2 * j + 1] = __builtin_elementwise_fma(hi, wn, o[2 * j + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (S + DEPTH < 8) hat_pass_request<(S + DEPTH < 8 ? S + DEPTH : 0), DEPTH>(rg, vx);
        HatPassSteps<S + 1, DEPTH>::run(rg, vx, o);
    }
};
template <int DEPTH>
struct HatPassSteps<8, DEPTH> {
    static __device__ __forceinline__ void run(HatRing<DEPTH>&, const HatVoxel&, f32x2 (&)[8]) {}
};

template <int DEPTH, int ROW>
__device__ __forceinline__ void hat_one_pass(const HatVoxel& vx, float* dst)
{
    HatRing<DEPTH> rg;
    f32x2 o[8];
    hat_pass_request<0, DEPTH>(rg, vx);
    if (DEPTH > 1) hat_pass_request<(DEPTH > 1 ? 1 : 0), DEPTH>(rg, vx);
    if (DEPTH > 2) hat_pass_request<(DEPTH > 2 ? 2 : 0), DEPTH>(rg, vx);
    if (DEPTH > 3) hat_pass_request<(DEPTH > 3 ? 3 : 0), DEPTH>(rg, vx);
    static_assert(DEPTH <= 4, "prologue written out for up to four slots");
    HatPassSteps<0, DEPTH>::run(rg, vx, o);
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[c * ROW] = o[c >> 1][c & 1];
}

// dW1 rows of ONE m-tile += du X^T on quarter Q, software-pipelined one chunk deep (operands of chunk k+1 requested
// before the MFMAs of chunk k; sched_barrier keeps hipcc from sinking the reads back to their uses).
//   chunks 0-7:  x and y slabs, k-step s = K / 2 (positions pl = 4 s + kq of tile Q), four k-tiles: 1 A value,
//                8 B values (4 k-tiles x {x, y}), 8 MFMAs
//   chunks 8-11: z slab, k-steps s = 4 (K-8) .. +3 (all 64 positions): 4 A values, 8 B values, 8 MFMAs
struct W1Chunk {
    float a[4];
    float b[8];
};
constexpr int kW1Chunks = 12;  // 8 half-steps of the x / y slabs + 4 groups of the z slab

template <int Q, int K>
__device__ __forceinline__ void w1_load(W1Chunk& ck, const float* da, const float* bx, const float* by, const float* bz)
{
    // da = dbuf + (16 m + row)*kDuStride + kq : A[row = o][k = pos];   bx = xbuf + i0*kXwStride + 8 kq + j;
    // by = xbuf + i0*kXwStride + kq + 8 j;   bz = xbuf + n*kXwStride + kq
    if (K < 8) {
        constexpr int s = K >> 1, k0 = 4 * (K & 1);  // k-step s (pa = s >> 1, pb = 4 (s & 1) + kq), k-tiles k0 .. k0 + 3
        ck.a[0] = da[16 * Q + 4 * s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ck.b[2 * i] = bx[2 * (k0 + i) * kXwStride + (s >> 1) * 64 + 32 * (s & 1)];     // X[c = 2 kt + i0][(pa, pb, e = j)]
            ck.b[2 * i + 1] = by[2 * (k0 + i) * kXwStride + (s >> 1) * 64 + 4 * (s & 1)];  // X[c = 2 kt + i0][(pa, b = j, pb)]
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = 4 * (K - 8) + i;
            ck.a[i] = da[4 * s];
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0) ck.b[2 * i + a0] = bz[a0 * 64 + 4 * s];  // X[c = n][voxel = a0*64 + 4 s + kq]
        }
    }
}

template <int Q, int K>
__device__ __forceinline__ void w1_mfma(f32x4 (&ax)[8], f32x4 (&ay)[8], f32x4 (&az)[4][2], const W1Chunk& ck)
{
#ifdef AHV_DIAG_W1_NO_MFMA  // diagnostic build of tools/kbench_bwd only (wrong results)
    asm volatile("" ::"v"(ck.a[0]), "v"(ck.a[3]), "v"(ck.b[0]), "v"(ck.b[1]), "v"(ck.b[7]));
#else
    if (K < 8) {
        constexpr int k0 = 4 * (K & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ax[k0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0], ck.b[2 * i], ax[k0 + i], 0, 0, 0);
            ay[k0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0], ck.b[2 * i + 1], ay[k0 + i], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0)
                az[Q][a0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i], ck.b[2 * i + a0], az[Q][a0], 0, 0, 0);
    }
#endif
}

template <int Q, int K>
struct W1Pipe {
    static __device__ __forceinline__ void run(f32x4 (&ax)[8], f32x4 (&ay)[8], f32x4 (&az)[4][2], W1Chunk& cur,
                                               const float* da, const float* bx, const float* by, const float* bz)
    {
        W1Chunk nxt;
        __builtin_amdgcn_sched_barrier(0);
        if (K + 1 < kW1Chunks) w1_load<Q, K + 1>(nxt, da, bx, by, bz);
        w1_mfma<Q, K>(ax, ay, az, cur);
        if (K + 1 < kW1Chunks) {  // the next chunk's LDS reads between this chunk's MFMAs (ahv_dual.h, G1Pipe)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (K + 1 < kW1Chunks) W1Pipe<Q, K + 1>::run(ax, ay, az, nxt, da, bx, by, bz);
    }
};
template <int Q>
struct W1Pipe<Q, kW1Chunks> {
    static __device__ __forceinline__ void run(f32x4 (&)[8], f32x4 (&)[8], f32x4 (&)[4][2], W1Chunk&, const float*,
                                               const float*, const float*, const float*) {}
};

template <int Q>
__device__ __forceinline__ void bwd_w1_quarter(f32x4 (&ax)[8], f32x4 (&ay)[8], f32x4 (&az)[4][2], const float* dbuf,
                                               const float* xbuf, int lane)
{
    const int n = lane & 15, kq = lane >> 4, row = lane & 15;
    const int i0 = n >> 3, j = n & 7;
    const float* da = dbuf + row * kDuStride + kq;  // the wave's 16 du rows
    const float* bx = xbuf + i0 * kXwStride + 8 * kq + j;
    const float* by = xbuf + i0 * kXwStride + kq + 8 * j;
    const float* bz = xbuf + n * kXwStride + kq;
    W1Chunk first;
    w1_load<Q, 0>(first, da, bx, by, bz);
    W1Pipe<Q, 0>::run(ax, ay, az, first, da, bx, by, bz);
}

// quarter Q of the rotated volume into this wave's X image, one pass (voxel) after the other
template <int Q>
__device__ __forceinline__ void w1_gather(float* xbuf, const float* srcT, const GatherHyp& gh, const GatherDst& dst)
{
#ifdef AHV_DIAG_W1_NO_GATHER  // diagnostic build of tools/kbench_bwd only (wrong results)
    asm volatile("" ::"v"(xbuf), "v"(dst.o0));
#else
    float* d0 = xbuf + dst.o0;
    __builtin_amdgcn_s_setprio(1);  // the gathering wave is latency-bound, its partner streams MFMAs (ahv_dual.h)
    HatVoxel vx;
    hat_voxel<Q>(vx, srcT, gh, 0);
    hat_one_pass<3, kXwStride>(vx, d0);
    hat_voxel<Q>(vx, srcT, gh, 1);
    hat_one_pass<3, kXwStride>(vx, d0 + 32);
    __builtin_amdgcn_s_setprio(0);
#endif
}

constexpr int kW1Threads = 512;
constexpr int kDuHalfFloats = 16 * kDuStride;

__global__ __launch_bounds__(kW1Threads, 2) void score_backward_w1_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ R, long r_batch_stride, int B, long N,
    const float* __restrict__ du_ws, float* __restrict__ dw1_partials)
{
    __shared__ __attribute__((aligned(1024))) float lds_src[kSrcFloats];  // at LDS address 0: see ahv_score.hip
    // per wave: du rows 16 role .. 16 role + 15 (local rows 0 .. 15) and the X image; after the hypothesis loop the
    // same pool is the scratch of the in-workgroup reduction of the accumulators
    __shared__ __attribute__((aligned(16))) float lds_pool[8 * kDuHalfFloats + 8 * kXwFloats];
    static_assert(8 * kDuHalfFloats + 8 * kXwFloats >= 4 * 96 * 64, "reduction scratch");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave & 3, role = wave >> 2;  // waves w and w + 4 share a SIMD and a hypothesis list
    const int n = lane & 15, kq = lane >> 4;
    float* dbuf = lds_pool + wave * kDuHalfFloats;
    float* xbuf = lds_pool + 8 * kDuHalfFloats + wave * kXwFloats;
    const GatherLane glane = gather_lane(lane);
    const GatherDst gdst = gather_dst_linear(lane);
    f32x4 ax[8], ay[8], az[4][2];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        ax[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        ay[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int a0 = 0; a0 < 2; ++a0) az[q][a0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long hstep = (long)gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kW1Threads);
        __syncthreads();
        const float* Rb = R + (long)b * r_batch_stride;
        long h = (long)slot * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        // this wave's half of du: the four fragments (t, m = role) of the hypothesis's 2048 floats (du_word) = rows 16 role ..
        f32x4 duh[4];
        if (h < N) {
            const f32x4* src = reinterpret_cast<const f32x4*>(du_ws + ((long)b * N + h) * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) duh[i] = __builtin_nontemporal_load(src + (2 * i + role) * 64 + lane);
        }
        for (; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i)  // this hypothesis's du rows, requested one iteration ago: fragment (t = i, m = role)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbuf[(4 * kq + r) * kDuStride + 16 * i + n] = duh[i][r];  // banks 16 kq + n + const
            {   // the next hypothesis's travel meanwhile
                const long hn = (h + hstep < N) ? h + hstep : h;
                const f32x4* src = reinterpret_cast<const f32x4*>(du_ws + ((long)b * N + hn) * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) duh[i] = __builtin_nontemporal_load(src + (2 * i + role) * 64 + lane);
            }
            GatherHyp gh;
            gather_hyp(gh, Rm, glane);
            w1_gather<0>(xbuf, lds_src, gh, gdst); wave_lds_fence();
            bwd_w1_quarter<0>(ax, ay, az, dbuf, xbuf, lane); wave_lds_fence();
            w1_gather<1>(xbuf, lds_src, gh, gdst); wave_lds_fence();
            bwd_w1_quarter<1>(ax, ay, az, dbuf, xbuf, lane); wave_lds_fence();
            w1_gather<2>(xbuf, lds_src, gh, gdst); wave_lds_fence();
            bwd_w1_quarter<2>(ax, ay, az, dbuf, xbuf, lane); wave_lds_fence();
            w1_gather<3>(xbuf, lds_src, gh, gdst); wave_lds_fence();
            bwd_w1_quarter<3>(ax, ay, az, dbuf, xbuf, lane); wave_lds_fence();
        }
    }
    // Reduce the four waves of each role inside the workgroup (two rounds through LDS), then ONE wave per role
    // writes the workgroup's partial dW1 rows to the workspace; score_backward_w1_reduce_kernel sums the partials.
    // (Flushing every wave's 96 accumulator registers with float atomics put 12.6 M atomic adds on 12 288
    // addresses: 0.37 ms of a 1.07-ms kernel, independent of N.)
    float* scratch = lds_pool;
    auto put = [&](int w) {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                scratch[((w * 96) + 4 * kt + r) * 64 + lane] = ax[kt][r];
                scratch[((w * 96) + 32 + 4 * kt + r) * 64 + lane] = ay[kt][r];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0)
#pragma unroll
                for (int r = 0; r < 4; ++r) scratch[((w * 96) + 64 + 4 * (2 * q + a0) + r) * 64 + lane] = az[q][a0][r];
    };
    auto get = [&](int w) {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ax[kt][r] += scratch[((w * 96) + 4 * kt + r) * 64 + lane];
                ay[kt][r] += scratch[((w * 96) + 32 + 4 * kt + r) * 64 + lane];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0)
#pragma unroll
                for (int r = 0; r < 4; ++r) az[q][a0][r] += scratch[((w * 96) + 64 + 4 * (2 * q + a0) + r) * 64 + lane];
    };
    __syncthreads();
    if (slot >= 2) put((slot - 2) + 2 * role);
    __syncthreads();
    if (slot < 2) get(slot + 2 * role);
    __syncthreads();
    if (slot == 1) put(role);
    __syncthreads();
    if (slot == 0) {
        get(role);
        // dW1[o = 16 role + 4 kq + r][k]: x: k = 16 kt + n; y: 128 + 16 kt + n; z: 256 + n*8 + 2 q + a0
        float* part = dw1_partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (32 * 384);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* g = part + (16 * role + 4 * kq + r) * 384;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                g[16 * kt + n] = ax[kt][r];
                g[128 + 16 * kt + n] = ay[kt][r];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a0 = 0; a0 < 2; ++a0) g[256 + n * 8 + 2 * q + a0] = az[q][a0][r];
        }
    }
}

// grad_W1[i] += sum over the workgroups' partials; blockIdx.y slices the workgroups so that the 50 MB-scale read is
// spread over the chip (grad_W1 is zeroed beforehand; 16 float atomics per element).
__global__ __launch_bounds__(256) void score_backward_w1_reduce_kernel(const float* __restrict__ partials, int count,
                                                                       float* __restrict__ grad_W1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = (count + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per, hi = min(lo + per, count);
    float acc = 0.0f;
    for (int w = lo; w < hi; ++w) acc += partials[(size_t)w * (32 * 384) + i];
    if (hi > lo) global_add(grad_W1 + i, acc);
}

// ---- kernel 2b building blocks -------------------------------------------------------------------------
// dX image of kernel 2b: dX[c][voxel], voxel = a0*64 + b*8 + e, rows kVxStride floats apart.  Deliberately NOT
// the XOR-swizzled layout of the forward's quarter image: here every access is "per-lane base + compile-time
// constant" (MFMA-result stores, their read-modify-writes and the scatter's reads), so the 16 stores / loads of
// an MFMA group cost one address register and immediate offsets.  With the swizzle the lane-dependent XOR had
// to be recomputed for each of them: ~80 VALU instructions per group of 16 MFMAs, more issue time than the
// MFMAs' own (fp32 MFMA and VALU do not overlap).  The 4-float pad spreads the four k-quads of a store over
// the banks; the residual 4-way conflicts of the x / y read-modify-writes cost LDS cycles nobody waits for.
constexpr int kVxStride = 132;
constexpr int kVxFloats = 16 * kVxStride;

// MFMA with the A operand taken straight from an ACCUMULATOR register.  Kernel 2b keeps the 192 W1^T fragments
// of a lane resident for the whole launch; with one wave per SIMD they fit the 256 AGPRs, but hipcc treats AGPRs
// as spill space and re-reads every fragment through v_accvgpr_read (+ hazard nops) in front of its MFMA --
// 768 extra VALU-slot instructions per hypothesis that the matrix pipe cannot overlap (measured: 0.79 ms of a
// 1.65-ms kernel for 0.38 ms of MFMA work).  The "a" constraint makes the fragment an AGPR operand of the MFMA
// itself (legal on gfx90a+: SrcA/SrcB may be AGPRs).  hipcc neither sees the instruction inside the statement nor
// pads its hazards, so the statement does: `s_nop 1` in front covers a VALU write of an operand immediately
// before it (2 wait states), and the LAST MFMA of an accumulation chain is followed by 12 wait states before
// anything may read D (8-pass MFMA -> VALU / LDS reader).  Back-to-back MFMAs on the same accumulator need none.
__device__ __forceinline__ void mfma_areg(f32x4& d, float a_in_agpr, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d) : "a"(a_in_agpr), "v"(b));
}

__device__ __forceinline__ void mfma_chain_end(f32x4& d0, f32x4& d1)
{
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(d0), "+v"(d1));
}

// du of one hypothesis as MFMA B operands, straight from the workspace into registers: lane (n, kq) holds
// du[o = 4 sp + kq][pos = 16 t + n] for sp < 8, t < 4 -- 32 floats.  The x / y slabs of quarter Q use tile t = Q,
// the z slab all four tiles, so no LDS image of du is needed in this kernel.
struct DuRegs {
    float v[4][8];
};

__device__ __forceinline__ void load_du_regs(DuRegs& d, const float* __restrict__ du_hyp, int lane)
{
    // o = 4 sp + kq = 16 m + 4 kq' + r  =>  m = sp >> 2, kq' = sp & 3, r = kq: word (2 t + m) * 256 + (16 (sp & 3) + n) * 4 + kq
    const float* p = du_hyp + (lane & 15) * 4 + (lane >> 4);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) d.v[t][sp] = p[(2 * t + (sp >> 2)) * 256 + (sp & 3) * 64];
}

// Kernel 2b runs TWO waves per SIMD, split by CHANNEL: wave role r = wave / 4 owns channels 8 r .. 8 r + 7 of
// dV.  Everything downstream of du splits cleanly along that axis -- the rows of dX = W1^T du are (channel,
// slab index), the scatter adds channel by channel -- so the pair shares nothing but the (atomic) dV image:
//   * W1^T fragments per wave: x and y slabs k-tiles 4 r .. 4 r + 3 (channel = 2 kt + (kq >> 1)), z slab with the
//     tile rows re-packed as (a0, channel - 8 r): 32 + 32 + 32 = 96 accumulator-file registers instead of 192,
//     384 MFMAs per wave and hypothesis instead of 768, none wasted;
//   * dX image per wave: 8 channels x 128 voxels; scatter lanes = 8 channels x 8 voxels per step.
// Waves w and w + 4 (same SIMD) walk the same hypotheses independently, like kernel 2a.  The one-wave-per-SIMD
// form of this kernel (1.44 ms) was the sum of its parts: 0.47 ms skeleton + 0.26 ms atomics + 0.69 ms for
// 0.38 ms worth of MFMAs, nothing overlapping anything.
constexpr int kVhFloats = 8 * kVxStride;  // half-channel dX image

// one MFMA output tile = one accumulator fed by 8 k-steps; two tiles side by side (see mfma_areg)
template <int Q>
__device__ __forceinline__ void bwd_vol_dx(const float (&wx)[4][8], const float (&wy)[4][8], const float (&wz)[4][8],
                                           const DuRegs& du, float* xbuf, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    // z slab: tile rows = (a0' = row >> 3, c8 = row & 7); row = 4 kq + r; column = position (b = 2 t + i0, e = j)
    {
        float* zb = xbuf + 4 * (kq & 1) * kVxStride + (kq >> 1) * 64 + 8 * i0 + j;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                mfma_areg(d0, wz[Q][sp], du.v[t][sp]);
                mfma_areg(d1, wz[Q][sp], du.v[t + 1][sp]);
            }
            mfma_chain_end(d0, d1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zb[r * kVxStride + 16 * t] = d0[r];
                zb[r * kVxStride + 16 * (t + 1)] = d1[r];
            }
        }
    }
    wave_lds_fence();
    // x: local channel = 2 kt + (kq >> 1), voxel = i0*64 + j*8 + 4 (kq & 1) + r;  y: voxel = i0*64 + (4 (kq & 1) + r)*8 + j
#pragma unroll
    for (int slab = 0; slab < 2; ++slab) {
        float* rb = xbuf + (kq >> 1) * kVxStride + 64 * i0 + (slab == 0 ? 8 * j + 4 * (kq & 1) : 32 * (kq & 1) + j);
        const int rs = slab == 0 ? 1 : 8;
#pragma unroll
        for (int kt = 0; kt < 4; kt += 2) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
            float o0[4], o1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[r] = rb[2 * kt * kVxStride + rs * r];
                o1[r] = rb[(2 * kt + 2) * kVxStride + rs * r];
            }
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                mfma_areg(d0, slab == 0 ? wx[kt][sp] : wy[kt][sp], du.v[Q][sp]);
                mfma_areg(d1, slab == 0 ? wx[kt + 1][sp] : wy[kt + 1][sp], du.v[Q][sp]);
            }
            mfma_chain_end(d0, d1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rb[2 * kt * kVxStride + rs * r] = o0[r] + d0[r];
                rb[(2 * kt + 2) * kVxStride + rs * r] = o1[r] + d1[r];
            }
        }
        wave_lds_fence();
    }
}

// Corner table of quarter Q (phase A, one voxel per lane and pass: the gather's map): 8 hat weights and the byte
// offset of row (jz, jy, jx) in the image of dV; the eight corners sit at constant offsets from it (the base index
// is clamped to [0, 6], see ahv_dual.h).  12 floats per voxel.
constexpr int kCtRow = 12;
// channel-last image of dV in 64-bit words, kDvRow = 17 words per voxel (16 channels + 1 pad): the voxels of a
// scatter instruction then start on different banks instead of all on bank 0 (136-byte rows)
constexpr int kDvRow = 17;
constexpr int kDvCornerBytes(int n) { return (((n & 1) ? 1 : 0) + ((n & 2) ? 8 : 0) + ((n & 4) ? 64 : 0)) * kDvRow * 8; }

template <int Q>
__device__ __forceinline__ void bwd_vol_corners(float* ctab, const GatherHyp& h, const GatherDst& dst)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float jx, jy, jz, wx0, wx1, wy0, wy1, wz0, wz1;
        hat_axis(gather_coord<Q>(h, 0, p), jx, wx0, wx1);
        hat_axis(gather_coord<Q>(h, 1, p), jy, wy0, wy1);
        hat_axis(gather_coord<Q>(h, 2, p), jz, wz0, wz1);
        const float w00 = wz0 * wy0, w01 = wz0 * wy1, w10 = wz1 * wy0, w11 = wz1 * wy1;
        const float base = fmaf(jz, 64.0f * (kDvRow * 8), fmaf(jy, 8.0f * (kDvRow * 8), jx * (float)(kDvRow * 8)));  // bytes
        float* row = ctab + (p ? dst.o1 : dst.o0) * kCtRow;
        *reinterpret_cast<f32x4*>(row + 0) = f32x4{w00 * wx0, w00 * wx1, w01 * wx0, w01 * wx1};
        *reinterpret_cast<f32x4*>(row + 4) = f32x4{w10 * wx0, w10 * wx1, w11 * wx0, w11 * wx1};
        row[8] = __uint_as_float((unsigned)base);
    }
}

// x (|x| < 2^31, scaled contribution) -> nearest integer, sign-extended to the 64-bit accumulator word.
// v_cvt_rpi_i32_f32 = floor(x + 0.5): one instruction, no bias towards zero (v_cvt_i32_f32 truncates).
__device__ __forceinline__ long long to_fixed32(float x)
{
    int i;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(x));
    return (long long)i;
}

// dV += trilinear^T dX (phase B): lane (c8, vq) adds local channel c8 of eight voxels per step, the voxels of a
// step 4 apart in b and 2 apart in e, so that for a rotation their 2x2x2 corner footprints rarely share a word
// (any R stays correct: the adds are atomic).  The image is 64-bit fixed point because ds_add_f32 is ~40x slower
// than the integer LDS atomics on gfx950 (tools/lds_atomic_probe.cpp: 771 vs 19 (u32) / 28 (u64) cycles per
// wave-instruction); every addend is rounded to a 32-bit integer (31 bits below the per-sample bound, finer than an
// fp32 mantissa for all but the largest terms), the sums are exact and do not depend on the order of the adds.
__device__ __forceinline__ void bwd_vol_scatter(const float* xbuf, const float* ctab, long long* dV, float fx_scale, int role, int lane)
{
    const int c8 = lane & 7, vq = lane >> 3;
    char* dvc = reinterpret_cast<char*>(dV + 8 * role + c8);
    const int vlane = 32 * (vq & 1) + 2 * (vq >> 1);  // b += 4 (vq & 1), e += 2 (vq >> 1)
#pragma unroll 2
    for (int st = 0; st < 16; ++st) {
        // step -> (a0, b' < 4, e' < 2): voxel = a0*64 + (b' + 4 (vq&1))*8 + e' + 2 (vq>>1)
        const int vox = vlane + (st >> 3) * 64 + ((st >> 1) & 3) * 8 + (st & 1);
        const float d = xbuf[c8 * kVxStride + vox] * fx_scale;
        const float* row = ctab + vox * kCtRow;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(row + 0), w1 = *reinterpret_cast<const f32x4*>(row + 4);
        char* base = dvc + __float_as_uint(row[8]);
        // unconditional: a zero weight adds 0 to a valid row.  Branching on the weight put every atomic in its
        // own basic block behind an s_waitcnt lgkmcnt(0), i.e. serialised their latencies.
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            lds_add_i64(reinterpret_cast<long long*>(base + kDvCornerBytes(nb)), to_fixed32(w0[nb] * d));
            lds_add_i64(reinterpret_cast<long long*>(base + kDvCornerBytes(4 + nb)), to_fixed32(w1[nb] * d));
        }
    }
}

constexpr int kVolThreads = 512;
// the scattering wave (VALU + LDS latency) issues first, its partner streams MFMAs: same reasoning as the forward's gather
#define AHV_VOL_PRIO(x) __builtin_amdgcn_s_setprio(x)

__global__ __launch_bounds__(kVolThreads, 2) void score_backward_volume_kernel(
    const float* __restrict__ R, long r_batch_stride, const float* __restrict__ W1, int B, long N,
    const float* __restrict__ du_ws, const unsigned* __restrict__ du_max_bits, float* __restrict__ grad_vol)
{
    __shared__ __attribute__((aligned(16))) long long lds_dv[512 * kDvRow];   // channel-last, 64-bit fixed point
    __shared__ __attribute__((aligned(16))) float lds_x[8 * kVhFloats];       // per wave: dX of its 8 channels
    __shared__ __attribute__((aligned(16))) float lds_ct[8 * 128 * kCtRow];   // per wave: corner table of a quarter
    __shared__ float lds_bound[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave & 3, role = wave >> 2;  // waves w and w + 4 share a SIMD and a hypothesis list
    const int kq = lane >> 4, row = lane & 15;
    float* xbuf = lds_x + wave * kVhFloats;
    float* ctab = lds_ct + wave * (128 * kCtRow);

    // |dX| <= (sum over the three slabs of max_k sum_o |W1[o][k]|) * max|du|: fixes the fixed-point scale per sample
    {
        float* colsum = lds_x;  // 384 floats of scratch before the hypothesis loop
        for (int k = tid; k < 384; k += kVolThreads) {
            float a = 0.0f;
            for (int o = 0; o < 32; ++o) a += fabsf(W1[o * 384 + k]);
            colsum[k] = a;
        }
        __syncthreads();
        if (tid < 3) {
            float m = 0.0f;
            for (int k = 0; k < 128; ++k) m = fmaxf(m, colsum[128 * tid + k]);
            lds_bound[tid] = m;
        }
        __syncthreads();
    }
    const float w1_bound = lds_bound[0] + lds_bound[1] + lds_bound[2];

    // W1^T fragments of this wave's channel half, A operands of dX: [tile][k-step over o]: W1[o = 4 s' + kq][k]
    //   x / y: k = 16 (4 role + kt) + row                 (channel 8 role + 2 kt + (row >> 3), slab index row & 7)
    //   z:     k = 256 + (8 role + (row & 7)) * 8 + 2 q + (row >> 3)      (tile rows = (a0, local channel))
    float wx[4][8], wy[4][8], wz[4][8];
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        const float* w = W1 + (4 * sp + kq) * 384;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            wx[kt][sp] = w[16 * (4 * role + kt) + row];
            wy[kt][sp] = w[128 + 16 * (4 * role + kt) + row];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) wz[q][sp] = w[256 + (8 * role + (row & 7)) * 8 + 2 * q + (row >> 3)];
    }
    const GatherLane glane = gather_lane(lane);
    const GatherDst gdst = gather_dst_linear(lane);

    const long hstep = (long)gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        for (int i = tid; i < 512 * kDvRow; i += kVolThreads) lds_dv[i] = 0ll;
        __syncthreads();
        // every addend w * dX (w <= 1) is rounded to a 32-bit integer in units of 2^-fx_exp with |addend| < 2^30;
        // the 64-bit words then have room for 2^33 of them
        const float bound = w1_bound * __uint_as_float(du_max_bits[b]);
        int ex = 0;
        (void)frexpf(bound, &ex);
        const bool usable = bound > 0.0f && bound < 3.0e38f;
        const int fx_exp = usable ? min(max(30 - ex, -80), 80) : 0;
        const float fx_scale = usable ? ldexpf(1.0f, fx_exp) : 0.0f;
        const float* Rb = R + (long)b * r_batch_stride;
        long h = (long)slot * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        DuRegs du;
        if (h < N) load_du_regs(du, du_ws + ((long)b * N + h) * 2048, lane);
        for (; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            GatherHyp gh;
            gather_hyp(gh, Rm, glane);
            DuRegs nxt;  // the next hypothesis's du travels from HBM / L2 while this one is processed
            const long hn = (h + hstep < N) ? h + hstep : h;
            load_du_regs(nxt, du_ws + ((long)b * N + hn) * 2048, lane);
            bwd_vol_dx<0>(wx, wy, wz, du, xbuf, lane);
            AHV_VOL_PRIO(1);
            bwd_vol_corners<0>(ctab, gh, gdst); wave_lds_fence();
            bwd_vol_scatter(xbuf, ctab, lds_dv, fx_scale, role, lane); wave_lds_fence();
            AHV_VOL_PRIO(0);
            bwd_vol_dx<1>(wx, wy, wz, du, xbuf, lane);
            AHV_VOL_PRIO(1);
            bwd_vol_corners<1>(ctab, gh, gdst); wave_lds_fence();
            bwd_vol_scatter(xbuf, ctab, lds_dv, fx_scale, role, lane); wave_lds_fence();
            AHV_VOL_PRIO(0);
            bwd_vol_dx<2>(wx, wy, wz, du, xbuf, lane);
            AHV_VOL_PRIO(1);
            bwd_vol_corners<2>(ctab, gh, gdst); wave_lds_fence();
            bwd_vol_scatter(xbuf, ctab, lds_dv, fx_scale, role, lane); wave_lds_fence();
            AHV_VOL_PRIO(0);
            bwd_vol_dx<3>(wx, wy, wz, du, xbuf, lane);
            AHV_VOL_PRIO(1);
            bwd_vol_corners<3>(ctab, gh, gdst); wave_lds_fence();
            bwd_vol_scatter(xbuf, ctab, lds_dv, fx_scale, role, lane); wave_lds_fence();
            AHV_VOL_PRIO(0);
            du = nxt;
        }
        __syncthreads();
        // non-finite upstream gradients: the bound is inf/NaN, nothing was accumulated -> report NaN like autograd would
        const float unscale = usable ? ldexpf(1.0f, -fx_exp) : (bound == 0.0f ? 0.0f : __builtin_nanf(""));
        float* gv = grad_vol + (long)b * (16 * 512);
        for (int i = tid; i < 16 * 512; i += kVolThreads) {
            const int c = i >> 9, v = i & 511;
            const long long a = lds_dv[v * kDvRow + c];
            if (a != 0ll || !usable) global_add(gv + i, (float)a * unscale + (usable ? 0.0f : unscale));
        }
    }
}


// -------------------------------------------------------------------------------------------------
// Kernel 2b, round 6: the same dV = sum_h trilinear_h^T (W1^T du_h), WITHOUT LDS atomics.
//
// What bounded score_backward_volume_kernel above (profiles/r06_training_pmc_summary.json, B = 32 x N = 9 000: 8.4 ms, 0.37 of
// the fp32 matrix peak) is its scatter: 1 024 ds_add_u64 wave-instructions per hypothesis on a pipe the four SIMDs share, and
// three vector instructions per atomic (scale, round to integer, sign-extend) that fp32 MFMAs cannot overlap -- 5 679 vector
// instructions per hypothesis against 768 MFMAs.  A plain read-modify-write moves the same bytes with ONE packed FMA per two
// channels, but it is only correct if no two lanes of an instruction (and no two waves) touch the same word.  Both can be
// arranged:
//   * between waves: a workgroup runs TWO hypotheses at a time (slots), each with a PRIVATE fp32 image of dV; the four waves
//     of a slot (one per SIMD) own four channels each, so they share the hypothesis' corner table and never an address.  (Four
//     private images -- one per hypothesis in flight, as kernel 2b has them in flight -- do not fit the 160 KB of LDS beside
//     the dX images; hence four waves per hypothesis, split by channel.)
//   * inside an instruction: lanes = 8 voxels x 8 corners, each lane the four channels of its wave (16 bytes).  The 8 corners of
//     a voxel are 8 different rows; the 8 voxels are 4 apart in z, y and x, and for a rotation two lattice points 4 apart land
//     >= 4 / sqrt(3) > 2 apart along some axis, so their 2 x 2 x 2 footprints are disjoint.  A matrix that is not a rotation
//     (|R^T R - I| > 0.04 anywhere; the ABI accepts any R) takes the same code one voxel at a time.
//   * the forward's clamped footprint (base row in [0, 6], hat weights) gives corners outside the volume a weight of exactly
//     0 on a row that may belong to another lane's voxel: such a corner is redirected to a trash row when the corner table is
//     built (its byte offset is part of the table), so a zero weight never writes a live word back.
// Sums are fp32 in the order the hypotheses arrive (the fixed-point image of kernel 2b was exact and order-free): the
// gradients agree with the fp64 reference to ~1e-6 of their largest entry, like those of the other kernels.
// Per hypothesis and wave: 192 MFMAs (x / y slabs: 2 row tiles of (channel, k); z slab: ONE row tile of (depth, channel) over
// the four depths of a half volume -- quarters {H, H + 2} -- so that no MFMA row is wasted on four channels), 64 scatter
// steps of (16-byte image read, two packed FMAs, 16-byte image write) with the step's operands (dX of the voxel, weight and
// row offset of the corner) requested two steps ahead.  Every LDS layout below is chosen so that the lanes of an access spread
// over the banks (SQ_LDS_BANK_CONFLICT went 12.7 k -> 2.9 k cycles per hypothesis on the way, tools/profile_kbench_bwd.sh).
// Measured (tools/kbench_bwd 32 9000): 7.6 ms against 8.4 ms; 2 736 vector and 1 606 LDS instructions per hypothesis (5 679 /
// 1 840).  What bounds it now is the LDS itself -- 10.0 k busy cycles per hypothesis of the 14.3 k the workgroup spends on
// it -- and the latency of the ONE read-modify-write chain a wave can have in flight (in-kernel stamps, -DAHV_RMW_STAMPS:
// 220-290 cycles per step; dX 5.0 k cycles per half for 3.1 k cycles of MFMAs).
// -------------------------------------------------------------------------------------------------
constexpr int kRmwThreads = 512;                  // 8 waves = 2 slots x 4 members, member m on SIMD m
// Private image of dV: channel-last, 16 words per voxel; a y row of 8 voxels is padded by 4 words and a depth plane by 8, so
// that the 8 rows of a footprint start at words {0, 16, 132, 148} (+ 1064 for z + 1) = banks {0, 16, 4, 20} and {8, 24, 12, 28}:
// a lane updates the FOUR channels of its wave at one corner (16 bytes), and the 8 corners of a voxel cover all 32 banks
// exactly once -- any 8 voxels of an instruction hit every bank 8 times, the LDS's own rate for 1 KiB.
constexpr int kImgRowBytes = 64;
constexpr int kImgYRowBytes = 8 * kImgRowBytes + 16;
constexpr int kImgPlaneBytes = 8 * kImgYRowBytes + 32;
constexpr int kImgTrashBytes = 8 * kImgPlaneBytes;      // the row behind the last plane: target of zero-weight corners
constexpr int kImgWords = (kImgTrashBytes + kImgRowBytes) / 4;
// dX image of a wave: channel-last, word address = al * kXPlane + b * kXRow + e * 4 + channel.  A y row (8 voxels x 4 channels)
// is padded to 34 words and a plane to 274: the x / y slab updates walk b (or e) across the lanes of an MFMA tile, and with
// the unpadded strides (32 and 288 words, both multiples of the 32 banks) all 64 lanes of such an access met in 4 banks --
// SQ_LDS_BANK_CONFLICT 12.7 k of 20.5 k LDS cycles per hypothesis in the first version of this kernel.
constexpr int kXRow = 34;                 // (2 mod 8, and the plane 2 mod 16: the eight voxels of a scatter step -- 4 apart in b, e and
constexpr int kXPlane = 8 * kXRow + 2;    //  depth -- then read eight different 4-bank groups; rows are 8-byte aligned only, so 16 bytes
constexpr int kXbufWords = 4 * kXPlane;   //  travel as two halves)
constexpr int kCt2Row = 12;                       // corner table: 8 weights, 8 x u16 byte offsets per voxel ...
constexpr int kCt2BRow = 8 * kCt2Row + 2;         // ... 8 voxels (one y row) + 2 words: voxels 4 apart in y then sit 8 banks apart,
constexpr int kCt2Words = 64 * kCt2BRow;          //     voxels 4 apart in x 16 -- the four voxels of a scatter step never share a bank

struct RmwSync {
    unsigned ready, done;
};

__device__ __forceinline__ void rmw_signal(unsigned* ctr, int lane)
{
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes and reads are complete
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void rmw_wait(unsigned* ctr, unsigned target)
{
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// |R^T R - I| <= 0.04 entrywise: the eigenvalues of R^T R are then >= 0.88, two lattice points 4 apart map >= 3.75 apart and
// >= 2.16 apart along some axis -- their footprints cannot share a row.  NaN compares false: not a rotation.
__device__ __forceinline__ bool rmw_rotation_like(const float* Rm)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            const float d = fmaf(Rm[i], Rm[j], fmaf(Rm[3 + i], Rm[3 + j], Rm[6 + i] * Rm[6 + j])) - (i == j ? 1.0f : 0.0f);
            ok = ok && (fabsf(d) <= 0.04f);
        }
    return ok;
}

// Corner table of quarter Q (one voxel per lane and pass, the gather's lane map): per voxel the 8 hat-weight products
// (index = 4 z + 2 y + x) and the BYTE offsets of the 8 rows in a private image -- the trash row where the weight is 0.
template <int Q>
__device__ __forceinline__ void rmw_corners(float* ctab, const GatherHyp& h, const GatherDst& dst)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float jx, jy, jz, wx0, wx1, wy0, wy1, wz0, wz1;
        hat_axis(gather_coord<Q>(h, 0, p), jx, wx0, wx1);
        hat_axis(gather_coord<Q>(h, 1, p), jy, wy0, wy1);
        hat_axis(gather_coord<Q>(h, 2, p), jz, wz0, wz1);
        const float w00 = wz0 * wy0, w01 = wz0 * wy1, w10 = wz1 * wy0, w11 = wz1 * wy1;
        const float w[8] = {w00 * wx0, w00 * wx1, w01 * wx0, w01 * wx1, w10 * wx0, w10 * wx1, w11 * wx0, w11 * wx1};
        const unsigned base = (unsigned)fmaf(jz, (float)kImgPlaneBytes, fmaf(jy, (float)kImgYRowBytes, jx * (float)kImgRowBytes));
        unsigned off[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const unsigned o = base + (unsigned)((n & 1) * kImgRowBytes + ((n & 2) ? kImgYRowBytes : 0) + ((n & 4) ? kImgPlaneBytes : 0));
            off[n] = (w[n] != 0.0f) ? o : (unsigned)kImgTrashBytes;   // (NaN != 0: a NaN weight writes NaN to its own row)
        }
        const int vox = Q * 128 + (p ? dst.o1 : dst.o0);
        float* row = ctab + (vox >> 3) * kCt2BRow + (vox & 7) * kCt2Row;   // 8-byte aligned
        *reinterpret_cast<f32x2*>(row + 0) = f32x2{w[0], w[1]};
        *reinterpret_cast<f32x2*>(row + 2) = f32x2{w[2], w[3]};
        *reinterpret_cast<f32x2*>(row + 4) = f32x2{w[4], w[5]};
        *reinterpret_cast<f32x2*>(row + 6) = f32x2{w[6], w[7]};
        *reinterpret_cast<f32x2*>(row + 8) = f32x2{__uint_as_float(off[0] | (off[1] << 16)), __uint_as_float(off[2] | (off[3] << 16))};
        *reinterpret_cast<f32x2*>(row + 10) = f32x2{__uint_as_float(off[4] | (off[5] << 16)), __uint_as_float(off[6] | (off[7] << 16))};
    }
}

// dX of this wave's four channels over half volume H (depth planes a = 2 H + (al & 1) + 4 (al >> 1), al = 0..3) into the
// channel-last image xbuf[al][b * 8 + e][4]: the z slab first (plain 16-byte stores: a lane's four accumulator rows ARE the
// four channels of one voxel), then the x and y slabs of the two quarters added on top.
template <int H>
__device__ __forceinline__ void rmw_dx_half(const float (&wx)[2][8], const float (&wy)[2][8], const float (&wz)[2][8],
                                            const DuRegs& du, float* xbuf, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    {
        float* zb = xbuf + kq * kXPlane + (n >> 3) * kXRow + (n & 7) * 4;  // voxel (al = kq, b = 2 t + (n >> 3), e = n & 7)
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[H][sp], du.v[t][sp], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[H][sp], du.v[t + 1][sp], d1, 0, 0, 0);
            }
            *reinterpret_cast<f32x2*>(zb + 2 * t * kXRow) = f32x2{d0[0], d0[1]};
            *reinterpret_cast<f32x2*>(zb + 2 * t * kXRow + 2) = f32x2{d0[2], d0[3]};
            *reinterpret_cast<f32x2*>(zb + 2 * (t + 1) * kXRow) = f32x2{d1[0], d1[1]};
            *reinterpret_cast<f32x2*>(zb + 2 * (t + 1) * kXRow + 2) = f32x2{d1[2], d1[3]};
        }
    }
    wave_lds_fence();
    // x and y slabs of the two quarters, added onto the z slab in LDS.  Block order x0, x1, y0, y1 (x / y slab of quarter
    // H + 2 j): the words a block adds onto are requested a block AHEAD -- 16 MFMAs hide the round trip; read just in front
    // of their MFMAs they cost the wave ~150 cycles per block on an LDS the scatters keep 70 % busy (in-kernel stamps:
    // 5.1 k cycles per half for 3.1 k cycles of MFMAs).  y j reads what x j wrote, so its request follows x j's stores.
    // x: channel 2 kt + (kq >> 1), voxel (al, b = n & 7, e = 4 (kq & 1) + r);  y: voxel (al, b = 4 (kq & 1) + r, e = n & 7)
    float* const rbx = xbuf + (kq >> 1) + (n >> 3) * kXPlane + (n & 7) * kXRow + 16 * (kq & 1);
    float* const rby = xbuf + (kq >> 1) + (n >> 3) * kXPlane + 4 * (kq & 1) * kXRow + (n & 7) * 4;
    auto rd = [&](float (&o0)[4], float (&o1)[4], const float* rb, int rs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = rb[rs * r];
            o1[r] = rb[rs * r + 2];
        }
    };
    auto mm = [&](f32x4& d0, f32x4& d1, const float (&w)[2][8], int t) {
        d0 = f32x4{0.f, 0.f, 0.f, 0.f};
        d1 = d0;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][sp], du.v[t][sp], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][sp], du.v[t][sp], d1, 0, 0, 0);
        }
    };
    auto wr = [&](float* rb, int rs, const float (&o0)[4], const float (&o1)[4], const f32x4& d0, const f32x4& d1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rb[rs * r] = o0[r] + d0[r];
            rb[rs * r + 2] = o1[r] + d1[r];
        }
        wave_lds_fence();   // (compiler-level: later reads of these words, by other lanes, stay behind the stores)
    };
    float ax0[4], ax1[4], bx0[4], bx1[4], ay0[4], ay1[4], by0[4], by1[4];
    f32x4 d0, d1;
    rd(ax0, ax1, rbx, 4);                          // x0
    rd(bx0, bx1, rbx + 2 * kXPlane, 4);            // x1
    mm(d0, d1, wx, H);
    wr(rbx, 4, ax0, ax1, d0, d1);
    rd(ay0, ay1, rby, kXRow);                      // y0 (behind x0's stores)
    mm(d0, d1, wx, H + 2);
    wr(rbx + 2 * kXPlane, 4, bx0, bx1, d0, d1);
    rd(by0, by1, rby + 2 * kXPlane, kXRow);        // y1 (behind x1's stores)
    mm(d0, d1, wy, H);
    wr(rby, kXRow, ay0, ay1, d0, d1);
    mm(d0, d1, wy, H + 2);
    wr(rby + 2 * kXPlane, kXRow, by0, by1, d0, d1);
}

struct RmwStep {   // what a lane needs for one step: its voxel's dX (4 channels), the weight of its corner, the byte offset of its row
    f32x2 d01, d23;
    float w;
    unsigned o;
};

template <int H, int S>
__device__ __forceinline__ void rmw_step_load(RmwStep& st, const float* xl, const float* tw, const unsigned short* to)
{
    constexpr int ap = S >> 4, b0 = (S >> 2) & 3, e0 = S & 3;       // step S = (ap, b0, e0)
    constexpr int t = (ap * 8 + b0) * kCt2BRow + e0 * kCt2Row;
    constexpr int x = ap * kXPlane + b0 * kXRow + e0 * 4;
    st.d01 = *reinterpret_cast<const f32x2*>(xl + x);
    st.d23 = *reinterpret_cast<const f32x2*>(xl + x + 2);
    st.w = tw[t];
    st.o = to[t * 2];
}

template <int H, int S, bool ROT>
struct RmwSteps {
    // read-modify-write of step S.  Its operands were requested TWO steps ago (the row address is ready when the step starts);
    // the operands of step S + 2 are requested behind the image read -- the LDS returns in order, so the image words arrive
    // first -- and ahead of the write.  One LDS round trip per step is exposed: the image read.
    static __device__ __forceinline__ void run(const RmwStep& cur, const RmwStep& nxt, const float* xl, const float* tw,
                                               const unsigned short* to, char* img_ch, int vg)
    {
        f32x4* p = reinterpret_cast<f32x4*>(img_ch + cur.o);
        RmwStep nn;
        if constexpr (ROT) {
            const f32x4 c = *p;
            rmw_step_load<H, (S + 2 < 32 ? S + 2 : 31)>(nn, xl, tw, to);
            const f32x2 lo = __builtin_elementwise_fma(f32x2{cur.w, cur.w}, cur.d01, f32x2{c[0], c[1]});
            const f32x2 hi = __builtin_elementwise_fma(f32x2{cur.w, cur.w}, cur.d23, f32x2{c[2], c[3]});
            *p = f32x4{lo[0], lo[1], hi[0], hi[1]};
        } else {   // any other matrix: footprints may overlap -- one voxel per instruction
            rmw_step_load<H, (S + 2 < 32 ? S + 2 : 31)>(nn, xl, tw, to);
#pragma unroll 1
            for (int g = 0; g < 8; ++g) {
                if (vg == g) {
                    const f32x4 c = *p;
                    *p = f32x4{fmaf(cur.w, cur.d01[0], c[0]), fmaf(cur.w, cur.d01[1], c[1]), fmaf(cur.w, cur.d23[0], c[2]),
                               fmaf(cur.w, cur.d23[1], c[3])};
                }
                wave_lds_fence();
            }
        }
        wave_lds_fence();   // the next step's image read stays behind this write (another lane's row may be this lane's next)
        RmwSteps<H, S + 1, ROT>::run(nxt, nn, xl, tw, to, img_ch, vg);
    }
};
template <int H, bool ROT>
struct RmwSteps<H, 32, ROT> {
    static __device__ __forceinline__ void run(const RmwStep&, const RmwStep&, const float*, const float*, const unsigned short*, char*, int) {}
};

// dV image += trilinear^T dX for half volume H: lane = (plane pair ab, voxel vx of 4, corner c of 8), all four channels of the
// wave.  A step covers the 8 voxels (depth plane al + 2 ab -- 4 further in z --, b0 + 4 (vx >> 1), e0 + 4 (vx & 1)): 32 steps.
template <int H>
__device__ __forceinline__ void rmw_scatter_half(const float* xbuf, const float* ctab, char* img_ch, bool rotation, int lane)
{
    const int c = lane & 7, vx = (lane >> 3) & 3, ab = lane >> 5;
    const float* xl = xbuf + 2 * ab * kXPlane + 4 * (vx >> 1) * kXRow + 16 * (vx & 1);
    // the lane's voxel of step 0: plane 2 H + 4 ab, y = 4 (vx >> 1), x = 4 (vx & 1)
    const float* trow = ctab + ((2 * H + 4 * ab) * 8 + 4 * (vx >> 1)) * kCt2BRow + 4 * (vx & 1) * kCt2Row;
    const float* tw = trow + c;                                    // weight of corner c
    const unsigned short* to = reinterpret_cast<const unsigned short*>(trow + 8) + c;
    RmwStep s0, s1;
    rmw_step_load<H, 0>(s0, xl, tw, to);
    rmw_step_load<H, 1>(s1, xl, tw, to);
    if (rotation) RmwSteps<H, 0, true>::run(s0, s1, xl, tw, to, img_ch, lane >> 3);
    else RmwSteps<H, 0, false>::run(s0, s1, xl, tw, to, img_ch, lane >> 3);
}

#ifdef AHV_RMW_STAMPS   // diagnostic build (tools/kbench_bwd): shader-clock time per phase, summed per wave
__device__ unsigned long long g_rmw_stamps[8 * 8];
#define AHV_RMW_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define AHV_RMW_T(i) do { } while (0)
#endif

__global__ __launch_bounds__(kRmwThreads, 2) void score_backward_volume_rmw_kernel(
    const float* __restrict__ R, long r_batch_stride, const float* __restrict__ W1, int B, long N,
    const float* __restrict__ du_ws, float* __restrict__ grad_vol)
{
    __shared__ __attribute__((aligned(16))) float lds_img[2 * kImgWords];        // private fp32 image of dV per slot
    __shared__ __attribute__((aligned(16))) float lds_x[8 * kXbufWords];         // per wave: dX of its 4 channels, half a volume
    __shared__ __attribute__((aligned(16))) float lds_ct[2 * kCt2Words];         // per slot: the hypothesis' corner table
    __shared__ RmwSync lds_sync[2];
    __shared__ unsigned lds_skew;   // slot 0 has reached its first scatter of this sample
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave >> 2, m = wave & 3;
    const int kq = lane >> 4, row = lane & 15;
    float* xbuf = lds_x + wave * kXbufWords;
    float* ctab = lds_ct + slot * kCt2Words;
    float* img = lds_img + slot * kImgWords;
    char* img_ch = reinterpret_cast<char*>(img) + 16 * m;   // this wave's four channels of a voxel row
    if (tid < 2) lds_sync[tid] = RmwSync{0u, 0u};

    // W1^T fragments of channels 4 m .. 4 m + 3 (A operands; lane (row, kq) holds W1[o = 4 sp + kq][k(row)]):
    //   x / y: k = 32 m + 16 kt + row (+ 128)            rows = (channel 2 kt + (row >> 3), slab index row & 7)
    //   z:     k = 256 + (4 m + (row & 3)) * 8 + depth   rows = (plane al = row >> 2, channel row & 3), depth = 2 H + (al & 1) + 4 (al >> 1)
    float wx[2][8], wy[2][8], wz[2][8];
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        const float* w = W1 + (4 * sp + kq) * 384;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            wx[kt][sp] = w[32 * m + 16 * kt + row];
            wy[kt][sp] = w[128 + 32 * m + 16 * kt + row];
        }
#pragma unroll
        for (int H = 0; H < 2; ++H) wz[H][sp] = w[256 + (4 * m + (row & 3)) * 8 + 2 * H + ((row >> 2) & 1) + 4 * (row >> 3)];
    }
    const GatherLane glane = gather_lane(lane);
    const GatherDst gdst = gather_dst_linear(lane);

#ifdef AHV_RMW_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    const long hstep = (long)gridDim.x * 2;
    unsigned iter = 0;  // hypotheses this slot has finished (its members count alike)
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        for (int i = tid; i < 2 * kImgWords; i += kRmwThreads) lds_img[i] = 0.0f;
        if (tid == 0) lds_skew = 0u;
        __syncthreads();
        bool first = true;
        const float* Rb = R + (long)b * r_batch_stride;
        long h = (long)slot * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        DuRegs du;
        if (h < N) load_du_regs(du, du_ws + ((long)b * N + h) * 2048, lane);
        for (; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            GatherHyp gh;
            gather_hyp(gh, Rm, glane);
            const bool rotation = rmw_rotation_like(Rm);
            DuRegs nxt;  // the next hypothesis's du travels from HBM / L2 while this one is processed
            const long hn = (h + hstep < N) ? h + hstep : h;
            load_du_regs(nxt, du_ws + ((long)b * N + hn) * 2048, lane);
            // The two slots run in ANTI-PHASE: a scatter is LDS traffic (~3.6 k LDS cycles per half for the slot's four waves), a dX
            // half is 96 MFMAs per wave (~3.1 k cycles) -- side by side they fill both pipes, in step they take turns idling them
            // (both slots start a sample at the same barrier and do the same work per hypothesis, so nothing separates them by
            // itself).  Slot 1 therefore starts a sample when slot 0 starts its first scatter, and the equal periods keep the offset.
            AHV_RMW_T(0);   // loop head: R, du request
            if (slot == 1 && first) rmw_wait(&lds_skew, 1u);
            // the slot's corner table: member m builds quarter m once everybody has finished with the previous table
            rmw_wait(&lds_sync[slot].done, 4u * iter);
            AHV_RMW_T(1);   // waiting for the team to finish the previous table
            if (m == 0) rmw_corners<0>(ctab, gh, gdst);
            else if (m == 1) rmw_corners<1>(ctab, gh, gdst);
            else if (m == 2) rmw_corners<2>(ctab, gh, gdst);
            else rmw_corners<3>(ctab, gh, gdst);
            rmw_signal(&lds_sync[slot].ready, lane);
            AHV_RMW_T(2);   // corner table
            rmw_dx_half<0>(wx, wy, wz, du, xbuf, lane);      // (the first half's MFMAs need no table: they run while the others arrive)
            AHV_RMW_T(3);   // dX half 0
            rmw_wait(&lds_sync[slot].ready, 4u * (iter + 1u));
            if (slot == 0 && first && m == 0 && lane == 0) __hip_atomic_store(&lds_skew, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = false;
            AHV_RMW_T(4);   // waiting for the table
            AHV_VOL_PRIO(1);   // the scattering wave is latency-bound: it issues first, its SIMD partner streams MFMAs
            rmw_scatter_half<0>(xbuf, ctab, img_ch, rotation, lane);
            AHV_VOL_PRIO(0);
            AHV_RMW_T(5);   // scatter half 0
            rmw_dx_half<1>(wx, wy, wz, du, xbuf, lane);
            AHV_RMW_T(6);   // dX half 1
            AHV_VOL_PRIO(1);
            rmw_scatter_half<1>(xbuf, ctab, img_ch, rotation, lane);
            AHV_VOL_PRIO(0);
            rmw_signal(&lds_sync[slot].done, lane);
            AHV_RMW_T(7);   // scatter half 1
            ++iter;
            du = nxt;
        }
        __syncthreads();
        float* gv = grad_vol + (long)b * (16 * 512);
        for (int i = tid; i < 16 * 512; i += kRmwThreads) {
            const int c = i >> 9, v = i & 511;
            const int w = (v >> 6) * (kImgPlaneBytes / 4) + ((v >> 3) & 7) * (kImgYRowBytes / 4) + (v & 7) * (kImgRowBytes / 4) + c;
            const float a = lds_img[w] + lds_img[kImgWords + w];
            if (a != 0.0f) global_add(gv + i, a);   // (NaN != 0: a poisoned sample reports NaN, like autograd)
        }
    }
#ifdef AHV_RMW_STAMPS
    if (blockIdx.x == 3 && blockIdx.y == 5 && lane == 0)
        for (int i = 0; i < 8; ++i) g_rmw_stamps[wave * 8 + i] = tsum[i];
#endif
}

hipError_t launch_zero_fill(void* const* ptrs, const size_t* bytes, int count, hipStream_t stream);  // ahv_ops.hip

// ---- host-side launcher -------------------------------------------------------------------------------
hipError_t launch_score_backward(const float* vol_src, const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                 const float* W1, const float* W2, const float* b2, int B, int64_t N,
                                 const float* grad_scores, float* du_ws, unsigned* du_max_bits, float* dw1_partials,
                                 float* grad_vol, float* grad_feat_tgt,
                                 float* grad_W1, float* grad_W2, float* grad_b2, int num_cu, hipStream_t stream, bool saved_u)
{
    hipError_t e;
    {   // accumulation targets (and the running maximum of |du|, bit pattern 0 = 0.0f): one zero-fill launch
        void* const ptrs[6] = {grad_vol, grad_feat_tgt, grad_W1, grad_W2, grad_b2, du_max_bits};
        const size_t bytes[6] = {sizeof(float) * (size_t)B * 8192, sizeof(float) * (size_t)B * 2048, sizeof(float) * 32 * 384,
                                 sizeof(float) * 32 * 32, sizeof(float) * 32, (B > 0 && N > 0) ? sizeof(unsigned) * (size_t)B : 0};
        if ((e = launch_zero_fill(ptrs, bytes, 6, stream)) != hipSuccess) return e;
    }
    if (B == 0 || N == 0) return hipSuccess;
    int gy = B < num_cu ? B : num_cu;
    int gx = num_cu / gy;
    const int64_t need = (N + 3) / 4;
    if (gx > need) gx = (int)need;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, gy);
    if (saved_u)   // du_ws holds u of every hypothesis (the training forward); the head kernel turns it into du in place
    {
        int gxs = num_cu / gy;                       // eight waves per workgroup, one hypothesis per wave and turn
        const int64_t need8 = (N + 7) / 8;
        if (gxs > need8) gxs = (int)need8;
        if (gxs < 1) gxs = 1;
        hipLaunchKernelGGL(score_backward_head_saved_kernel, dim3(gxs, gy), dim3(kSavedThreads), 0, stream, feat_tgt, W2, b2, B,
                           (long)N, grad_scores, du_ws, du_max_bits, grad_feat_tgt, grad_W2, grad_b2);
    }
    else
        hipLaunchKernelGGL(score_backward_head_kernel, grid, dim3(kBwdThreads), 0, stream, vol_src, feat_tgt, R,
                           (long)r_batch_stride, W1, W2, b2, B, (long)N, grad_scores, du_ws, du_max_bits, grad_feat_tgt, grad_W2,
                           grad_b2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(score_backward_w1_kernel, grid, dim3(kW1Threads), 0, stream, vol_src, R, (long)r_batch_stride,
                       B, (long)N, du_ws, dw1_partials);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(score_backward_w1_reduce_kernel, dim3(32 * 384 / 256, 16), dim3(256), 0, stream, dw1_partials,
                       gx * gy, grad_W1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
#ifdef AHV_BWD_VOLUME_ATOMICS  // rounds 2-5: the scatter as 64-bit fixed-point LDS atomics (kept for A/B builds: tools/kbench_bwd)
    hipLaunchKernelGGL(score_backward_volume_kernel, grid, dim3(kVolThreads), 0, stream, R, (long)r_batch_stride, W1,
                       B, (long)N, du_ws, du_max_bits, grad_vol);
#else
    {   // two hypotheses per workgroup at a time: spread a short list over more workgroups
        int gxv = num_cu / gy;
        const int64_t need2 = (N + 1) / 2;
        if (gxv > need2) gxv = (int)need2;
        if (gxv < 1) gxv = 1;
        hipLaunchKernelGGL(score_backward_volume_rmw_kernel, dim3(gxv, gy), dim3(kRmwThreads), 0, stream, R, (long)r_batch_stride,
                           W1, B, (long)N, du_ws, grad_vol);
    }
#endif
    return hipGetLastError();
}

}  // namespace ahv
