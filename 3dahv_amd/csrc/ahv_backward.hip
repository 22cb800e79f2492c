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
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ahv_device.h"
#include "ahv_dual.h"

namespace ahv {

constexpr int kBwdThreads = 256;  // 4 waves, one per SIMD: 512 registers per wave

// du image in LDS: du[o][pos] at o*64 + (pos ^ ((o & 7) << 2)).  The XOR keeps aligned groups of four
// positions together (float4 fills) and spreads rows over banks for the transposed reads (k = pos).
__device__ __forceinline__ int dimg(int o, int pos) { return o * 64 + (pos ^ ((o & 7) << 2)); }

__device__ __forceinline__ void lds_add_i64(long long* p, long long v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    __shared__ __attribute__((aligned(16))) float lds_src[kSrcFloats];
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
                hat_prologue<0>(st, lds_src, gh);
                hat_body(st, buf, lane); wave_lds_fence();
                gemm1_quarter_pipe<0>(acc, lds_w1, buf, lane, [&] { hat_prologue<1>(st, lds_src, gh); }); wave_lds_fence();
                hat_body(st, buf, lane); wave_lds_fence();
                gemm1_quarter_pipe<1>(acc, lds_w1, buf, lane, [&] { hat_prologue<2>(st, lds_src, gh); }); wave_lds_fence();
                hat_body(st, buf, lane); wave_lds_fence();
                gemm1_quarter_pipe<2>(acc, lds_w1, buf, lane, [&] { hat_prologue<3>(st, lds_src, gh); }); wave_lds_fence();
                hat_body(st, buf, lane); wave_lds_fence();
                gemm1_quarter_pipe<3>(acc, lds_w1, buf, lane, [] {}); wave_lds_fence();
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
                        dst[(16 * m + 4 * kq + r) * 64 + 16 * t + n] = x;
                        // max |du| of the sample (kernel 2b sizes its fixed-point scale with it); NaN / inf poison it
                        du_amax = (x == x) ? fmaxf(du_amax, fabsf(x)) : __builtin_inff();
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
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) du_amax = fmaxf(du_amax, __shfl_xor(du_amax, sft, 64));
        if (lane == 0) atomicMax(du_max_bits + b, __float_as_uint(du_amax));  // non-negative floats order like uints
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
// Kernels 2a / 2b.  Both walk the hypotheses again with du from the workspace.  They are separate launches
// because each needs 192 registers of persistent state per wave (2a: the dW1 accumulators, 2b: the W1^T
// fragments) next to a gather / scatter that wants ~100 more: together they spill, apart they do not.
//   2a  score_backward_w1_kernel      re-gathers each quarter X of the rotated volume, dW1 += du X^T
//   2b  score_backward_volume_kernel  dX = W1^T du per quarter, dV += trilinear^T dX in an LDS image of the
//                                     sample's volume gradient, flushed once per sample
// ---------------------------------------------------------------------------------------------------

// X image of 2a: plane c rotated by 4c floats, so that the z slab's B operand (16 lanes = 16 channels of one
// voxel) spreads over 8 banks instead of one; all other accesses have c uniform per instruction.
__device__ __forceinline__ int xoff(int c, int a0, int b, int e) { return c * 128 + ((qoff(a0, b, e) + 4 * c) & 127); }

__device__ __forceinline__ void load_du_image(float* dbuf, const float* __restrict__ du_hyp, int lane)
{
    const f32x4* src = reinterpret_cast<const f32x4*>(du_hyp);
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // coalesced float4 loads, aligned float4 stores (dimg keeps groups of 4 together)
        const int idx = 4 * (i * 64 + lane), o = idx >> 6, pos = idx & 63;
        *reinterpret_cast<f32x4*>(dbuf + dimg(o, pos)) = src[i * 64 + lane];
    }
}

template <int Q>
__device__ __forceinline__ void bwd_w1_quarter(f32x4 (&ax)[2][8], f32x4 (&ay)[2][8], f32x4 (&az)[2][4][2],
                                               const float* dbuf, float* xbuf, const float* srcT, const float* Rm,
                                               int lane)
{
    const int n = lane & 15, kq = lane >> 4, row = lane & 15;
    const int i0 = n >> 3, j = n & 7;
    {   // gather quarter Q (same lane -> voxel map as tri_quarter) into the rotated-plane image
        const int e = lane & 7, a0 = (lane >> 3) & 1, b0 = (lane >> 4) & 1, b1 = (lane >> 5) & 1;
        const float x = (2.0f * e + 1.0f) * 0.125f - 1.0f;
        const float z = (2.0f * (2 * Q + a0) + 1.0f) * 0.125f - 1.0f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int b = 4 * p + 2 * b1 + b0;
            const float y = (2.0f * b + 1.0f) * 0.125f - 1.0f;
            TriCoefP k;
            tri_coef_ptr(k, srcT, Rm, x, y, z);
            float o[16];
            tri_blend_ptr(o, k);
            const int q0 = qoff(a0, b, e);
#pragma unroll
            for (int c = 0; c < 16; ++c) xbuf[c * 128 + ((q0 + 4 * c) & 127)] = o[c];
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // x and y slabs: the 16 positions of tile Q, four per k-step
        const int pl = 4 * s + kq, pa = pl >> 3, pb = pl & 7;
        const float a0v = dbuf[dimg(row, 16 * Q + pl)], a1v = dbuf[dimg(16 + row, 16 * Q + pl)];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const float bx = xbuf[xoff(2 * kt + i0, pa, pb, j)];  // X[k = (c, e = j)][pos = (a0, b)]
            ax[0][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, bx, ax[0][kt], 0, 0, 0);
            ax[1][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, bx, ax[1][kt], 0, 0, 0);
            const float by = xbuf[xoff(2 * kt + i0, pa, j, pb)];  // X[k = (c, b = j)][pos = (a0, e)]
            ay[0][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, by, ay[0][kt], 0, 0, 0);
            ay[1][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, by, ay[1][kt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // z slab: all 64 positions (b, e); k = (c = column, a0)
        const int pl = 4 * s + kq;
        const float a0v = dbuf[dimg(row, pl)], a1v = dbuf[dimg(16 + row, pl)];
#pragma unroll
        for (int a0 = 0; a0 < 2; ++a0) {
            const float bz = xbuf[xoff(n, a0, pl >> 3, pl & 7)];
            az[0][Q][a0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, bz, az[0][Q][a0], 0, 0, 0);
            az[1][Q][a0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, bz, az[1][Q][a0], 0, 0, 0);
        }
    }
    wave_lds_fence();
}

__global__ __launch_bounds__(kBwdThreads, 1) void score_backward_w1_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ R, long r_batch_stride, int B, long N,
    const float* __restrict__ du_ws, float* __restrict__ grad_W1)
{
    __shared__ __attribute__((aligned(16))) float lds_src[kSrcFloats];
    __shared__ __attribute__((aligned(16))) float lds_du[4 * 2048];
    __shared__ __attribute__((aligned(16))) float lds_x[4 * kQuarterFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    float* dbuf = lds_du + wave * 2048;
    float* xbuf = lds_x + wave * kQuarterFloats;
    f32x4 ax[2][8], ay[2][8], az[2][4][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            ax[m][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            ay[m][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0) az[m][q][a0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long hstep = (long)gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        stage_src_volume(lds_src, vol_src + (long)b * (16 * 512), tid, kBwdThreads);
        __syncthreads();
        const float* Rb = R + (long)b * r_batch_stride;
        for (long h = (long)wave * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y); h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            load_du_image(dbuf, du_ws + ((long)b * N + h) * 2048, lane);
            wave_lds_fence();
            bwd_w1_quarter<0>(ax, ay, az, dbuf, xbuf, lds_src, Rm, lane);
            bwd_w1_quarter<1>(ax, ay, az, dbuf, xbuf, lds_src, Rm, lane);
            bwd_w1_quarter<2>(ax, ay, az, dbuf, xbuf, lds_src, Rm, lane);
            bwd_w1_quarter<3>(ax, ay, az, dbuf, xbuf, lds_src, Rm, lane);
        }
    }
    // dW1[o = 16 m + 4 kq + r][k]: x: k = 16 kt + n; y: 128 + 16 kt + n; z: 256 + n*8 + 2 q + a0
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* g = grad_W1 + (16 * m + 4 * kq + r) * 384;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                global_add(g + 16 * kt + n, ax[m][kt][r]);
                global_add(g + 128 + 16 * kt + n, ay[m][kt][r]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a0 = 0; a0 < 2; ++a0) global_add(g + 256 + n * 8 + 2 * q + a0, az[m][q][a0][r]);
        }
}

template <int Q>
__device__ __forceinline__ void bwd_vol_quarter(const float (&wx)[8][8], const float (&wy)[8][8],
                                                const float (&wz)[4][2][8], float* dbuf, float* xbuf,
                                                long long* dV, float fx_scale, const float* Rm, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    // ---- dX = W1^T du for the voxels of quarter Q ---------------------------------------------------
    float bq[8];  // du[o = 4 s' + kq][pos = 16 Q + n]: B operand of the x and y slabs
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) bq[sp] = dbuf[dimg(4 * sp + kq, 16 * Q + n)];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[kt][sp], bq[sp], d, 0, 0, 0);
        // rows k = 16 kt + 4 kq + r -> c = 2 kt + (kq >> 1), e = 4 (kq & 1) + r; column = position (a0 = i0, b = j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xbuf[xoff(2 * kt + (kq >> 1), i0, j, 4 * (kq & 1) + r)] = d[r];
    }
    wave_lds_fence();
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wy[kt][sp], bq[sp], d, 0, 0, 0);
        // rows k -> c = 2 kt + (kq >> 1), b = 4 (kq & 1) + r; column = position (a0 = i0, e = j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xbuf[xoff(2 * kt + (kq >> 1), i0, 4 * (kq & 1) + r, j)] += d[r];
    }
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float bt[8];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) bt[sp] = dbuf[dimg(4 * sp + kq, 16 * t + n)];
#pragma unroll
        for (int a0 = 0; a0 < 2; ++a0) {
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[Q][a0][sp], bt[sp], d, 0, 0, 0);
            // rows c = 4 kq + r; column = position (b = 2 t + i0, e = j)
#pragma unroll
            for (int r = 0; r < 4; ++r) xbuf[xoff(4 * kq + r, a0, 2 * t + i0, j)] += d[r];
        }
    }
    wave_lds_fence();
    // ---- dV += trilinear^T dX ----------------------------------------------------------------------------
    // Phase A, one voxel per lane (the gather's map): the 8 corner weights and row offsets go to a small table
    // that takes the place of the du image (reloaded at the start of the next quarter).
    float* ctab = dbuf;
    {
        const int e = lane & 7, a0 = (lane >> 3) & 1, b0 = (lane >> 4) & 1, b1 = (lane >> 5) & 1;
        const float x = (2.0f * e + 1.0f) * 0.125f - 1.0f;
        const float z = (2.0f * (2 * Q + a0) + 1.0f) * 0.125f - 1.0f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int b = 4 * p + 2 * b1 + b0;
            const float y = (2.0f * b + 1.0f) * 0.125f - 1.0f;
            float w[8];
            int a[8];
            tri_coef_dense(w, a, Rm, x, y, z);
            float* row = ctab + (a0 * 64 + b * 8 + e) * 16;
            *reinterpret_cast<f32x4*>(row + 0) = f32x4{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<f32x4*>(row + 4) = f32x4{w[4], w[5], w[6], w[7]};
            *reinterpret_cast<f32x4*>(row + 8) = f32x4{__int_as_float(a[0]), __int_as_float(a[1]), __int_as_float(a[2]),
                                                       __int_as_float(a[3])};
            *reinterpret_cast<f32x4*>(row + 12) = f32x4{__int_as_float(a[4]), __int_as_float(a[5]), __int_as_float(a[6]),
                                                        __int_as_float(a[7])};
        }
    }
    wave_lds_fence();
    // Phase B, one channel per lane: lane (c, vq) adds channel c of four voxels per step.  The four voxels of a
    // step are 4 apart in b and/or e, so their 2x2x2 corner footprints are disjoint (a rotation preserves
    // distances): no two lanes of an atomic ever hit the same address, and the 16 channels of a corner row are
    // 16 consecutive words.  The image is 64-bit fixed point because ds_add_f32 is ~40x slower than the integer
    // LDS atomics on gfx950 (tools/lds_atomic_probe.cpp: 771 vs 19 (u32) / 28 (u64) cycles per wave-instruction);
    // as a bonus the per-workgroup sums do not depend on the order of the adds.
    {
        const int c = lane & 15, vq = lane >> 4;
#pragma unroll 1
        for (int a0 = 0; a0 < 2; ++a0)
#pragma unroll 2
            for (int be = 0; be < 16; ++be) {
                const int b = (be >> 2) + 4 * (vq & 1), e = (be & 3) + 4 * (vq >> 1);
                const float d = xbuf[xoff(c, a0, b, e)] * fx_scale;
                const float* row = ctab + (a0 * 64 + b * 8 + e) * 16;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(row + 0), w1 = *reinterpret_cast<const f32x4*>(row + 4);
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(row + 8), o1 = *reinterpret_cast<const f32x4*>(row + 12);
                // unconditional: a zero weight adds 0 to a clamped (valid) row.  Branching on the weight put every
                // atomic in its own basic block behind an s_waitcnt lgkmcnt(0), i.e. serialised their latencies.
                long long fx[8];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    fx[nb] = to_fixed(w0[nb] * d);
                    fx[4 + nb] = to_fixed(w1[nb] * d);
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    lds_add_i64(dV + __float_as_int(o0[nb]) + c, fx[nb]);
                    lds_add_i64(dV + __float_as_int(o1[nb]) + c, fx[4 + nb]);
                }
            }
    }
    wave_lds_fence();
}

__global__ __launch_bounds__(kBwdThreads, 1) void score_backward_volume_kernel(
    const float* __restrict__ R, long r_batch_stride, const float* __restrict__ W1, int B, long N,
    const float* __restrict__ du_ws, const unsigned* __restrict__ du_max_bits, float* __restrict__ grad_vol)
{
    __shared__ __attribute__((aligned(16))) long long lds_dv[512 * 16];  // dense channel-last, 64-bit fixed point
    __shared__ __attribute__((aligned(16))) float lds_du[4 * 2048];      // per wave: du image, then the corner table
    __shared__ __attribute__((aligned(16))) float lds_x[4 * kQuarterFloats];
    __shared__ float lds_bound[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, row = lane & 15;
    float* dbuf = lds_du + wave * 2048;
    float* xbuf = lds_x + wave * kQuarterFloats;

    // |dX| <= (sum over the three slabs of max_k sum_o |W1[o][k]|) * max|du|: fixes the fixed-point scale per sample
    {
        float* colsum = lds_x;  // 384 floats of scratch before the hypothesis loop
        for (int k = tid; k < 384; k += kBwdThreads) {
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

    // W1^T fragments, A operands of dX: [k-tile][k-step over o]: W1[o = 4 s' + kq][k = base + row]
    float wx[8][8], wy[8][8], wz[4][2][8];
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        const float* w = W1 + (4 * sp + kq) * 384;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            wx[kt][sp] = w[16 * kt + row];
            wy[kt][sp] = w[128 + 16 * kt + row];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a0 = 0; a0 < 2; ++a0) wz[q][a0][sp] = w[256 + row * 8 + 2 * q + a0];
    }

    const long hstep = (long)gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        for (int i = tid; i < 512 * 16; i += kBwdThreads) lds_dv[i] = 0ll;
        __syncthreads();
        // every contribution w * dX (w <= 1) stays below 2^40 in units of 2^-fx_exp: 23 bits of headroom for the sum
        const float bound = w1_bound * __uint_as_float(du_max_bits[b]);
        int ex = 0;
        (void)frexpf(bound, &ex);
        const bool usable = bound > 0.0f && bound < 3.0e38f;
        // a word collects at most 8 corners x (hypotheses of this workgroup): give up precision bits, never range,
        // when that exceeds the 2^22 adds the 63-bit word has room for (only beyond ~500 000 hypotheses per workgroup)
        const long adds = 8 * ((N + gridDim.x - 1) / gridDim.x);
        const int spare = (adds > (1l << 22)) ? (64 - __builtin_clzl((unsigned long)(adds - 1))) - 22 : 0;
        const int fx_exp = usable ? min(max(40 - spare - ex, -80), 80) : 0;
        const float fx_scale = usable ? ldexpf(1.0f, fx_exp) : 0.0f;
        const float* Rb = R + (long)b * r_batch_stride;
        for (long h = (long)wave * gridDim.x + xcd_residue(blockIdx.x, gridDim.x, gridDim.y); h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
            const float* du_h = du_ws + ((long)b * N + h) * 2048;
            load_du_image(dbuf, du_h, lane); wave_lds_fence();
            bwd_vol_quarter<0>(wx, wy, wz, dbuf, xbuf, lds_dv, fx_scale, Rm, lane);
            load_du_image(dbuf, du_h, lane); wave_lds_fence();
            bwd_vol_quarter<1>(wx, wy, wz, dbuf, xbuf, lds_dv, fx_scale, Rm, lane);
            load_du_image(dbuf, du_h, lane); wave_lds_fence();
            bwd_vol_quarter<2>(wx, wy, wz, dbuf, xbuf, lds_dv, fx_scale, Rm, lane);
            load_du_image(dbuf, du_h, lane); wave_lds_fence();
            bwd_vol_quarter<3>(wx, wy, wz, dbuf, xbuf, lds_dv, fx_scale, Rm, lane);
        }
        __syncthreads();
        // non-finite upstream gradients: the bound is inf/NaN, nothing was accumulated -> report NaN like autograd would
        const float unscale = usable ? ldexpf(1.0f, -fx_exp) : (bound == 0.0f ? 0.0f : __builtin_nanf(""));
        float* gv = grad_vol + (long)b * (16 * 512);
        for (int i = tid; i < 16 * 512; i += kBwdThreads) {
            const int c = i >> 9, v = i & 511;
            const long long a = lds_dv[v * 16 + c];
            if (a != 0ll || !usable) global_add(gv + i, (float)a * unscale + (usable ? 0.0f : unscale));
        }
    }
}

// ---- host-side launcher -------------------------------------------------------------------------------
hipError_t launch_score_backward(const float* vol_src, const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                 const float* W1, const float* W2, const float* b2, int B, int64_t N,
                                 const float* grad_scores, float* du_ws, unsigned* du_max_bits, float* grad_vol,
                                 float* grad_feat_tgt,
                                 float* grad_W1, float* grad_W2, float* grad_b2, int num_cu, hipStream_t stream)
{
    hipError_t e;
    if ((e = hipMemsetAsync(grad_vol, 0, sizeof(float) * (size_t)B * 8192, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(grad_feat_tgt, 0, sizeof(float) * (size_t)B * 2048, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(grad_W1, 0, sizeof(float) * 32 * 384, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(grad_W2, 0, sizeof(float) * 32 * 32, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(grad_b2, 0, sizeof(float) * 32, stream)) != hipSuccess) return e;
    if (B == 0 || N == 0) return hipSuccess;
    if ((e = hipMemsetAsync(du_max_bits, 0, sizeof(unsigned) * (size_t)B, stream)) != hipSuccess) return e;
    int gy = B < num_cu ? B : num_cu;
    int gx = num_cu / gy;
    const int64_t need = (N + 3) / 4;
    if (gx > need) gx = (int)need;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, gy);
    hipLaunchKernelGGL(score_backward_head_kernel, grid, dim3(kBwdThreads), 0, stream, vol_src, feat_tgt, R,
                       (long)r_batch_stride, W1, W2, b2, B, (long)N, grad_scores, du_ws, du_max_bits, grad_feat_tgt, grad_W2,
                       grad_b2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(score_backward_w1_kernel, grid, dim3(kBwdThreads), 0, stream, vol_src, R, (long)r_batch_stride,
                       B, (long)N, du_ws, grad_W1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(score_backward_volume_kernel, grid, dim3(kBwdThreads), 0, stream, R, (long)r_batch_stride, W1,
                       B, (long)N, du_ws, du_max_bits, grad_vol);
    return hipGetLastError();
}

}  // namespace ahv
