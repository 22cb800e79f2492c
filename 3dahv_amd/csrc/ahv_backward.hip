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
//   du[o][pos], kDuStride = 68 floats per row (multiple of 4: the image is filled with aligned float4 stores).
constexpr int kXwStride = 129;
constexpr int kXwFloats = 16 * kXwStride;
constexpr int kDuStride = 68;
constexpr int kDuFloats = 32 * kDuStride;

struct DuImage {  // one hypothesis's du (2048 floats) as the wave loads it: 8 coalesced float4 per lane
    f32x4 v[8];
};

__device__ __forceinline__ void load_du_image(DuImage& d, const float* __restrict__ du_hyp, int lane)
{
    const f32x4* src = reinterpret_cast<const f32x4*>(du_hyp);
#pragma unroll
    for (int i = 0; i < 8; ++i) d.v[i] = __builtin_nontemporal_load(src + i * 64 + lane);
}

__device__ __forceinline__ void store_du_image(float* dbuf, const DuImage& d, int lane)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = 4 * (i * 64 + lane), o = idx >> 6, pos = idx & 63;
        *reinterpret_cast<f32x4*>(dbuf + o * kDuStride + pos) = d.v[i];
    }
}

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
        // this wave's half of du: float4 chunks 4 role .. 4 role + 3 of the hypothesis's 2048 floats = rows 16 role ..
        f32x4 duh[4];
        if (h < N) {
            const f32x4* src = reinterpret_cast<const f32x4*>(du_ws + ((long)b * N + h) * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) duh[i] = __builtin_nontemporal_load(src + (4 * role + i) * 64 + lane);
        }
        for (; h < N; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = Rb[h * 9 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // this hypothesis's du rows, requested one iteration ago
                const int idx = 4 * (i * 64 + lane), o = idx >> 6, pos = idx & 63;
                *reinterpret_cast<f32x4*>(dbuf + o * kDuStride + pos) = duh[i];
            }
            {   // the next hypothesis's travel meanwhile
                const long hn = (h + hstep < N) ? h + hstep : h;
                const f32x4* src = reinterpret_cast<const f32x4*>(du_ws + ((long)b * N + hn) * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) duh[i] = __builtin_nontemporal_load(src + (4 * role + i) * 64 + lane);
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
    const float* p = du_hyp + (lane >> 4) * 64 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) d.v[t][sp] = p[sp * 256 + 16 * t];
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

hipError_t launch_zero_fill(void* const* ptrs, const size_t* bytes, int count, hipStream_t stream);  // ahv_ops.hip

// ---- host-side launcher -------------------------------------------------------------------------------
hipError_t launch_score_backward(const float* vol_src, const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                 const float* W1, const float* W2, const float* b2, int B, int64_t N,
                                 const float* grad_scores, float* du_ws, unsigned* du_max_bits, float* dw1_partials,
                                 float* grad_vol, float* grad_feat_tgt,
                                 float* grad_W1, float* grad_W2, float* grad_b2, int num_cu, hipStream_t stream)
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
    hipLaunchKernelGGL(score_backward_volume_kernel, grid, dim3(kVolThreads), 0, stream, R, (long)r_batch_stride, W1,
                       B, (long)N, du_ws, du_max_bits, grad_vol);
    return hipGetLastError();
}

}  // namespace ahv
