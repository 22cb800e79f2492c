// ahv_wide.h -- the 32x32x2 fp32-MFMA formulation of the per-hypothesis work.
//
// Measured on MI355X (tools/mfma_peak.cpp, one wave per SIMD): v_mfma_f32_32x32x2_f32 issues
// back to back at its full 64 cycles (32 per 2048 flops), v_mfma_f32_16x16x4_f32 only every
// ~40.5, and -- unlike the bf16 matrix pipe -- an fp32 MFMA does NOT hide VALU work: every
// VALU/DS instruction beside it costs its ~4 issue cycles on top.  So the fastest stream is
// the one with the fewest instructions, latencies hidden by in-wave ILP (prefetch distance),
// not a VALU/MFMA overlap schedule.
//
// One wave owns one hypothesis.  Per HALF volume (d in 4H..4H+3: 256 voxels x 16 channels =
// 16 KiB private LDS image) it gathers (4 passes x 64 voxels) and then contracts:
//   x slab: positions (d,h) of this half = n-tile H, k=(c,w)          64 MFMA 32x32x2
//   y slab: positions (d,w) of this half = n-tile H, k=(c,h)          64
//   z slab: all positions (h,w) = both n-tiles, k=(c,d in this half)  64
// Rows of every MFMA = the 32 head channels (A = W1 fragments, 192 VGPRs resident),
// columns = 32 positions (tile T: i in 4T..4T+3, j < 8).
#pragma once
#include "../../3dahv_amd/csrc/ahv_device.h"

namespace ahv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Half-volume image: [16 c][256] floats; voxel (a = d&3, b = h, e = w) of a plane at
//   P = ((e^b)&7) | (((a^b)&3)<<3) | (b<<5)
// -> bank bits [e0^b0, e1^b1, e2^b2, a0^b0, a1^b1]: the x / y / z B-operand reads (32 lanes
// varying (a,b) / (a,e) / (b&3,e)) and the gather's writes (lanes varying (e,a)) are all
// ds_*_b32 bank-conflict free.
constexpr int kHalfFloats = 16 * 256;  // 16 KiB

__device__ __forceinline__ int hoff(int a, int b, int e)
{
    return ((e ^ b) & 7) | (((a ^ b) & 3) << 3) | (b << 5);
}

// v_mfma_f32_32x32x2_f32: A[row = lane&31][k = lane>>5], B[k = lane>>5][col = lane&31],
// D[row = 8*(reg>>2) + 4*(lane>>5) + (reg&3)][col = lane&31], reg < 16.
struct WideFrags {
    float ax[16][4];  // [c][eh]: W1[row][      c*8 + 2*eh + half]
    float ay[16][4];  // [c][bh]: W1[row][128 + c*8 + 2*bh + half]
    float az[16][4];  // [c][ah]: W1[row][256 + c*8 + 2*ah + half]
    float a2[16];     // [r]    : W2[row][8*(r>>2) + 4*half + (r&3)]
    f32x16 bias;      // b2[8*(r>>2) + 4*half + (r&3)]
};

__device__ __forceinline__ void load_wide_frags(WideFrags& f, const float* __restrict__ W1,
                                                const float* __restrict__ W2,
                                                const float* __restrict__ b2, int lane)
{
    const int row = lane & 31, half = lane >> 5;
    const float* w = W1 + row * 384;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f.ax[c][p] = w[c * 8 + 2 * p + half];
            f.ay[c][p] = w[128 + c * 8 + 2 * p + half];
            f.az[c][p] = w[256 + c * 8 + 2 * p + half];
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 8 * (r >> 2) + 4 * half + (r & 3);
        f.a2[r] = W2[row * 32 + o];
        f.bias[r] = b2[o];
    }
}

struct WideLane {
    int xb[4], yb[4], zb[2][2];  // B-operand offsets (floats) inside a half image: [pair], z: [tile][local pair]
    int wr[4];                   // gather write offset per pass
    float x, y[4], z0;           // voxel-centre coordinates of this lane: w; h per pass; d of half 0
};

__device__ __forceinline__ WideLane make_wide_lane(int lane)
{
    WideLane L;
    const int half = lane >> 5, n = lane & 31, il = n >> 3, j = n & 7;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        L.xb[p] = hoff(il, j, 2 * p + half);
        L.yb[p] = hoff(il, 2 * p + half, j);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int al = 0; al < 2; ++al) L.zb[t][al] = hoff(2 * al + half, 4 * t + il, j);
    // gather: lane -> (w = l&7, d&3 = (l>>3)&3, h = 2*pass + (l>>5))
    const int e = lane & 7, a = (lane >> 3) & 3, b0 = lane >> 5;
    L.x = (2.0f * e + 1.0f) * 0.125f - 1.0f;
    L.z0 = (2.0f * a + 1.0f) * 0.125f - 1.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int b = 2 * p + b0;
        L.y[p] = (2.0f * b + 1.0f) * 0.125f - 1.0f;
        L.wr[p] = hoff(a, b, e);
    }
    return L;
}

// ---- gather of one half volume ---------------------------------------------------------
// 4 passes x 8 corners, flattened into 32 steps; the 4x ds_read_b128 of step s+AHEAD are issued
// before the 16 FMAs of step s, so the LDS latency hides behind the wave's own arithmetic.
#ifndef AHV_GATHER_AHEAD
#define AHV_GATHER_AHEAD 3
#endif

template <int H>
__device__ __forceinline__ void gather_half(float* buf, const float* srcT, const float* Rm, const WideLane& L)
{
    constexpr int AHEAD = AHV_GATHER_AHEAD, RING = AHEAD + 1;
    TriCoef k[4];
    const float z = L.z0 + (float)H;  // d += 4 per half -> +1.0 in normalised units
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_coef(k[p], Rm, L.x, L.y[p], z);
#ifdef AHV_DIAG_LINEAR_GATHER  // diagnostic only: conflict-free addresses (wrong results) to price bank conflicts
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 8; ++n) k[p].a[n] = (((int)threadIdx.x & 63) + 64 * n) * kSrcStride;  // rows 0..511 < 8 * kSrcPlaneRows
#endif
    f32x4 ld[RING][4];
    // Source order = loads of step s+AHEAD, then the FMAs of step s.  AHV_GATHER_PIN pins that order with
    // sched_barrier(0); measured slower than letting hipcc re-place the loads (it also hoists the next
    // half's coordinate math into the GEMM phase), so it is off by default.
#ifdef AHV_GATHER_PIN
#define AHV_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define AHV_PIN()
#endif
    AHV_PIN();
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
        const f32x4* row = reinterpret_cast<const f32x4*>(srcT + k[s >> 3].a[s & 7]);
#pragma unroll
        for (int q = 0; q < 4; ++q) ld[s % RING][q] = row[q];
    }
    AHV_PIN();
    float out[16];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        if (s + AHEAD < 32) {
            const f32x4* row = reinterpret_cast<const f32x4*>(srcT + k[(s + AHEAD) >> 3].a[(s + AHEAD) & 7]);
#pragma unroll
            for (int q = 0; q < 4; ++q) ld[(s + AHEAD) % RING][q] = row[q];
        }
        AHV_PIN();
        const float w = k[s >> 3].w[s & 7];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((s & 7) == 0) out[4 * q + e] = w * ld[s % RING][q][e];
                else out[4 * q + e] += w * ld[s % RING][q][e];
            }
        if ((s & 7) == 7) {
            float* dst = buf + L.wr[s >> 3];
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c * 256] = out[c];
        }
        AHV_PIN();
    }
}

// ---- GEMM1 on one half -----------------------------------------------------------------
template <int H>
__device__ __forceinline__ void gemm1_half(f32x16 (&acc)[2], const WideFrags& f, const float* buf,
                                           const WideLane& L)
{
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            acc[H] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.ax[c][p], buf[c * 256 + L.xb[p]], acc[H], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            acc[H] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.ay[c][p], buf[c * 256 + L.yb[p]], acc[H], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int al = 0; al < 2; ++al)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.az[c][2 * H + al], buf[c * 256 + L.zb[t][al]],
                                                              acc[t], 0, 0, 0);
}

// ---- ReLU + GEMM2 + bias from the accumulators (no lane movement) ------------------------
__device__ __forceinline__ void gemm2_wide(f32x16 (&v)[2], const f32x16 (&acc)[2], const WideFrags& f)
{
    v[0] = f.bias;
    v[1] = f.bias;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // relu: med3(x, 0, +inf) = max(x, 0) in one instruction
            const float u = __builtin_amdgcn_fmed3f(acc[t][r], 0.0f, __builtin_inff());
            v[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a2[r], u, v[t], 0, 0, 0);
        }
}

// F.normalize over channels, dot with the unit-norm target, mean over the 64 positions.
// A lane holds 16 of the 32 channels of position (32t + lane&31); the other 16 are in lane^32.
__device__ __forceinline__ float score_wide(const f32x16 (&v)[2], const f32x16 (&tg)[2])
{
    float tot = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float ss = 0.0f, dt = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ss += v[t][r] * v[t][r];
            dt += v[t][r] * tg[t][r];
        }
        ss += __shfl_xor(ss, 32, 64);
        dt += __shfl_xor(dt, 32, 64);
        tot += dt / fmaxf(sqrtf(ss), 1e-12f);
    }
    // every position is now counted twice (both halves hold its total): sum / 128
    return wave_sum_dpp(tot) * (1.0f / 128.0f);
}

}  // namespace ahv
