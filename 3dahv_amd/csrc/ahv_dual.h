// ahv_dual.h -- two-waves-per-SIMD formulation of the fused scorer ("dual", score_variant 3).
//
// Why: on gfx950 an fp32 MFMA and VALU work never overlap (tools/valu_probe.cpp: an MFMA-only wave and
// a VALU-only wave on one SIMD take the SUM of their times), a lone wave issues a VALU instruction only
// every ~3.6 cycles (2.76 with two waves) and v_mfma_f32_16x16x4_f32 only every ~40 cycles (32 with
// two waves).  Two waves per SIMD therefore buy issue rate and hide each other's LDS / scalar latencies
// without any hand-built overlap schedule.  Two waves per SIMD means 256 registers per wave, so the
// 32x384 head matrix cannot stay in registers: it lives in LDS as ready-made A fragments
// (lane-linear, one conflict-free ds_read_b128 = the A operands of 4 MFMAs).
//
// Workgroup = 512 threads = 8 waves, one hypothesis per wave, no barrier on the hot path.
// LDS (152 KiB): source image 40 KiB + W1 fragment table 48 KiB + 8 x 8 KiB quarter images.
#pragma once
#include "ahv_device.h"

namespace ahv {

constexpr int kW1TableFloats = 48 * 64 * 4;  // 48 fragment groups x 64 lanes x 4 fragments

// Fragment group g, lane l, slot j (use order of gemm1_quarter_lds):
//   g <  16: x slab, c = g      : j -> (eh = j>>1, m = j&1)   W1[16m+row][      c*8 + 4eh + kq]
//   g <  32: y slab, c = g-16   : j -> (bh = j>>1, m = j&1)   W1[16m+row][128 + c*8 + 4bh + kq]
//   g >= 32: z slab, q = (g-32)>>2, cp = 2*((g-32)&3) + (j>>1), m = j&1
//                                                             W1[16m+row][256 + (2cp+(kq>>1))*8 + 2q+(kq&1)]
__device__ __forceinline__ float w1_table_entry(const float* __restrict__ W1, int g, int lane, int j)
{
    const int row = lane & 15, kq = lane >> 4, m = j & 1, hh = j >> 1;
    const float* w = W1 + (16 * m + row) * 384;
    if (g < 16) return w[g * 8 + 4 * hh + kq];
    if (g < 32) return w[128 + (g - 16) * 8 + 4 * hh + kq];
    const int q = (g - 32) >> 2, cp = 2 * ((g - 32) & 3) + hh;
    return w[256 + (2 * cp + (kq >> 1)) * 8 + 2 * q + (kq & 1)];
}

__device__ __forceinline__ void stage_w1_table(float* table, const float* __restrict__ W1, int tid, int nthreads)
{
    for (int i = tid; i < kW1TableFloats; i += nthreads) {
        const int j = i & 3, lane = (i >> 2) & 63, g = i >> 8;
        table[i] = w1_table_entry(W1, g, lane, j);
    }
}

struct DualFrags {
    float a2[2][4][2];  // GEMM2 [m][r][m2]: W2[16m2+row][16m + 4kq + r]
    f32x4 bias[2];      // [m2]: b2[16m2 + 4kq + r]
};

__device__ __forceinline__ void load_dual_frags(DualFrags& f, const float* __restrict__ W2,
                                                const float* __restrict__ b2, int lane)
{
    const int row = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) f.a2[m][r][m2] = W2[(16 * m2 + row) * 32 + 16 * m + 4 * kq + r];
#pragma unroll
        for (int r = 0; r < 4; ++r) f.bias[m2][r] = b2[16 * m2 + 4 * kq + r];
    }
}

// GEMM1 on quarter Q with the A operands streamed from the LDS fragment table.
template <int Q>
__device__ __forceinline__ void gemm1_quarter_lds(f32x4 (&acc)[2][4], const float* table, const float* buf, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    const f32x4* T = reinterpret_cast<const f32x4*>(table) + lane;  // group g at T[g * 64]
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x4 a = T[c * 64];
#pragma unroll
        for (int eh = 0; eh < 2; ++eh) {
            const float bx = buf[c * 128 + qoff(i0, j, 4 * eh + kq)];
            acc[0][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * eh + 0], bx, acc[0][Q], 0, 0, 0);
            acc[1][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * eh + 1], bx, acc[1][Q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x4 a = T[(16 + c) * 64];
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            const float by = buf[c * 128 + qoff(i0, 4 * bh + kq, j)];
            acc[0][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * bh + 0], by, acc[0][Q], 0, 0, 0);
            acc[1][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * bh + 1], by, acc[1][Q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int cpp = 0; cpp < 4; ++cpp) {
        const f32x4 a = T[(32 + 4 * Q + cpp) * 64];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cp = 2 * cpp + ci;
                const float bz = buf[(2 * cp + (kq >> 1)) * 128 + qoff(kq & 1, 2 * t + i0, j)];
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * ci + 0], bz, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * ci + 1], bz, acc[1][t], 0, 0, 0);
            }
    }
}

__device__ __forceinline__ void gemm2_dual(f32x4 (&v)[2][4], const f32x4 (&acc)[2][4], const DualFrags& f)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[0][t] = f.bias[0];
        v[1][t] = f.bias[1];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float u = __builtin_amdgcn_fmed3f(acc[m][t][r], 0.0f, __builtin_inff());  // relu, one instruction
                v[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][0], u, v[0][t], 0, 0, 0);
                v[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][1], u, v[1][t], 0, 0, 0);
            }
}

}  // namespace ahv
