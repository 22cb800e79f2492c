// ahv_pipeline.h -- software-pipelined body of the fused scorer.
//
// One wave per SIMD owns the matrix pipe alone, so nothing but its own instruction stream can
// fill the shadow of an MFMA (a v_mfma_f32_16x16x4_f32 occupies the pipe for 32 cycles but the
// issue port for only a few).  A pipeline STAGE therefore contracts quarter Q (192 MFMAs) while
// it gathers quarter Q+1 (2 passes x 64 voxels x 16 channels of trilinear blending), cut into
// 24 MICRO-STEPS of 8 MFMAs.  Each micro-step is its own scheduling region
// (__builtin_amdgcn_sched_barrier(0) is a scheduling boundary), small enough for
// sched_group_barrier to lay out "1 MFMA : 2 VALU : 1 DS" quickly and deterministically.
//
// micro-step I of a stage     MFMA side                         gather side (pass p = I / 12, i = I % 12)
//   every I                   8 MFMAs on B operands loaded in   i = 0      coordinates, floor/frac, masks
//                             step I-1; ds_read the 4 B          i = 1      8 corner weights + 8 row offsets
//                             operands of step I+1               i = 2..10  corner j=i-2: 4x ds_read_b128 (j<8),
//                                                                           16 FMAs of corner j-1 (j>0)
//                                                                i = 11     16x ds_write_b32 into the other buffer
#pragma once
#include "../../3dahv_amd/csrc/ahv_device.h"

namespace ahv {

struct LaneConst {
    int xb[2], yb[2], zb[4];  // B-operand offsets (floats) of this lane inside a quarter buffer
    int wr[2];                // gather write offsets of this lane for pass 0 / 1
    float x, y[2];            // normalised voxel-centre coordinates of this lane (w; h for pass 0 / 1)
    int a0;                   // d parity of this lane's voxel
};

__device__ __forceinline__ LaneConst make_lane_const(int lane)
{
    LaneConst lc;
    const int n = lane & 15, kq = lane >> 4, i0 = n >> 3, j = n & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        lc.xb[h] = qoff(i0, j, 4 * h + kq);
        lc.yb[h] = qoff(i0, 4 * h + kq, j);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) lc.zb[t] = (kq >> 1) * 128 + qoff(kq & 1, 2 * t + i0, j);
    const int e = lane & 7, a0 = (lane >> 3) & 1, b0 = (lane >> 4) & 1, b1 = (lane >> 5) & 1;
    lc.a0 = a0;
    lc.x = (2.0f * e + 1.0f) * 0.125f - 1.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int b = 4 * p + 2 * b1 + b0;
        lc.y[p] = (2.0f * b + 1.0f) * 0.125f - 1.0f;
        lc.wr[p] = qoff(a0, b, e);
    }
    return lc;
}

// B operand of k-step KS (0..95) of quarter Q: x slab KS<32, y slab KS<64, z slab otherwise.
template <int KS>
__device__ __forceinline__ float load_b(const float* rd, const LaneConst& lc)
{
    if constexpr (KS < 32) {
        return rd[(KS >> 1) * 128 + lc.xb[KS & 1]];
    } else if constexpr (KS < 64) {
        return rd[((KS - 32) >> 1) * 128 + lc.yb[KS & 1]];
    } else {
        return rd[((KS - 64) >> 2) * 256 + lc.zb[(KS - 64) & 3]];
    }
}

template <int Q, int KS>
__device__ __forceinline__ void mfma_pair(f32x4 (&acc)[2][4], const HeadFrags& f, float b)
{
    if constexpr (KS < 32) {
        acc[0][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ax[KS >> 1][KS & 1][0], b, acc[0][Q], 0, 0, 0);
        acc[1][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ax[KS >> 1][KS & 1][1], b, acc[1][Q], 0, 0, 0);
    } else if constexpr (KS < 64) {
        acc[0][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ay[(KS - 32) >> 1][KS & 1][0], b, acc[0][Q], 0, 0, 0);
        acc[1][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ay[(KS - 32) >> 1][KS & 1][1], b, acc[1][Q], 0, 0, 0);
    } else {
        constexpr int cp = (KS - 64) >> 2, t = (KS - 64) & 3;
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.az[Q][cp][0], b, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.az[Q][cp][1], b, acc[1][t], 0, 0, 0);
    }
}

// Gather state carried across micro-steps (all in registers).
struct TriState {
    float w0[3], w1[3];  // per-axis weights (x, y, z), zeroed when the neighbour is out of range
    int o0[3], o1[3];    // per-axis clamped row offsets
    TriCoef k;
    float out[16];
    f32x4 ld[2][4];
};

__device__ __forceinline__ void tri_step_axes(TriState& s, const float* Rm, float x, float y, float z)
{
    const float gx = Rm[0] * x + Rm[1] * y + Rm[2] * z;
    const float gy = Rm[3] * x + Rm[4] * y + Rm[5] * z;
    const float gz = Rm[6] * x + Rm[7] * y + Rm[8] * z;
    axis_coef(gx, s.w0[0], s.w1[0], s.o0[0], s.o1[0], kSrcStride);
    axis_coef(gy, s.w0[1], s.w1[1], s.o0[1], s.o1[1], 8 * kSrcStride);
    axis_coef(gz, s.w0[2], s.w1[2], s.o0[2], s.o1[2], kSrcPlaneRows * kSrcStride);
}

__device__ __forceinline__ void tri_step_corners(TriState& s)
{
    const float w00 = s.w0[2] * s.w0[1], w01 = s.w0[2] * s.w1[1], w10 = s.w1[2] * s.w0[1], w11 = s.w1[2] * s.w1[1];
    s.k.w[0] = w00 * s.w0[0]; s.k.w[1] = w00 * s.w1[0]; s.k.w[2] = w01 * s.w0[0]; s.k.w[3] = w01 * s.w1[0];
    s.k.w[4] = w10 * s.w0[0]; s.k.w[5] = w10 * s.w1[0]; s.k.w[6] = w11 * s.w0[0]; s.k.w[7] = w11 * s.w1[0];
    const int a00 = s.o0[2] + s.o0[1], a01 = s.o0[2] + s.o1[1], a10 = s.o1[2] + s.o0[1], a11 = s.o1[2] + s.o1[1];
    s.k.a[0] = a00 + s.o0[0]; s.k.a[1] = a00 + s.o1[0]; s.k.a[2] = a01 + s.o0[0]; s.k.a[3] = a01 + s.o1[0];
    s.k.a[4] = a10 + s.o0[0]; s.k.a[5] = a10 + s.o1[0]; s.k.a[6] = a11 + s.o0[0]; s.k.a[7] = a11 + s.o1[0];
}

template <int J>
__device__ __forceinline__ void tri_step_load(TriState& s, const float* srcT)
{
    const f32x4* row = reinterpret_cast<const f32x4*>(srcT + s.k.a[J]);
#pragma unroll
    for (int q = 0; q < 4; ++q) s.ld[J & 1][q] = row[q];
}

template <int J>
__device__ __forceinline__ void tri_step_fma(TriState& s)
{
    const float w = s.k.w[J];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (J == 0) s.out[4 * q + e] = w * s.ld[J & 1][q][e];
            else s.out[4 * q + e] += w * s.ld[J & 1][q][e];
        }
}

// Gather work of micro-step I for the quarter QN being produced.
template <int QN, int I>
__device__ __forceinline__ void tri_micro(TriState& s, float* wr, const float* srcT, const float* Rm,
                                          const LaneConst& lc)
{
    constexpr int p = I / 12, i = I % 12;
    if constexpr (i == 0) {
        const float z = (2.0f * (2 * QN + lc.a0) + 1.0f) * 0.125f - 1.0f;
        tri_step_axes(s, Rm, lc.x, lc.y[p], z);
    } else if constexpr (i == 1) {
        tri_step_corners(s);
    } else if constexpr (i <= 10) {
        constexpr int j = i - 2;
        if constexpr (j < 8) tri_step_load<(j < 8 ? j : 0)>(s, srcT);
        if constexpr (j > 0) tri_step_fma<(j > 0 ? j - 1 : 0)>(s);
    } else {
        float* dst = wr + lc.wr[p];
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c * 128] = s.out[c];
    }
}

// One micro-step: 8 MFMAs of quarter Q (k-steps 4I..4I+3), prefetch of the next 4 B operands,
// 1/24 of the gather of quarter QN.
template <int Q, int QN, int I>
__device__ __forceinline__ void micro_step(f32x4 (&acc)[2][4], float (&bcur)[4], const HeadFrags& f,
                                           const float* rd, float* wr, const float* srcT, const float* Rm,
                                           const LaneConst& lc, TriState& s)
{
    float bnext[4];
    if constexpr (I < 23) {
        bnext[0] = load_b<(I < 23 ? 4 * I + 4 : 0)>(rd, lc);
        bnext[1] = load_b<(I < 23 ? 4 * I + 5 : 0)>(rd, lc);
        bnext[2] = load_b<(I < 23 ? 4 * I + 6 : 0)>(rd, lc);
        bnext[3] = load_b<(I < 23 ? 4 * I + 7 : 0)>(rd, lc);
    }
    tri_micro<QN, I>(s, wr, srcT, Rm, lc);
    mfma_pair<Q, 4 * I + 0>(acc, f, bcur[0]);
    mfma_pair<Q, 4 * I + 1>(acc, f, bcur[1]);
    mfma_pair<Q, 4 * I + 2>(acc, f, bcur[2]);
    mfma_pair<Q, 4 * I + 3>(acc, f, bcur[3]);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);  // 1 DS
    }
    if constexpr (I < 23) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bcur[q] = bnext[q];
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int Q, int QN, int I>
struct MicroRun {
    static __device__ __forceinline__ void run(f32x4 (&acc)[2][4], float (&bcur)[4], const HeadFrags& f,
                                               const float* rd, float* wr, const float* srcT, const float* Rm,
                                               const LaneConst& lc, TriState& s)
    {
        micro_step<Q, QN, I>(acc, bcur, f, rd, wr, srcT, Rm, lc, s);
        if constexpr (I + 1 < 24) MicroRun<Q, QN, I + 1>::run(acc, bcur, f, rd, wr, srcT, Rm, lc, s);
    }
};

// Stage: contract quarter Q from `rd` while gathering quarter (Q+1)&3 (of the hypothesis whose
// rotation is Rm) into `wr`.
template <int Q>
__device__ __forceinline__ void pipelined_stage(f32x4 (&acc)[2][4], const HeadFrags& f, const float* rd,
                                                float* wr, const float* srcT, const float* Rm,
                                                const LaneConst& lc, TriState& s)
{
    float bcur[4];
    bcur[0] = load_b<0>(rd, lc);
    bcur[1] = load_b<1>(rd, lc);
    bcur[2] = load_b<2>(rd, lc);
    bcur[3] = load_b<3>(rd, lc);
    __builtin_amdgcn_sched_barrier(0);
    MicroRun<Q, (Q + 1) & 3, 0>::run(acc, bcur, f, rd, wr, srcT, Rm, lc, s);
    wave_lds_fence();
}

}  // namespace ahv
