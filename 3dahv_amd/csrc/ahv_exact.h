// ahv_exact.h -- the fused scorer's path for NON-FINITE inputs (a NaN / inf voxel or head weight).
//
// The hot loop (ahv_dual.h) clamps the 2x2x2 footprint of a sample INTO the volume and gives the corners that zeros padding
// drops a weight of exactly 0, and its ReLU is an integer max: both are exact for finite data and wrong next to a NaN or
// an inf -- 0 * inf = NaN where F.grid_sample (utils.py:129) SKIPS the out-of-range corner, and a NaN with the sign bit set
// becomes +0 where F.relu (modules/modules.py:68) propagates it.  Every workgroup reads all of a sample's voxels and the
// head weights while it stages them, so it knows for free whether the sample is finite (NonFinite below: one LDS word per
// kind, set by whichever thread sees a non-finite value); a sample that is not goes through the code in this file instead
// of the hot loop, one hypothesis per wave:
//   * the gather restates ATen's grid_sampler_3d corner by corner -- neighbours floor(i), floor(i) + 1 per axis, weights
//     (1 - t, t), a corner accumulated if and only if it lies inside the volume, in ATen's corner order, with ATen's weight
//     product (x * y) * z -- on the coordinate arithmetic of the CPU oracle (oracle/ahv_oracle.c: no contraction), so that
//     "inside" means the same thing in both;
//   * ReLU is  x < 0 ? 0 : x  (false for a NaN of either sign: it stays);
//   * GEMM1 / GEMM2 are the same fp32 MFMAs (IEEE: inf and NaN propagate), the normalisation and the score the hot loop's.
// Finite samples never come here, so nothing in this file is tuned: compact loops, W1 fragments from the LDS table (fp32
// instances) or straight from global memory (split-f16 instance, whose table holds f16 pairs).
// Pinned by tests/golden/nonfinite.npz (the REFERENCE's scores on inputs with one non-finite element, NaN mask and
// torch.max's index included): tests/test_gpu_verify.py::test_non_finite_inputs_match_the_reference.
#pragma once
#include "ahv_team.h"

namespace ahv {

// v_cmp_class_f32 mask: signalling NaN | quiet NaN | -inf | +inf
constexpr int kClassNonFinite = 0x1 | 0x2 | 0x4 | 0x200;

__device__ __forceinline__ bool non_finite(float x) { return __builtin_amdgcn_classf(x, kClassNonFinite); }

// Written while a sample (generation `gen` = samples this workgroup has staged so far + 1) or the head weights are staged,
// read behind the staging barrier.  Zeroed once per launch; a generation never repeats, so nothing is reset per sample.
struct NonFinite {
    unsigned src_gen;  // == gen: the source volume being staged holds a NaN / inf
    unsigned tgt_gen;  // == gen: the target volume (ahv_verify_pair_f32) does
    unsigned weights;  // != 0:   W1, W2 or b2 does (sticky for the launch)
};

// torch.relu: a NaN of either sign stays a NaN, -inf -> 0
__device__ __forceinline__ f32x4 relu4_exact(f32x4 x)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = x[r] < 0.0f ? 0.0f : x[r];
    return x;
}

// gemm2_dual with the propagating ReLU
__device__ __forceinline__ void gemm2_dual_exact(f32x4 (&v)[2][4], const f32x4 (&acc)[2][4], const DualFrags& f)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[0][t] = f.bias[0];
        v[1][t] = f.bias[1];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x4 u[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) u[t] = relu4_exact(acc[m][t]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][0], u[t][r], v[0][t], 0, 0, 0);
                v[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][1], u[t][r], v[1][t], 0, 0, 0);
            }
    }
}

// team_head with the propagating ReLU (the in-launch target features of a non-finite target volume)
__device__ __forceinline__ void team_head_exact(f32x4 (&v)[2], const f32x4 (&u)[2], const DualFrags& f)
{
    v[0] = f.bias[0];
    v[1] = f.bias[1];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x4 x = relu4_exact(u[m]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][0], x[r], v[0], 0, 0, 0);
            v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][1], x[r], v[1], 0, 0, 0);
        }
    }
}

// One axis as grid_sample sees it (oracle/ahv_oracle.c, unnormalize + floor): neighbours floor(i) and floor(i) + 1 with
// weights (1 - t, t); a neighbour counts if it lies in [0, 7].  The clamp keeps the integer conversion defined for any
// coordinate: inf lands beyond both neighbours' range, NaN is dropped by v_max (-> -2) -- out of range like ATen's cast.
struct ExactAxis {
    float w[2];
    int off[2];    // byte offset of the neighbour's row / plane in the source image (clamped into the image)
    bool in[2];
};

__device__ __forceinline__ void exact_axis(ExactAxis& a, float g, int scale_bytes)
{
#pragma clang fp contract(off)
    float i = ((g + 1.0f) * 8.0f - 1.0f) / 2.0f;
    i = fminf(fmaxf(i, -2.0f), 9.0f);
    const float fl = floorf(i);
    const float t = i - fl;
    const int i0 = (int)fl;
    a.w[0] = 1.0f - t;
    a.w[1] = t;
    a.in[0] = (unsigned)i0 < 8u;
    a.in[1] = (unsigned)(i0 + 1) < 8u;
    a.off[0] = min(max(i0, 0), 7) * scale_bytes;
    a.off[1] = min(max(i0 + 1, 0), 7) * scale_bytes;
}

// Quarter q (d in {2q, 2q+1}) of the rotated volume into the swizzled quarter image `buf`, corner by corner as ATen does
// it.  Lane -> voxel as in the round-1 gather: (w = l & 7, d0 = (l >> 3) & 1, h = 4 p + 2 (l >> 5) + ((l >> 4) & 1)).
__device__ __forceinline__ void exact_gather_quarter(float* buf, const float* srcT, const float (&Rm)[9], int q, int lane)
{
#pragma clang fp contract(off)
    const int e = lane & 7, a0 = (lane >> 3) & 1, b0 = (lane >> 4) & 1, b1 = lane >> 5;
    const float x = (2.0f * (float)e + 1.0f) / 8.0f - 1.0f;
    const float z = (2.0f * (float)(2 * q + a0) + 1.0f) / 8.0f - 1.0f;
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        const int b = 4 * p + 2 * b1 + b0;
        const float y = (2.0f * (float)b + 1.0f) / 8.0f - 1.0f;
        const float gx = Rm[0] * x + Rm[1] * y + Rm[2] * z;
        const float gy = Rm[3] * x + Rm[4] * y + Rm[5] * z;
        const float gz = Rm[6] * x + Rm[7] * y + Rm[8] * z;
        ExactAxis ax, ay, az;
        exact_axis(ax, gx, 4 * kSrcStride);
        exact_axis(ay, gy, 4 * kSrcRowsY * kSrcStride);
        exact_axis(az, gz, 4 * kSrcPlaneRows * kSrcStride);
        float o[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) o[c] = 0.0f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {  // ATen's corner order: (dz, dy, dx)
            const int dx = n & 1, dy = (n >> 1) & 1, dz = n >> 2;
            if (ax.in[dx] && ay.in[dy] && az.in[dz]) {  // an out-of-range corner is SKIPPED, whatever the voxel it would clamp to
                const float w = ax.w[dx] * ay.w[dy] * az.w[dz];
                const f32x4* row = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(srcT) + az.off[dz] + ay.off[dy] + ax.off[dx]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = row[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[4 * j + k] = o[4 * j + k] + w * v[k];
                }
            }
        }
        float* dst = buf + qoff(a0, b, e);
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c * 128] = o[c];
    }
}

// The same straight to global memory (rotate_volume): `oq` = out[n][0][128 q ...], channel planes 512 floats apart.
__device__ __forceinline__ void exact_gather_quarter_global(float* oq, const float* srcT, const float (&Rm)[9], int q, int lane)
{
#pragma clang fp contract(off)
    const int e = lane & 7, a0 = (lane >> 3) & 1, b0 = (lane >> 4) & 1, b1 = lane >> 5;
    const float x = (2.0f * (float)e + 1.0f) / 8.0f - 1.0f;
    const float z = (2.0f * (float)(2 * q + a0) + 1.0f) / 8.0f - 1.0f;
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        const int b = 4 * p + 2 * b1 + b0;
        const float y = (2.0f * (float)b + 1.0f) / 8.0f - 1.0f;
        const float gx = Rm[0] * x + Rm[1] * y + Rm[2] * z;
        const float gy = Rm[3] * x + Rm[4] * y + Rm[5] * z;
        const float gz = Rm[6] * x + Rm[7] * y + Rm[8] * z;
        ExactAxis ax, ay, az;
        exact_axis(ax, gx, 4 * kSrcStride);
        exact_axis(ay, gy, 4 * kSrcRowsY * kSrcStride);
        exact_axis(az, gz, 4 * kSrcPlaneRows * kSrcStride);
        float o[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) o[c] = 0.0f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int dx = n & 1, dy = (n >> 1) & 1, dz = n >> 2;
            if (ax.in[dx] && ay.in[dy] && az.in[dz]) {
                const float w = ax.w[dx] * ay.w[dy] * az.w[dz];
                const f32x4* row = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(srcT) + az.off[dz] + ay.off[dy] + ax.off[dx]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = row[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[4 * j + k] = o[4 * j + k] + w * v[k];
                }
            }
        }
        float* dst = oq + a0 * 64 + b * 8 + e;
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c * 512] = o[c];
    }
}

// W1 A fragments of table group g for this lane: from the LDS fragment table, or (split-f16 instance) from W1 itself
struct W1FragsLds {
    const f32x4* T;  // table + lane
    __device__ __forceinline__ f32x4 operator()(int g) const { return T[g * 64]; }
};
struct W1FragsGlobal {
    const float* W1;
    int lane;
    __device__ __forceinline__ f32x4 operator()(int g) const
    {
        return f32x4{w1_table_entry(W1, g, lane, 0), w1_table_entry(W1, g, lane, 1), w1_table_entry(W1, g, lane, 2),
                     w1_table_entry(W1, g, lane, 3)};
    }
};

// GEMM1 on the quarter in `buf`, quarter index q known at run time: gemm1_quarter_lds's MFMAs in compact loops
template <typename AF>
__device__ __forceinline__ void gemm1_quarter_exact(f32x4 (&acc)[2][4], int q, const AF& afrag, const float* buf, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    f32x4 xy[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
    for (int c = 0; c < 32; ++c) {  // x slab (groups 0-15), then y slab (16-31)
        const f32x4 a = afrag(c);
        const int cc = c & 15;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float bv = c < 16 ? buf[cc * 128 + qoff(i0, j, 4 * hh + kq)] : buf[cc * 128 + qoff(i0, 4 * hh + kq, j)];
            xy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * hh + 0], bv, xy[0], 0, 0, 0);
            xy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * hh + 1], bv, xy[1], 0, 0, 0);
        }
    }
#pragma unroll 1
    for (int cpp = 0; cpp < 4; ++cpp) {
        const f32x4 a = afrag(32 + 4 * q + cpp);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bz = buf[(2 * (2 * cpp + ci) + (kq >> 1)) * 128 + qoff(kq & 1, 2 * t + i0, j)];
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * ci + 0], bz, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * ci + 1], bz, acc[1][t], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t == q) {  // x / y slabs of quarter q feed position tile q
            acc[0][t] += xy[0];
            acc[1][t] += xy[1];
        }
}

// Pre-activations of one hypothesis, exact path: acc[m][t] = rows of m-tile m at the positions of tile t
template <typename AF>
__device__ __forceinline__ void exact_hypothesis_gemm1(f32x4 (&acc)[2][4], float* buf, const float* srcT, const float (&Rm)[9],
                                                       const AF& afrag, int lane)
{
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        exact_gather_quarter(buf, srcT, Rm, q, lane);
        wave_lds_fence();
        gemm1_quarter_exact(acc, q, afrag, buf, lane);
        wave_lds_fence();
    }
}

}  // namespace ahv
