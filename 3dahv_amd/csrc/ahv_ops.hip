// ahv_ops.hip -- op-level drop-in kernels: each materialises the tensor the
// reference's corresponding call returns, so the reference's own call sequence
// (rotate_volume -> forward_3d2d -> mul/sum/mean -> max) runs unchanged on HIP.
// These are the HBM-bound siblings of the fused scorer (ahv_score.hip).
#include "ahv_device.h"
#include "ahv_dual.h"
#include "ahv_exact.h"

namespace ahv {

// ---------------------------------------------------------------------------------
// rotate_volume, fast path: volume (16,8,8,8) shared by all N hypotheses (the stride-0
// expand of test_co3d.py:137).  HBM: 36 B in + 32 KiB out per hypothesis -> write-bandwidth bound,
// provided the kernel needs less than the ~3 500 cycles per hypothesis and CU that 5.7 TB/s leave it.
// Rounds 1-4 (tri_coef + tri_blend per voxel) needed ~2 100 vector instructions per hypothesis and reached 5.14 TB/s.
// Round 5: the fused scorer's gather -- hat weights on a clamped base row, one base address per voxel, the request ring,
// packed FMAs, the point-mirror quarters (ahv_dual.h: ~900 vector instructions per hypothesis) -- with a store that goes
// straight to global memory: one non-temporal 4-byte store per channel and pass, two whole 128-byte lines per instruction.
// No LDS besides the source image (47.5 KiB) -> three workgroups = 12 waves per CU at <= 168 registers (ring depth 2:
// 3 and 4 rows spill).  First built with a per-wave LDS image and 16-byte stores (8 waves per CU): 1.245 ms against
// 1.19-1.21 ms for this one on the same box, N = 200 000 (5.27 vs 5.43-5.52 TB/s; the minimum of ten launches 5.8).
// What bounds it is the store stream itself: the same kernel with the gather removed (-DAHV_DIAG_ROT_STORE_ONLY) and
// tools/store_probe.cpp (this store pattern, others, and a plain fill, from the same persistent grid) write 6.55 GB at
// 5.3-6.1 TB/s whatever the pattern -- one 32 KiB region per wave, ~3 000 regions open at once -- and the gather with its
// bank conflicts switched off (-DAHV_DIAG_LINEAR_GATHER) is no faster.
// A NaN / inf voxel: the workgroup sees it while staging and every hypothesis of the launch goes through
// exact_gather_quarter_global (ahv_exact.h: grid_sample's per-corner zeros padding, utils.py:129) instead.
// ---------------------------------------------------------------------------------
constexpr int kRotThreads = 256;
#ifndef AHV_DIAG_ROT_DEPTH
#define AHV_DIAG_ROT_DEPTH 2
#endif
constexpr int kRotDepth = AHV_DIAG_ROT_DEPTH;  // rows the gather requests ahead: 2 -> 160 registers, 3 / 4 -> 8 / 22 spills and slower

// Where a blended voxel goes: straight to out[n][c][...], one non-temporal 4-byte store per channel.  In a pass the 64 lanes
// own the voxels (a0, 4 p + bq, e) of the quarter -- two runs of 32 consecutive floats per channel plane -- so every store
// instruction writes two whole 128-byte lines.
struct RotStoreGlobal {
    static constexpr bool kXdlKernel = kFp32LowHalf;
    static constexpr int kDepth = kRotDepth;
    float* d[2];  // the lane's voxel of pass 0 / pass 1 in channel plane 0 of this quarter
    __device__ __forceinline__ void operator()(int p, const f32x2 (&o)[8]) const
    {
        float* dst = d[p];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#ifdef AHV_DIAG_ROT_TEMPORAL
            dst[c * 512] = o[c >> 1][c & 1];
#else
            __builtin_nontemporal_store(o[c >> 1][c & 1], dst + c * 512);
#endif
        }
    }
};

template <bool MIR>
__device__ __forceinline__ void rot_quarter(HatState& st, float* oq, const GatherDst& dst)
{
    f32x2 o[8];
    const RotStoreGlobal store = {{oq + (MIR ? dst.m0 : dst.o0), oq + (MIR ? dst.m1 : dst.o1)}};
#ifdef AHV_DIAG_ROT_STORE_ONLY   // the store stream alone (wrong results): what the gather costs on top of it
    for (int c = 0; c < 8; ++c) o[c] = f32x2{st.vx[0].w[0], st.vx[1].w[0]};
    store(0, o);
    store(1, o);
#else
    HatSteps<0, RotStoreGlobal, MIR>::run(st, o, store);
#endif
}

__global__ __launch_bounds__(kRotThreads, 3) void rotate_volume_16x8_kernel(
    const float* __restrict__ vol, const float* __restrict__ R, long N, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(1024))) float srcT[kSrcFloats];
    __shared__ unsigned nf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) nf = 0u;
    __syncthreads();
    {
        bool bad = false;
        for (int i = tid; i < 16 * 512; i += kRotThreads) {
            const int c = i >> 9, v = i & 511;
            const float x = vol[i];
            bad = bad || non_finite(x);
            srcT[((v >> 6) * kSrcPlaneRows + ((v >> 3) & 7) * kSrcRowsY + (v & 7)) * kSrcStride + c] = x;
        }
        if (bad) nf = 1u;
    }
    __syncthreads();
    const bool exact = __builtin_amdgcn_readfirstlane((int)nf) != 0;
    const GatherLane glane = gather_lane(lane);
    GatherDst gdst = gather_dst_linear(lane);
    asm volatile("" : "+v"(gdst.m0), "+v"(gdst.m1));
    const int rl = lane < 9 ? lane : 8;
    const long nstep = (long)gridDim.x * 4;
    long n = (long)blockIdx.x * 4 + wave;
    float Rn = n < N ? R[n * 9 + rl] : 0.0f;
    for (; n < N; n += nstep) {
        float Rm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Rn), i));
        Rn = R[(n + nstep < N ? n + nstep : n) * 9 + rl];
        float* o = out + n * (16 * 512);
        if (!exact) {
            GatherHyp gh;
            gather_hyp(gh, Rm, glane);
            HatState st;
            // quarters in the order 0, 3, 1, 2: quarter 3 - Q is the point mirror of quarter Q and reuses its set-up
#ifdef AHV_DIAG_ROT_STORE_ONLY
            st.vx[0].w[0] = gh.ixy[0][0]; st.vx[1].w[0] = gh.izp[1];
            rot_quarter<false>(st, o, gdst); rot_quarter<true>(st, o + 3 * 128, gdst);
            rot_quarter<false>(st, o + 128, gdst); rot_quarter<true>(st, o + 2 * 128, gdst);
            continue;
#endif
            hat_prologue<0, kFp32LowHalf, kRotDepth>(st, srcT, gh);
            rot_quarter<false>(st, o, gdst);
            hat_prologue_mirror<kRotDepth>(st, srcT);
            rot_quarter<true>(st, o + 3 * 128, gdst);
            hat_prologue<1, kFp32LowHalf, kRotDepth>(st, srcT, gh);
            rot_quarter<false>(st, o + 128, gdst);
            hat_prologue_mirror<kRotDepth>(st, srcT);
            rot_quarter<true>(st, o + 2 * 128, gdst);
        } else {
            // a NaN / inf voxel: every hypothesis corner by corner as grid_sample does it (ahv_exact.h), through the
            // same lane -> voxel map as the stores above expect nothing of: each lane writes its own voxels
#pragma unroll 1
            for (int q = 0; q < 4; ++q) exact_gather_quarter_global(o + q * 128, srcT, Rm, q, lane);
        }
    }
}

// ---------------------------------------------------------------------------------
// rotate_volume, generic path: any C, D, H, W and any batch stride (utils.py:113-131
// accepts every 5-D volume).  One thread per output voxel, channels looped; the source
// is read through the caches.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void axis_generic(float g, int size, float& w0, float& w1, long& i0, long& i1, bool& in0, bool& in1)
{
    float i = ((g + 1.0f) * (float)size - 1.0f) * 0.5f;
    i = fminf(fmaxf(i, -2.0f), (float)size + 1.0f);
    const float fl = floorf(i);
    const float t = i - fl;
    const int a = (int)fl, b = a + 1;
    in0 = a >= 0 && a < size;
    in1 = b >= 0 && b < size;
    w0 = in0 ? 1.0f - t : 0.0f;
    w1 = in1 ? t : 0.0f;
    i0 = min(max(a, 0), size - 1);
    i1 = min(max(b, 0), size - 1);
}

__global__ __launch_bounds__(256) void rotate_volume_generic_kernel(
    const float* __restrict__ vol, long vol_batch_stride, const float* __restrict__ R, long N, int C, int D,
    int H, int W, float* __restrict__ out)
{
    const long plane = (long)D * H * W;
    const long total = N * plane;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / plane;
        const long v = i - n * plane;
        const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
        const float* r = R + n * 9;
        const float x = (2.0f * w + 1.0f) / (float)W - 1.0f;
        const float y = (2.0f * h + 1.0f) / (float)H - 1.0f;
        const float z = (2.0f * d + 1.0f) / (float)D - 1.0f;
        const float gx = r[0] * x + r[1] * y + r[2] * z;
        const float gy = r[3] * x + r[4] * y + r[5] * z;
        const float gz = r[6] * x + r[7] * y + r[8] * z;
        float wx[2], wy[2], wz[2];
        long ox[2], oy[2], oz[2];
        bool ix[2], iy[2], iz[2];
        axis_generic(gx, W, wx[0], wx[1], ox[0], ox[1], ix[0], ix[1]);
        axis_generic(gy, H, wy[0], wy[1], oy[0], oy[1], iy[0], iy[1]);
        axis_generic(gz, D, wz[0], wz[1], oz[0], oz[1], iz[0], iz[1]);
        float wgt[8];
        long off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            wgt[k] = wz[dz] * wy[dy] * wx[dx];
            // zeros padding is per corner: an out-of-range corner is SKIPPED (its weight here is 0, but 0 * a non-finite
            // voxel would be NaN where F.grid_sample leaves the voxel out); an in-range corner counts even with weight 0
            off[k] = (iz[dz] && iy[dy] && ix[dx]) ? (oz[dz] * H + oy[dy]) * W + ox[dx] : -1;
        }
        const float* src = vol + n * vol_batch_stride;
        float* o = out + n * C * plane + v;
        for (int c = 0; c < C; ++c) {
            const float* sc = src + c * plane;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (off[k] >= 0) acc += wgt[k] * sc[off[k]];
            o[c * plane] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// Adjoint of rotate_volume w.r.t. the volume (the reference's rotate_volume is differentiable and
// infoNCE_loss back-propagates through it, modules/model_co3d.py:49-54): every output voxel scatters
// its gradient to the 8 trilinear corners it was blended from, with the forward's weights (zero for
// corners outside the volume).  grad_vol is zeroed by the caller; sums use float atomics, so the
// result is reproducible to rounding, not bitwise.  With vol_batch_stride = 0 (the stride-0 expand
// the reference passes) all N hypotheses accumulate into ONE volume.  Compatibility path for
// unmodified reference scripts; the training path proper is the fused backward (ahv_backward.hip).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotate_volume_backward_kernel(
    const float* __restrict__ grad_out, long vol_batch_stride, const float* __restrict__ R, long N, int C, int D,
    int H, int W, float* __restrict__ grad_vol)
{
    const long plane = (long)D * H * W;
    const long total = N * plane;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / plane;
        const long v = i - n * plane;
        const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
        const float* r = R + n * 9;
        const float x = (2.0f * w + 1.0f) / (float)W - 1.0f;
        const float y = (2.0f * h + 1.0f) / (float)H - 1.0f;
        const float z = (2.0f * d + 1.0f) / (float)D - 1.0f;
        const float gx = r[0] * x + r[1] * y + r[2] * z;
        const float gy = r[3] * x + r[4] * y + r[5] * z;
        const float gz = r[6] * x + r[7] * y + r[8] * z;
        float wx[2], wy[2], wz[2];
        long ox[2], oy[2], oz[2];
        bool ix[2], iy[2], iz[2];
        axis_generic(gx, W, wx[0], wx[1], ox[0], ox[1], ix[0], ix[1]);
        axis_generic(gy, H, wy[0], wy[1], oy[0], oy[1], iy[0], iy[1]);
        axis_generic(gz, D, wz[0], wz[1], oz[0], oz[1], iz[0], iz[1]);
        float wgt[8];
        long off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            wgt[k] = wz[dz] * wy[dy] * wx[dx];
            off[k] = (oz[dz] * H + oy[dy]) * W + ox[dx];
            if (!(iz[dz] && iy[dy] && ix[dx])) wgt[k] = 0.0f;  // skipped below: an out-of-range corner receives nothing
        }
        float* dst = grad_vol + n * vol_batch_stride;
        const float* g = grad_out + n * C * plane + v;
        for (int c = 0; c < C; ++c) {
            const float go = g[c * plane];
            float* dc = dst + c * plane;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (wgt[k] != 0.0f)
                    __hip_atomic_fetch_add(dc + off[k], wgt[k] * go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------
// forward_3d2d (modules/modules.py:112-124) on materialised volumes [M][16][8][8][8].
// Same wave-per-item MFMA contraction as the fused scorer; the quarter buffers are
// filled from HBM instead of by the trilinear gather.  32 KiB in + 8 KiB out per item.
// ---------------------------------------------------------------------------------
constexpr int kF32Threads = 256;
constexpr int kF32LdsFloats = 4 * 2 * kQuarterFloats;

template <int Q>
__device__ __forceinline__ void stage_quarter(float* buf, const float* __restrict__ vol, int lane)
{
    // quarter Q of channel c = 128 contiguous floats at c*512 + Q*128; a lane moves 2 of them
    const int i = 2 * lane;
    const int a0 = i >> 6, b = (i >> 3) & 7, e = i & 7;
    const int o0 = qoff(a0, b, e), o1 = qoff(a0, b, e + 1);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float2 v = *reinterpret_cast<const float2*>(vol + c * 512 + Q * 128 + i);
        buf[c * 128 + o0] = v.x;
        buf[c * 128 + o1] = v.y;
    }
}

__global__ __launch_bounds__(kF32Threads, 1) void forward_3d2d_kernel(
    const float* __restrict__ vol, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, long M, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* buf0 = smem + wave * (2 * kQuarterFloats);
    float* buf1 = buf0 + kQuarterFloats;
    HeadFrags f;
    load_head_frags(f, W1, W2, b2, lane);
    const int n16 = lane & 15, kq = lane >> 4;
    for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
        const float* V = vol + m * (16 * 512);
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        stage_quarter<0>(buf0, V, lane);
        stage_quarter<1>(buf1, V, lane);
        wave_lds_fence();
        gemm1_quarter<0>(acc, f, buf0, lane);
        gemm1_quarter<1>(acc, f, buf1, lane);
        wave_lds_fence();
        stage_quarter<2>(buf0, V, lane);
        stage_quarter<3>(buf1, V, lane);
        wave_lds_fence();
        gemm1_quarter<2>(acc, f, buf0, lane);
        gemm1_quarter<3>(acc, f, buf1, lane);
        wave_lds_fence();
        f32x4 v[2][4];
        gemm2(v, acc, f);
        float* o = out + m * (32 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float ss = 0.0f;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += v[m2][t][r] * v[m2][t][r];
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize clamp_min(eps)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n16] = v[m2][t][r] / nrm;
        }
    }
}

// ---------------------------------------------------------------------------------
// forward_3d2d, throughput path: the "dual" structure of the fused scorer (ahv_dual.h): 512 threads = two
// waves per SIMD, one item per wave, W1 as an LDS fragment table, one 8-KiB quarter image per wave filled
// straight from HBM (next quarter's loads are in flight while the current one is contracted).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void forward_3d2d_dual_kernel(
    const float* __restrict__ vol, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, long M, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float lds_w1[kW1TableFloats];
    __shared__ __attribute__((aligned(16))) float lds_q[8 * kQuarterFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* buf = lds_q + wave * kQuarterFloats;
    stage_w1_table(lds_w1, W1, tid, 512);
    DualFrags f;
    load_dual_frags(f, W2, b2, lane);
    __syncthreads();
    const int n16 = lane & 15, kq = lane >> 4;
    const int i2 = 2 * lane, sa0 = i2 >> 6, sb = (i2 >> 3) & 7, se = i2 & 7;
    const int o0 = qoff(sa0, sb, se), o1 = qoff(sa0, sb, se + 1);
    const long mstep = (long)gridDim.x * 8;
    long m = (long)wave * gridDim.x + blockIdx.x;
    f32x2 cur[16], nxt[16];
    if (m < M) {
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(vol + m * (16 * 512) + i2 + c * 512));
    }
    for (; m < M; m += mstep) {
        const float* V = vol + m * (16 * 512) + i2;
        // quarter 3's prefetch is the NEXT item's quarter 0 (the last item re-reads its own): no first-touch wait per item
        const float* Vn = vol + (m + mstep < M ? m + mstep : m) * (16 * 512) + i2;
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#define AHV_F3_QUARTER(Q)                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 16; ++c)                                                      \
            nxt[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>((Q < 3 ? V + (Q + 1) * 128 : Vn) + c * 512)); \
        _Pragma("unroll") for (int c = 0; c < 16; ++c) { buf[c * 128 + o0] = cur[c][0]; buf[c * 128 + o1] = cur[c][1]; } \
        wave_lds_fence();                                                                                   \
        gemm1_quarter_pipe<Q>(acc, lds_w1, buf, lane, [] {});                                               \
        wave_lds_fence();                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
        AHV_F3_QUARTER(0)
        AHV_F3_QUARTER(1)
        AHV_F3_QUARTER(2)
        AHV_F3_QUARTER(3)
#undef AHV_F3_QUARTER
        f32x4 v[2][4];
        gemm2_dual_exact(v, acc, f);  // the op-level drop-in materialises the reference's tensor: F.relu's NaN propagation too
        float* o = out + m * (32 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float ss = 0.0f;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += v[m2][t][r] * v[m2][t][r];
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            // v / max(|v|, eps) as v * (1 / max(|v|, eps)): ONE correctly rounded division per position instead of eight per
            // lane (an IEEE division is ~10 vector instructions, and fp32 MFMAs overlap none of them: 320 of an item's ~900);
            // each feature within 1 ulp of the quotient
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n16] = v[m2][t][r] * inv;
        }
    }
}

// ---------------------------------------------------------------------------------
// forward_3d2d for a handful of items (the per-pair target feature, test_co3d.py:141): latency matters,
// not throughput.  One 512-thread workgroup per item.  W1 is copied once, coalesced, into LDS with a
// 386-float row stride (the fragment reads W1[row][k0 + kq] of a half-wave then hit 32 different banks); wave
// (q, kh) contracts quarter q with the x slab + half of the z slab (kh = 0) or the y slab + the other half
// (kh = 1); the eight partial accumulators meet in LDS and waves 0-3 each finish one position tile
// (ReLU, GEMM2, bias, normalise).
// ---------------------------------------------------------------------------------
constexpr int kW1PadStride = 386;  // bank = (2 row + k) mod 32: 16 rows x 2 k-groups of a half-wave hit 32 banks

__global__ __launch_bounds__(512) void forward_3d2d_small_kernel(
    const float* __restrict__ vol, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float qbuf[4][kQuarterFloats];
    __shared__ __attribute__((aligned(16))) float part[8][8][64][4];
    __shared__ float w1s[32 * kW1PadStride];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave & 3, kh = wave >> 2;
    const int n = lane & 15, kq = lane >> 4, i0 = n >> 3, j = n & 7, row = lane & 15;
    const float* V = vol + (long)blockIdx.x * (16 * 512);
    for (int i = tid; i < 32 * 96; i += 512) {  // 96 float4 per W1 row
        const int r = i / 96, k4 = i - r * 96;
        const f32x4 w = *reinterpret_cast<const f32x4*>(W1 + r * 384 + 4 * k4);
        float* d = w1s + r * kW1PadStride + 4 * k4;
        d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    {   // quarter q of channel c = 128 contiguous floats at c*512 + q*128; the two waves of a quarter take 8 channels each
        float* buf = qbuf[q];
        const int i = 2 * lane, a0 = i >> 6, bb = (i >> 3) & 7, e = i & 7;
        const int o0 = qoff(a0, bb, e), o1 = qoff(a0, bb, e + 1);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int c = 8 * kh + cc;
            const float2 v = *reinterpret_cast<const float2*>(V + c * 512 + q * 128 + i);
            buf[c * 128 + o0] = v.x;
            buf[c * 128 + o1] = v.y;
        }
    }
    __syncthreads();
    const float* buf = qbuf[q];
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w0 = w1s + row * kW1PadStride;
    const float* w1 = w1s + (16 + row) * kW1PadStride;
    // the x / y slab lands in n-tile q; which tile that is must be a compile-time register index
#define AHV_XY(T)                                                                                          \
    _Pragma("unroll") for (int c = 0; c < 16; ++c) _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {        \
        const int k = 128 * kh + c * 8 + 4 * hh + kq;                                                      \
        const float bv = kh ? buf[c * 128 + qoff(i0, 4 * hh + kq, j)] : buf[c * 128 + qoff(i0, j, 4 * hh + kq)]; \
        acc[0][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[k], bv, acc[0][T], 0, 0, 0);                    \
        acc[1][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[k], bv, acc[1][T], 0, 0, 0);                    \
    }
    if (q == 0) { AHV_XY(0) } else if (q == 1) { AHV_XY(1) } else if (q == 2) { AHV_XY(2) } else { AHV_XY(3) }
#undef AHV_XY
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int cp = 4 * kh + cc;
        const int kz = 256 + (2 * cp + (kq >> 1)) * 8 + 2 * q + (kq & 1);
        const float a0 = w0[kz], a1 = w1[kz];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bz = buf[(2 * cp + (kq >> 1)) * 128 + qoff(kq & 1, 2 * t + i0, j)];
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bz, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bz, acc[1][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(part[wave][m * 4 + t][lane]) = acc[m][t];
    __syncthreads();
    if (wave >= 4) return;
    const int t = wave;  // this wave finishes positions 16t .. 16t+15
    f32x4 u[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        u[m] = *reinterpret_cast<const f32x4*>(part[0][m * 4 + t][lane]);
#pragma unroll
        for (int w = 1; w < 8; ++w) u[m] += *reinterpret_cast<const f32x4*>(part[w][m * 4 + t][lane]);
    }
    f32x4 v[2];
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m2][r] = b2[16 * m2 + 4 * kq + r];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a20 = W2[row * 32 + 16 * m + 4 * kq + r], a21 = W2[(16 + row) * 32 + 16 * m + 4 * kq + r];
            const float x = u[m][r] < 0.0f ? 0.0f : u[m][r];  // F.relu: a NaN stays a NaN (v_max would drop it)
            v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, x, v[0], 0, 0, 0);
            v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, x, v[1], 0, 0, 0);
        }
    float ss = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss += v[m2][r] * v[m2][r];
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float* o = out + (long)blockIdx.x * (32 * 64);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(16 * m2 + 4 * kq + r) * 64 + 16 * t + n] = v[m2][r] / nrm;
}

// ---------------------------------------------------------------------------------
// score (test_co3d.py:143): one wave per (b, n); 8 KiB of f_src streamed per hypothesis.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_features_kernel(const float* __restrict__ f_src,
                                                             const float* __restrict__ f_tgt, int B, long N,
                                                             float* __restrict__ scores)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    for (long i = wave; i < (long)B * N; i += nw) {
        const long b = i / N;
        const f32x4* s = reinterpret_cast<const f32x4*>(f_src + i * 2048);
        const f32x4* t = reinterpret_cast<const f32x4*>(f_tgt + b * 2048);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 a = __builtin_nontemporal_load(s + k * 64 + lane);
            const f32x4 c = t[k * 64 + lane];
            acc += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
        if (lane == 0) scores[i] = acc * (1.0f / 64.0f);
    }
}

// ---------------------------------------------------------------------------------
// arg-max over materialised scores (test_co3d.py:145), same packed key as the fused path.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ scores, int B, long N,
                                                     long n_offset, key_t* __restrict__ best_key)
{
    const int b = blockIdx.y;
    const float* s = scores + (long)b * N;
    key_t best = kKeyEmpty;
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
        const key_t k = pack_key(s[n], (unsigned)(n_offset + n));
        best = k > best ? k : best;
    }
    best = wave_max_key(best);
    if ((threadIdx.x & 63) == 0 && best != kKeyEmpty) atomicMax(best_key + b, best);
}

// ---------------------------------------------------------------------------------
// Coarse-to-fine support (BASELINE.json configs[4]; build-defined, the reference scores one flat
// set): refinement hypotheses R_fine[b][n] = R[idx_b] * D[n], where idx_b is decoded on the device from
// the packed key of the coarse stage and D is a fixed set of small rotations.  Graph-capturable.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compose_rotations_kernel(const key_t* __restrict__ best_key,
                                                                const float* __restrict__ R, long r_batch_stride,
                                                                long n_offset, long N, const float* __restrict__ D,
                                                                long N2, int B, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * N2) return;
    const int b = (int)(i / N2);
    const long n = i - (long)b * N2;
    const key_t key = best_key[b];
    long idx = key_index(key) - n_offset;
    idx = (key == kKeyEmpty || idx < 0 || idx >= N) ? 0 : idx;  // nothing scored / foreign shard: stay in bounds
    const float* r = R + (long)b * r_batch_stride + idx * 9;
    const float* d = D + n * 9;
    float* o = out + i * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[a * 3 + c] = r[a * 3] * d[c] + r[a * 3 + 1] * d[3 + c] + r[a * 3 + 2] * d[6 + c];
}

// unpack + gather in one launch: (best score, global index, R_pred = R[idx]) of test_co3d.py:145-146.
// reset: the key is handed back EMPTY, ready for the next verify step's atomic max (the step then needs no launch
// of its own to clear it: stream order puts this kernel between the two scorers).
__global__ void select_rotation_kernel(key_t* __restrict__ best_key, const float* __restrict__ R, long r_batch_stride,
                                       long n_offset, long N, int B, float* __restrict__ R_out,
                                       float* __restrict__ best_score, long* __restrict__ best_idx, bool reset)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const key_t k = best_key[b];
    if (reset) best_key[b] = kKeyEmpty;
    const long gidx = (k == kKeyEmpty) ? -1l : key_index(k);
    if (best_score) best_score[b] = (k == kKeyEmpty) ? -INFINITY : key_score(k);
    if (best_idx) best_idx[b] = gidx;
    if (R_out) {
        const long loc = gidx - n_offset;
        const bool mine = (k != kKeyEmpty) && loc >= 0 && loc < N;  // with sharding only the owner rank holds the row
        const float* r = R + (long)b * r_batch_stride + (mine ? loc : 0) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) R_out[b * 9 + e] = mine ? r[e] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------
// Haar-uniform rotation hypotheses generated on the device (replaces the host call
// pytorch3d.transforms.random_rotations(N), test_co3d.py:106 / modules/model.py:184; only the
// distribution matters -- hypotheses are inputs of the hot path).  Counter-based: rotation n
// depends on (seed, offset + n) alone, so any shard of the set can be generated anywhere,
// reproducibly, and the call is graph-capturable.  Philox-4x32-10 -> 4 uniforms -> Box-Muller
// -> normalised Gaussian quaternion -> matrix.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void random_rotations_kernel(unsigned long long seed, unsigned long long offset, long N,
                                                               float* __restrict__ out)
{
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const unsigned long long ctr = offset + (unsigned long long)n;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0x3D4148u, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    // uniforms in (0,1]: never log(0)
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u1, &s0, &c0);
    sincosf(6.28318530717958647692f * u3, &s1, &c1);
    const float qr = r0 * c0, qi = r0 * s0, qj = r1 * c1, qk = r1 * s1;
    const float t = 2.0f / fmaxf(qr * qr + qi * qi + qj * qj + qk * qk, 1e-30f);
    float* o = out + n * 9;
    o[0] = 1.0f - t * (qj * qj + qk * qk); o[1] = t * (qi * qj - qk * qr);        o[2] = t * (qi * qk + qj * qr);
    o[3] = t * (qi * qj + qk * qr);        o[4] = 1.0f - t * (qi * qi + qk * qk); o[5] = t * (qj * qk - qi * qr);
    o[6] = t * (qi * qk - qj * qr);        o[7] = t * (qj * qk + qi * qr);        o[8] = 1.0f - t * (qi * qi + qj * qj);
}

// Super-Fibonacci SO(3) grid (Alexa, CVPR 2022): point i of n is the quaternion
// (sqrt(t) sin a, sqrt(t) cos a, sqrt(1-t) sin b, sqrt(1-t) cos b), t = (i + 1/2)/n, a = 2 pi (i + 1/2)/sqrt(2),
// b = 2 pi (i + 1/2)/psi.  The angles reach 10^6 rad, so their turn fraction is taken in fp64 before sin/cos.
__global__ __launch_bounds__(256) void so3_grid_kernel(long n_total, long offset, long N, float* __restrict__ out)
{
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double s = (double)(offset + n) + 0.5;
    const double t = s / (double)n_total;
    double fa = s * 0.70710678118654752440, fb = s * (1.0 / 1.533751168755204288118041);
    fa -= floor(fa);
    fb -= floor(fb);
    float sa, ca, sb, cb;
    sincosf(6.28318530717958647692f * (float)fa, &sa, &ca);
    sincosf(6.28318530717958647692f * (float)fb, &sb, &cb);
    const float r = sqrtf((float)t), Rr = sqrtf((float)(1.0 - t));
    const float qr = r * sa, qi = r * ca, qj = Rr * sb, qk = Rr * cb;  // unit norm by construction
    float* o = out + n * 9;
    o[0] = 1.0f - 2.0f * (qj * qj + qk * qk); o[1] = 2.0f * (qi * qj - qk * qr);        o[2] = 2.0f * (qi * qk + qj * qr);
    o[3] = 2.0f * (qi * qj + qk * qr);        o[4] = 1.0f - 2.0f * (qi * qi + qk * qk); o[5] = 2.0f * (qj * qk - qi * qr);
    o[6] = 2.0f * (qi * qk - qj * qr);        o[7] = 2.0f * (qj * qk + qi * qr);        o[8] = 1.0f - 2.0f * (qi * qi + qj * qj);
}

// ---- launchers ----------------------------------------------------------------------
hipError_t launch_rotate_volume(const float* vol, int64_t vol_batch_stride, const float* R, int64_t N, int C,
                                int D, int H, int W, float* out, int num_cu, hipStream_t stream)
{
    if (C == 16 && D == 8 && H == 8 && W == 8 && vol_batch_stride == 0) {
        long blocks = (N + 3) / 4;
        const long cap = (long)num_cu * 3;  // 47.5 KiB of LDS per workgroup: three per CU are resident
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(rotate_volume_16x8_kernel, dim3((unsigned)blocks), dim3(kRotThreads), 0, stream, vol,
                           R, (long)N, out);
    } else {
        const long total = (long)N * D * H * W;
        long blocks = (total + 255) / 256;
        const long cap = (long)num_cu * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(rotate_volume_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, vol,
                           (long)vol_batch_stride, R, (long)N, C, D, H, W, out);
    }
    return hipGetLastError();
}

hipError_t launch_rotate_volume_backward(const float* grad_out, int64_t vol_batch_stride, const float* R, int64_t N,
                                         int C, int D, int H, int W, float* grad_vol, int num_cu, hipStream_t stream)
{
    const long total = (long)N * D * H * W;
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rotate_volume_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, grad_out,
                       (long)vol_batch_stride, R, (long)N, C, D, H, W, grad_vol);
    return hipGetLastError();
}

hipError_t launch_forward_3d2d(const float* vol, const float* W1, const float* W2, const float* b2, int64_t M,
                               float* out, int num_cu, hipStream_t stream)
{
    const size_t lds = sizeof(float) * kF32LdsFloats;
    static thread_local int attr_dev = -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(forward_3d2d_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    if (M <= 64) {  // latency path: one workgroup per item, quarters and slabs split over its 8 waves
        hipLaunchKernelGGL(forward_3d2d_small_kernel, dim3((unsigned)M), dim3(512), 0, stream, vol, W1, W2, b2, out);
        return hipGetLastError();
    }
    if (M >= 4096) {  // throughput path
        long blocks = (M + 7) / 8;
        if (blocks > num_cu) blocks = num_cu;
        hipLaunchKernelGGL(forward_3d2d_dual_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, vol, W1, W2, b2,
                           (long)M, out);
        return hipGetLastError();
    }
    long blocks = (M + 3) / 4;
    if (blocks > num_cu) blocks = num_cu;
    hipLaunchKernelGGL(forward_3d2d_kernel, dim3((unsigned)blocks), dim3(kF32Threads), lds, stream, vol, W1, W2,
                       b2, (long)M, out);
    return hipGetLastError();
}

hipError_t launch_score_features(const float* f_src, const float* f_tgt, int B, int64_t N, float* scores,
                                 int num_cu, hipStream_t stream)
{
    long blocks = ((long)B * N + 3) / 4;
    const long cap = (long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(score_features_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f_src, f_tgt, B,
                       (long)N, scores);
    return hipGetLastError();
}

hipError_t launch_compose_rotations(const int64_t* best_key, const float* R, int64_t r_batch_stride,
                                    int64_t n_offset, int64_t N, const float* D, int64_t N2, int B, float* out,
                                    hipStream_t stream)
{
    const long total = (long)B * N2;
    hipLaunchKernelGGL(compose_rotations_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const key_t*>(best_key), R, (long)r_batch_stride, (long)n_offset, (long)N, D,
                       (long)N2, B, out);
    return hipGetLastError();
}

hipError_t launch_select_rotation(int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset,
                                  int64_t N, int B, float* R_out, float* best_score, int64_t* best_idx, bool reset,
                                  hipStream_t stream)
{
    hipLaunchKernelGGL(select_rotation_kernel, dim3((B + 63) / 64), dim3(64), 0, stream,
                       reinterpret_cast<key_t*>(best_key), R, (long)r_batch_stride, (long)n_offset, (long)N, B, R_out,
                       best_score, reinterpret_cast<long*>(best_idx), reset);
    return hipGetLastError();
}

hipError_t launch_random_rotations(uint64_t seed, uint64_t offset, int64_t N, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(random_rotations_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream,
                       (unsigned long long)seed, (unsigned long long)offset, (long)N, out);
    return hipGetLastError();
}

hipError_t launch_so3_grid(int64_t n_total, int64_t offset, int64_t N, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(so3_grid_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, (long)n_total,
                       (long)offset, (long)N, out);
    return hipGetLastError();
}

hipError_t launch_argmax(const float* scores, int B, int64_t N, int64_t n_offset, int64_t* best_key,
                         int num_cu, hipStream_t stream)
{
    long bx = (N + 255) / 256;
    const long cap = num_cu * 4 / (B < num_cu ? B : num_cu) + 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, scores, B, (long)N,
                       (long)n_offset, reinterpret_cast<key_t*>(best_key));
    return hipGetLastError();
}

// best_key[0..B) = EMPTY (below every real key): one tiny launch, graph-capturable
__global__ void fill_keys_kernel(key_t* __restrict__ k, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) k[b] = kKeyEmpty;
}

hipError_t launch_fill_keys(int64_t* best_key, int B, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_keys_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, reinterpret_cast<key_t*>(best_key), B);
    return hipGetLastError();
}

// ---- zero fill ------------------------------------------------------------------------------------------
// The library's accumulation targets (gradients, best keys) are zeroed by THIS kernel, not by hipMemsetAsync: a run
// of memset nodes captured into a hipGraph (six in front of the scorer backward) did not reliably take effect on
// replay on ROCm 7.2 -- replay 0 was right because fresh pool memory is zero, every later replay accumulated onto
// whatever the pool block held (tests/test_gpu_graph_replay.py).  One launch for up to six spans.
struct ZeroSpans {
    void* p[6];
    unsigned long long bytes[6];  // multiples of 4
};

__global__ __launch_bounds__(256) void zero_fill_kernel(const ZeroSpans a)
{
    const unsigned long long i0 = (unsigned long long)blockIdx.x * 256 + threadIdx.x, step = (unsigned long long)gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned long long nb = a.bytes[k];
        if (nb == 0) continue;
        unsigned long long n16 = (reinterpret_cast<unsigned long long>(a.p[k]) & 15) ? 0 : nb >> 4;
        uint4* q = static_cast<uint4*>(a.p[k]);
        for (unsigned long long i = i0; i < n16; i += step) q[i] = uint4{0u, 0u, 0u, 0u};
        unsigned* d = static_cast<unsigned*>(a.p[k]) + 4 * n16;
        const unsigned long long nd = (nb >> 2) - 4 * n16;
        for (unsigned long long i = i0; i < nd; i += step) d[i] = 0u;
    }
}

hipError_t launch_zero_fill(void* const* ptrs, const size_t* bytes, int count, hipStream_t stream)
{
    if (count < 0 || count > 6) return hipErrorInvalidValue;
    ZeroSpans a;
    unsigned long long most = 0;
    for (int k = 0; k < 6; ++k) {
        a.p[k] = k < count ? ptrs[k] : nullptr;
        a.bytes[k] = (k < count && ptrs[k]) ? bytes[k] : 0;
        if (a.bytes[k] & 3) return hipErrorInvalidValue;
        if (a.bytes[k] > most) most = a.bytes[k];
    }
    if (most == 0) return hipSuccess;
    unsigned long long blocks = (most / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ahv
