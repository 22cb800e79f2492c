// ahv_encoder.hip -- HIP kernels for the transformer blocks of the reference's "3D-aware encoder"
// (transformer/attention.py: BasicTransformerBlock :240-258, CrossAttention :196-237, FeedForward/GEGLU
// :81-108, BidirectionTransformerBlock :260-274).  Once per image pair; 64 tokens x 256 channels per
// stream, so every GEMM is "skinny" (M = 64*B rows) and its cost is streaming the weights once.
//
//   linear_partial_kernel   P[ks][M][N] = X[M][K-slice ks] . W[N][K-slice]^T     fp32 MFMA 16x16x4, split-K
//   attention_kernel        softmax(Q K^T / 8) V for one (sample, head) per wave, scores kept transposed
//                           so that softmax runs down registers and P feeds the second MFMA in place
//   finish_*_kernel         sum of split-K slabs + bias, fused with LayerNorm / residual / GEGLU / concat
//
// fp32 throughout (the reference runs set_float32_matmul_precision("highest")).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ahv.h"

namespace ahv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// Skinny linear.  grid = (N / (16*NT), KS, M / 64); 256 threads.  The workgroup owns 64 rows x 16*NT
// columns and the K range [ks*Kc, (ks+1)*Kc); its 4 waves split that range (each streams its own part of
// the W rows exactly once from HBM), stage their X slice in LDS, and meet in LDS at the end.
// k ordering inside a 16-wide step: MFMA k-step s, lane group kq <-> column k0 + 4*kq + s, so that both
// operands are one aligned float4 per lane.
// -------------------------------------------------------------------------------------------------
constexpr int kLinMaxKw = 128;            // K columns per wave
constexpr int kLinLdx = kLinMaxKw + 4;    // +4 floats: ds_read_b128 of 16 rows hits 16 distinct slots

template <int NT>
__global__ __launch_bounds__(256) void linear_partial_kernel(const float* __restrict__ X, long ldx,
                                                             const float* __restrict__ W, long ldw,
                                                             float* __restrict__ P, int M, int N, int Kc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16 * NT;
    const int ks = blockIdx.y;
    const int m0 = blockIdx.z * 64;
    const int Kw = Kc >> 2;
    const int kbase = ks * Kc + wave * Kw;
    float* Xs = lds + wave * (64 * kLinLdx);
    // stage X[m0..m0+63][kbase..kbase+Kw) -> Xs[row][k]
    {
        const int q4 = Kw >> 2;  // float4 per row
        for (int i = lane; i < 64 * q4; i += 64) {
            const int r = i / q4, c = i - r * q4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(X + (long)(m0 + r) * ldx + kbase + 4 * c);
            *reinterpret_cast<f32x4*>(Xs + r * kLinLdx + 4 * c) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    asm volatile("" ::: "memory");
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 acc[4][NT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (long)(n0 + nt * 16 + r16) * ldw + kbase + 4 * kq;
    f32x4 bnext[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bnext[nt] = *reinterpret_cast<const f32x4*>(wrow[nt]);
    for (int k = 0; k < Kw; k += 16) {
        f32x4 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = bnext[nt];
        if (k + 16 < Kw) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bnext[nt] = *reinterpret_cast<const f32x4*>(wrow[nt] + k + 16);
        }
        f32x4 a[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(Xs + (rt * 16 + r16) * kLinLdx + k + 4 * kq);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][s], b[nt][s], acc[rt][nt], 0, 0, 0);
    }
    // cross-wave reduction through LDS (re-using the X staging area), then one coalesced store
    __syncthreads();
    float* red = lds;  // [4 waves][64 rows][16*NT]
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 64 + rt * 16 + 4 * kq + r) * (16 * NT) + nt * 16 + r16] = acc[rt][nt][r];
    __syncthreads();
    float* out = P + ((long)ks * M + m0) * N + n0;
    for (int i = tid; i < 64 * 16 * NT; i += 256) {
        const int r = i / (16 * NT), c = i - r * (16 * NT);
        const float v = red[i] + red[i + 64 * 16 * NT] + red[i + 2 * 64 * 16 * NT] + red[i + 3 * 64 * 16 * NT];
        out[(long)r * N + c] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// Attention for 64 queries x 64 keys x 64 head dims: one wave per (sample, head).
// S^T[j][i] = sum_d K[j][d] Q[i][d]   (keys on rows/registers, queries on lanes)
// softmax over j = down the registers (+ lane groups l^16, l^32);  O^T[d][i] = sum_j V[j][d] P^T[j][i],
// whose B operand is the S^T accumulator itself.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ Q, long ldq,
                                                        const float* __restrict__ K, long ldk,
                                                        const float* __restrict__ V, long ldv,
                                                        float* __restrict__ O, long ldo, int heads, float scale)
{
    const int lane = threadIdx.x & 63;
    const int h = threadIdx.x >> 6;  // blockDim = 64 * heads
    const int b = blockIdx.x;
    const int c16 = lane & 15, kq = lane >> 4;
    const float* q = Q + (long)b * 64 * ldq + h * 64;
    const float* k = K + (long)b * 64 * ldk + h * 64;
    const float* v = V + (long)b * 64 * ldv + h * 64;
    f32x4 st[4][4];  // [jt][it]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) st[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d0 = 0; d0 < 64; d0 += 16) {
        f32x4 ka[4], qb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ka[t] = *reinterpret_cast<const f32x4*>(k + (long)(t * 16 + c16) * ldk + d0 + 4 * kq);
            qb[t] = *reinterpret_cast<const f32x4*>(q + (long)(t * 16 + c16) * ldq + d0 + 4 * kq);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[jt][s], qb[it][s], st[jt][it], 0, 0, 0);
    }
    // softmax over keys (rows) for each query column i = it*16 + c16
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[jt][it][r] *= scale;
                m = fmaxf(m, st[jt][it][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.0f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = expf(st[jt][it][r] - m);
                st[jt][it][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[jt][it][r] *= inv;
    }
    // O^T[d][i]: k-step (jt, r) has k-values j = jt*16 + 4*kq + r
    f32x4 ot[4][4];  // [dt][it]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it) ot[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float va[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) va[dt] = v[(long)(jt * 16 + 4 * kq + r) * ldv + dt * 16 + c16];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    ot[dt][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[dt], st[jt][it][r], ot[dt][it], 0, 0, 0);
        }
    float* o = O + (long)b * 64 * ldo + h * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<f32x4*>(o + (long)(it * 16 + c16) * ldo + dt * 16 + 4 * kq) = ot[dt][it];
}

// -------------------------------------------------------------------------------------------------
// Finish kernels: one wave per row.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// y = LayerNorm(sum_ks P[ks][row][:] + bias) * g + be   (256 columns, eps 1e-5, biased variance);
// RESIDUAL: out[row][0:256] = res[row] + y        (BasicTransformerBlock :258)
// else    : out[row][col_off : col_off+256] = y    (the "message" half of cat([x, m]), :256)
template <bool RESIDUAL>
__global__ __launch_bounds__(256) void finish_ln_kernel(const float* __restrict__ P, int KS, int M,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ g, const float* __restrict__ be,
                                                        const float* __restrict__ res, long ldres,
                                                        float* __restrict__ out, long ldo, int col_off)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 x = *reinterpret_cast<const f32x4*>(bias + 4 * lane);
    for (int ks = 0; ks < KS; ++ks) x += *reinterpret_cast<const f32x4*>(P + ((long)ks * M + row) * 256 + 4 * lane);
    const float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * (1.0f / 256.0f);
    const f32x4 d = x - mean;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * lane);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(be + 4 * lane);
    f32x4 y = d * rstd * gg + bb;
    if (RESIDUAL) y += *reinterpret_cast<const f32x4*>(res + (long)row * ldres + 4 * lane);
    *reinterpret_cast<f32x4*>(out + (long)row * ldo + col_off + 4 * lane) = y;
}

// GEGLU (transformer/attention.py:81-88): h = sum_ks P[ks][row][0:2H] + bias; out = h[:H] * gelu(h[H:]), exact erf.
__global__ __launch_bounds__(256) void finish_geglu_kernel(const float* __restrict__ P, int KS, int M, int H,
                                                           const float* __restrict__ bias, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of the output
    const long total = (long)M * (H / 4);
    if (i >= total) return;
    const long row = i / (H / 4);
    const int c = (int)(i - row * (H / 4)) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 gt = *reinterpret_cast<const f32x4*>(bias + H + c);
    for (int ks = 0; ks < KS; ++ks) {
        const float* p = P + ((long)ks * M + row) * (2 * H);
        a += *reinterpret_cast<const f32x4*>(p + c);
        gt += *reinterpret_cast<const f32x4*>(p + H + c);
    }
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = a[e] * (0.5f * gt[e] * (1.0f + erff(gt[e] * 0.70710678118654752f)));
    *reinterpret_cast<f32x4*>(out + row * H + c) = y;
}

// copy x into the first half of the concat buffer: cat[row][0:256] = x[row]
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out,
                                                        long ldo, int M)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    *reinterpret_cast<f32x4*>(out + (long)row * ldo + 4 * lane) = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + 4 * lane);
}

// ---- host side -------------------------------------------------------------------------------------
static hipError_t launch_linear(const float* X, long ldx, const float* W, long ldw, float* P, int M, int N, int K,
                                int KS, hipStream_t s)
{
    const int Kc = K / KS;
    const size_t lds = sizeof(float) * 4 * 64 * kLinLdx;
    static thread_local int attr_dev = -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_partial_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_partial_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    if (N % 32 == 0 && N >= 1024)
        hipLaunchKernelGGL(linear_partial_kernel<2>, dim3(N / 32, KS, M / 64), dim3(256), lds, s, X, ldx, W, ldw, P, M, N, Kc);
    else
        hipLaunchKernelGGL(linear_partial_kernel<1>, dim3(N / 16, KS, M / 64), dim3(256), lds, s, X, ldx, W, ldw, P, M, N, Kc);
    return hipGetLastError();
}

// split-K factor: keep the per-wave K slice <= 128 and a multiple of 16, and get >= ~64 workgroups
static int pick_ks(int N, int K)
{
    int ks = 1;
    while (K / ks / 4 > kLinMaxKw) ks *= 2;
    while ((N / 16) * ks < 64 && K / (ks * 2) / 4 >= 16 && (K / (ks * 2)) % 64 == 0) ks *= 2;
    return ks;
}

size_t transformer_workspace_floats(int B)
{
    const size_t M = (size_t)64 * (size_t)(B > 0 ? B : 0);
    // qkv 768 + kv 768 + attn 256 + cat 512 + gate 2048 + split-K slabs (max N*KS over the linears = 4096)
    // + two ping-pong token buffers
    return M * (768 + 768 + 256 + 512 + 2048 + 4096 + 2 * 256) + 64;
}

static int run_block(const ahv_block_weights* w, const float* x, const float* ctx, int self_attn, float* out, int M,
                     float* ws, hipStream_t s, const char** what)
{
    const int B = M / 64;
    float* qkv = ws;                       // [M][768]  (self: q|k|v of x;  cross: q of x in cols 0..255)
    float* kv = qkv + (size_t)M * 768;     // [M][768]  (cross: k|v of ctx in cols 256..767)
    float* att = kv + (size_t)M * 768;     // [M][256]
    float* cat = att + (size_t)M * 256;    // [M][512]
    float* gate = cat + (size_t)M * 512;   // [M][2048]
    float* part = gate + (size_t)M * 2048; // split-K slabs
    hipError_t e;
#define AHV_TRY(call, name) do { e = (call); if (e != hipSuccess) { *what = name; return (int)e; } } while (0)
    const float *Q, *Kp, *Vp;
    long ldq, ldkv;
    if (self_attn) {
        AHV_TRY(launch_linear(x, 256, w->w_qkv, 256, qkv, M, 768, 256, 1, s), "qkv projection");
        Q = qkv; Kp = qkv + 256; Vp = qkv + 512; ldq = 768; ldkv = 768;
    } else {
        AHV_TRY(launch_linear(x, 256, w->w_qkv, 256, qkv, M, 256, 256, 1, s), "q projection");
        AHV_TRY(launch_linear(ctx, 256, w->w_qkv + 256 * 256, 256, kv, M, 512, 256, 1, s), "kv projection");
        Q = qkv; Kp = kv; Vp = kv + 256; ldq = 256; ldkv = 512;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(B), dim3(256), 0, s, Q, ldq, Kp, ldkv, Vp, ldkv, att, 256L, 4, 0.125f);
    AHV_TRY(hipGetLastError(), "attention");
    int ks = pick_ks(256, 256);
    AHV_TRY(launch_linear(att, 256, w->w_out, 256, part, M, 256, 256, ks, s), "out projection");
    hipLaunchKernelGGL(copy_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, 256L, cat, 512L, M);
    hipLaunchKernelGGL(finish_ln_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, part, ks, M, w->b_out, w->ln1_g,
                       w->ln1_b, (const float*)nullptr, 0L, cat, 512L, 256);
    AHV_TRY(hipGetLastError(), "norm1");
    AHV_TRY(launch_linear(cat, 512, w->w_ff1, 512, part, M, 4096, 512, 1, s), "ff in");
    hipLaunchKernelGGL(finish_geglu_kernel, dim3((unsigned)(((long)M * 512 + 255) / 256)), dim3(256), 0, s, part, 1, M,
                       2048, w->b_ff1, gate);
    AHV_TRY(hipGetLastError(), "geglu");
    ks = pick_ks(256, 2048);
    AHV_TRY(launch_linear(gate, 2048, w->w_ff2, 2048, part, M, 256, 2048, ks, s), "ff out");
    hipLaunchKernelGGL(finish_ln_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, part, ks, M, w->b_ff2, w->ln2_g,
                       w->ln2_b, x, 256L, out, 256L, 0);
    AHV_TRY(hipGetLastError(), "norm2");
#undef AHV_TRY
    return 0;
}

int transformer_blocks(const ahv_block_weights* blocks, int depth, float* x_src, float* x_tgt, int B, float* ws,
                       hipStream_t s, const char** what)
{
    const int M = 64 * B;
    // the two ping-pong token buffers live at the end of the workspace
    float* tmp_src = ws + (size_t)M * (768 + 768 + 256 + 512 + 2048 + 4096);
    float* tmp_tgt = tmp_src + (size_t)M * 256;
    for (int d = 0; d < depth; ++d) {
        const ahv_block_weights* w = blocks + 4 * d;  // order: attn_self_1, attn_self_2, attn_cross_1, attn_cross_2
        int rc;
        // x = self_1(x); ctx = self_2(ctx)
        if ((rc = run_block(w + 0, x_src, x_src, 1, tmp_src, M, ws, s, what))) return rc;
        if ((rc = run_block(w + 1, x_tgt, x_tgt, 1, tmp_tgt, M, ws, s, what))) return rc;
        // x_out = cross_1(x, ctx); ctx_out = cross_2(ctx, x)  -- both read the post-self tensors
        if ((rc = run_block(w + 2, tmp_src, tmp_tgt, 0, x_src, M, ws, s, what))) return rc;
        if ((rc = run_block(w + 3, tmp_tgt, tmp_src, 0, x_tgt, M, ws, s, what))) return rc;
    }
    return 0;
}

}  // namespace ahv
