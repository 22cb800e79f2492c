// ahv_encoder.hip -- HIP kernels for the transformer blocks of the reference's "3D-aware encoder"
// (transformer/attention.py: BasicTransformerBlock :240-258, CrossAttention :196-237, FeedForward/GEGLU
// :81-108, BidirectionTransformerBlock :260-274).  Once per image pair; 64 tokens x 256 channels per
// stream, so every GEMM is "skinny" (M = 64*B rows): its cost is streaming the weights once and the
// launch count.  The two streams of a layer (self_1 | self_2, then cross_1 | cross_2) are independent, so
// every launch carries BOTH blocks (and, for cross-attention, both the q and the k|v projections) as
// separate "problems": 6 launches per half layer, 48 for the whole depth-4 encoder.
//
//   linear_staged_kernel  P[ks][M][N] = X[M][K-slice ks] . W[N][K-slice]^T   fp32 MFMA 16x16x4, split-K, up to 4
//                         problems per launch, K walked in stages through LDS; optional bias + GEGLU epilogue
//   linear_kernel         the register-resident form of the same (3-D convolution taps, odd K slices)
//   linear_tile_kernel    128 x 128 tiles for M >= 1024
//   attention_heads_kernel  softmax(Q K^T / 8) V and the output projection, four waves per (sample, head, 16 queries)
//   ln_kernel             LayerNorm(sum_ks P + bias) fused with cat([x, .]) or x + .   (attention.py:255-258)
//
// fp32 throughout (the reference runs set_float32_matmul_precision("highest")).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ahv.h"

// Every launch in this file goes through AHV_ENC_LAUNCH so that tools/kbench_enc.cpp (built with -DAHV_ENC_PROBE) can
// cut a forward after k launches -- the marginal cost of every launch with no profiler attached -- and collect an
// in-kernel timeline of the linear kernels (8 realtime stamps, 100 MHz, per workgroup).
#ifdef AHV_ENC_PROBE
#include <string.h>
static int g_enc_probe_budget = 1 << 30;
static int g_enc_probe_count = 0;
static const char* g_enc_probe_names[256];
static unsigned long long* g_enc_probe_stamps = nullptr;
// g_enc_probe_dup: every launch that reads weights (the linears and the attention + output projection, all pure functions
// of their inputs) is issued TWICE: the second one finds its weights in the L2 of the XCDs that need them -- its marginal
// cost is what a perfect weight prefetch could make of the first (tools/kbench_enc.cpp --each --dup)
static int g_enc_probe_dup = 0;
#define AHV_ENC_LAUNCH(k, ...) do { \
    const int reps_ = (g_enc_probe_dup && (strstr(#k, "linear") || strstr(#k, "attention"))) ? 2 : 1; \
    for (int r_ = 0; r_ < reps_; ++r_) \
        if (g_enc_probe_budget-- > 0) { g_enc_probe_names[g_enc_probe_count++ & 255] = r_ ? #k " [again: weights hot]" : #k; hipLaunchKernelGGL(k, __VA_ARGS__); } \
    } while (0)
#define AHV_ENC_STAMP_PTR (g_enc_probe_stamps ? g_enc_probe_stamps + (size_t)(g_enc_probe_count & 255) * 8192 : nullptr)
#define AHV_ENC_STAMP(slot) do { if (a.stamps && threadIdx.x == 0) { \
    a.stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define AHV_ENC_LAUNCH(...) hipLaunchKernelGGL(__VA_ARGS__)
#define AHV_ENC_STAMP(slot) do { } while (0)
#endif

namespace ahv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Output stores of the big launches (experiment knobs, A/B on one box with tools/kbench_enc): non-temporal stores leave less
// dirty data in the L2s for the write-back at the end of a kernel, which sits between every two launches of this forward.
#ifndef AHV_TILE_NT_STORES
#define AHV_TILE_NT_STORES 1
#endif
#ifndef AHV_ROW_NT_STORES
#define AHV_ROW_NT_STORES 0
#endif
#if AHV_TILE_NT_STORES
#define AHV_TILE_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define AHV_TILE_STORE(p, v) (*(p) = (v))
#endif
#if AHV_ROW_NT_STORES
#define AHV_ROW_STORE4(p, v) __builtin_nontemporal_store((v), reinterpret_cast<f32x4*>(p))
#else
#define AHV_ROW_STORE4(p, v) (*reinterpret_cast<f32x4*>(p) = (v))
#endif


constexpr int kMaxProb = 4;

struct LinProb {
    const float* X;     // [M][ldx]  (GEGLU mode: the [M][2H] slab of the previous linear)
    const float* W;     // [N][ldw]
    float* P;           // [KS][M][N]
    const float* bias;  // GEGLU_OUT: bias of this projection [2H]
    int N;
    int tile0;          // first n-tile (of 16*NT columns) of this problem in blockIdx.x
};

struct LinArgs {
    LinProb p[kMaxProb];
    int nprob, M, Kc;
    long ldx, ldw;
    int geglu_h;  // H > 0: GEGLU epilogue, W has 2H rows (value rows, then gate rows), output [M][H]
#ifdef AHV_ENC_PROBE
    unsigned long long* stamps;
#endif
};

// GELU(g) = g Phi(g), exact form (nn.GELU() default = 0.5 g (1 + erf(g / sqrt 2)), attention.py:81-88), for a PAIR of values
// on the packed fp32 pipe.  A 128 x 128 GEGLU tile ends in 64 of them per lane, and fp32 MFMAs do not overlap VALU work on
// gfx950: libdevice's erff (a branch between two polynomials, 35 vector instructions per value) cost the FF-in launch as
// much as 300 MFMAs per wave and tile, both ranges of a < 1-ulp erf on v_pk_fma_f32 (round 6, first form) 18 per value.
// This form needs 8:  GELU(g) = max(g, 0) - |g| Phi(-|g|)  with  Phi(-t) = exp2(p(t)),  p(t) = -1 + c1 t + ... + c8 t^8
// -- log2 of the normal tail is smooth on t >= 0, so ONE polynomial serves every t, c8 < 0 and p falls monotonically
// beyond the fitted [0, 6], so large |g| need no clamp (exp2 -> 0), and nothing cancels: for g > 0 the tail is subtracted
// from g, for g < 0 it IS the result (0.5 g (1 + erf) loses digits there).  tools/fit_gelu.py regenerates the
// coefficients and the error with fp32 rounding after every operation: |gelu_pk - fp64| <= 2.6e-7 on [-300, 300], where
// the expression torch evaluates in fp32 is off by up to 4.5e-7; relative to max(|GELU|, 1e-3) 1.7e-5 against 8.3e-5.
// (g = +inf gives NaN here -- inf * 0 -- and inf there; LayerNorm turns either into NaN one launch later.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define AHV_PK(c) (f32x2{c, c})
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f32x2 gelu_pk(f32x2 g)
{
    const f32x2 t = __builtin_elementwise_abs(g);
    f32x2 r = pk_fma(AHV_PK(-2.357509857e-06f), t, AHV_PK(3.389861013e-05f));
    r = pk_fma(r, t, AHV_PK(-1.614213979e-04f));
    r = pk_fma(r, t, AHV_PK(-1.932584128e-04f));
    r = pk_fma(r, t, AHV_PK(7.131781429e-03f));
    r = pk_fma(r, t, AHV_PK(-5.253926665e-02f));
    r = pk_fma(r, t, AHV_PK(-4.591956735e-01f));
    r = pk_fma(r, t, AHV_PK(-1.151106358e+00f));
    r = pk_fma(r, t, AHV_PK(-1.0f));
    const f32x2 tail = {__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};   // Phi(-|g|)
    const f32x2 pos = {fmaxf(g[0], 0.0f), fmaxf(g[1], 0.0f)};
    return pk_fma(-t, tail, pos);
}

// (v0, v1) * GELU(g0, g1)
__device__ __forceinline__ f32x2 geglu_pk(f32x2 v, f32x2 g) { return v * gelu_pk(g); }

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// -------------------------------------------------------------------------------------------------
// Skinny linear.  grid = (sum of n-tiles, KS, M / 64); 512 threads.  The workgroup owns 64 rows x 16*NT
// columns and the K range [ks*Kc, (ks+1)*Kc); its 8 waves split that range (KW columns each, every W row
// segment is streamed exactly once).  Both MFMA operands go straight from memory to registers: with the
// k ordering "MFMA k-step s, lane group kq <-> column k0 + 4*kq + s" a lane needs exactly one aligned
// float4 of its X row and one of its W row per 16-wide step, and nobody else needs them.  ALL loads of a
// wave are issued before its first MFMA (sched_barrier keeps hipcc from sinking them to their uses), so
// the memory latency is paid once.  The 8 partial tiles meet in LDS.
// -------------------------------------------------------------------------------------------------
// GEGLU_OUT (NT = 2, KS = 1): the two n-tiles of the workgroup are the VALUE columns [16t, 16t+16) and the GATE
// columns [H+16t, H+16t+16) of the GEGLU projection, and the epilogue writes
// out[r][16t+c] = (value + b[16t+c]) * gelu(gate + b[H+16t+c]) straight into a [M][H] buffer (attention.py:81-88).
// MODE 1 / 2: implicit GEMM of a 3x3 convolution over 8x8 tokens (C = 256 channels) / a 3x3x3 convolution over
// the 8^3 voxels of a channel-last volume (C = KW channels): the K axis is (tap, channel) with one tap (or a
// part of one) per wave, so the row a lane reads is its own row shifted by the tap, or zero outside the image
// (zero padding).  Taps >= 27 (3-D) are K padding: their weights are zero and nothing is loaded.
template <int NT, int KW, bool GEGLU_OUT, int MODE = 0>
__global__ __launch_bounds__(512) void linear_kernel(const LinArgs a)
{
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 16 * NT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    AHV_ENC_STAMP(0);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxProb; ++i)
        if (i < a.nprob && (int)blockIdx.x >= a.p[i].tile0) pi = i;
    const LinProb pr = a.p[pi];
    const int tile = (int)blockIdx.x - pr.tile0;
    const int n0 = GEGLU_OUT ? tile * 16 : tile * 16 * NT;
    const int nstep = GEGLU_OUT ? a.geglu_h : 16;  // distance between the workgroup's n-tiles
    const int ks = blockIdx.y;
    const int m0 = blockIdx.z * 64;
    const int kbase = ks * a.Kc + wave * KW;
    const int r16 = lane & 15, kq = lane >> 4;
    constexpr int STEPS = KW / 16;

    f32x4 w[STEPS][NT], x[STEPS][4];
    const float* xsrc[4];
    bool xok[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int p = rt * 16 + r16;  // row of this lane inside the 64-row tile
        if (MODE == 0) {
            xsrc[rt] = pr.X + (long)(m0 + p) * a.ldx + kbase + 4 * kq;
            xok[rt] = true;
        } else if (MODE == 1) {
            const int tap = kbase >> 8, ci0 = kbase & 255;
            const int ny = (p >> 3) + tap / 3 - 1, nx = (p & 7) + tap % 3 - 1;
            xok[rt] = (unsigned)ny < 8u && (unsigned)nx < 8u;
            xsrc[rt] = pr.X + (long)(m0 + (xok[rt] ? ny * 8 + nx : p)) * 256 + ci0 + 4 * kq;
        } else {
            const int tap = kbase / KW;
            const int v = (m0 & 511) + p;  // voxel inside the sample
            const int nd = (v >> 6) + tap / 9 - 1, nh = ((v >> 3) & 7) + (tap / 3) % 3 - 1, nw = (v & 7) + tap % 3 - 1;
            xok[rt] = tap < 27 && (unsigned)nd < 8u && (unsigned)nh < 8u && (unsigned)nw < 8u;
            xsrc[rt] = pr.X + (long)((m0 - (m0 & 511)) + (xok[rt] ? nd * 64 + nh * 8 + nw : v)) * KW + 4 * kq;
        }
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            w[st][nt] = *reinterpret_cast<const f32x4*>(pr.W + (long)(n0 + nt * nstep + r16) * a.ldw + kbase + 16 * st + 4 * kq);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            x[st][rt] = *reinterpret_cast<const f32x4*>(xsrc[rt] + 16 * st);
            if (MODE != 0 && !xok[rt]) x[st][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4][NT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[st][rt][s4], w[st][nt][s4], acc[rt][nt], 0, 0, 0);
    AHV_ENC_STAMP(4);
    // cross-wave reduction through LDS, then one coalesced store
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 64 + rt * 16 + 4 * kq + r) * (16 * NT) + nt * 16 + r16] = acc[rt][nt][r];
    __syncthreads();
    AHV_ENC_STAMP(5);
    if (GEGLU_OUT) {
        float* out = pr.P + (long)m0 * a.geglu_h + n0;
        for (int i = tid; i < 64 * 16; i += 512) {
            const int r = i >> 4, c = i & 15;
            float v = pr.bias[n0 + c], gt = pr.bias[a.geglu_h + n0 + c];
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) {
                v += red[(wv * 64 + r) * 32 + c];
                gt += red[(wv * 64 + r) * 32 + 16 + c];
            }
            out[(long)r * a.geglu_h + c] = geglu_pk(f32x2{v, 0.0f}, f32x2{gt, 0.0f})[0];
        }
    } else {
        float* out = pr.P + ((long)ks * a.M + m0) * pr.N + n0;
        for (int i = tid; i < 64 * 16 * NT; i += 512) {
            const int r = i / (16 * NT), c = i - r * (16 * NT);
            float v = 0.0f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += red[i + wv * 64 * 16 * NT];
            out[(long)r * pr.N + c] = v;
        }
    }
    AHV_ENC_STAMP(6);
}

// -------------------------------------------------------------------------------------------------
// The same linear, K-PIPELINED through LDS (MODE 0 / 1; grid, LinArgs and 512 threads as linear_kernel).
// linear_kernel requests its whole operand set at once and as MFMA fragments: the four lanes of a quad then sit
// in four different 128-byte lines and the texture path delivers ~16 B/clk (tools/launch_floor.cpp: 128 KB re-read
// per workgroup 3.3 us as fragments, 1.15 us row-contiguous), so a workgroup of the FF-in projection spends ~5 us
// receiving 192 KB before its last wave can start its MFMAs, and ends with an 8-way reduction through LDS.
// Here the workgroup walks K in stages of 64 columns: X[64][64] and W[16 NT][64] are fetched row-contiguously
// (16 lanes = one 256-byte row segment), parked in a double-buffered LDS stage (row stride 68 floats: a fragment
// read of 16 rows x one float4 covers all 64 banks once) while the MFMAs of the previous stage run, and wave w
// owns output rows 16 w .. 16 w + 15 over ALL of the workgroup's columns and ALL of K -- no cross-wave reduction,
// and the value and gate tiles of the GEGLU projection meet in one lane's registers.
// -------------------------------------------------------------------------------------------------
constexpr int kStageK = 64;    // K columns per stage
constexpr int kStageLd = 68;   // floats per staged row

template <int NT, int KC, bool GEGLU_OUT, int MODE = 0>
__global__ __launch_bounds__(512) void linear_staged_kernel(const LinArgs a)
{
    static_assert(MODE == 0 || MODE == 1, "MODE 2 (3-D taps) stays with linear_kernel");
    static_assert(NT == 1 || NT == 2, "one or two n-tiles per workgroup");
    constexpr int NS = KC / kStageK;  // stages
    constexpr int DEPTH = 2 < NS ? 2 : NS;  // stages requested ahead of their use (3, 4, 8 measured: 319.7 / 323.9 / 323.9 us
                                            // per forward against 318.5 at 2, round 3)
    // two stage buffers (26 KB each at NT = 2): 2 workgroups per CU once M > 64.  (Three buffers with the next stage's
    // fragments read between this stage's MFMAs: equal at B = 1 -- 309.2 against 309.4 us per forward --, 4-10 % slower at
    // B = 2..8, measured again in round 5 on the interleaved issue order below.)
    __shared__ __attribute__((aligned(16))) float sx[2][64 * kStageLd];
    __shared__ __attribute__((aligned(16))) float sw[2][16 * NT * kStageLd];
    __shared__ __attribute__((aligned(16))) float comb[4 * 64 * 4];  // the second wave of each pair hands its tile over
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Two waves per SIMD (a lone wave issues an fp32 MFMA only every ~40 cycles): wave = (row tile rt, half h).
    // NT = 2 (GEGLU): h = n-tile (0 value, 1 gate).  NT = 1: h = half of every stage's K (k-steps 2h, 2h + 1).
    const int rt = wave & 3, half = wave >> 2;
    AHV_ENC_STAMP(0);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxProb; ++i)
        if (i < a.nprob && (int)blockIdx.x >= a.p[i].tile0) pi = i;
    const LinProb pr = a.p[pi];
    const int tile = (int)blockIdx.x - pr.tile0;
    const int n0 = GEGLU_OUT ? tile * 16 : tile * 16 * NT;
    const int nstep = GEGLU_OUT ? a.geglu_h : 16;  // distance between the workgroup's n-tiles
    const int ks = blockIdx.y;
    const int m0 = blockIdx.z * 64;
    const int r16 = lane & 15, kq = lane >> 4;

    // staging map: thread -> (row tid / 16 (+32), float4 column tid % 16); W: 16 NT rows, the first 256 NT threads
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    const float* xg[2];
    bool xok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = srow + 32 * j;  // row inside the 64-row tile
        if (MODE == 0) {
            xg[j] = pr.X + (long)(m0 + p) * a.ldx + (long)ks * KC + sc4;
            xok[j] = true;
        } else {  // 3x3 convolution over 8x8 tokens: this K slice is tap ks, all 256 channels
            const int ny = (p >> 3) + ks / 3 - 1, nx = (p & 7) + ks % 3 - 1;
            xok[j] = (unsigned)ny < 8u && (unsigned)nx < 8u;
            xg[j] = pr.X + (long)(m0 + (xok[j] ? ny * 8 + nx : p)) * 256 + sc4;
        }
    }
    const bool wload = srow < 16 * NT;
    const int wrow = wload ? srow : 0;
    const float* wg = pr.W + (long)(n0 + (wrow >> 4) * nstep + (wrow & 15)) * a.ldw + (long)ks * KC + sc4;

    f32x4 gx[NS][2], gw[NS];
    auto request = [&](int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            gx[st][j] = xok[j] ? *reinterpret_cast<const f32x4*>(xg[j] + kStageK * st) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (wload) gw[st] = *reinterpret_cast<const f32x4*>(wg + kStageK * st);
    };
    auto park = [&](int st, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&sx[buf][(srow + 32 * j) * kStageLd + sc4]) = gx[st][j];
        if (wload) *reinterpret_cast<f32x4*>(&sw[buf][srow * kStageLd + sc4]) = gw[st];
    };
#pragma unroll
    for (int st = 0; st < DEPTH; ++st) request(st);
    float gbias_v = 0.0f, gbias_g = 0.0f;  // GEGLU epilogue: this lane's column
    if (GEGLU_OUT) {
        gbias_v = pr.bias[n0 + r16];
        gbias_g = pr.bias[a.geglu_h + n0 + r16];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int KSTEPS = NT == 2 ? 4 : 2;        // k-steps of 16 per stage and wave
    const int k40 = NT == 2 ? 0 : 2 * half;        // first of them
    const int wtile = NT == 2 ? half : 0;          // n-tile of this wave
    park(0, 0);
    __syncthreads();
    AHV_ENC_STAMP(3);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int cur = st & 1;
        // issue order: this stage's fragment reads first, then its (dependent, ~40 cycles apart) MFMAs with the requests of
        // stage st + DEPTH and the LDS stores of stage st + 1 in the gaps between them (hard fences: sched_group_barrier
        // does not move instructions the MFMAs' own operands depend on)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 fx[KSTEPS], fw[KSTEPS];
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            fx[k] = *reinterpret_cast<const f32x4*>(&sx[cur][(16 * rt + r16) * kStageLd + 16 * (k40 + k) + 4 * kq]);
            fw[k] = *reinterpret_cast<const f32x4*>(&sw[cur][(16 * wtile + r16) * kStageLd + 16 * (k40 + k) + 4 * kq]);
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = 4 * KSTEPS;  // MFMAs of this stage
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[q >> 2][q & 3], fw[q >> 2][q & 3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int slot = q;  // one memory instruction behind each of the first six MFMAs
            if (st + DEPTH < NS) {
                const int sr = st + DEPTH;
                if (slot == 0) gx[sr][0] = xok[0] ? *reinterpret_cast<const f32x4*>(xg[0] + kStageK * sr) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (slot == 1) gx[sr][1] = xok[1] ? *reinterpret_cast<const f32x4*>(xg[1] + kStageK * sr) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (slot == 2 && wload) gw[sr] = *reinterpret_cast<const f32x4*>(wg + kStageK * sr);
            }
            if (st + 1 < NS) {
                const int sp = st + 1, buf = cur ^ 1;
                if (slot == 3) *reinterpret_cast<f32x4*>(&sx[buf][srow * kStageLd + sc4]) = gx[sp][0];
                if (slot == 4) *reinterpret_cast<f32x4*>(&sx[buf][(srow + 32) * kStageLd + sc4]) = gx[sp][1];
                if (slot == 5 && wload) *reinterpret_cast<f32x4*>(&sw[buf][srow * kStageLd + sc4]) = gw[sp];
            }
            if (slot < 6) __builtin_amdgcn_sched_barrier(0);
        }
        if (st + 1 < NS) __syncthreads();
    }
    AHV_ENC_STAMP(4);
    // D layout: acc[r] = C[m0 + 16 rt + 4 kq + r][column r16 of this wave's n-tile]; waves 4..7 hand over
    if (half == 1) *reinterpret_cast<f32x4*>(&comb[(rt * 64 + lane) * 4]) = acc;
    __syncthreads();
    if (half == 0) {
        const f32x4 other = *reinterpret_cast<const f32x4*>(&comb[(rt * 64 + lane) * 4]);
        if (GEGLU_OUT) {
            float* out = pr.P + (long)(m0 + 16 * rt + 4 * kq) * a.geglu_h + n0 + r16;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 y = geglu_pk(f32x2{acc[r] + gbias_v, acc[r + 1] + gbias_v}, f32x2{other[r] + gbias_g, other[r + 1] + gbias_g});
                out[(long)r * a.geglu_h] = y[0];
                out[(long)(r + 1) * a.geglu_h] = y[1];
            }
        } else if (NT == 1) {
            float* out = pr.P + ((long)ks * a.M + m0 + 16 * rt + 4 * kq) * pr.N + n0 + r16;
            const float bias = pr.bias ? pr.bias[n0 + r16] : 0.0f;  // a 1x1 conv's bias (single K slice only: host)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)r * pr.N] = acc[r] + other[r] + bias;
        } else {  // two plain n-tiles
            float* out = pr.P + ((long)ks * a.M + m0 + 16 * rt + 4 * kq) * pr.N + n0 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[(long)r * pr.N] = acc[r];
                out[(long)r * pr.N + 16] = other[r];
            }
        }
    }
    AHV_ENC_STAMP(6);
}

// -------------------------------------------------------------------------------------------------
// Tiled linear for MANY rows (M >= 1024, i.e. B >= 16): C[M][N] = X[M][K] . W[N][K]^T, fp32 MFMA 16x16x4.
// The skinny kernel above splits K over the waves of a workgroup and pays an 8-way LDS reduction per 64 x 32
// output tile: right for 64..512 rows (weights are the traffic), 44 % of the matrix peak at 2048 rows.  Here a
// workgroup (256 threads, 2 x 2 waves) owns a 128 x 128 output tile and walks K in steps of 16 through three
// LDS stages; a wave owns 64 x 64 (16 accumulator tiles).  Both operands are K-contiguous, so
// a tile row in LDS is one 64-byte line [16 k] and, with the k ordering "MFMA k-step s, lane group kq <-> k = 4 kq
// + s", an operand fragment is one lane-linear (conflict-free) ds_read_b128 serving four MFMA k-steps.
// GEGLU (the FF-in projection, attention.py:81-88): the 128 tile columns are, per wave column wc, 32 VALUE columns
// and their 32 GATE columns, so value and gate of one output element sit in the same lane and the epilogue writes
// (v + b_v) * gelu(g + b_g) straight into the [M][H] buffer.
// -------------------------------------------------------------------------------------------------
struct TileArgs {
    LinProb p[kMaxProb];  // tile0 = first column tile of the problem in blockIdx.x
    int nprob, M, K;      // K = K range of ONE split (blockIdx.z = ks)
    long ldx, ldw;
    int geglu_h;
    const float* zero;    // CONV: 64 bytes of zeros (what a tap outside the 8 x 8 image reads)
#ifdef AHV_ENC_PROBE
    unsigned long long* stamps;
#endif
};

// K pipeline (round 5): four LDS stages filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write --
// the stage image is lane-linear, thread t owns bytes [16 t, 16 t + 16) of each 4 KiB piece, which is exactly what one
// such instruction per wave writes), two fragment sets.  In k-step kt a wave requests tile kt + 3, reads the fragments of
// tile kt + 1 from LDS, issues the MFMAs of tile kt (whose fragments it read a k-step ago), waits until its own pieces of
// tile kt + 2 have landed (vmcnt: tile kt + 3 stays in flight across the barrier) and meets the others at the one raw
// s_barrier -- behind which the next MFMAs can issue at once.  The memory instructions of a k-step are issued BETWEEN its
// MFMAs (one per TM MFMAs), not in front of them.  Measured on the way (tools/tile_probe.cpp, M = 2048, K = 2048, N = 4096,
// two problems): two stages + fragments read after the barrier 0.77 of the fp32 matrix peak; fragments read a k-step
// ahead 0.81; LDS-DMA 0.83; the issue order 0.90 (profiles/r05_tile_kernel_diag.txt).
// TM = 4: 128 x 128 tiles (a wave owns 64 x 64), M >= 1024 rows of the FF projections.  TM = 2: 64 x 64 tiles (a wave owns
// 32 x 32) for the 256-wide projections, whose 128-tiles would not fill the chip.
#ifndef AHV_DIAG_TILE  // diagnostic only (wrong results): bit 1 no tile loads, 2 no fragment reads, 3 no barrier
#define AHV_DIAG_TILE 0
#endif
template <int TM>
struct TileFrags {
    f32x4 a[TM], b[TM];
};

__device__ __forceinline__ void glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// CONV (TM = 2, round 6): the 3 x 3 convolutions of the 2-D res-block (modules/modules.py:126-164) as an implicit GEMM on
// this kernel: K = (tap, channel) = 9 x 256, a 64-row tile is exactly one sample's 8 x 8 tokens, and the A row a thread
// fetches for k-tile kt is its own token shifted by the tap of that k-tile -- or the 64 zero bytes of `a.zero` when the tap
// leaves the image (zero padding; LDS-DMA cannot mask a lane, it can be pointed somewhere else).  The tap of a k-tile is
// scalar arithmetic on the loop counter, the select two vector instructions per k-step: no branch in the K loop.  Round 5
// ran these two launches on the skinny kernel, one tap per K slice: 9 216 workgroups that each re-read a 64 x 256 X tile
// for 16 output columns, 64.5 us per launch at B = 32 (0.48 of the fp32 matrix peak).
template <bool GEGLU, int TM, bool CONV = false>
__global__ __launch_bounds__(256, 2) void linear_tile_kernel(const TileArgs a)
{
    static_assert(TM == 2 || TM == 4, "64 x 64 or 128 x 128 tiles");
    static_assert(!GEGLU || TM == 4, "the GEGLU column pairing is laid out for 128-column tiles");
    static_assert(!CONV || (TM == 2 && !GEGLU), "the implicit-GEMM convolution walks one sample (64 tokens) per tile");
    constexpr int T = 32 * TM;          // tile rows = tile columns
    constexpr int HALF = T * 16;        // floats of one operand tile [T][16]
    constexpr int STAGE = 2 * HALF;     // [A | B]
    constexpr int NP = TM / 2;          // 1 KiB pieces per wave, operand and k-step
    // ONE LDS object (a second one beside an LDS-DMA target makes hipcc drain the DMA before every ds_read)
    __shared__ __attribute__((aligned(1024))) float smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxProb; ++i)
        if (i < a.nprob && (int)blockIdx.x >= a.p[i].tile0) pi = i;
    const LinProb pr = a.p[pi];
    const int tile = (int)blockIdx.x - pr.tile0;
    const int m0 = blockIdx.y * T;
    AHV_ENC_STAMP(0);
    const int r16 = lane & 15, kq = lane >> 4;
    // staging: thread -> (row = tid / 4 (+64), 16-byte chunk = tid % 4) of both tiles
    const int srow = tid >> 2, sch = tid & 3;
    const float* xg[NP];
    const float* wg[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int j = srow + 64 * i;  // tile row
        xg[i] = pr.X + (long)(m0 + j) * a.ldx + (long)blockIdx.z * a.K + 4 * sch;
        int wrow;
        if (GEGLU) wrow = ((j >> 5) & 1) * a.geglu_h + tile * 64 + 32 * (j >> 6) + (j & 31);
        else wrow = tile * T + j;
        wg[i] = pr.W + (long)wrow * a.ldw + (long)blockIdx.z * a.K + 4 * sch;
    }
    f32x4 acc[TM][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 16;
    const int soff = srow * 16 + 4 * sch;                        // this thread's 16 bytes of a piece (= 4 tid floats)
    const int foffa = (16 * TM * wr + r16) * 16 + 4 * kq;        // fragment rows of this lane (+ 16 rows per i)
    const int foffb = HALF + (16 * TM * wc + r16) * 16 + 4 * kq;
    // CONV: this thread's token (y, x) inside the sample and the sample's first row
    const int cty = srow >> 3, ctx = srow & 7;
    const float* xs = pr.X + (long)m0 * 256 + 4 * sch;
    const int zoff = CONV ? (int)((a.zero + 4 * sch) - xs) : 0;   // the zero line, as an offset from this thread's row base
    auto gload = [&](int kt, int st) {  // tile kt -> stage st, 2 NP pieces of 1 KiB per wave
        float* dst = smem + st * STAGE + soff;
        if constexpr (CONV) {
            const int kg = (int)blockIdx.z * a.K + 16 * kt;   // uniform: the k-tile's first column in (tap, channel) order
            const int tap = kg >> 8, ci = kg & 255;
            const int t3 = (tap * 11) >> 5;                   // tap / 3 for tap < 9
            const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
            const int ny = cty + dy, nx = ctx + dx;
            // an arithmetic select on a 32-bit offset (the workspace is far below 8 GB): written as `inside ? p : q` on pointers
            // hipcc turned the k-step into exec-masked blocks, i.e. a K loop of several basic blocks -- and then waits for
            // every DMA and fragment read in front of the MFMAs (tests/test_isa_hazard.py checks the loop's shape)
            const int keep = -(int)((unsigned)(ny | nx) < 8u);   // all ones inside the image (a negative coordinate sets the OR's sign bit)
            const int off = (((ny * 8 + nx) * 256 + ci) & keep) | (zoff & ~keep);
            glds16(xs + off, dst);
            glds16(wg[0] + 16 * kt, dst + HALF);
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                glds16(xg[i] + 16 * kt, dst + 1024 * i);
                glds16(wg[i] + 16 * kt, dst + HALF + 1024 * i);
            }
        }
    };
    auto fread = [&](TileFrags<TM>& f, int st) {
        const float* src = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.a[i] = *reinterpret_cast<const f32x4*>(src + foffa + 256 * i);
            f.b[i] = *reinterpret_cast<const f32x4*>(src + foffb + 256 * i);
        }
    };
    auto mma = [&](const TileFrags<TM>& f) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][s4], f.b[j][s4], acc[i][j], 0, 0, 0);
    };
    // vmcnt(TM) lgkmcnt(0): this wave's pieces of tile kt + 2 are in LDS, the TM pieces of tile kt + 3 may still be on
    // their way when the barrier is met; `nxt` has long arrived
    constexpr int kWait = 0x0070 | TM;
    // one k-step: `cur` holds tile kt's fragments, `nxt` receives tile kt + 1's; tile kt + 3 goes into the stage tile
    // kt - 1 occupied (its fragments were read two barriers ago).  No conditions in here: behind the last tile the loads
    // repeat tile nk - 1 into a stage nobody reads again and the fragment read fetches a tile nobody multiplies -- every
    // branch would cost hipcc its count of what is in flight (it then waits for the fragment reads in front of the MFMAs).
    auto kstep = [&](int kt, const TileFrags<TM>& cur, TileFrags<TM>& nxt) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(AHV_DIAG_TILE & 2)) gload(kt + 3 < nk ? kt + 3 : nk - 1, (kt + 3) & 3);
        if (!(AHV_DIAG_TILE & 4)) fread(nxt, (kt + 1) & 3);
        mma(cur);
        // issue order: the 3 TM memory instructions go BETWEEN the MFMAs (one per TM), where their issue slots are free --
        // all of them in front of the first MFMA leave the matrix pipe waiting whenever the SIMD's other wave stands at
        // the same point
#pragma unroll
        for (int g = 0; g < TM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);  // TM MFMAs
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 VMEM (an LDS-DMA piece)
        }
#pragma unroll
        for (int g = 0; g < 2 * TM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);  // TM MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TM, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kWait);
        if (!(AHV_DIAG_TILE & 8)) __builtin_amdgcn_s_barrier();
    };
    // prologue: tiles 0, 1, 2 requested; tiles 0 and 1 landed before the first barrier
    TileFrags<TM> f0, f1;
    gload(0, 0);
    gload(1, 1);
    gload(nk > 2 ? 2 : nk - 1, 2);
    __builtin_amdgcn_s_waitcnt(kWait);
    __builtin_amdgcn_s_barrier();
    AHV_ENC_STAMP(1);
    fread(f0, 0);
    if (AHV_DIAG_TILE & 4) fread(f1, 0);
    // (a wait here, or hipcc carries "f0 pending" into the loop)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    for (int kt = 0; kt < nk; kt += 2) {  // nk is even (tile_eligible)
        kstep(kt, f0, f1);
        kstep(kt + 1, f1, f0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the repeated loads have landed before the LDS is given up
    AHV_ENC_STAMP(4);
    // D layout: acc[i][j][r] = C[m0 + 16 TM wr + 16 i + 4 kq + r][column 16 TM wc + 16 j + r16 of the tile]: 4-byte stores, four
    // 64-byte row segments per instruction.  (The transposed tiles -- W rows as the A operand, a lane then holds four
    // consecutive columns and stores 16 bytes, sixteen 64-byte row segments per instruction -- measured 1-3 % SLOWER.)
    if constexpr (GEGLU) {
        const int H = a.geglu_h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tile * 64 + 32 * wc + 16 * j + r16;
            const float bv = pr.bias[col], bg = pr.bias[H + col];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const long row = m0 + 64 * wr + 16 * i + 4 * kq + r;
                    const f32x2 y = geglu_pk(f32x2{acc[i][j][r] + bv, acc[i][j][r + 1] + bv},
                                             f32x2{acc[i][j + 2][r] + bg, acc[i][j + 2][r + 1] + bg});
                    AHV_TILE_STORE(pr.P + row * H + col, y[0]);
                    AHV_TILE_STORE(pr.P + (row + 1) * H + col, y[1]);
                }
        }
    } else {
        float* P = pr.P + (long)blockIdx.z * a.M * pr.N;  // split-K slab [ks][M][N]
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int col = tile * T + 16 * TM * wc + 16 * j + r16;
            const float bias = (pr.bias != nullptr && gridDim.z == 1) ? pr.bias[col] : 0.0f;  // a bias only without split-K (host)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    AHV_TILE_STORE(P + (long)(m0 + 16 * TM * wr + 16 * i + 4 * kq + r) * pr.N + col, acc[i][j][r] + bias);
        }
    }
    AHV_ENC_STAMP(6);
}

// -------------------------------------------------------------------------------------------------
// Attention + output projection in one launch (CrossAttention.forward, transformer/attention.py:214-237, up to but
// not including to_out's bias).
// -------------------------------------------------------------------------------------------------
struct AttnOutProb {
    const float *Q, *K, *V;
    const float* Wo;  // [256][256] to_out.0.weight
    float* P;         // [M][256]
    long ldq, ldkv;
};
struct AttnOutArgs {
    AttnOutProb p[2];
    int B;
    float scale;
#ifdef AHV_ENC_PROBE
    unsigned long long* stamps;
#endif
};

// -------------------------------------------------------------------------------------------------
// The four waves of a workgroup cooperate on ONE head of one 16-query tile: grid = (B * nprob, 4 query tiles,
// 4 heads).  (Round 2's first form gave every wave a whole head and repeated the attention per column quarter: 192
// dependent MFMAs per wave, 9.2 us at B = 1; this one 6.5 us.)  Wave w scores key tile w
// (16 MFMAs), the four partial softmaxes are merged through 128 bytes of LDS with the online-softmax identity
// (p_w = exp(s - m_w); m = max m_w; L = sum l_w exp(m_w - m); P = p_w exp(m_w - m) / L: ONE barrier), wave w forms its
// keys' share of O^T (16 MFMAs), the shares meet in LDS, and wave w then contracts the head's O with column quarter
// w of W_out (64 MFMAs).  96 dependent MFMAs per wave instead of 192, a quarter of the K / V loads.  The result is
// one split-K slab PER HEAD, P[h][M][256]; ln_kernel sums the four.
// -------------------------------------------------------------------------------------------------
// QT query tiles per workgroup (round 6 experiment, QT = 1 ships): a workgroup can walk several 16-query tiles of its
// (sample, head) with the K, V and W_out fragments it holds in registers (96 of the ~100 KB a workgroup reads; only Q changes
// per tile, requested a tile ahead).  At B = 32 it changes nothing (see AHV_ATTN_QT at the launch): the launch is bound by the
// dependent chain of a tile (16 + 16 + 64 MFMAs, two barriers, a softmax), not by its 100 MB of L2 reads.
// (sample, head) pairs from which attention_sample_head_kernel runs (tools/kbench_enc, both kernels alternating on one box, us per
// forward: B = 8 -- 64 pairs -- 751 with the tile-wise kernel / 784 with this one; B = 16: 1 106 / 1 111; B = 24: 1 606 / 1 596; B = 32: 1 919 / 1 895)
#ifndef AHV_ATTN_PAIR_MIN_WGS
#define AHV_ATTN_PAIR_MIN_WGS 192
#endif
template <int QT>
__global__ __launch_bounds__(256) void attention_heads_kernel(const AttnOutArgs a, int M)
{
    __shared__ float sm[4][16][2];                                   // per wave and query: (max, sum)
    __shared__ __attribute__((aligned(16))) float so[4][4][64][4];   // per wave: O^T share [dt][lane][r]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pi = (int)blockIdx.x / a.B, b = (int)blockIdx.x - pi * a.B;
    const AttnOutProb pr = a.p[pi];
    const int it0 = blockIdx.y * QT, h = blockIdx.z;
    const int c16 = lane & 15, kq = lane >> 4;
    const float* q = pr.Q + (long)(b * 64 + it0 * 16 + c16) * pr.ldq + h * 64 + 4 * kq;
    const float* k = pr.K + (long)(b * 64 + w * 16 + c16) * pr.ldkv + h * 64 + 4 * kq;
    const float* v = pr.V + (long)(b * 64 + w * 16 + 4 * kq) * pr.ldkv + h * 64 + c16;
    AHV_ENC_STAMP(0);
    // every operand is requested before the first MFMA
    f32x4 ka[4], qb[4], wf[4][4];
    float va[4][4];  // [r][dt]: V[16 w + 4 kq + r][16 dt + c16]
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        qb[ds] = *reinterpret_cast<const f32x4*>(q + 16 * ds);
        ka[ds] = *reinterpret_cast<const f32x4*>(k + 16 * ds);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) va[r][dt] = v[(long)r * pr.ldkv + dt * 16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            wf[nt][dt] = *reinterpret_cast<const f32x4*>(pr.Wo + (long)(w * 64 + nt * 16 + c16) * 256 + h * 64 + 16 * dt + 4 * kq);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int t = 0; t < QT; ++t) {
        f32x4 qn[4];   // the next tile's queries travel while this one is processed (the last tile re-reads itself)
        if (QT > 1) {
            const float* qq = q + (long)(t + 1 < QT ? t + 1 : t) * 16 * pr.ldq;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) qn[ds] = *reinterpret_cast<const f32x4*>(qq + 16 * ds);
        }
        // S^T[j = 16 w + 4 kq + r][i = c16]
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[ds][s], qb[ds][s], st, 0, 0, 0);
        if (t == 0) AHV_ENC_STAMP(1);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[r] *= a.scale;
            m = fmaxf(m, st[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[r] = expf(st[r] - m);
            l += st[r];
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (kq == 0) {
            sm[w][c16][0] = m;
            sm[w][c16][1] = l;
        }
        if (t == 0) AHV_ENC_STAMP(2);
        __syncthreads();
        if (t == 0) AHV_ENC_STAMP(3);
        float mg = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) mg = fmaxf(mg, sm[u][c16][0]);
        float L = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) L += sm[u][c16][1] * expf(sm[u][c16][0] - mg);
        const float f = expf(m - mg) / L;
        // this wave's share of O^T[d = 16 dt + 4 kq + r][i = c16]
        f32x4 ot[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = st[r] * f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[r][dt], p, ot[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(&so[w][dt][lane][0]) = ot[dt];
        if (t == 0) AHV_ENC_STAMP(4);
        __syncthreads();   // (also: everybody has read sm, the next tile may overwrite it; so is rewritten behind the next tile's first barrier)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ot[dt] = *reinterpret_cast<const f32x4*>(&so[0][dt][lane][0]);
#pragma unroll
            for (int u = 1; u < 4; ++u) ot[dt] += *reinterpret_cast<const f32x4*>(&so[u][dt][lane][0]);
        }
        // column quarter w of the head's output projection: D^T[n = 64 w + 16 nt + 4 kq + r][q = c16]
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][dt][r], ot[dt][r], acc[nt], 0, 0, 0);
        float* out = pr.P + ((long)h * M + b * 64 + (it0 + t) * 16 + c16) * 256 + w * 64 + 4 * kq;
        if (t == 0) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); AHV_ENC_STAMP(5); }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) AHV_ROW_STORE4(out + nt * 16, acc[nt]);
        if (t == QT - 1) AHV_ENC_STAMP(6);
        if (QT > 1) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) qb[ds] = qn[ds];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The same launch for MANY samples (B >= 24, round 6): one workgroup per (sample, head), wave w owns the 16 queries of
// tile w against ALL 64 keys.  The in-kernel timeline of the kernel above at B = 32 (tools/kbench_enc.bin 32 --stamps,
// profiles/r06_attention_timeline_b32.txt) showed what its 17.9 us are: 1 024 workgroups of ~150 registers run three per CU,
// so a quarter of them waits 11 us for a slot; every workgroup spends its first 4.2 us fetching 100 KB of operands with
// fragment-shaped loads (64 KB of them W_out: 64 MB over the grid for a 256 KB matrix) and then 6.3 us in a chain of 96
// MFMAs, two barriers and a softmax whose every vector instruction queues behind the other waves' MFMAs.  Here a wave has
// whole softmax rows (no merge, no barrier between scores and P V), four independent MFMA chains in every phase (64 + 64 +
// 256 MFMAs per wave), every operand arrives ONCE per workgroup as full 256-byte rows by LDS-DMA, and W_out's head slice
// travels while softmax and P V run: 256 workgroups at B = 32, one per CU, 29 MB of operand traffic.
// Same math as above (softmax over the 64 keys in one pass: p = exp2((s - max) scale log2 e) / sum), same output: one
// split-K slab per head.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_sample_head_kernel(const AttnOutArgs a, int M)
{
    // ONE LDS object (see linear_tile_kernel): row-major images [row][64 d] of the head's Q, K, V (64 rows each) and W_out
    // slice (256 rows), 16-byte chunks XOR-swizzled per row on the SOURCE side (the images are filled by LDS-DMA: lane-linear)
    constexpr int kQ = 0, kK = 64 * 64, kV = 2 * 64 * 64, kW = 3 * 64 * 64;
    __shared__ __attribute__((aligned(1024))) float img[kW + 256 * 64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pi = (int)blockIdx.x / a.B, b = (int)blockIdx.x - pi * a.B;
    const AttnOutProb pr = a.p[pi];
    const int h = blockIdx.y;
    const int c16 = lane & 15, kq = lane >> 4;
    AHV_ENC_STAMP(0);
    // Every operand comes through LDS in 1 KiB pieces of four 256-byte rows (lane -> row lane / 16, chunk lane % 16): the
    // fragment-shaped loads of the kernel above put 16 lanes on 16 different rows (16 quads x 4 segments = 64 cycles of the
    // CU's address unit per instruction; with 36 of them per wave the first 3.5 us of the workgroup), these take 16.
    // Chunk swizzles (conflict-free for the ds_read_b128 lane groups, checked in DESIGN 4.3): Q, K: c ^ (row % 16);
    // V: c ^ (row % 4); W_out: c ^ rot(row % 16), rot = the nibble's two bit pairs exchanged.
    const int prow = lane >> 4, pch = lane & 15;
    {
        const long r0 = b * 64 + 16 * w;   // wave w fetches rows 16 w .. 16 w + 15 of Q, K and V
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * i + prow;   // row % 16
            glds16(pr.Q + (r0 + row) * pr.ldq + h * 64 + 4 * (pch ^ row), img + kQ + (16 * w + 4 * i) * 64 + 4 * lane);
            glds16(pr.K + (r0 + row) * pr.ldkv + h * 64 + 4 * (pch ^ row), img + kK + (16 * w + 4 * i) * 64 + 4 * lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * i + prow;
            glds16(pr.V + (r0 + row) * pr.ldkv + h * 64 + 4 * (pch ^ (row & 3)), img + kV + (16 * w + 4 * i) * 64 + 4 * lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 qb[4], ka[4][4], va[4][4];   // va[j][r][dt] = V[key 16 j + 4 kq + r][d = 4 c16 + dt]
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        qb[ds] = *reinterpret_cast<const f32x4*>(img + kQ + (16 * w + c16) * 64 + 4 * ((4 * ds + kq) ^ c16));
#pragma unroll
        for (int j = 0; j < 4; ++j) ka[j][ds] = *reinterpret_cast<const f32x4*>(img + kK + (16 * j + c16) * 64 + 4 * ((4 * ds + kq) ^ c16));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) va[j][r] = *reinterpret_cast<const f32x4*>(img + kV + (16 * j + 4 * kq + r) * 64 + 4 * (c16 ^ r));
    // The head dimension is walked in two orders: d = 16 ds + 4 kq + s for Q K^T (a lane's four consecutive d are MFMA
    // k-steps) and d = 16 kq + 4 r + dt for P V and the projection (a lane's four consecutive d are the four d-TILES: tile
    // dt = {d : d mod 4 = dt}) -- any order serves a contraction as long as both operands use it.
    // S^T[key = 16 j + 4 kq + r][query = c16], four independent chains
    f32x4 st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
#pragma unroll
        for (int sx = 0; sx < 4; ++sx)
#pragma unroll
            for (int j = 0; j < 4; ++j) st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j][ds][sx], qb[ds][sx], st[j], 0, 0, 0);
    // W_out rows 64 w .. 64 w + 63 of the head's slice (16 pieces).  Requested HERE, with every fragment of Q, K and V in
    // registers: hipcc drains LDS-DMA in front of any ds_read of the same object, so nothing is read between this point and
    // the barrier in front of the projection -- the pieces travel while softmax and P V run.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(va[j][r]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = 64 * w + 4 * i + prow, r16 = row & 15;
        glds16(pr.Wo + (long)row * 256 + h * 64 + 4 * (pch ^ (((r16 & 3) << 2) | (r16 >> 2))), img + kW + (64 * w + 4 * i) * 64 + 4 * lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    AHV_ENC_STAMP(1);
    const float c = a.scale * 1.44269504088896341f;
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[j][r] *= c;
            m = fmaxf(m, st[j][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[j][r] = __builtin_amdgcn_exp2f(st[j][r] - m);
            l += st[j][r];
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    AHV_ENC_STAMP(2);
    // O^T[d = 4 (4 kq + r) + dt][query = c16] in tile dt
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = st[j][r] * inv;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j][r][dt], p, ot[dt], 0, 0, 0);
        }
    AHV_ENC_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);    // (MFMAs on registers would otherwise be free to move across the barrier)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's W_out pieces are in LDS
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    AHV_ENC_STAMP(4);
    // the head's output projection for this wave's 16 queries: D^T[n = 16 nt + 4 kq + r][query = c16], four n-tiles at a time;
    // k-step (dt, r) contracts d = 16 kq + 4 r + dt: B = ot[dt][r], A = component dt of W_out[16 nt + c16][16 kq + 4 r ..]
    float* out = pr.P + ((long)h * M + b * 64 + 16 * w + c16) * 256 + 4 * kq;
    const float* wrow = img + kW + c16 * 64;
    const int wsw = ((c16 & 3) << 2) | (c16 >> 2);
#pragma unroll
    for (int ng = 0; ng < 4; ++ng) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wrow + 16 * (4 * ng + i) * 64 + 4 * ((4 * kq + r) ^ wsw));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][dt], ot[dt][r], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) AHV_ROW_STORE4(out + 16 * (4 * ng + i), acc[i]);
    }
    AHV_ENC_STAMP(6);
}

// -------------------------------------------------------------------------------------------------
// Row kernels (one wave per 256-wide row; blockIdx.y = problem).
// -------------------------------------------------------------------------------------------------
struct LnProb {
    const float* P;     // [KS][M][256] split-K slabs of the preceding linear
    const float* bias;  // [256]
    const float* g;     // LayerNorm weight
    const float* be;    // LayerNorm bias
    const float* x;     // [M][256] block input (copied into cat / added as residual)
    float* out;         // CONCAT: cat [M][512];  else: block output [M][256]
};
struct LnArgs {
    LnProb p[2];
    int KS, M;
};

// LayerNorm(256, eps 1e-5, biased variance) of (sum_ks P + bias);
// CONCAT: out[row] = [x[row], y]   (attention.py:255-256)   else: out[row] = x[row] + y   (:257-258)
// KS is a template parameter so that the slab loads are independent instructions issued together: with a run-time
// trip count hipcc keeps the loop rolled and every slab costs its own memory round trip.
template <bool CONCAT, int KS>
__global__ __launch_bounds__(256) void ln_kernel(const LnArgs a)
{
    const LnProb pr = a.p[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    f32x4 slab[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) slab[ks] = *reinterpret_cast<const f32x4*>(pr.P + ((long)ks * a.M + row) * 256 + 4 * lane);
    const f32x4 xin = *reinterpret_cast<const f32x4*>(pr.x + (long)row * 256 + 4 * lane);
    const f32x4 gam = *reinterpret_cast<const f32x4*>(pr.g + 4 * lane), bet = *reinterpret_cast<const f32x4*>(pr.be + 4 * lane);
    f32x4 t = *reinterpret_cast<const f32x4*>(pr.bias + 4 * lane);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) t += slab[ks];
    const float mean = wave_sum(t[0] + t[1] + t[2] + t[3]) * (1.0f / 256.0f);
    const f32x4 d = t - mean;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const f32x4 y = d * rstd * gam + bet;
    if (CONCAT) {
        AHV_ROW_STORE4(pr.out + (long)row * 512 + 4 * lane, xin);
        AHV_ROW_STORE4(pr.out + (long)row * 512 + 256 + 4 * lane, y);
    } else {
        AHV_ROW_STORE4(pr.out + (long)row * 256 + 4 * lane, xin + y);
    }
}

// -------------------------------------------------------------------------------------------------
// Kernels around the token stage (Feature_Aligner.forward_2d3d, modules/modules.py:86-101).
// Activations live token-major [M = 64*B][C] ("channel-last"); blockIdx.y = stream.
// -------------------------------------------------------------------------------------------------
struct Nchw2TokArgs {
    const float* in[2];  // [B][C][64]
    float* out[2];       // [B*64][C]
    int C, B;
    float* zero;         // 16 floats the first workgroup clears: the zero row of the implicit-GEMM convolutions
};

__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const Nchw2TokArgs a)
{
    __shared__ float tile[64][65];
    const float* in = a.in[blockIdx.z];
    float* out = a.out[blockIdx.z];
    const int b = blockIdx.y, c0 = blockIdx.x * 64;
    if (a.zero != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 16) a.zero[threadIdx.x] = 0.0f;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, m = i & 63;
        tile[c][m] = in[((long)b * a.C + c0 + c) * 64 + m];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int m = i >> 6, c = i & 63;
        out[((long)b * 64 + m) * a.C + c0 + c] = tile[c][m];
    }
}

// GroupNorm(32 groups, eps 1e-6, biased variance) over (8 channels x 64 tokens) of one sample
// (transformer/attention.py:120-121,378,383): one wave per (sample, group), lane = token.
struct GnArgs {
    const float* x[2];
    float* y[2];
    const float *g, *be;
    int B;
};

__global__ __launch_bounds__(256) void groupnorm_kernel(const GnArgs a)
{
    const int lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6);  // 0..31
    const int b = blockIdx.y;
    const float* x = a.x[blockIdx.z] + ((long)b * 64 + lane) * 256 + grp * 8;
    float* y = a.y[blockIdx.z] + ((long)b * 64 + lane) * 256 + grp * 8;
    const f32x4 u = *reinterpret_cast<const f32x4*>(x), v = *reinterpret_cast<const f32x4*>(x + 4);
    const float mean = wave_sum(u[0] + u[1] + u[2] + u[3] + v[0] + v[1] + v[2] + v[3]) * (1.0f / 512.0f);
    const f32x4 du = u - mean, dv = v - mean;
    const float var = wave_sum(du[0] * du[0] + du[1] * du[1] + du[2] * du[2] + du[3] * du[3] + dv[0] * dv[0] +
                               dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * (1.0f / 512.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.g + grp * 8), g1 = *reinterpret_cast<const f32x4*>(a.g + grp * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.be + grp * 8), b1 = *reinterpret_cast<const f32x4*>(a.be + grp * 8 + 4);
    *reinterpret_cast<f32x4*>(y) = du * rstd * g0 + b0;
    *reinterpret_cast<f32x4*>(y + 4) = dv * rstd * g1 + b1;
}

// Generic finish: v = sum_ks P[ks][m][n] (+ bias[n]); relu on columns < relu_cols; + res[m][n]; + pe[m % 64][n];
// stored as  layout 0: out[m][ldo] token-major
//            layout 1: channel-last volume from tokens, out[b][d][h][w][c'] with token m = (b, h*8+w), n = c'*8 + d
//                      (the reshape(bs, 32, 8, 8, 8) of modules/modules.py:97)
//            layout 2: NCDHW, out[b][n][voxel] with row m = (b, voxel)            (the scorer's volume layout)
//            layout 3: columns [0, N/2) to out[m][N/2], columns [N/2, N) to out2[m][N/2]   (the 3-D res-block's first
//                      conv carries its 1x1x1 skip as extra output columns: one pass splits them)
struct FinProb {
    const float* P;
    const float* bias;
    const float* res;
    float* out;
    float* out2;
};
struct FinArgs {
    FinProb p[2];
    const float* pe;
    int KS, M, N, ldp, col0;  // P slabs are [KS][M][ldp]; columns col0 .. col0+N-1 are used
    int relu_cols, ldres, ldo, layout;
};

template <int KS>  // compile-time: every slab (and the optional operands) is requested before the first add
__global__ __launch_bounds__(256) void finish_kernel(const FinArgs a)
{
    const FinProb pr = a.p[blockIdx.y];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = a.N >> 2;
    if (i >= (long)a.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 slab[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) slab[ks] = *reinterpret_cast<const f32x4*>(pr.P + ((long)ks * a.M + m) * a.ldp + a.col0 + n);
    const f32x4 res = pr.res ? *reinterpret_cast<const f32x4*>(pr.res + (long)m * a.ldres + n) : zero;
    const f32x4 pe = a.pe ? *reinterpret_cast<const f32x4*>(a.pe + (long)(m & 63) * a.N + n) : zero;
    f32x4 v = pr.bias ? *reinterpret_cast<const f32x4*>(pr.bias + n) : zero;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) v += slab[ks];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (n + e < a.relu_cols) v[e] = fmaxf(v[e], 0.0f);
    v += res;
    v += pe;
    if (a.layout == 0) {
        *reinterpret_cast<f32x4*>(pr.out + (long)m * a.ldo + n) = v;
    } else if (a.layout == 1) {
        const int b = m >> 6, hw = m & 63, cp = n >> 3, d0 = n & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) pr.out[(((long)b * 8 + d0 + e) * 64 + hw) * 32 + cp] = v[e];
    } else if (a.layout == 3) {
        const int h = a.N >> 1;
        float* dst = n < h ? pr.out + (long)m * h + n : pr.out2 + (long)m * h + (n - h);
        *reinterpret_cast<f32x4*>(dst) = v;
    } else {
        const int b = m >> 9, vox = m & 511;
#pragma unroll
        for (int e = 0; e < 4; ++e) pr.out[((long)b * a.N + n + e) * 512 + vox] = v[e];
    }
}

// ---- host side -------------------------------------------------------------------------------------
struct LinSpec {
    const float* X;
    const float* W;
    float* P;
    const float* bias;
    int N;
};

static bool tile64_eligible(const LinSpec* specs, int nprob, int M, int K);
static hipError_t launch_linear_tile(const LinSpec* specs, int nprob, long ldx, long ldw, int M, int K, int KS, int geglu_h,
                                     hipStream_t s, int tile, const float* zero = nullptr);

static hipError_t launch_linear(const LinSpec* specs, int nprob, long ldx, long ldw, int M, int K, int KS, int geglu_h,
                                hipStream_t s, int mode = 0)
{
    if (mode == 0 && KS == 1 && geglu_h == 0 && K == 256 && tile64_eligible(specs, nprob, M, K))
        return launch_linear_tile(specs, nprob, ldx, ldw, M, K, 1, 0, s, 64);
    const bool wide = geglu_h > 0;  // the GEGLU projection: value tile + gate tile per workgroup
    const int cols = wide ? 32 : 16;
    const int Kw = K / KS / 8;
    LinArgs a;
    a.nprob = nprob; a.M = M; a.Kc = K / KS; a.ldx = ldx; a.ldw = ldw; a.geglu_h = geglu_h;
#ifdef AHV_ENC_PROBE
    a.stamps = AHV_ENC_STAMP_PTR;
#endif
    int tiles = 0;
    for (int i = 0; i < kMaxProb; ++i) {
        const LinSpec& sp = specs[i < nprob ? i : 0];
        a.p[i] = LinProb{sp.X, sp.W, sp.P, sp.bias, sp.N, tiles};
        if (i < nprob) tiles += sp.N / cols;
    }
    const dim3 grid(tiles, KS, M / 64);
#ifndef AHV_ENC_NO_STAGED
    if (mode != 2 && (a.Kc == 256 || (a.Kc == 512 && mode == 0)) && (!wide || (KS == 1 && a.Kc == 512))) {  // K-pipelined form (same grid)
        if (mode == 1 && a.Kc == 256) AHV_ENC_LAUNCH((linear_staged_kernel<1, 256, false, 1>), grid, dim3(512), 0, s, a);
        else if (mode == 0 && wide && a.Kc == 512) AHV_ENC_LAUNCH((linear_staged_kernel<2, 512, true>), grid, dim3(512), 0, s, a);
        else if (mode == 0 && !wide && a.Kc == 512) AHV_ENC_LAUNCH((linear_staged_kernel<1, 512, false>), grid, dim3(512), 0, s, a);
        else if (mode == 0 && !wide && a.Kc == 256) AHV_ENC_LAUNCH((linear_staged_kernel<1, 256, false>), grid, dim3(512), 0, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#endif
    if (mode == 1 && Kw == 32) AHV_ENC_LAUNCH((linear_kernel<1, 32, false, 1>), grid, dim3(512), 0, s, a);
    else if (mode == 2 && Kw == 32) AHV_ENC_LAUNCH((linear_kernel<1, 32, false, 2>), grid, dim3(512), 0, s, a);
    else if (mode == 2 && Kw == 16) AHV_ENC_LAUNCH((linear_kernel<1, 16, false, 2>), grid, dim3(512), 0, s, a);
    else if (mode != 0) return hipErrorInvalidValue;
    else if (wide && Kw == 64 && KS == 1) AHV_ENC_LAUNCH((linear_kernel<2, 64, true>), grid, dim3(512), 0, s, a);
    else if (!wide && Kw == 64) AHV_ENC_LAUNCH((linear_kernel<1, 64, false>), grid, dim3(512), 0, s, a);
    else if (!wide && Kw == 32 && !geglu_h) AHV_ENC_LAUNCH((linear_kernel<1, 32, false>), grid, dim3(512), 0, s, a);
    else if (!wide && Kw == 16 && !geglu_h) AHV_ENC_LAUNCH((linear_kernel<1, 16, false>), grid, dim3(512), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Many-row path of two-problem plain / GEGLU linears (both streams in one launch); KS = 1 output layout.
static bool tile_eligible(int M, int K, int N, int geglu_h)
{
#ifndef AHV_TILE_MIN_M
#define AHV_TILE_MIN_M 1024
#endif
    if (M < AHV_TILE_MIN_M || (M & 127) || (K & 31)) return false;  // K per split: an even number of 16-wide k-steps
    return geglu_h > 0 ? (geglu_h % 64 == 0 && N == 2 * geglu_h) : (N % 128 == 0);
}

// tile = 128: the two-problem FF projections; tile = 64: up to four plain problems (q / k / v, proj_in / proj_out)
// 256-wide projections (K = 256, N = 256 / 512 / 768) at many rows: 64 x 64 tiles instead of the skinny kernel, whose
// workgroups re-read their 64 x 256 X tile once per 16 output columns.  Measured per forward (tools/kbench_enc, graph):
// B = 32: 2040 -> 2012 us (17.3 against 21.5 us per launch); B = 16: 1238 -> 1253; B = 8: 753 -> 761; B = 4: 490 -> 508.
#ifndef AHV_TILE64_MIN_M
#define AHV_TILE64_MIN_M 2048
#endif
static bool tile64_eligible(const LinSpec* specs, int nprob, int M, int K)
{
    if (M < AHV_TILE64_MIN_M || (M & 63) || (K & 31) || nprob > kMaxProb) return false;
    for (int i = 0; i < nprob; ++i)
        if (specs[i].N & 63) return false;
    return true;
}

static hipError_t launch_linear_tile(const LinSpec* specs, int nprob, long ldx, long ldw, int M, int K, int KS, int geglu_h,
                                     hipStream_t s, int tile, const float* zero)
{
    TileArgs a;
    a.nprob = nprob; a.M = M; a.K = K / KS; a.ldx = ldx; a.ldw = ldw; a.geglu_h = geglu_h;
    int tiles = 0;
    for (int i = 0; i < kMaxProb; ++i) {
        const LinSpec& sp = specs[i < nprob ? i : 0];
        a.p[i] = LinProb{sp.X, sp.W, sp.P, sp.bias, sp.N, tiles};
        if (i < nprob) tiles += geglu_h > 0 ? geglu_h / 64 : sp.N / tile;
    }
    a.zero = zero;
#ifdef AHV_ENC_PROBE
    a.stamps = AHV_ENC_STAMP_PTR;
#endif
    const dim3 grid(tiles, M / tile, geglu_h > 0 ? 1 : KS);
    if (zero != nullptr) {  // the 3 x 3 convolution as an implicit GEMM (K = 9 x 256)
        if (tile != 64 || geglu_h > 0 || K != 2304 || ldx != 256) return hipErrorInvalidValue;
        AHV_ENC_LAUNCH((linear_tile_kernel<false, 2, true>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (geglu_h > 0 && tile == 128) AHV_ENC_LAUNCH((linear_tile_kernel<true, 4>), grid, dim3(256), 0, s, a);
    else if (geglu_h > 0) return hipErrorInvalidValue;
    else if (tile == 128) AHV_ENC_LAUNCH((linear_tile_kernel<false, 4>), grid, dim3(256), 0, s, a);
    else if (tile == 64) AHV_ENC_LAUNCH((linear_tile_kernel<false, 2>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

size_t transformer_workspace_floats(int B)
{
    const size_t M = (size_t)64 * (size_t)(B > 0 ? B : 0);
    // per stream: qkv 768 + kv 512 + (spare 256) + cat 512 + slabs / gated activations 4096 + ping-pong tokens 256;
    // two streams
    return 2 * M * (768 + 512 + 256 + 512 + 4096 + 256) + 64;
}

struct StreamWs {
    float *qkv, *kv, *att, *cat, *part, *tmp;
};

static StreamWs carve(float* ws, int M, int which)
{
    float* base = ws + (size_t)which * M * (768 + 512 + 256 + 512 + 4096 + 256);
    StreamWs w;
    w.qkv = base;
    w.kv = w.qkv + (size_t)M * 768;
    w.att = w.kv + (size_t)M * 512;
    w.cat = w.att + (size_t)M * 256;
    w.part = w.cat + (size_t)M * 512;
    w.tmp = w.part + (size_t)M * 4096;
    return w;
}

// One BasicTransformerBlock on EACH stream in the same launches: stream s uses weights w[s], input x[s],
// attention context ctx[s] (== x[s] for self-attention) and writes out[s].
static int run_block_pair(const ahv_block_weights* const w[2], const float* const x[2], const float* const ctx[2],
                          bool self_attn, float* const out[2], int M, const StreamWs ws[2], hipStream_t s,
                          const char** what)
{
    const int B = M / 64;
    hipError_t e;
#define AHV_TRY(call, name) do { e = (call); if (e != hipSuccess) { *what = name; return (int)e; } } while (0)
    AttnOutArgs ao;  // q, k, v of each stream (row strides ldq / ldkv), W_out, one output slab per head in `part`
    ao.B = B;
    ao.scale = 0.125f;  // dim_head ** -0.5
    if (self_attn) {
        LinSpec sp[2];
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{x[i], w[i]->w_qkv, ws[i].qkv, nullptr, 768};
        AHV_TRY(launch_linear(sp, 2, 256, 256, M, 256, 1, 0, s), "qkv projection");
        for (int i = 0; i < 2; ++i)
            ao.p[i] = AttnOutProb{ws[i].qkv, ws[i].qkv + 256, ws[i].qkv + 512, w[i]->w_out, ws[i].part, 768, 768};
    } else {
        LinSpec sq[2], skv[2];
        for (int i = 0; i < 2; ++i) {
            sq[i] = LinSpec{x[i], w[i]->w_qkv, ws[i].qkv, nullptr, 256};
            skv[i] = LinSpec{ctx[i], w[i]->w_qkv + 256 * 256, ws[i].kv, nullptr, 512};
        }
        LinSpec sp[4] = {sq[0], sq[1], skv[0], skv[1]};
        AHV_TRY(launch_linear(sp, 4, 256, 256, M, 256, 1, 0, s), "q / kv projections");
        for (int i = 0; i < 2; ++i)
            ao.p[i] = AttnOutProb{ws[i].qkv, ws[i].kv, ws[i].kv + 256, w[i]->w_out, ws[i].part, 256, 512};
    }
    {   // attention and the output projection in ONE launch (attention_heads_kernel), then norm1 + concat
#ifndef AHV_ATTN_QT   // A/B knob (tools/kbench_enc): query tiles per workgroup at B >= 32.  Measured in round 6: 1 / 2 / 4 tiles per
#define AHV_ATTN_QT 1   // workgroup give 1 964 / 1 958 / 1 960 us per forward (17.6-18.1 us per launch either way): not the re-reads
#endif                  // of K, V and W_out bound this launch but the dependent chain of one tile; the one-tile form stays
#ifdef AHV_ENC_PROBE
        ao.stamps = AHV_ENC_STAMP_PTR;
#endif
        // from 192 (sample, head) pairs on (B >= 24): one workgroup per pair (attention_sample_head_kernel)
        if (2 * B * 4 >= AHV_ATTN_PAIR_MIN_WGS) AHV_ENC_LAUNCH(attention_sample_head_kernel, dim3(2 * B, 4), dim3(256), 0, s, ao, M);
        else if (B >= 32 && AHV_ATTN_QT == 4) AHV_ENC_LAUNCH(attention_heads_kernel<4>, dim3(2 * B, 1, 4), dim3(256), 0, s, ao, M);
        else if (B >= 32 && AHV_ATTN_QT == 2) AHV_ENC_LAUNCH(attention_heads_kernel<2>, dim3(2 * B, 2, 4), dim3(256), 0, s, ao, M);
        else AHV_ENC_LAUNCH(attention_heads_kernel<1>, dim3(2 * B, 4, 4), dim3(256), 0, s, ao, M);
        AHV_TRY(hipGetLastError(), "attention + out projection");
        LnArgs ln;
        ln.KS = 4; ln.M = M;  // one slab per head
        for (int i = 0; i < 2; ++i) ln.p[i] = LnProb{ws[i].part, w[i]->b_out, w[i]->ln1_g, w[i]->ln1_b, x[i], ws[i].cat};
        AHV_ENC_LAUNCH((ln_kernel<true, 4>), dim3((M + 3) / 4, 2), dim3(256), 0, s, ln);
        AHV_TRY(hipGetLastError(), "norm1 + concat");
    }
    {
        LinSpec sp[2];
        // FF in + GEGLU: writes the gated activations [M][2048] into `part`
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{ws[i].cat, w[i]->w_ff1, ws[i].part, w[i]->b_ff1, 4096};
        // B >= 16, even: 128 x 128 tiles; FF-out (N = 256: only 64 tiles per stream) keeps its 4-way split-K
        const bool tiled = tile_eligible(M, 512, 4096, 2048);
        if (tiled) AHV_TRY(launch_linear_tile(sp, 2, 512, 512, M, 512, 1, 2048, s, 128), "ff in + geglu (tiled)");
        else AHV_TRY(launch_linear(sp, 2, 512, 512, M, 512, 1, 2048, s), "ff in + geglu");
        LnArgs ln;
        ln.M = M;
#ifndef AHV_FFOUT_KS8_MAX_M
#define AHV_FFOUT_KS8_MAX_M 128
#endif
        // FF out on tiles: N = 256 gives only 2 column tiles per stream, so the K split has to fill the chip -- 8 ways up to
        // M = 1024 (B = 16: 256 workgroups instead of 128; forward 1 213 -> 1 117 us), 4 ways from M = 2048 (B = 32: 256
        // workgroups; 8 ways measured 1.3 % SLOWER there: 16 k-steps per workgroup and an 8-slab LayerNorm)
#ifndef AHV_FFOUT_TILED_KS8_MAX_M
#define AHV_FFOUT_TILED_KS8_MAX_M 1024
#endif
        if (tiled && M <= AHV_FFOUT_TILED_KS8_MAX_M) {
            for (int i = 0; i < 2; ++i) sp[i] = LinSpec{ws[i].part, w[i]->w_ff2, ws[i].part + (size_t)M * 2048, nullptr, 256};
            AHV_TRY(launch_linear_tile(sp, 2, 2048, 2048, M, 2048, 8, 0, s, 128), "ff out (tiled, split-K 8)");
            ln.KS = 8;
            for (int i = 0; i < 2; ++i)
                ln.p[i] = LnProb{ws[i].part + (size_t)M * 2048, w[i]->b_ff2, w[i]->ln2_g, w[i]->ln2_b, x[i], out[i]};
            AHV_ENC_LAUNCH((ln_kernel<false, 8>), dim3((M + 3) / 4, 2), dim3(256), 0, s, ln);
        } else if (tiled || M > AHV_FFOUT_KS8_MAX_M) {
            // FF out: 4 K-splits x [M][256] = [M][1024] floats: fits the qkv + kv scratch (1280 per row), free by now
            for (int i = 0; i < 2; ++i) sp[i] = LinSpec{ws[i].part, w[i]->w_ff2, ws[i].qkv, nullptr, 256};
            if (tiled) AHV_TRY(launch_linear_tile(sp, 2, 2048, 2048, M, 2048, 4, 0, s, 128), "ff out (tiled, split-K 4)");
            else AHV_TRY(launch_linear(sp, 2, 2048, 2048, M, 2048, 4, 0, s), "ff out");
            ln.KS = 4;
            for (int i = 0; i < 2; ++i) ln.p[i] = LnProb{ws[i].qkv, w[i]->b_ff2, w[i]->ln2_g, w[i]->ln2_b, x[i], out[i]};
            AHV_ENC_LAUNCH((ln_kernel<false, 4>), dim3((M + 3) / 4, 2), dim3(256), 0, s, ln);
        } else {
            // B <= 2: 8 K-splits so that the 4 MB of W_ff2 stream through 256 workgroups instead of 128 (-3.6 % per
            // forward at B = 1, even at B = 2, slower from B = 4 on); the slabs
            // ([8][M][256]) go behind the gated activations in `part` (2048 of its 4096 floats per row are free)
            for (int i = 0; i < 2; ++i) sp[i] = LinSpec{ws[i].part, w[i]->w_ff2, ws[i].part + (size_t)M * 2048, nullptr, 256};
            AHV_TRY(launch_linear(sp, 2, 2048, 2048, M, 2048, 8, 0, s), "ff out (split-K 8)");
            ln.KS = 8;
            for (int i = 0; i < 2; ++i)
                ln.p[i] = LnProb{ws[i].part + (size_t)M * 2048, w[i]->b_ff2, w[i]->ln2_g, w[i]->ln2_b, x[i], out[i]};
            AHV_ENC_LAUNCH((ln_kernel<false, 8>), dim3((M + 3) / 4, 2), dim3(256), 0, s, ln);
        }
        AHV_TRY(hipGetLastError(), "norm2 + residual");
    }
#undef AHV_TRY
    return 0;
}

int transformer_blocks(const ahv_block_weights* blocks, int depth, float* x_src, float* x_tgt, int B, float* ws,
                       hipStream_t s, const char** what)
{
    const int M = 64 * B;
    const StreamWs w2[2] = {carve(ws, M, 0), carve(ws, M, 1)};
    for (int d = 0; d < depth; ++d) {
        const ahv_block_weights* w = blocks + 4 * d;  // order: attn_self_1, attn_self_2, attn_cross_1, attn_cross_2
        int rc;
        {   // x = self_1(x); ctx = self_2(ctx)
            const ahv_block_weights* const ww[2] = {w + 0, w + 1};
            const float* const xin[2] = {x_src, x_tgt};
            float* const out[2] = {w2[0].tmp, w2[1].tmp};
            if ((rc = run_block_pair(ww, xin, xin, true, out, M, w2, s, what))) return rc;
        }
        {   // x_out = cross_1(x, ctx); ctx_out = cross_2(ctx, x): both read the post-self tensors
            const ahv_block_weights* const ww[2] = {w + 2, w + 3};
            const float* const xin[2] = {w2[0].tmp, w2[1].tmp};
            const float* const cin[2] = {w2[1].tmp, w2[0].tmp};
            float* const out[2] = {x_src, x_tgt};
            if ((rc = run_block_pair(ww, xin, cin, false, out, M, w2, s, what))) return rc;
        }
    }
    return 0;
}

// ---- whole forward_2d3d (modules/modules.py:86-101, inference form) --------------------------------------
static const size_t kRowFloats = 768 + 2304 + 6 * 256 + 128 + 128;  // per token row and stream

size_t forward_2d3d_workspace_floats(int B)
{
    const size_t M = (size_t)64 * (size_t)(B > 0 ? B : 0);
    return 2 * M * kRowFloats + transformer_workspace_floats(B) + 64;
}

struct EncWs {
    float *T0, *S, *E0, *H1, *Xin, *Gn, *Xs, *Vol0, *H3, *Skip;
};

static EncWs carve_enc(float* ws, size_t M, int which)
{
    float* b = ws + (size_t)which * M * kRowFloats;
    EncWs w;
    w.T0 = b;                 b += M * 768;
    w.S = b;                  b += M * 2304;
    w.E0 = b;                 b += M * 256;
    w.H1 = b;                 b += M * 256;
    w.Xin = b;                b += M * 256;
    w.Gn = b;                 b += M * 256;
    w.Xs = b;                 b += M * 256;
    w.Vol0 = b;               b += M * 256;
    w.H3 = b;                 b += M * 128;
    w.Skip = b;
    return w;
}

static hipError_t launch_finish(const float* const P[2], const float* const bias[2], const float* const res[2],
                                float* const out[2], const float* pe, int KS, int M, int N, int ldp, int col0,
                                int relu_cols, int ldres, int ldo, int layout, hipStream_t s,
                                float* const out2[2] = nullptr)
{
    FinArgs a;
    for (int i = 0; i < 2; ++i)
        a.p[i] = FinProb{P[i], bias ? bias[i] : nullptr, res ? res[i] : nullptr, out[i], out2 ? out2[i] : nullptr};
    a.pe = pe; a.KS = KS; a.M = M; a.N = N; a.ldp = ldp; a.col0 = col0;
    a.relu_cols = relu_cols; a.ldres = ldres; a.ldo = ldo; a.layout = layout;
    const long n = (long)M * (N / 4);
    const dim3 grid((unsigned)((n + 255) / 256), 2);
    if (KS == 1) AHV_ENC_LAUNCH(finish_kernel<1>, grid, dim3(256), 0, s, a);
    else if (KS == 2) AHV_ENC_LAUNCH(finish_kernel<2>, grid, dim3(256), 0, s, a);
    else if (KS == 3) AHV_ENC_LAUNCH(finish_kernel<3>, grid, dim3(256), 0, s, a);
    else if (KS == 4) AHV_ENC_LAUNCH(finish_kernel<4>, grid, dim3(256), 0, s, a);
    else if (KS == 9) AHV_ENC_LAUNCH(finish_kernel<9>, grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

int forward_2d3d(const ahv_aligner_weights* w, const float* l4_src, const float* l4_tgt, int B, float* ws,
                 float* vol_src, float* vol_tgt, hipStream_t s, const char** what)
{
    const int M = 64 * B, MV = 512 * B;
    const EncWs e[2] = {carve_enc(ws, M, 0), carve_enc(ws, M, 1)};
    float* tws = ws + 2 * (size_t)M * kRowFloats;
    float* zero_row = ws + forward_2d3d_workspace_floats(B) - 64;   // the workspace's last 256 bytes
    // many rows (B >= 16): the two 3 x 3 convolutions on the tile kernel, K split so that 512 workgroups fill the chip
    const int conv_ks = (M >= 1024 && (M & 63) == 0) ? (M >= 4096 ? 1 : 4096 / M) : 0;   // 4 (B = 16), 2 (B = 32), 1
    hipError_t err;
#define AHV_TRY(call, name) do { err = (call); if (err != hipSuccess) { *what = name; return (int)err; } } while (0)
#define PAIR(T, name, a0, a1) T const name[2] = {a0, a1}
    {   // (B,768,8,8) -> tokens [M][768]
        Nchw2TokArgs a;
        a.in[0] = l4_src; a.in[1] = l4_tgt; a.out[0] = e[0].T0; a.out[1] = e[1].T0; a.C = 768; a.B = B;
        a.zero = zero_row;
        AHV_ENC_LAUNCH(nchw_to_tokens_kernel, dim3(768 / 64, B, 2), dim3(256), 0, s, a);
        AHV_TRY(hipGetLastError(), "layout");
    }
    PAIR(const float*, S, e[0].S, e[1].S);
    PAIR(float*, E0, e[0].E0, e[1].E0);
    PAIR(float*, H1, e[0].H1, e[1].H1);
    PAIR(float*, Xin, e[0].Xin, e[1].Xin);
    PAIR(float*, Vol0, e[0].Vol0, e[1].Vol0);
    PAIR(float*, H3, e[0].H3, e[1].H3);
    PAIR(float*, Skip, e[0].Skip, e[1].Skip);
    {   // feature_embedding: conv1x1 768->256, then res-block (conv3x3, relu, conv3x3, + skip), then + pos-emb
        LinSpec sp[2];
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].T0, w->w_emb, e[i].S, nullptr, 256};
        AHV_TRY(launch_linear(sp, 2, 768, 768, M, 768, 3, 0, s), "embedding conv1x1");
        AHV_TRY(launch_finish(S, nullptr, nullptr, E0, nullptr, 3, M, 256, 256, 0, 0, 0, 256, 0, s), "embedding sum");
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].E0, w->w_conv1, e[i].S, nullptr, 256};
        const int cks = conv_ks > 0 ? conv_ks : 9;
        if (conv_ks > 0) AHV_TRY(launch_linear_tile(sp, 2, 256, 2304, M, 2304, conv_ks, 0, s, 64, zero_row), "res-block conv1 (tiles)");
        else AHV_TRY(launch_linear(sp, 2, 256, 2304, M, 2304, 9, 0, s, 1), "res-block conv1");
        AHV_TRY(launch_finish(S, nullptr, nullptr, H1, nullptr, cks, M, 256, 256, 0, 256, 0, 256, 0, s), "res-block relu");
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].H1, w->w_conv2, e[i].S, nullptr, 256};
        if (conv_ks > 0) AHV_TRY(launch_linear_tile(sp, 2, 256, 2304, M, 2304, conv_ks, 0, s, 64, zero_row), "res-block conv2 (tiles)");
        else AHV_TRY(launch_linear(sp, 2, 256, 2304, M, 2304, 9, 0, s, 1), "res-block conv2");
        PAIR(const float*, resE0, e[0].E0, e[1].E0);
        AHV_TRY(launch_finish(S, nullptr, resE0, Xin, w->posemb, cks, M, 256, 256, 0, 0, 256, 256, 0, s), "res-block skip + posemb");
    }
    {   // BidirectionTransformer: shared GroupNorm, per-stream proj_in, blocks, per-stream proj_out + residual
        GnArgs g;
        g.x[0] = e[0].Xin; g.x[1] = e[1].Xin; g.y[0] = e[0].Gn; g.y[1] = e[1].Gn; g.g = w->gn_g; g.be = w->gn_b; g.B = B;
        AHV_ENC_LAUNCH(groupnorm_kernel, dim3(8, B, 2), dim3(256), 0, s, g);
        AHV_TRY(hipGetLastError(), "groupnorm");
        LinSpec sp[2];
        // proj_in with its bias added in the linear's epilogue (one K slice): straight into the token buffer
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].Gn, w->w_in[i], e[i].Xs, w->b_in[i], 256};
        AHV_TRY(launch_linear(sp, 2, 256, 256, M, 256, 1, 0, s), "proj_in + bias");
        const int rc = transformer_blocks(w->blocks, w->depth, e[0].Xs, e[1].Xs, B, tws, s, what);
        if (rc) return rc;
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].Xs, w->w_out[i], e[i].S, nullptr, 256};
        AHV_TRY(launch_linear(sp, 2, 256, 256, M, 256, 1, 0, s), "proj_out");
        PAIR(const float*, bout, w->b_out[0], w->b_out[1]);
        PAIR(const float*, resX, e[0].Xin, e[1].Xin);
        // + bias + x_in, written as the channel-last volume [B][8][8][8][32] (channel 256 = c' * 8 + d)
        AHV_TRY(launch_finish(S, bout, resX, Vol0, nullptr, 1, M, 256, 256, 0, 0, 256, 0, 1, s), "proj_out + residual");
    }
    {   // feature_embedding_3d: conv3d 32->16, relu, conv3d 16->16, + 1x1x1 skip (folded into conv1 as columns 16..31)
        LinSpec sp[2];
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].Vol0, w->w3d_1, e[i].S, nullptr, 32};
        AHV_TRY(launch_linear(sp, 2, 32, 1024, MV, 1024, 4, 0, s, 2), "conv3d 1 + skip");
        // columns 0..15 = conv1 (ReLU) -> H3, columns 16..31 = the 1x1x1 skip -> Skip: one pass over the 4 slabs
        AHV_TRY(launch_finish(S, nullptr, nullptr, H3, nullptr, 4, MV, 32, 32, 0, 16, 0, 0, 3, s, Skip), "conv3d relu | skip");
        PAIR(float*, S2, e[0].S + (size_t)MV * 32 * 4, e[1].S + (size_t)MV * 32 * 4);
        for (int i = 0; i < 2; ++i) sp[i] = LinSpec{e[i].H3, w->w3d_2, S2[i], nullptr, 16};
        AHV_TRY(launch_linear(sp, 2, 16, 512, MV, 512, 4, 0, s, 2), "conv3d 2");
        PAIR(const float*, S2c, S2[0], S2[1]);
        PAIR(const float*, resS, e[0].Skip, e[1].Skip);
        PAIR(float*, vout, vol_src, vol_tgt);
        AHV_TRY(launch_finish(S2c, nullptr, resS, vout, nullptr, 4, MV, 16, 16, 0, 0, 16, 0, 2, s), "volume");
    }
#undef PAIR
#undef AHV_TRY
    return 0;
}

}  // namespace ahv
