// ahv_split.h -- GEMM1 of the fused scorer on the f16 matrix pipe with split operands (OPT-IN per call through
// AHV_SCORE_SPLIT_F16: the default stays the all-fp32 dual kernel).
//
// Why: fp32 MFMA tops out at 157 TFLOP/s and shares its issue slots with VALU work (ahv_dual.h), so the
// fp32 kernel sits at ~80 % of a roofline that is itself 16x below the f16 matrix pipe.  Each fp32
// operand x is written as hi + lo with hi = x truncated to 11 significant bits and lo = x - hi rounded
// to f16; the three products hi*hi + hi*lo + lo*hi are accumulated in fp32 by v_mfma_f32_16x16x32_f16
// (hi*hi is exact in fp32, the dropped lo*lo term is 2^-22 relative).  Measured against the fp32
// oracle the scores agree to ~1e-6, two orders inside the 1e-4 parity bar (tests/test_gpu_split.py), but
// the arithmetic is NOT IEEE fp32 -- hence opt-in.  Operands are prescaled by powers of two taken from their
// largest magnitude (split_prescale_exp), so any finite fp32 range works.
//
// K order inside a slab is (spatial index, channel) with the channel fastest: a lane's own 16 blended
// channels are then two complete B fragments, the gather stores them with four ds_write_b128 and ONE
// image serves all three slab views with plain ds_read_b128.
//
// v_mfma_f32_16x16x32_f16: A[row = lane&15][k = 8*(lane>>4) + j], B[k = 8*(lane>>4) + j][col = lane&15],
// D[row = 4*(lane>>4) + reg][col = lane&15]  (j = 0..7, the eight halfs of the operand register quad).
#pragma once
#include "ahv_device.h"
#include "ahv_dual.h"

namespace ahv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Both operands are prescaled by a power of two chosen from their largest magnitude (per launch for W1,
// per sample for the source volume) so that max |x| * 2^e lies in [2^13, 2^14): hi never overflows f16 and
// lo stays a normal f16 for everything within 2^-13 of the maximum.  The exact 2^-(eW+eV) comes back out
// through the GEMM2 A operand.
__device__ __forceinline__ int split_prescale_exp(float amax)
{
    if (!(amax > 0.0f) || amax > 3.0e38f) return 0;  // all-zero, NaN or inf operands: nothing to gain
    int ex;
    (void)frexpf(amax, &ex);  // amax = m * 2^ex, m in [0.5, 1)
    const int e = 14 - ex;
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}

// max over the workgroup; `scratch` = 8 floats of LDS nobody else is using right now
__device__ __forceinline__ float block_absmax(float x, float* scratch, int tid)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) x = fmaxf(x, __shfl_xor(x, sft, 64));
    if ((tid & 63) == 0) scratch[tid >> 6] = x;
    __syncthreads();
    float m = scratch[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, scratch[i]);
    __syncthreads();
    return m;
}
constexpr int kSplitTableFrags = 48;       // 12 groups (slab, k-step) x {m0 hi, m0 lo, m1 hi, m1 lo}
constexpr int kSplitTableBytes = kSplitTableFrags * 64 * 16;  // 48 KiB
constexpr int kSplitImageBytes = 128 * 64;  // per wave: 128 voxels x (16 hi + 16 lo halfs)

// ---- W1 fragment table -----------------------------------------------------------------
// Group G = 4*s + ks (s = slab x,y,z; ks = k-step of 32), fragment i = 2*m + part at f16x8 slot
// (4*G + i)*64 + lane.  Lane (row, kq), half j holds
//   W1[16m + row][128 s + c*8 + idx],  idx = 2 ks + (kq >> 1),  c = 8 (kq & 1) + j
// i.e. k = idx*16 + c inside the slab; for the z slab idx = a, so k-step ks = the quarter.
__device__ __forceinline__ void stage_w1_split(f16x8* table, const float* __restrict__ W1, float scale, int tid, int nthreads)
{
    for (int i = tid; i < kSplitTableFrags * 64; i += nthreads) {
        const int lane = i & 63, frag = i >> 6;
        const int G = frag >> 2, m = (frag >> 1) & 1, part = frag & 1;
        const int s = G >> 2, ks = G & 3;
        const int row = lane & 15, kq = lane >> 4;
        const float* w = W1 + (16 * m + row) * 384 + 128 * s + 2 * ks + (kq >> 1);
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = w[(8 * (kq & 1) + j) * 8] * scale;
            const _Float16 hi = (_Float16)x;
            v[j] = part ? (_Float16)(x - (float)hi) : hi;
        }
        table[i] = v;
    }
}

// Source volume -> LDS, prescaled per sample; returns the exponent used (512 threads, 16 values each).
__device__ __forceinline__ int stage_src_volume_scaled(float* srcT, const float* __restrict__ vol, float* scratch, int tid)
{
    float x[16];
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        x[k] = vol[tid + 512 * k];
        m = fmaxf(m, fabsf(x[k]));
    }
    const int e = split_prescale_exp(block_absmax(m, scratch, tid));
    const float scale = ldexpf(1.0f, e);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = tid + 512 * k, c = i >> 9, v = i & 511;
        srcT[((v >> 6) * kSrcPlaneRows + ((v >> 3) & 7) * kSrcRowsY + (v & 7)) * kSrcStride + c] = x[k] * scale;
    }
    return e;
}

// ---- rotated quarter image ---------------------------------------------------------------
// Voxel (a0, b, e) of the quarter -> 64-byte row a0 + 2e + 16b = [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]; the 16-byte
// slot of a chunk inside its 256-byte LDS line is address bits 4-7 = (chunk, row & 3), and those four bits are XORed
// with four parities of the voxel's bits v = a0 | (b & 3) << 1 | e << 4:
//     bit 4 ^= a0      bit 5 ^= b1 ^ e1 ^ e2      bit 6 ^= e0      bit 7 ^= b0
// (b >> 2 is the gather's pass and stays out, so that pass 1 remains "+ 4096 bytes"; bit 5 moves with the chunk only, so
// that lo = chunk + 2 remains "address ^ 32"; bits 6-7 are XORed with functions of OTHER bits, so the map stays one to
// one).  What it has to satisfy (MI355X_MICROARCH.md, LDS): a ds_read_b128 is served in four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32), one cycle each if the 16 lanes hit 16 different slots of the 256-byte
// line; a ds_write_b128 in eight groups of 8 CONTIGUOUS lanes over 128 bytes (slot = address bits 4-6).  Round 3's
// layout XORed the chunk bits only: its stores were 2-way conflicts (256 extra cycles per hypothesis) and so were the
// x- and z-slab fragment reads (128 each) -- 512 of the 1 167 conflict cycles SQ_LDS_BANK_CONFLICT counted per
// hypothesis, the other 655 being the gather's.  tools/split_image_sim.py models both group shapes (it reproduces
// the counter for the old layout, and for a first attempt that wrongly used the read groups for the stores: 768 + 655
// = the 1 423 measured) and searches the GF(2)-linear swizzles: 61 440 of them are conflict-free for the stores of both
// passes, plain and mirrored, and for all three slab reads; this is one of the cheapest (the compiler hoists every
// per-lane base address out of the hypothesis loop: no instruction added, + 7 registers).  Measured, one box, binaries
// alternating: conflict cycles 1 167 -> 655 per hypothesis (0.339 -> 0.223 of the LDS cycles), 0.3770 -> 0.3688 ms
// (profiles/r04m_split_swizzle_ab.txt).
constexpr int kSwzChunk0 = 0x01, kSwzChunk1 = 0x64, kSwzRow0 = 0x10, kSwzRow1 = 0x02;  // masks over v
__device__ __forceinline__ int split_addr(int a0, int b, int e, int chunk)
{
    const int v = a0 | ((b & 3) << 1) | (e << 4);
    const int x = (__builtin_popcount(v & kSwzChunk0) & 1) | ((__builtin_popcount(v & kSwzChunk1) & 1) << 1) |
                  ((__builtin_popcount(v & kSwzRow0) & 1) << 2) | ((__builtin_popcount(v & kSwzRow1) & 1) << 3);
    return ((a0 + 2 * e + 16 * b) * 64 + (chunk << 4)) ^ (x << 4);
}

__device__ __forceinline__ unsigned pk_rtz(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// The gather is the fp32 kernel's (ahv_dual.h: hat weights on a clamped base row, one base address per voxel, a
// request ring six rows deep, the 4 x 2 x 2-box lane map); only the store differs: the sixteen blended channels are
// split into hi (the top 11 significant bits, exactly an f16 in the normal range) and lo (the rest) and leave as
// 16-byte stores, conflict-free under the image's swizzle (split_addr above).
struct SplitDst {
    int chunk[4];    // byte offsets of [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] of the lane's pass-0 voxel (pass 1: + 4096)
    int chunk_m[4];  // the same for the lane's voxel in a MIRRORED quarter, (7 - x, 7 - y, 1 - z) (hat_mirror, ahv_dual.h)
};

__device__ __forceinline__ SplitDst split_dst(int lane)
{
    const LaneVox lv = lane_vox(lane);
    SplitDst d;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        d.chunk[c] = split_addr(lv.a0, lv.bq, lv.e, c);  // b = 4 p + bq: the pass keeps b & 1
        d.chunk_m[c] = split_addr(1 - lv.a0, 3 - lv.bq, 7 - lv.e, c);
    }
    asm volatile("" : "+v"(d.chunk[0]), "+v"(d.chunk[1]), "+v"(d.chunk[2]), "+v"(d.chunk[3]));
    asm volatile("" : "+v"(d.chunk_m[0]), "+v"(d.chunk_m[1]), "+v"(d.chunk_m[2]), "+v"(d.chunk_m[3]));
    return d;
}

template <bool MIR>
struct HatStoreSplit {
    static constexpr bool kXdlKernel = true;  // GEMM1 runs on XDL MFMAs here: broadcast scalars stay in low halves (low_half, ahv_dual.h)
    char* img;
    SplitDst d;
    __device__ __forceinline__ void operator()(int p, const f32x2 (&o)[8]) const
    {
        // eight channels at a time, each half stored before the next is converted: the request ring of the other pass
        // is live here and sixteen conversion temporaries on top of it do not fit in 256 registers
        char* dst = img + p * (64 * 64);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // scalars first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this
                // hipcc (ROCm 7.2) whatever the index
                const float x0 = o[4 * half + i][0], x1 = o[4 * half + i][1];
                const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x0) & 0xFFFFE000u);
                const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x1) & 0xFFFFE000u);
                hi[i] = pk_rtz(h0, h1);
                lo[i] = pk_rtz(x0 - h0, x1 - h1);
            }
            *reinterpret_cast<u32x4*>(dst + (MIR ? d.chunk_m[half] : d.chunk[half])) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(dst + (MIR ? d.chunk_m[2 + half] : d.chunk[2 + half])) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// the 16 blend steps of a quarter whose prologue (hat_prologue, ahv_dual.h) has been issued
template <bool MIR = false>
__device__ __forceinline__ void hat_body_split(HatState& st, char* img, const SplitDst& dst)
{
    f32x2 o[8];
    const HatStoreSplit<MIR> store = {img, dst};
    __builtin_amdgcn_s_setprio(1);
    HatSteps<0, HatStoreSplit<MIR>, MIR>::run(st, o, store);
    __builtin_amdgcn_s_setprio(0);
}

// ---- GEMM1 on one quarter: 72 x v_mfma_f32_16x16x32_f16 -----------------------------------------
#define AHV_MFMA_F16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0)

// Operands of one k-step.  Steps 0-3 = x slab (ks), 4-7 = y slab (ks), 8-11 = z slab (tile t).
struct SplitStep {
    f16x8 bh, bl;    // B fragment pair of the step
    f16x8 a[4];      // {m0 hi, m0 lo, m1 hi, m1 lo}; the z steps share one quad
};

template <int Q, int S>
__device__ __forceinline__ void split_load(SplitStep& o, const f16x8* T, const char* img, int i0, int j, int kh, int kc)
{
    constexpr int ks = S & 3;
    int off;
    if (S < 4) off = split_addr(i0, j, 2 * ks + kh, kc);        // position (a0 = i0, b = j), k = (e, c)
    else if (S < 8) off = split_addr(i0, 2 * ks + kh, j, kc);   // position (a0 = i0, e = j), k = (b, c)
    else off = split_addr(kh, 2 * ks + i0, j, kc);              // position (b = 2t + i0, e = j), k = (a0, c)
    o.bh = *reinterpret_cast<const f16x8*>(img + off);
    o.bl = *reinterpret_cast<const f16x8*>(img + (off ^ 32));   // chunk + 2: the lo halves
    if (S <= 8) {
        constexpr int G = S < 4 ? ks : (S < 8 ? 4 + ks : 8 + Q);
#pragma unroll
        for (int i = 0; i < 4; ++i) o.a[i] = T[(4 * G + i) * 64];
    }
}

// Stress knob of tools/first_launch.cpp, never set in the product: AHV_DIAG_MFMA_GAP="s_nop 7" leaves the matrix pipe idle
// between any two MFMAs, which is what exposes the packed-fp32 op_sel hazard (low_half, ahv_dual.h) on every hypothesis
// instead of on the few that meet an instruction-fetch stall in a process' first launch.
#ifdef AHV_DIAG_MFMA_GAP
#define AHV_MFMA_GAP __builtin_amdgcn_sched_barrier(0); asm volatile(AHV_DIAG_MFMA_GAP ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#else
#define AHV_MFMA_GAP
#endif
template <int Q, int S>
__device__ __forceinline__ void split_mfma(f32x4 (&acc)[2][4], const SplitStep& o, const f16x8 (&az)[4])
{
    constexpr int t = S < 8 ? Q : (S & 3);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f16x8 ah = S < 8 ? o.a[2 * m] : az[2 * m], al = S < 8 ? o.a[2 * m + 1] : az[2 * m + 1];
        acc[m][t] = AHV_MFMA_F16(al, o.bh, acc[m][t]);
        AHV_MFMA_GAP
        acc[m][t] = AHV_MFMA_F16(ah, o.bl, acc[m][t]);
        AHV_MFMA_GAP
        acc[m][t] = AHV_MFMA_F16(ah, o.bh, acc[m][t]);
        AHV_MFMA_GAP
    }
}

// Twelve k-steps, software-pipelined by hand one step deep: the operands of step S+1 are requested before
// the six MFMAs of step S are issued, and sched_barrier keeps the compiler from hoisting every load of the
// quarter to the top (which costs ~100 registers and spills).
// `hook` runs once, ahead of the MFMAs of step 8 (the next quarter's gather prologue, as in gemm1_quarter_pipe).
template <int Q, typename Hook>
__device__ __forceinline__ void gemm1_quarter_split(f32x4 (&acc)[2][4], const f16x8* table, const char* img, int lane,
                                                    Hook hook)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7, kh = kq >> 1, kc = kq & 1;
    const f16x8* T = table + lane;
    SplitStep s0, s1;
    f16x8 az[4];
#define AHV_SPLIT_PAIR(S)                                            \
    split_load<Q, S + 1>(s1, T, img, i0, j, kh, kc);                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    split_mfma<Q, S>(acc, s0, az);                                   \
    __builtin_amdgcn_sched_barrier(0);                               \
    if (S + 2 < 12) split_load<Q, (S + 2 < 12 ? S + 2 : 0)>(s0, T, img, i0, j, kh, kc); \
    if (S + 2 == 8) { az[0] = s0.a[0]; az[1] = s0.a[1]; az[2] = s0.a[2]; az[3] = s0.a[3]; } \
    if (S == 8) hook();                                              \
    __builtin_amdgcn_sched_barrier(0);                               \
    split_mfma<Q, S + 1>(acc, s1, az);                               \
    __builtin_amdgcn_sched_barrier(0);
    split_load<Q, 0>(s0, T, img, i0, j, kh, kc);
    AHV_SPLIT_PAIR(0)
    AHV_SPLIT_PAIR(2)
    AHV_SPLIT_PAIR(4)
    AHV_SPLIT_PAIR(6)
    AHV_SPLIT_PAIR(8)
    AHV_SPLIT_PAIR(10)
#undef AHV_SPLIT_PAIR
}

}  // namespace ahv
