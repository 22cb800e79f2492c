// ahv_dual.h -- two-waves-per-SIMD formulation of the fused scorer (score_hypotheses_dual_kernel, the library's
// fp32 kernel) and the pieces the backward and forward_3d2d share with it.
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
// LDS (158 KiB): source image 46 KiB (80-byte voxel rows, y rows 640 B and z planes 5 920 B apart; the pads hold
// the target fragments) + W1 fragment table 48 KiB + 8 x 8 KiB quarter images.
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

// Table slot of W1[o][k] (the inverse of w1_table_entry).
__device__ __forceinline__ int w1_table_slot(int o, int k)
{
    const int m = o >> 4, row = o & 15, slab = k >> 7, kk = k & 127;
    if (slab < 2) {
        const int c = kk >> 3, hh = (kk >> 2) & 1, kq = kk & 3;
        return ((16 * slab + c) * 64 + 16 * kq + row) * 4 + 2 * hh + m;
    }
    const int cc = kk >> 3, d = kk & 7;  // channel, depth
    const int q = d >> 1, kq = 2 * (cc & 1) + (d & 1), cp = cc >> 1;
    return ((32 + 4 * q + (cp >> 1)) * 64 + 16 * kq + row) * 4 + 2 * (cp & 1) + m;
}

// W1 [32][384] -> fragment table.  Read as 16-byte rows (six per thread of a 512-thread workgroup, all in flight at
// once) and scattered into LDS; the gather form (one 4-byte load per table entry, 24 dependent-address iterations
// per thread) took most of the kernel's 7-us prologue.
__device__ __forceinline__ void stage_w1_table(float* table, const float* __restrict__ W1, int tid, int nthreads)
{
    if ((reinterpret_cast<unsigned long long>(W1) & 15ull) == 0) {
        for (int i = tid; i < 32 * 96; i += nthreads) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W1 + 4 * i);
            const int o = i / 96, k = 4 * (i - 96 * o);
#pragma unroll
            for (int e = 0; e < 4; ++e) table[w1_table_slot(o, k + e)] = w[e];
        }
    } else {  // a view that starts off a 16-byte boundary: same map, one float at a time
        for (int i = tid; i < 32 * 384; i += nthreads) table[w1_table_slot(i / 384, i % 384)] = W1[i];
    }
}

// The same in two halves, so that a kernel can issue the six loads ahead of other work and scatter them later
// (16-byte-aligned W1 only).
struct W1Regs {
    f32x4 w[6];
};

__device__ __forceinline__ void w1_regs_load(W1Regs& r, const float* __restrict__ W1, int tid)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) r.w[k] = *reinterpret_cast<const f32x4*>(W1 + 4 * (tid + 512 * k));
}

__device__ __forceinline__ void w1_regs_store(float* table, const W1Regs& r, int tid)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int i = tid + 512 * k, o = i / 96, kk = 4 * (i - 96 * o);
#pragma unroll
        for (int e = 0; e < 4; ++e) table[w1_table_slot(o, kk + e)] = r.w[k][e];
    }
}

// Source volume of one sample, split the same way: thread v < 512 owns voxel v -- its 16 channels are 16 coalesced
// 4-byte loads (one per channel plane) and ONE 64-byte row of the channel-last image (four ds_write_b128).
struct SrcRegs {
    float x[16];
};

__device__ __forceinline__ void src_regs_load(SrcRegs& r, const float* __restrict__ vol, int tid)
{
#pragma unroll
    for (int c = 0; c < 16; ++c) r.x[c] = vol[c * 512 + tid];
}

__device__ __forceinline__ void src_regs_store(float* srcT, const SrcRegs& r, int tid)
{
    f32x4* row = reinterpret_cast<f32x4*>(srcT + ((tid >> 6) * kSrcPlaneRows + ((tid >> 3) & 7) * kSrcRowsY + (tid & 7)) * kSrcStride);
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = f32x4{r.x[4 * j], r.x[4 * j + 1], r.x[4 * j + 2], r.x[4 * j + 3]};
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

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ReLU in ONE instruction per register: a signed-integer max of the float's bits with 0 (negative floats, -0
// included, are negative integers; non-negative floats order like their bits).  fmaxf / fmed3 compile to v_max(x, x)
// (canonicalise) + v_max(0, x) because the compiler cannot know that an MFMA result needs no canonicalisation, and
// every vector instruction costs ~4 issue cycles beside the partner wave's MFMA stream (32 ReLUs per hypothesis).
// Same value for every non-NaN input; a NaN with a clear sign bit stays a NaN (torch's relu propagates it too).
// Not inline asm: the compiler does not pad the MFMA -> VALU read hazard in front of an instruction it cannot see
// into (the split-f16 kernel read stale accumulators that way).
__device__ __forceinline__ f32x4 relu4(f32x4 x)
{
    return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, x), i32x4{0, 0, 0, 0}));
}

__device__ __forceinline__ void gemm2_dual(f32x4 (&v)[2][4], const f32x4 (&acc)[2][4], const DualFrags& f)
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
        for (int t = 0; t < 4; ++t) u[t] = relu4(acc[m][t]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][0], u[t][r], v[0][t], 0, 0, 0);
                v[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][1], u[t][r], v[1][t], 0, 0, 0);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Gather, second formulation (fused fp32 scorer).  Same sampling semantics as tri_coef_ptr / tri_blend_ptr
// (F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=False), utils.py:123-129), restated so that the
// per-voxel VALU work outside the blend drops from ~100 to ~40 instructions (fp32 MFMA and VALU share the
// issue slots on gfx950, so every instruction removed here is time returned to the matrix pipe):
//
//  * affine coordinates: the sample coordinate along axis A is  i_A = 4 g_A + 3.5,  g = R p, and a lane's eight
//    voxels differ only by p.y += 1 (pass) and p.z += 1/2 (quarter), so  i_A(pass, Q) = i_A(0,0) + pass * 4 R[A][1]
//    + Q * 2 R[A][2]:  nine FMAs per HYPOTHESIS plus three per voxel instead of nine + six per voxel;
//  * "hat" weights on a base clamped to [0, 6]: with j = clamp(floor(i), 0, 6) the two neighbours j, j+1 are
//    ALWAYS inside the volume and their weights are the hat function  max(0, 1 - |i - j|), max(0, 1 - |i - j - 1|)
//    -- exactly grid_sample's weights where a neighbour is inside, and exactly 0 for what zeros-padding drops
//    (i in [-1,0): j = 0 gets 1 + i, j+1 gets 0;  i in [7,8): j = 6 gets 0, j+1 = 7 gets 8 - i;  beyond: both 0).
//    No range compares, no selects, no index clamps;  max(0, .) is the free VOP3 clamp modifier;
//  * one base address: because j+1 is always valid the eight corner rows sit at CONSTANT byte offsets
//    {0,80} + {0,640} + {0,5920} from the row of (jz, jy, jx) -- one address register and immediate offsets
//    instead of eight computed pointers;
//  * the blend runs on v_pk_fma_f32 (two channels per lane and instruction, the corner weight broadcast
//    through op_sel): half the blend instructions, same FMA order per channel, bit-identical sums.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Wave priority: s_setprio 1 during the gather (hat_body).  The two waves of a SIMD run the same program with no
// barrier; the one that is gathering is latency-bound (LDS round trips, short VALU bursts) while its partner streams
// MFMAs, so letting the gathering wave issue FIRST whenever it is ready shortens its critical path at no cost to the
// matrix pipe: measured -3.4 % kernel time (none / gather / GEMM-high / all-but-GEMM high: 0.743 / 0.718 / 0.734 /
// 0.727 ms per 50 000 hypotheses, round 2).

// Lane -> output voxel of a gather pass.  The 16 lanes that the LDS serves together in a ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) own a compact 4 (x) x 2 (y) x 2 (z) block of the quarter, so that
// after any rotation their source rows form a compact block too (fewest bank conflicts: table in ahv_device.h).
// Group g = 2 (lane >> 5) + (second group of its half-wave), slot k = rank of the lane inside its group:
//   z = 2 Q + (k & 1),  x = 4 (g & 1) + ((k >> 1) & 3),  y = 2 (g >> 1) + (k >> 3) [+ 4 in pass 1].
// z is the FASTEST bit of k so that the contiguous lanes an LDS store is served by hold both z of their voxels: the
// split-f16 kernel's 16-byte image stores (ahv_split.h) were then 2-way instead of 4-way bank conflicts (measured: conflict
// cycles 1 680 -> 1 168 of 3 447 LDS cycles per hypothesis, profiles/r03e_split_pmc_summary.json), and conflict-free
// once the image's swizzle took the 8-lane store groups into account (round 4, split_addr).
// The two groups of a half-wave differ in x bit 2, which keeps the ds_write_b32 of the blended voxels into the
// XOR-swizzled quarter image (qoff) on 32 distinct banks.
struct LaneVox {
    int e, a0, bq;  // x (0..7), z inside the quarter (0..1), y inside the pass (0..3)
};

__device__ __forceinline__ LaneVox lane_vox(int lane)
{
    const unsigned l = lane & 31;
    const unsigned kFirst = 0x0FF0F00Fu;  // lanes {0-3, 12-15, 20-27} of a half-wave: the first b128 group
    const bool first = (kFirst >> l) & 1u;
    const int k = __builtin_popcount((first ? kFirst : ~kFirst) & ((1u << l) - 1u));  // rank inside the group
    LaneVox v;
    v.a0 = k & 1;
    v.e = (first ? 0 : 4) + ((k >> 1) & 3);
    v.bq = 2 * (lane >> 5) + (k >> 3);
    return v;
}

// Offsets (floats) of the lane's two voxels (pass 0, pass 1) inside a channel plane of the destination image,
// computed ONCE per kernel: passed through an empty asm so that the compiler keeps the two registers instead of
// re-deriving the lane map inside the hypothesis loop (it did: 12 compares and 15 exec-mask updates per hypothesis).
struct GatherDst {
    int o0, o1;  // pass 0 / pass 1 voxel of the lane
    int m0, m1;  // pass 0 / pass 1 voxel of the lane in a MIRRORED quarter: (7 - x, 7 - y, 1 - z)
};

// the forward's XOR-swizzled quarter image (qoff)
__device__ __forceinline__ GatherDst gather_dst_swizzled(int lane)
{
    const LaneVox lv = lane_vox(lane);
    GatherDst d = {qoff(lv.a0, lv.bq, lv.e), qoff(lv.a0, 4 + lv.bq, lv.e), qoff(1 - lv.a0, 3 - lv.bq, 7 - lv.e),
                   qoff(1 - lv.a0, 7 - lv.bq, 7 - lv.e)};
    asm volatile("" : "+v"(d.o0), "+v"(d.o1), "+v"(d.m0), "+v"(d.m1));
    return d;
}

// a linear image X[c][voxel = a0 * 64 + b * 8 + e] (backward kernels)
__device__ __forceinline__ GatherDst gather_dst_linear(int lane)
{
    const LaneVox lv = lane_vox(lane);
    GatherDst d = {lv.a0 * 64 + lv.bq * 8 + lv.e, lv.a0 * 64 + (4 + lv.bq) * 8 + lv.e,
                   (1 - lv.a0) * 64 + (3 - lv.bq) * 8 + 7 - lv.e, (1 - lv.a0) * 64 + (7 - lv.bq) * 8 + 7 - lv.e};
    asm volatile("" : "+v"(d.o0), "+v"(d.o1));  // the linear images (backward) do not use the mirrored slots
    return d;
}

struct GatherLane {
    float x4, y4, z4;  // 4 * voxel-centre coordinate of this lane's (pass 0, quarter 0) voxel: (2 i + 1) / 2 - 4
};

__device__ __forceinline__ GatherLane gather_lane(int lane)
{
    const LaneVox v = lane_vox(lane);
    GatherLane g;
    g.x4 = (float)(2 * v.e + 1) * 0.5f - 4.0f;
    g.y4 = (float)(2 * v.bq + 1) * 0.5f - 4.0f;
    g.z4 = (float)(2 * v.a0 + 1) * 0.5f - 4.0f;
    return g;
}

struct GatherHyp {
    f32x2 ixy[2];  // [pass]: (x, y) sample coordinates of the lane's voxel in quarter 0 (a register pair: v_pk_* operands)
    f32x2 dqxy;    // per-quarter increment 2 R[axis][2] of x and y
    f32x2 izp;     // z of pass 0 and pass 1 (a pair too: the two voxels of a lane are set up together)
    float dqz;
};

__device__ __forceinline__ void gather_hyp(GatherHyp& h, const float* Rm, const GatherLane& g)
{
    // x and y as one packed chain (rows 0 and 1 of R), z on its own
    const f32x2 i0xy = __builtin_elementwise_fma(
        f32x2{Rm[0], Rm[3]}, f32x2{g.x4, g.x4},
        __builtin_elementwise_fma(f32x2{Rm[1], Rm[4]}, f32x2{g.y4, g.y4},
                                  __builtin_elementwise_fma(f32x2{Rm[2], Rm[5]}, f32x2{g.z4, g.z4}, f32x2{3.5f, 3.5f})));
    const float i0z = fmaf(Rm[6], g.x4, fmaf(Rm[7], g.y4, fmaf(Rm[8], g.z4, 3.5f)));
    h.ixy[0] = i0xy;
    h.ixy[1] = __builtin_elementwise_fma(f32x2{4.0f, 4.0f}, f32x2{Rm[1], Rm[4]}, i0xy);
    h.dqxy = f32x2{Rm[2], Rm[5]} + f32x2{Rm[2], Rm[5]};
    h.izp = f32x2{i0z, fmaf(4.0f, Rm[7], i0z)};
    h.dqz = Rm[8] + Rm[8];
}

// sample coordinate of the lane's voxel (pass p of quarter Q) along axis a (0 = x, 1 = y, 2 = z)
template <int Q>
__device__ __forceinline__ float gather_coord(const GatherHyp& h, int a, int p)
{
    return a == 2 ? fmaf((float)Q, h.dqz, h.izp[p]) : fmaf((float)Q, h.dqxy[a], h.ixy[p][a]);
}

__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// base index (as a float, 0..6) and the hat weights of rows j, j+1 for sample coordinate i
__device__ __forceinline__ void hat_axis(float i, float& jf, float& w0, float& w1)
{
    jf = __builtin_amdgcn_fmed3f(floorf(i), 0.0f, 6.0f);
    const float u = i - jf;
    w0 = clamp01(1.0f - fabsf(u));
    w1 = clamp01(1.0f - fabsf(u - 1.0f));
}

// A hazard of gfx950 that neither hipcc nor the guides know (found in round 3; profiles/r03_pk_opsel_hazard.txt holds the
// experiments, tools/pk_opsel_hazard.cpp reproduces it in 60 lines without this library).
// A packed-fp32 VALU instruction whose LOW result lane reads the HIGH half of a source pair -- op_sel:[..1..], which is
// what hipcc emits to broadcast the odd element of a register pair, e.g. v_pk_fma_f32 o, row, w[2:3] op_sel:[0,1,0] --
// can lose that operand in lanes 48-63 (the product comes out 0) when an XDL MFMA (v_mfma_f32_16x16x32_f16) of EITHER wave
// of the SIMD starts on a matrix pipe that has been idle for ~24 cycles or more: an instruction-fetch stall, a branch or an
// s_nop between two MFMAs is enough.  It is the op_sel bit of SRC1 alone (v_pk_mul / v_pk_fma / v_pk_add alike): cross-half
// reads of src0 or src2, the high lane reading a low half (op_sel_hi:[..0..]) and the straight forms are not affected, XDL
// MFMAs on the other SIMDs of the CU do not matter, and neither does a kernel whose MFMAs are all fp32 (v_mfma_f32_16x16x4_f32 does not overlap VALU work at all: bit-identical
// results with 8..64-cycle gaps forced between its MFMAs).  In the split-f16 scorer this showed up as ~1 % of the first
// hypotheses of a process' first launch being 1e-3 off -- the cold instruction cache supplies the gaps -- and with gaps forced
// between the MFMAs as 80 % of all scores wrong.
// A kernel that issues XDL MFMAs therefore keeps every scalar it broadcasts over a register pair in the LOW half: low_half(x)
// hides x from hipcc, which then has to hold it in a register of its own and broadcasts it with op_sel_hi:[..0..].
// Kernels that issue fp32 MFMAs alone are not exposed by THEMSELVES (an fp32 MFMA never overlaps VALU work), only to an XDL
// wave of another kernel sharing their SIMD.  Rounds 3-4 ruled that out for the scorers by occupancy -- two waves of 249-256
// registers take all 512 of a SIMD (tests/test_isa_hazard.py still checks the allocation) -- but the waves of a workgroup
// retire one by one, and beside the LAST wave of a SIMD a foreign wave fits.  Since round 5 every kernel that uses this
// gather defines AHV_FP32_LOW_HALF (ahv_score.hip, ahv_backward.hip): measured cost 0.2-0.7 % (tools/kbench_lowhalf,
// kbench_bwd_lowhalf: 0.6809 -> 0.6828 ms per 50 000 hypotheses, 2.576 -> 2.588 ms per backward; profiles/r04d_low_half_ab.txt).
#if defined(AHV_FP32_LOW_HALF) || defined(AHV_DIAG_FP32_LOW_HALF)
constexpr bool kFp32LowHalf = true;
#else
constexpr bool kFp32LowHalf = false;
#endif

__device__ __forceinline__ float low_half(float x)
{
#ifndef AHV_DIAG_NO_LOW_HALF  // tools/first_launch_sweep.sh builds the unprotected kernel once, to show the hazard itself
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// Weights and base row of one voxel (pass p of quarter Q).
struct HatVoxel {
    float w[8];        // corner weights, order (dz, dy, dx)
    const char* base;  // row (jz, jy, jx) of the source image; corner n sits at base + kHatOff(n)
};

__device__ __forceinline__ constexpr int hat_off(int n)
{
    return ((n & 1) ? 4 * kSrcStride : 0) + ((n & 2) ? 4 * kSrcRowsY * kSrcStride : 0) + ((n & 4) ? 4 * kSrcPlaneRows * kSrcStride : 0);
}

// Weights and base row from the sample coordinates (x, y) and z.  XDL: the kernel issues XDL MFMAs (see low_half).
template <bool XDL>
__device__ __forceinline__ void hat_voxel_at(HatVoxel& v, const float* srcT, f32x2 ixy, float iz)
{
    float jx, jy, jz, wx0, wx1, wy0, wy1, wz0, wz1;
    // x and y ride on packed instructions where one exists (coordinate, offset from the base row, offset - 1);
    // floor, med3 and the |.|-with-clamp subtraction have no packed form
    jx = __builtin_amdgcn_fmed3f(floorf(ixy[0]), 0.0f, 6.0f);
    jy = __builtin_amdgcn_fmed3f(floorf(ixy[1]), 0.0f, 6.0f);
    const f32x2 uxy = ixy - f32x2{jx, jy};
    const f32x2 txy = uxy - f32x2{1.0f, 1.0f};
    wx0 = clamp01(1.0f - fabsf(uxy[0]));
    wy0 = clamp01(1.0f - fabsf(uxy[1]));
    wx1 = clamp01(1.0f - fabsf(txy[0]));
    wy1 = clamp01(1.0f - fabsf(txy[1]));
    hat_axis(iz, jz, wz0, wz1);
    // the outer product of the three weight pairs on v_pk_mul_f32 (6 instead of 12 multiplications, same products;
    // 0.7081 -> 0.7033 ms per 50 000 hypotheses against scalar multiplications, round 3)
    const f32x2 wy = {wy0, wy1}, wx = {wx0, wx1};
    const f32x2 w0y = wz0 * wy, w1y = wz1 * wy;
    const float s0 = XDL ? low_half(w0y[1]) : w0y[1], s1 = XDL ? low_half(w1y[1]) : w1y[1];
    const f32x2 a = w0y[0] * wx, b = s0 * wx, c = w1y[0] * wx, d = s1 * wx;
    v.w[0] = a[0]; v.w[1] = a[1]; v.w[2] = b[0]; v.w[3] = b[1];
    v.w[4] = c[0]; v.w[5] = c[1]; v.w[6] = d[0]; v.w[7] = d[1];
    // byte offset of row (jz, jy, jx): exact in fp32 (< 2^24), one conversion
    const float af = fmaf(jz, (float)(4 * kSrcPlaneRows * kSrcStride),
                          fmaf(jy, (float)(4 * kSrcRowsY * kSrcStride), jx * (float)(4 * kSrcStride)));
    v.base = reinterpret_cast<const char*>(srcT) + (unsigned)af;
#ifdef AHV_DIAG_LINEAR_GATHER  // diagnostic only (wrong results): row = lane, so every b128 lane group of every corner
    // request covers the 16 slots once -- the conflict-free bound of the gather (tools/kbench, profiles/r03_scorer_segments.txt)
    v.base = reinterpret_cast<const char*>(srcT) + (threadIdx.x & 63) * (4 * kSrcStride);
#endif
}

// pass p of quarter Q, Q a compile-time constant (the one-wave-per-hypothesis kernels)
template <int Q, bool XDL = kFp32LowHalf>
__device__ __forceinline__ void hat_voxel(HatVoxel& v, const float* srcT, const GatherHyp& h, int p)
{
    hat_voxel_at<XDL>(v, srcT, Q == 0 ? h.ixy[p] : __builtin_elementwise_fma(f32x2{(float)Q, (float)Q}, h.dqxy, h.ixy[p]),
                      Q == 0 ? h.izp[p] : fmaf((float)Q, h.dqz, h.izp[p]));
}

// the same with the quarter known only at run time (wave-uniform: ahv_team.h, one quarter per wave)
// Same values as hat_voxel<Q> for Q = qf, bit for bit: quarter 0 takes the coordinates as they are (hat_voxel<0> does not
// multiply: 0 * inf would differ), and fma(1, d, i) is the rounded sum d + i that hat_voxel<1> computes.
__device__ __forceinline__ void hat_voxel_rt(HatVoxel& v, const float* srcT, const GatherHyp& h, int p, float qf)
{
    const f32x2 ixy = __builtin_elementwise_fma(f32x2{qf, qf}, h.dqxy, h.ixy[p]);
    const float iz = fmaf(qf, h.dqz, h.izp[p]);
    const bool q0 = qf == 0.0f;
    hat_voxel_at<kFp32LowHalf>(v, srcT, f32x2{q0 ? h.ixy[p][0] : ixy[0], q0 ? h.ixy[p][1] : ixy[1]}, q0 ? h.izp[p] : iz);
}

// Quarter Q of the rotated volume into `buf`.  The two voxels of a lane (passes 0, 1) are blended as ONE stream
// of 16 corner steps whose source rows are requested kHatDepth steps ahead (a ring of 4 x ds_read_b128 per
// step): hipcc on its own keeps only 2-3 reads in flight, about a third of an LDS round trip under bank
// conflicts, and leaves the rest to the partner wave.  The stream is cut in two so that its head can be issued
// from inside the PREVIOUS quarter's GEMM (hat_prologue: coordinates, weights and the first kHatDepth row
// requests, placed ahead of that GEMM's last MFMA chunk), which hides the gather's start-up round trip as well.
// The depth is the fp32 kernels' (4 and 8 measured slower there, round 2).  The split-f16 kernel is bound by what the LDS
// gets through, not by its round trips: there every depth from 1 to 7 times the same, and it runs at depth 2 and spends
// the registers on W1 fragments instead (kSplitHatDepth, ahv_split.h).  A kernel only touches the slots below its depth.
constexpr int kHatDepth = 6;  // fp32 kernel, round 4: 2 / 3 / 4 / 5 / 6 rows = 0.690 / 0.687 / 0.684 / 0.681 / 0.681 ms
static_assert(kHatDepth >= 1 && kHatDepth <= 8, "the prologue requests rows of pass 0 only");

struct HatState {
    HatVoxel vx[2];
    f32x4 ring[kHatDepth][4];
};

// MIR: the quarter being gathered is the point mirror of the one whose voxels st.vx was set up for (hat_mirror below):
// its pass-p voxel is the mirror of the other quarter's pass-(1 - p) voxel.
template <int S, bool MIR = false, int DEPTH = kHatDepth>
__device__ __forceinline__ void hat_request(HatState& st)
{
    static_assert(DEPTH >= 1 && DEPTH <= kHatDepth, "ring slots");
    const f32x4* row = reinterpret_cast<const f32x4*>(st.vx[MIR ? 1 - (S >> 3) : (S >> 3)].base + hat_off(S & 7));
#pragma unroll
    for (int j = 0; j < 4; ++j) st.ring[S % DEPTH][j] = row[j];
}

template <int S, int END, bool MIR = false>
struct HatRequests {  // the first END requests of a quarter = a ring of depth END being filled
    static __device__ __forceinline__ void run(HatState& st)
    {
        hat_request<S, MIR, END>(st);
        HatRequests<S + 1, END, MIR>::run(st);
    }
};
template <int END, bool MIR>
struct HatRequests<END, END, MIR> {
    static __device__ __forceinline__ void run(HatState&) {}
};

// Point symmetry.  The sample coordinate of the voxel mirrored through the volume's centre is i' = 7 - i on every axis
// (i = 4 (R p) + 3.5 and p' = -p), so its clamped base row is 6 - j and, the hat function being even, its corner weights
// are the SAME eight numbers in reverse corner order: w'[n] = w[7 - n].  The mirror of the pass-p voxel of quarter Q
// that a lane owns is the pass-(1 - p) voxel (7 - x, 7 - y, 7 - z) of quarter 3 - Q.  So the quarters are gathered in the
// order 0, 3, 1, 2 and the second quarter of each pair costs TWO instructions of set-up (the mirrored base addresses)
// instead of ~60: the weights are read in reverse, the two voxels swap passes, and the lane writes the mirrored voxels'
// slots of the image (GatherDst::m0 / m1).  Where i sits within an ulp of an integer the mirrored floor may differ from
// a direct evaluation by one row with weights (1, 0) against (0, 1): the same sample to rounding.
constexpr int kSrcMirrorBytes = 6 * 4 * kSrcStride * (kSrcPlaneRows + kSrcRowsY + 1);  // byte offset of row (6, 6, 6)

__device__ __forceinline__ void hat_mirror(HatState& st, const float* srcT)
{
    const char* top = reinterpret_cast<const char*>(srcT) + kSrcMirrorBytes;
#pragma unroll
    for (int k = 0; k < 2; ++k) st.vx[k].base = top - (st.vx[k].base - reinterpret_cast<const char*>(srcT));
#ifdef AHV_DIAG_LINEAR_GATHER
#pragma unroll
    for (int k = 0; k < 2; ++k) st.vx[k].base = reinterpret_cast<const char*>(srcT) + (threadIdx.x & 63) * (4 * kSrcStride);
#endif
}

// head of the mirrored quarter's gather (the counterpart of hat_prologue)
template <int DEPTH = kHatDepth>
__device__ __forceinline__ void hat_prologue_mirror(HatState& st, const float* srcT)
{
    hat_mirror(st, srcT);
    HatRequests<0, DEPTH, true>::run(st);
}

template <int Q, bool XDL = kFp32LowHalf, int DEPTH = kHatDepth>
__device__ __forceinline__ void hat_prologue(HatState& st, const float* srcT, const GatherHyp& h)
{
    // voxel 0, its first row requests, THEN voxel 1: setting both voxels up together (packed over the two passes, 24
    // instructions fewer per hypothesis) was measured slower -- 0.6911 vs 0.6880 ms per 50 000 hypotheses, round 3 -- the
    // first six row requests then wait for both set-ups
    hat_voxel<Q, XDL>(st.vx[0], srcT, h, 0);
    HatRequests<0, DEPTH>::run(st);
    hat_voxel<Q, XDL>(st.vx[1], srcT, h, 1);
}

__device__ __forceinline__ void hat_prologue_rt(HatState& st, const float* srcT, const GatherHyp& h, float qf)
{
    hat_voxel_rt(st.vx[0], srcT, h, 0, qf);
    HatRequests<0, kHatDepth>::run(st);
    hat_voxel_rt(st.vx[1], srcT, h, 1, qf);
}

// Where a blended voxel goes.  ROW = floats between the channel planes of the destination image (128: the
// forward's swizzled quarter image).  The split-f16 kernel has a store of its own (HatStoreSplit, ahv_split.h).
template <int ROW>
struct HatStoreF32 {
    static constexpr bool kXdlKernel = kFp32LowHalf;  // fp32 MFMAs only: no protection needed (see low_half)
    static constexpr int kDepth = kHatDepth;
    float* d[2];  // the lane's voxel of pass 0 / pass 1 in channel plane 0
    __device__ __forceinline__ void operator()(int p, const f32x2 (&o)[8]) const
    {
        float* dst = d[p];
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c * ROW] = o[c >> 1][c & 1];
    }
};

template <int S, typename Store, bool MIR = false>
struct HatSteps {
    static __device__ __forceinline__ void run(HatState& st, f32x2 (&o)[8], const Store& store)
    {
        constexpr int p = S >> 3, n = S & 7;
        constexpr int vi = MIR ? 1 - p : p, wi = MIR ? 7 - n : n;  // mirrored quarter: the other voxel, weights reversed
        __builtin_amdgcn_sched_barrier(0);
        // the odd weights sit in the high half of their register pair (XDL kernels: see low_half)
        const float wsc = (Store::kXdlKernel && (wi & 1)) ? low_half(st.vx[vi].w[wi]) : st.vx[vi].w[wi];
        const f32x2 wn = {wsc, wsc};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = st.ring[S % Store::kDepth][j];
            const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
            if (n == 0) {
                o[2 * j] = lo * wn;
                o[2 * j + 1] = hi * wn;
            } else {
                o[2 * j] = __builtin_elementwise_fma(lo, wn, o[2 * j]);
                o[2 * j + 1] = __builtin_elementwise_fma(hi, wn, o[2 * j + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (S + Store::kDepth < 16) hat_request<(S + Store::kDepth < 16 ? S + Store::kDepth : 0), MIR, Store::kDepth>(st);
        if (n == 7) store(p, o);
        HatSteps<S + 1, Store, MIR>::run(st, o, store);
    }
};
template <typename Store, bool MIR>
struct HatSteps<16, Store, MIR> {
    static __device__ __forceinline__ void run(HatState&, f32x2 (&)[8], const Store&) {}
};

// the 16 blend steps of a quarter whose prologue has been issued (MIR: hat_prologue_mirror)
template <bool MIR = false>
__device__ __forceinline__ void hat_body(HatState& st, float* buf, const GatherDst& dst)
{
    f32x2 o[8];
    const HatStoreF32<128> store = {{buf + (MIR ? dst.m0 : dst.o0), buf + (MIR ? dst.m1 : dst.o1)}};
    __builtin_amdgcn_s_setprio(1);
    HatSteps<0, HatStoreF32<128>, MIR>::run(st, o, store);
    __builtin_amdgcn_s_setprio(0);
}

// the same into a LINEAR image X[c][voxel = a0*64 + b*8 + e] with ROW floats per channel plane (backward kernels)
template <int ROW, bool MIR = false>
__device__ __forceinline__ void hat_body_linear(HatState& st, float* img, const GatherDst& dst)
{
    f32x2 o[8];
    const HatStoreF32<ROW> store = {{img + (MIR ? dst.m0 : dst.o0), img + (MIR ? dst.m1 : dst.o1)}};
    HatSteps<0, HatStoreF32<ROW>, MIR>::run(st, o, store);
}

}  // namespace ahv

namespace ahv {

// ---------------------------------------------------------------------------------------------------------
// GEMM1 on quarter Q, software-pipelined by hand one chunk deep.  hipcc schedules gemm1_quarter_lds just in
// time (fragment reads, s_waitcnt, 8 MFMAs, next reads ...): every group of MFMAs starts with an exposed LDS
// round trip that only the partner wave can cover.  Here the 192 MFMAs are cut into 12 chunks of 16; the A
// fragments (LDS table) and B operands (quarter image) of chunk k+1 are requested before the MFMAs of chunk k
// are done (512 matrix-pipe cycles, several LDS latencies), and sched_barrier / sched_group_barrier keep hipcc from
// sinking the reads back to their uses.  Same MFMAs, same order per accumulator as gemm1_quarter_lds: bit-identical sums.
//   chunks 0-3: x slab, channels 4k..4k+3;  4-7: y slab;  8-11: z slab, channel pairs cpp = k - 8.
// ---------------------------------------------------------------------------------------------------------
struct G1Chunk {
    f32x4 a[4];
    float b[8];
};

template <int Q, int K>
__device__ __forceinline__ void g1_load(G1Chunk& ck, const f32x4* T, const float* buf, int i0, int j, int kq)
{
    if (K < 8) {
        const int c0 = 4 * (K & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i;
            ck.a[i] = T[((K < 4 ? 0 : 16) + c) * 64];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                ck.b[2 * i + hh] = (K < 4) ? buf[c * 128 + qoff(i0, j, 4 * hh + kq)]    // x slab: k = (c, w)
                                           : buf[c * 128 + qoff(i0, 4 * hh + kq, j)];   // y slab: k = (c, h)
        }
    } else {
        const int cpp = K - 8;
        ck.a[0] = T[(32 + 4 * Q + cpp) * 64];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                ck.b[4 * ci + t] = buf[(2 * (2 * cpp + ci) + (kq >> 1)) * 128 + qoff(kq & 1, 2 * t + i0, j)];
    }
}

template <int Q, int K>
__device__ __forceinline__ void g1_mfma(f32x4 (&acc)[2][4], const G1Chunk& ck)
{
    if (K < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                acc[0][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 0], ck.b[2 * i + hh], acc[0][Q], 0, 0, 0);
                acc[1][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 1], ck.b[2 * i + hh], acc[1][Q], 0, 0, 0);
            }
    } else {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0][2 * ci + 0], ck.b[4 * ci + t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0][2 * ci + 1], ck.b[4 * ci + t], acc[1][t], 0, 0, 0);
            }
    }
}

// `hook` runs once, ahead of the MFMAs of chunk kG1HookChunk (the fused scorer passes the next quarter's
// gather prologue, whose LDS round trip then elapses under the remaining MFMAs of this quarter).
constexpr int kG1HookChunk = 10;  // chunks 8 and 11 measured slower (round 2)

template <int Q, int K>
struct G1Pipe {
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[2][4], G1Chunk& cur, const f32x4* T, const float* buf,
                                               int i0, int j, int kq, Hook& hook)
    {
        G1Chunk nxt;
        if (K == kG1HookChunk) {
            if (K + 1 < 12) g1_load<Q, K + 1>(nxt, T, buf, i0, j, kq);
            hook();
            __builtin_amdgcn_sched_barrier(0);
            g1_mfma<Q, K>(acc, cur);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // Issue order (round 5): the next chunk's LDS reads go BETWEEN this chunk's MFMAs, one behind each -- an MFMA
            // holds the matrix pipe for 32 cycles in which the wave's issue slot is free.  All of them in front of the
            // first MFMA (rounds 2-4) left the pipe waiting for their issue whenever the SIMD's other wave was not in a
            // GEMM: +1.0 % on the whole kernel (tools/kbench, three alternating runs on two boxes).
            __builtin_amdgcn_sched_barrier(0);
            if (K + 1 < 12) g1_load<Q, K + 1>(nxt, T, buf, i0, j, kq);
            g1_mfma<Q, K>(acc, cur);
            if (K + 1 < 12) {
                constexpr int NL = (K + 1 < 8) ? 12 : 9;  // LDS reads of the next chunk
#pragma unroll
                for (int g = 0; g < NL; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16 - NL, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (K + 1 < 12) G1Pipe<Q, K + 1>::run(acc, nxt, T, buf, i0, j, kq, hook);
    }
};

template <int Q>
struct G1Pipe<Q, 12> {
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&)[2][4], G1Chunk&, const f32x4*, const float*, int, int, int, Hook&) {}
};

template <int Q, typename Hook>
__device__ __forceinline__ void gemm1_quarter_pipe(f32x4 (&acc)[2][4], const float* table, const float* buf, int lane,
                                                   Hook hook)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    const f32x4* T = reinterpret_cast<const f32x4*>(table) + lane;
    G1Chunk first;
    g1_load<Q, 0>(first, T, buf, i0, j, kq);
    G1Pipe<Q, 0>::run(acc, first, T, buf, i0, j, kq, hook);
}

}  // namespace ahv
