// ahv_team.h -- ONE volume shared by the four waves of a "team" (fused scorer, fp32 path).
//
// The fused scorer gives every hypothesis to one wave (ahv_score.hip): ~27 us of latency per hypothesis and a
// persistent grid of 2 048 wave slots.  Two places want a hypothesis' worth of work done FASTER than that:
//  * the remainder of a launch, N mod 2048 hypotheses that fill a last round only partly (a 6 250-hypothesis shard of a
//    strong-scaling run ends with 106 hypotheses on 2 048 slots; test_co3d.py:137-146 split over 8 GPUs);
//  * the target features forward_3d2d(vol_tgt) (test_co3d.py:141), which every workgroup of ahv_verify_pair_f32 builds
//    for itself instead of reading them from a launch of their own.
// A team = waves {4T .. 4T+3} of a workgroup (one wave per SIMD; the two teams of a workgroup interleave on the SIMDs
// like the two waves of a SIMD do in the one-wave formulation).  Member Q produces quarter Q of the volume (d in
// {2Q, 2Q+1}) in its private 8-KiB image -- with the one-wave kernel's own gather, point mirror included: members 3 and 2
// set up quarters 0 and 1 and gather their mirror images -- and the members meet (`arrive`).  Member t then owns position
// tile t and runs the one-wave kernel's accumulation chain OF THAT TILE: quarters in the order 0, 3, 1, 2; the x / y slabs
// of quarter t (its own image, 128 MFMAs) where the chain has them; the z slab of every quarter, restricted to tile t, from
// the image of the member that gathered it (4 x 16 MFMAs).  Same MFMAs with the same operands in the same order per
// accumulator as gemm1_quarter_pipe<0>, <3>, <1>, <2>: the pre-activations are BIT-IDENTICAL to a lone wave's.  ReLU, GEMM2,
// bias (modules/modules.py:68-69) are per tile anyway; the sums of squares, the dot products with the target and the mean
// over positions are associated exactly as hyp_score_rs / hyp_score_tail associate them (ahv_score.hip), the four partial
// sums meet in LDS and member 0 adds them in the order of the one-wave kernel's last two DPP steps.  A team's score is
// therefore the lone wave's score bit for bit: whether a hypothesis goes to a team is a scheduling decision that no result
// depends on (rounds 1-4 exchanged partial z sums instead: 1e-7 apart, and sharded runs disagreed with unsharded ones at
// rounding; tests/test_gpu_parity.py::test_team_scores_are_bit_identical).
//
// Synchronisation: gfx950 has one s_barrier per workgroup and the two teams must not run in lockstep, so the members meet
// on LDS counters (TeamSync): `arrive` = "my quarter is in my image", `done` = "I have read everybody's image, the
// images may be overwritten".  Counters only grow (4 per round); a wave's DS operations execute in order, so an image
// written before the counter update is visible to whoever sees the update.  Termination: all four members of a team are
// resident in the same workgroup and run the same number of rounds (team-uniform trip counts), so every wait is met.
#pragma once
#include "ahv_dual.h"

namespace ahv {

struct TeamSync {
    unsigned arrive[2];   // per team: members that have published this round's partials (monotonic, 4 per round)
    unsigned done[2];     // per team: members that have finished reading their teammates' images (monotonic)
    unsigned tgt_ready;   // ahv_verify_pair_f32: members of team 1 that have written their target rows (4 per sample)
    float part[2][2][4];  // [team][round parity][member]: mean-cosine partial of 16 positions
};

__device__ __forceinline__ void team_signal(unsigned* ctr, int lane)
{
    // earlier DS writes of this wave are ahead of the add in the LDS queue; the wait makes "earlier reads have
    // returned" true as well (done = my reads of the teammates' images are complete)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void team_wait(unsigned* ctr, unsigned target)
{
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- GEMM1 of ONE position tile, the one-wave kernel's chain ----------------------------------------------------
// z-slab operands of quarter Q for position tile t: the A fragments of g1_load<Q, 8..11> and, of their B operands, the
// ones of tile t -- read from the image of the member that gathered quarter Q.
struct TeamZ {
    f32x4 a[4];  // [cpp]: table group 32 + 4 Q + cpp
    float b[8];  // [2 cpp + ci]
};

__device__ __forceinline__ void team_z_load(TeamZ& z, const f32x4* T, const float* bufs, int Q, int zoff)
{
    const f32x4* Tz = T + (32 + 4 * Q) * 64;
    const float* img = bufs + Q * kQuarterFloats + zoff;
#pragma unroll
    for (int cpp = 0; cpp < 4; ++cpp) {
        z.a[cpp] = Tz[cpp * 64];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) z.b[2 * cpp + ci] = img[2 * (2 * cpp + ci) * 128];
    }
}

__device__ __forceinline__ void team_z_mfma(f32x4 (&u)[2], const TeamZ& z)
{
#pragma unroll
    for (int cpp = 0; cpp < 4; ++cpp)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            u[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(z.a[cpp][2 * ci + 0], z.b[2 * cpp + ci], u[0], 0, 0, 0);
            u[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(z.a[cpp][2 * ci + 1], z.b[2 * cpp + ci], u[1], 0, 0, 0);
        }
}

// x / y slabs of the quarter in `buf` into u: chunks 0-7 of gemm1_quarter_pipe, operands one chunk ahead
template <int K>
__device__ __forceinline__ void g1_load_xy(G1Chunk& ck, const f32x4* T, const float* buf, int i0, int j, int kq)
{
    static_assert(K < 8, "x and y slabs");
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
}

__device__ __forceinline__ void g1_mfma_xy(f32x4 (&u)[2], const G1Chunk& ck)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            u[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 0], ck.b[2 * i + hh], u[0], 0, 0, 0);
            u[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 1], ck.b[2 * i + hh], u[1], 0, 0, 0);
        }
}

template <int K>
struct G1PipeXY {
    static __device__ __forceinline__ void run(f32x4 (&u)[2], G1Chunk& cur, const f32x4* T, const float* buf, int i0, int j, int kq)
    {
        G1Chunk nxt;
        __builtin_amdgcn_sched_barrier(0);
        if (K + 1 < 8) g1_load_xy<(K + 1 < 8 ? K + 1 : 0)>(nxt, T, buf, i0, j, kq);
        g1_mfma_xy(u, cur);
        if (K + 1 < 8) {  // the next chunk's 12 LDS reads between the MFMAs, one behind each (G1Pipe, ahv_dual.h)
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (K + 1 < 8) G1PipeXY<K + 1>::run(u, nxt, T, buf, i0, j, kq);
    }
};
template <>
struct G1PipeXY<8> {
    static __device__ __forceinline__ void run(f32x4 (&)[2], G1Chunk&, const f32x4*, const float*, int, int, int) {}
};

// quarter order of the one-wave kernel (point mirror: 0, 3, 1, 2; ahv_dual.h)
__device__ __forceinline__ int team_quarter_at(int step) { return (0x2130 >> (4 * step)) & 3; }

// u[m] = pre-activation rows of m-tile m at the 16 positions of tile `member`, accumulated exactly as the one-wave kernel
// accumulates acc[m][member].  `bufs` = image of member 0 of this team; all four images must be complete (arrive point).
__device__ __forceinline__ void team_tile_gemm(f32x4 (&u)[2], const float* table, const float* bufs, int member, int lane)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    const f32x4* T = reinterpret_cast<const f32x4*>(table) + lane;  // group g at T[g * 64]
    // B operand of z chunk (cpp, ci) for tile t: image[(2 (2 cpp + ci) + (kq >> 1)) * 128 + qoff(kq & 1, 2 t + i0, j)]
    const int zoff = (kq >> 1) * 128 + qoff(kq & 1, 2 * member + i0, j);
    u[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    u[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    TeamZ zc;
    team_z_load(zc, T, bufs, 0, zoff);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int Q = team_quarter_at(s);
        TeamZ zn;
        team_z_load(zn, T, bufs, team_quarter_at(s < 3 ? s + 1 : 3), zoff);  // next quarter's operands (the last step reloads its own)
        if (Q == member) {
            const float* buf = bufs + member * kQuarterFloats;
            G1Chunk first;
            g1_load_xy<0>(first, T, buf, i0, j, kq);
            G1PipeXY<0>::run(u, first, T, buf, i0, j, kq);
        }
        __builtin_amdgcn_sched_barrier(0);
        team_z_mfma(u, zc);
        __builtin_amdgcn_sched_barrier(0);
        zc = zn;
    }
}

// ---- one quarter with the quarter index known at run time (the exact path of ahv_score.hip) -----------------------
struct TeamAcc {
    f32x4 xy[2];    // [m]: x and y slabs of this wave's quarter -> position tile Q (complete)
    f32x4 z[2][4];  // [m][t]: z slab restricted to this wave's two depths -> all four position tiles (partial)
};

// GEMM1 on the quarter in `buf` with the quarter index known at run time (it only selects the z slab's A fragments).
// Same MFMAs, operands and hand-made pipeline as gemm1_quarter_pipe (ahv_dual.h: 12 chunks of 16 MFMAs, the operands of
// chunk k + 1 requested before the MFMAs of chunk k are issued); x / y accumulate in their own tile.
template <int K>
__device__ __forceinline__ void g1_load_team(G1Chunk& ck, const f32x4* T, const f32x4* Tz, const float* buf, int i0, int j, int kq)
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
        ck.a[0] = Tz[cpp * 64];  // Tz = T + (32 + 4 Q) * 64: the z-slab fragments of THIS wave's quarter
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                ck.b[4 * ci + t] = buf[(2 * (2 * cpp + ci) + (kq >> 1)) * 128 + qoff(kq & 1, 2 * t + i0, j)];
    }
}

template <int K>
__device__ __forceinline__ void g1_mfma_team(TeamAcc& a, const G1Chunk& ck)
{
    if (K < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                a.xy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 0], ck.b[2 * i + hh], a.xy[0], 0, 0, 0);
                a.xy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[i][2 * hh + 1], ck.b[2 * i + hh], a.xy[1], 0, 0, 0);
            }
    } else {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a.z[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0][2 * ci + 0], ck.b[4 * ci + t], a.z[0][t], 0, 0, 0);
                a.z[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ck.a[0][2 * ci + 1], ck.b[4 * ci + t], a.z[1][t], 0, 0, 0);
            }
    }
}

template <int K>
struct G1PipeTeam {
    static __device__ __forceinline__ void run(TeamAcc& a, G1Chunk& cur, const f32x4* T, const f32x4* Tz, const float* buf,
                                               int i0, int j, int kq)
    {
        G1Chunk nxt;
        if (K + 1 < 12) g1_load_team<K + 1>(nxt, T, Tz, buf, i0, j, kq);
        __builtin_amdgcn_sched_barrier(0);
        g1_mfma_team<K>(a, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (K + 1 < 12) G1PipeTeam<K + 1>::run(a, nxt, T, Tz, buf, i0, j, kq);
    }
};
template <>
struct G1PipeTeam<12> {
    static __device__ __forceinline__ void run(TeamAcc&, G1Chunk&, const f32x4*, const f32x4*, const float*, int, int, int) {}
};

__device__ __forceinline__ void gemm1_quarter_team(TeamAcc& a, const float* table, const float* buf, int lane, int Q)
{
    const int n = lane & 15, kq = lane >> 4;
    const int i0 = n >> 3, j = n & 7;
    const f32x4* T = reinterpret_cast<const f32x4*>(table) + lane;  // group g at T[g * 64]
    const f32x4* Tz = T + (32 + 4 * Q) * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        a.xy[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) a.z[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    G1Chunk first;
    g1_load_team<0>(first, T, Tz, buf, i0, j, kq);
    G1PipeTeam<0>::run(a, first, T, Tz, buf, i0, j, kq);
}

// What a thread brings in for the TARGET of a sample, in two halves (loads issued early, stored ahead of the staging
// barrier).  TGT: the target VOLUME -- members of team 1 (threads 256..511) load their un-rotated quarter (what the
// gather produces for R = I, without the gather: channel c of quarter Q is 128 contiguous floats; a thread owns two
// neighbouring voxels, neighbours in the swizzled image too); else the target FEATURES [32][64] as the score's per-lane fragments: thread (t, m2, l) holds
// rows 16 m2 + 4 (l >> 4) + r of position 16 t + (l & 15) and parks them in the pad of source row `tid`.
struct TgtRegs {
    float2 q[16];
};

template <bool TGT>
__device__ __forceinline__ void tgt_regs_load(TgtRegs& r, const float* __restrict__ tgt, int b, int tid)
{
    if (TGT) {
        if (tid >= 256) {
            const float* V = tgt + (long)b * (16 * 512) + ((tid >> 6) & 3) * 128 + 2 * (tid & 63);
#pragma unroll
            for (int c = 0; c < 16; ++c) r.q[c] = *reinterpret_cast<const float2*>(V + c * 512);
        } else {  // (defined on every path: a conditional definition keeps the OLD values alive across the hypothesis loop)
#pragma unroll
            for (int c = 0; c < 16; ++c) r.q[c] = float2{0.0f, 0.0f};
        }
    } else {
        const float* ft = tgt + (long)b * (32 * 64);
        const int l = tid & 63, t = tid >> 7, m2 = (tid >> 6) & 1;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            r.q[k] = float2{ft[(16 * m2 + 4 * (l >> 4) + 2 * k) * 64 + 16 * t + (l & 15)],
                            ft[(16 * m2 + 4 * (l >> 4) + 2 * k + 1) * 64 + 16 * t + (l & 15)]};
    }
}

// `buf` = the calling wave's private quarter image
template <bool TGT>
__device__ __forceinline__ void tgt_regs_store(float* srcT, float* buf, const TgtRegs& r, int tid)
{
    if (TGT) {
        if (tid >= 256) {
            const int i = 2 * (tid & 63), o = qoff(i >> 6, (i >> 3) & 7, i & 7);
#pragma unroll
            for (int c = 0; c < 16; ++c) *reinterpret_cast<float2*>(buf + c * 128 + o) = r.q[c];
        }
    } else {
        *reinterpret_cast<f32x4*>(srcT + tid * kSrcStride + 16) = f32x4{r.q[0].x, r.q[0].y, r.q[1].x, r.q[1].y};
    }
}

// The round of a team: meet (every member's quarter is in its image), build this member's tile, release the images.
// On return `done` has been signalled, i.e. the caller may NOT touch its image again before
// team_wait(done, 4 * rounds) (the next round's first image store).
__device__ __forceinline__ void team_round(f32x4 (&u)[2], const float* table, float* bufs, TeamSync& ts, int team, int member,
                                           unsigned rounds_before, int lane)
{
    team_signal(&ts.arrive[team], lane);
    team_wait(&ts.arrive[team], 4u * (rounds_before + 1u));
    team_tile_gemm(u, table, bufs, member, lane);
    team_signal(&ts.done[team], lane);
}

// v = W2 relu(u) + b2 for one position tile: the tile-sized piece of gemm2_dual, same MFMAs in the same order
__device__ __forceinline__ void team_head(f32x4 (&v)[2], const f32x4 (&u)[2], const DualFrags& f)
{
    v[0] = f.bias[0];
    v[1] = f.bias[1];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x4 x = relu4(u[m]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][0], x[r], v[0], 0, 0, 0);
            v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[m][r][1], x[r], v[1], 0, 0, 0);
        }
    }
}

// squared norm over the 32 channels of each position of the tile (all four lane rows of a column end up with the sum)
__device__ __forceinline__ float team_sumsq(const f32x4 (&v)[2])
{
    float ss = 0.0f;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(v[m2][r], v[m2][r], ss);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    return ss;
}

}  // namespace ahv
