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
// {2Q, 2Q+1}) in its private 8-KiB image and contracts it against W1:
//     x / y slabs of quarter Q feed position tile Q only        -> TeamAcc::xy, complete after this wave's 128 MFMAs
//     the z slab of quarter Q feeds all four position tiles     -> TeamAcc::z[t], a PARTIAL sum over d in {2Q, 2Q+1}
// The members exchange the z partials through their images (the image is dead once its GEMM has run), member t then owns
// position tile t: u = xy + z_0 + z_1 + z_2 + z_3 (fixed order), ReLU, GEMM2, bias, normalise (modules/modules.py:68-69,
// 122) -- and for a hypothesis the dot product with the target and the mean over its 16 positions; the four partial
// means meet in LDS and member 0 adds them in a fixed order.  The sums are associated differently from the one-wave
// kernel's (there the z slab is ONE chain over the quarters), so a team's score agrees with a lone wave's to rounding
// (~1e-7), not bit for bit; AHV_SCORE_NO_TEAMS switches teams off for callers who need scores that do not depend on N.
//
// Synchronisation: gfx950 has one s_barrier per workgroup and the two teams must not run in lockstep, so the members meet
// on LDS counters (TeamSync): `arrive` = "my partials are in my image", `done` = "I have read everybody's partials, the
// images may be overwritten".  Counters only grow (4 per round); a wave's DS operations execute in order, so a partial
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

// Quarter Q of an UN-rotated volume V[16][8][8][8] (global memory) into the swizzled quarter image: what the gather
// produces for R = I, without the gather.  Channel c of the quarter is 128 contiguous floats; a lane owns two
// neighbouring voxels (x even, x + 1), which are neighbours in the image as well (qoff only XORs bits 1-2 of x).
__device__ __forceinline__ void stage_quarter_global(float* buf, const float* __restrict__ V, int Q, int lane)
{
    const int i = 2 * lane, a0 = i >> 6, bb = (i >> 3) & 7, e = i & 7;
    const int o = qoff(a0, bb, e);
    float2 x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = *reinterpret_cast<const float2*>(V + c * 512 + Q * 128 + i);
#pragma unroll
    for (int c = 0; c < 16; ++c) *reinterpret_cast<float2*>(buf + c * 128 + o) = x[c];
}

// The exchange: publish the z partials, meet, and collect position tile Q.  `bufs` = image of member 0 of this team.
// On return u[m] = pre-activation rows of m-tile m at the 16 positions of tile Q; `done` has been signalled, i.e. the
// caller may NOT touch its image again before team_wait(done, 4 * rounds) (the next round's first image store).
__device__ __forceinline__ void team_exchange(f32x4 (&u)[2], const TeamAcc& a, float* bufs, TeamSync& ts, int team, int Q,
                                              unsigned rounds_before, int lane)
{
    f32x4* mine = reinterpret_cast<f32x4*>(bufs + Q * kQuarterFloats);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) mine[(m * 4 + t) * 64 + lane] = a.z[m][t];
    team_signal(&ts.arrive[team], lane);
    team_wait(&ts.arrive[team], 4u * (rounds_before + 1u));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        u[m] = a.xy[m];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            u[m] += reinterpret_cast<const f32x4*>(bufs + q * kQuarterFloats)[(m * 4 + Q) * 64 + lane];
    }
    team_signal(&ts.done[team], lane);
}

// v = W2 relu(u) + b2 for one position tile (the tile-sized piece of gemm2_dual), and the squared norm over the 32
// channels of each position (all four lane rows of a column end up with the complete sum).
__device__ __forceinline__ float team_head(f32x4 (&v)[2], const f32x4 (&u)[2], const DualFrags& f)
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
