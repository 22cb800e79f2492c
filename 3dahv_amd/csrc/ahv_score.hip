// ahv_score.hip -- fused rotation-hypothesis scorer (the north-star kernel).
//
// One launch scores N hypotheses for B volume pairs:
//   rotate_volume (utils.py:113-131) -> forward_3d2d (modules/modules.py:112-124)
//   -> (f_src*f_tgt).sum(ch).mean(pos) (test_co3d.py:143) -> running max (test_co3d.py:145)
// HBM traffic per hypothesis: 36 B of R in, 4 B of score out.  Everything else
// (source volume, head weights, target feature) is resident on chip.
//
// Launch shape: 512-thread workgroups (8 waves, two per SIMD, 256 registers each), a persistent grid of
// ~one workgroup per CU; blockIdx.y strides over the batch, (blockIdx.x, wave) strides over hypotheses;
// one wave owns one hypothesis, no barrier on the hot path.  LDS per workgroup (158 KiB): source image
// 46 KiB + W1 fragment table 48 KiB + 8 x 8 KiB private quarter images (ahv_dual.h).
// Three instances: score_hypotheses_dual_kernel<false, false> (all fp32, the default), <false, true> (the same with the
// target features forward_3d2d(vol_tgt) built inside the launch: ahv_verify_pair_f32) and <true, false> (GEMM1 as
// split-f16 MFMA products, opt-in through AHV_SCORE_SPLIT_F16; ahv_split.h).  The remainder of a launch that fills
// less than a quarter of the wave slots is scored by TEAMS of four waves per hypothesis (ahv_team.h).
#ifndef AHV_DIAG_NO_FP32_LOW_HALF  // (tools/kbench A/B build of the unprotected fp32 scorers)
// The fp32 scorers keep every scalar they broadcast over a register pair in the LOW half too (low_half, ahv_dual.h).  Rounds 3-4
// relied on occupancy instead (two waves of > 248 registers own the SIMD: no foreign XDL wave fits) -- true while both waves
// run, but the older wave of a SIMD retires ~110 us before its partner, and in that tail a wave of another stream's XDL
// kernel CAN land beside the remaining one.  75 more vector instructions per hypothesis (2 555 -> 2 630): +0.3 %.
#define AHV_FP32_LOW_HALF
#endif
#include "ahv_dual.h"
#include "ahv_split.h"
#include "ahv_team.h"
#include "ahv_exact.h"
#include "ahv_launch.h"

namespace ahv {

// a value the compiler cannot see through: keeps address arithmetic where it is written (not hoisted out of a loop)
__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ float swap_add32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float swap_add16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// F.normalize(dim=1), dot with the unit-norm target, mean over the 64 positions (modules/modules.py:122,
// test_co3d.py:143) as a reduce-scatter: a lane holds 8 of the 32 channels of position (16t + lane&15) for the
// four tiles t; after two exchange steps lane (col, kq) owns the complete sums of ONE position (tile kq, column
// col), so the normalisation runs once per lane.  Result uniform (read from lane 63).
// One tile's share: sums of squares and dot products of a lane's 8 channels of position (16 t + lane & 15).
template <bool XDL>  // XDL: the kernel issues XDL MFMAs (low_half, ahv_dual.h)
__device__ __forceinline__ void hyp_tile_sums(float& ss, float& dt, const f32x4& v0, const f32x4& v1, const f32x4& g0, const f32x4& g1)
{
    // on v_pk_fma_f32: even and odd registers accumulate side by side and meet in one addition (8 packed instead of 16
    // scalar FMAs per tile; the order of the eight additions differs from a sequential sum in the last bit at most)
    f32x2 s2 = {0.0f, 0.0f}, d2 = {0.0f, 0.0f};
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4& vv = m2 ? v1 : v0;
            const f32x4& gg = m2 ? g1 : g0;
            const f32x2 x = {vv[2 * hh], vv[2 * hh + 1]};
            const f32x2 g = {gg[2 * hh], gg[2 * hh + 1]};
            s2 = __builtin_elementwise_fma(x, x, s2);
            d2 = __builtin_elementwise_fma(x, g, d2);
        }
    // hipcc adds the halves of a pair with v_pk_add_f32 op_sel:[0,1]: the low lane reads a high half
    ss = s2[0] + (XDL ? low_half(s2[1]) : s2[1]);
    dt = d2[0] + (XDL ? low_half(d2[1]) : d2[1]);
}

// From the per-tile sums to the hypothesis' score.  Result uniform (read from lane 63).
__device__ __forceinline__ float hyp_score_tail(const float (&ss)[4], const float (&dt)[4])
{
    // No contraction here: hipcc fused the cosine's multiplication into the first addition of wave_sum_dpp
    // (v_mul, v_mov_dpp, v_fmac: the sum then holds one UNROUNDED product per lane), which a team -- whose cosines pass
    // through a select first -- did not: the one place where a team's score was an ulp off a lone wave's (round 5,
    // tools/team_stage_diff.py).  With the product rounded in both, the two are the same sums.  One instruction fewer, too.
#pragma clang fp contract(off)
    // v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second: with
    // (first, second) = (tile i, tile i+2) every lane then holds, in the two registers, its own half-sum of the
    // tile it keeps (i below lane 32, i+2 above) and the partner lane's half-sum of that same tile.  One VALU
    // instruction per exchange, no LDS crossbar round trip (ds_bpermute) and no selects.
    float s2[2], d2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        s2[i] = swap_add32(ss[i], ss[2 + i]);
        d2[i] = swap_add32(dt[i], dt[2 + i]);
    }
    // same one level down: rows of 16 lanes, odd rows keep the second tile of their pair
    const float s1 = swap_add16(s2[0], s2[1]);
    const float d1 = swap_add16(d2[0], d2[1]);
    // <v, tg> / max(|v|, 1e-12) (F.normalize's eps) as <v, tg> * rsq(max(|v|^2, 1e-24)): v_max + v_rsq + v_mul instead of a
    // correctly rounded sqrt and an IEEE division (~20 instructions of scaling and fix-up); v_rsq_f32 is good to 1 ulp,
    // the cosine to ~2e-7 relative, three orders inside the parity bar
    const float c = d1 * __builtin_amdgcn_rsqf(fmaxf(s1, 1e-24f));
    const float tot = wave_sum_dpp(c);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63)) * (1.0f / 64.0f);
}

// F.normalize(dim=1), dot with the unit-norm target, mean over the 64 positions (modules/modules.py:122,
// test_co3d.py:143) as a reduce-scatter: a lane holds 8 of the 32 channels of position (16t + lane&15) for the
// four tiles t; after two exchange steps lane (col, kq) owns the complete sums of ONE position (tile kq, column
// col), so the normalisation runs once per lane.
template <bool XDL>
__device__ __forceinline__ float hyp_score_rs(const f32x4 (&v)[2][4], const f32x4 (&tg)[4][2], int lane)
{
    float ss[4], dt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) hyp_tile_sums<XDL>(ss[t], dt[t], v[0][t], v[1][t], tg[t][0], tg[t][1]);
    return hyp_score_tail(ss, dt);
}

#ifdef AHV_DIAG_STAGE
// Diagnostic builds only (tools/team_stage_diff.py): instead of the score a hypothesis returns an order-independent XOR
// checksum of the BITS of one intermediate stage -- 1: GEMM1 pre-activations u, 2: head outputs v, 3: per-position sums of
// squares and dot products, 4: per-position cosines -- computed the same way by a lone wave and by a team, so that the first
// stage at which the two formulations differ can be read off.
__device__ __forceinline__ unsigned diag_wave_xor(unsigned x)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) x ^= (unsigned)__shfl_xor((int)x, sft, 64);
    return x;
}
__device__ __forceinline__ unsigned diag_bits4(const f32x4& a)
{
    return __float_as_uint(a[0]) ^ (__float_as_uint(a[1]) * 3u) ^ (__float_as_uint(a[2]) * 5u) ^ (__float_as_uint(a[3]) * 7u);
}
#endif

// A team member's share of hyp_score_rs (ahv_team.h): the sums of position tile t alone, associated exactly as
// hyp_score_tail associates them -- lane rows (kq 0 + kq 2) + (kq 1 + kq 3), the normalisation per position, the row_shr
// steps of wave_sum_dpp over the tile's 16 positions (rows 1-3 enter as zeros: x + 0 is exact).  Lane 63 returns P_t; the
// one-wave kernel's last two DPP steps make ((P_3 + P_2) + (P_1 + P_0)) of them, which is what member 0 does.
template <bool XDL>
__device__ __forceinline__ float team_tile_score(const f32x4 (&v)[2], const f32x4& g0, const f32x4& g1, int lane)
{
#pragma clang fp contract(off)
    float ss, dt;
    hyp_tile_sums<XDL>(ss, dt, v[0], v[1], g0, g1);
    ss += __shfl_xor(ss, 32, 64);
    dt += __shfl_xor(dt, 32, 64);
    ss += __shfl_xor(ss, 16, 64);
    dt += __shfl_xor(dt, 16, 64);
    const float c = lane < 16 ? dt * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)) : 0.0f;
#ifdef AHV_DIAG_STAGE
    if (AHV_DIAG_STAGE == 3) return __uint_as_float(diag_wave_xor(lane < 16 ? (__float_as_uint(ss) ^ (__float_as_uint(dt) * 3u)) : 0u));
    if (AHV_DIAG_STAGE == 4) return __uint_as_float(diag_wave_xor(lane < 16 ? __float_as_uint(c) : 0u));
#endif
    return wave_sum_dpp(c);
}

// Quarter `member` of the rotated volume into the wave's image `buf`, gathered exactly as the one-wave kernel gathers it:
// quarters 3 and 2 are the point mirrors of quarters 0 and 1 (hat_mirror, ahv_dual.h) -- same weights, same rows, same
// sums, bit for bit.  `done_ctr` / `done_target`: the image is free once the team's last round has been read.  Two
// straight-line paths (a merge between the row requests and the blend made hipcc keep two copies of the request ring).
__device__ __forceinline__ void team_gather(float* buf, const float* srcT, const GatherHyp& gh, const GatherDst& gdst, int member,
                                            unsigned* done_ctr, unsigned done_target)
{
    HatState st;
    if (member < 2) {
        hat_prologue_rt(st, srcT, gh, (float)member);
        team_wait(done_ctr, done_target);
        hat_body(st, buf, gdst);
    } else {
        const float qf = (float)(3 - member);
        hat_voxel_rt(st.vx[0], srcT, gh, 0, qf);
        hat_voxel_rt(st.vx[1], srcT, gh, 1, qf);
        hat_prologue_mirror(st, srcT);
        team_wait(done_ctr, done_target);
        hat_body<true>(st, buf, gdst);
    }
}

#ifdef AHV_STAMPS
// Diagnostic build only (tools/kbench.cpp): per-wave cycle sums of the loop segments.  The stamps
// leave the kernel through this buffer alone; no output value depends on them.
__device__ unsigned long long g_stamps[2048 * 16];
// per workgroup: 100 MHz real-time stamps at kernel entry, hypothesis-loop start and end, and the XCC id
__device__ unsigned long long g_wgstamps[1024 * 4];
__device__ unsigned long long g_wgclk[1024 * 2];  // s_memtime at loop start / end of the same workgroups
#define AHV_STAMP(var)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#else
#define AHV_STAMP(var)
#endif

// ---------------------------------------------------------------------------------------
// Dual variant: 512 threads = two waves per SIMD, 16x16x4 MFMA, W1 fragments in LDS (ahv_dual.h).
// ---------------------------------------------------------------------------------------
constexpr int kDualThreads = 512;

// SPLIT = true: GEMM1 on the f16 matrix pipe with hi/lo split operands (ahv_split.h; AHV_SCORE_SPLIT_F16).
// TGT = true: `tgt` is the target VOLUME [B][16][8][8][8] and team 1 of every workgroup builds forward_3d2d of it while
// the other waves are on their first hypothesis (ahv_verify_pair_f32); else `tgt` = the features [B][32][64].
// Hypotheses [0, n_main) go to single waves, [n_main, N) to teams of four (n_main = N: no teams).
// C2F = true (coarse_to_fine_kernel, ahv_coarse_to_fine_f32): TWO stages in one launch.  Stage 0 is the verify step above on
// the coarse set R; the workgroups of a sample then meet at a device-wide counter, read the winner's rotation R* and
// stage 1 scores the refinement set R* D[n] (composed per hypothesis from the 36 bytes of D[n], the arithmetic of
// compose_rotations_kernel) against the target features stage 0 left in LDS; the workgroup that finishes a sample last
// decodes both keys (what the two select launches did) and hands keys and counters back empty.
struct C2fArgs {
    const float* D;          // [N2][3][3] refinement rotations
    long N2, n_main2;        // stage 1's hypotheses and how many of them go to single waves
    key_t* best_key2;        // [B] stage 1's key (EMPTY on entry and on exit, like best_key)
    float* scores2;          // [B][N2] or NULL
    unsigned* sync;          // [2 B + 1]: per sample {arrived at the meeting point, finished}, then an error word; 0 on entry
    float* R_pred;           // [B][3][3]   = R* D[n*]
    float *fine_score, *coarse_score;  // [B]
    long *fine_idx, *coarse_idx;       // [B]
};
constexpr unsigned kC2fSpinLimit = 1u << 20;  // polls (~1 us each) before a workgroup gives the meeting point up: the
                                              // launch then ends with the error word set instead of hanging the device

// The exact path's share of one workgroup wave (ahv_exact.h): hypotheses h0, h0 + hstep, ... < N of sample-relative set R
// (Rstar != null: the coarse-to-fine stage-1 composition R* d).  Everything it needs comes in as arguments or from LDS --
// it reloads the GEMM2 fragments itself -- so that the call site keeps nothing alive for it.  Returns the running best key.
struct ExactCompose {  // by value: taking the address of the caller's R* would move it to scratch for the hot loop too
    float r[9];
    bool on;
};

struct NoCompose {  // the instances without a second stage pass nothing
    static constexpr bool on = false;
    static constexpr float r[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
};

template <bool SPLIT, typename Compose>
__device__ __attribute__((noinline)) key_t score_share_exact(const float* lds_src, const float* lds_w1, float* buf, const float* W1,
                                                             const float* W2, const float* b2, const float* Rb, Compose Rstar,
                                                             long N, long h0, long hstep, float* scores_b, long n_offset, key_t best,
                                                             int lane)
{
    // (`lane` comes in as an argument: a callee that reads threadIdx makes every caller keep the launch's work-item-id
    // register alive for it -- two spilled registers per lane at kernel entry, 1 MB of scratch stores per launch)
    const int rl = lane < 9 ? lane : 8;
    DualFrags f;
    load_dual_frags(f, W2, b2, lane);
    for (long he = h0; he < N; he += hstep) {
        const float Re = Rb[he * 9 + rl];
        float Rm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Re), i));
        if (Rstar.on) {  // compose_rotations_kernel's arithmetic, as the hot loop's hyp_rotation
            float d[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) d[i] = Rm[i];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Rm[a * 3 + c] = Rstar.r[a * 3] * d[c] + Rstar.r[a * 3 + 1] * d[3 + c] + Rstar.r[a * 3 + 2] * d[6 + c];
        }
        f32x4 acc[2][4];
        if constexpr (SPLIT) exact_hypothesis_gemm1(acc, buf, lds_src, Rm, W1FragsGlobal{W1, lane}, lane);
        else exact_hypothesis_gemm1(acc, buf, lds_src, Rm, W1FragsLds{reinterpret_cast<const f32x4*>(lds_w1) + lane}, lane);
        f32x4 v[2][4];
        gemm2_dual_exact(v, acc, f);
        f32x4 tg[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
                tg[t][m2] = *reinterpret_cast<const f32x4*>(lds_src + ((2 * t + m2) * 64 + lane) * kSrcStride + 16);
        const float s = hyp_score_rs<SPLIT || kFp32LowHalf>(v, tg, lane);
        if (scores_b != nullptr && lane == 0) scores_b[he] = s;
        const key_t key = pack_key(s, (unsigned)(n_offset + he));
        best = key > best ? key : best;
    }
    return best;
}

// SAVE_U (round 6, the TRAINING forward: ahv_score_hypotheses_train_f32): every hypothesis also leaves its pre-activations
// u = W1 slabs(rot(V, R_n)) -- the 32 accumulator registers GEMM1 ends with -- in `feat_tgt_out` (here: the u buffer), 8 KB per
// hypothesis in the wave's own fragment layout [t][m][lane][r] (ahv_backward.hip::du_word) (eight 16-byte stores per lane, 1 KB contiguous each).  The
// backward then reads them instead of recomputing gather + GEMM1 (ahv_backward.hip, score_backward_head_saved_kernel).
template <bool SPLIT, bool TGT, bool C2F, bool SAVE_U = false>
__device__ __forceinline__ void score_hypotheses_body(
    const float* __restrict__ vol_src, const float* __restrict__ tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, long n_main, float* __restrict__ scores,
    key_t* __restrict__ best_key, float* __restrict__ feat_tgt_out, unsigned long long* __restrict__ clk, const C2fArgs& cf)
{
    static_assert(!(SPLIT && TGT), "the split-f16 kernel takes ready-made target features");
    static_assert(!C2F || (TGT && !SPLIT), "the two-stage launch is the fp32 verify kernel");
    static_assert(!SAVE_U || (!SPLIT && !TGT && !C2F), "u is saved by the fp32 scorer with ready-made target features");
    // The fp32 instances own their SIMDs: touching v255 makes the kernel's register allocation 256 per wave whatever the
    // allocator needs, so the two waves of a SIMD hold all 512 registers and no wave of another kernel can be resident beside
    // them -- which is what keeps hipcc's packed op_sel forms safe here (low_half, ahv_dual.h; tests/test_isa_hazard.py).
    if constexpr (!SPLIT) asm volatile("" ::: "v255");
    // The source image goes FIRST in LDS (the backend lays the arrays out by descending alignment): the gather builds
    // a row's byte offset in fp32 and converts it once, so with the image at LDS address 0 the offset IS the address
    // (it sat at 0x1c000, too far for the 16-bit immediate of ds_read: one v_add_u32 per voxel).
    __shared__ __attribute__((aligned(1024))) float lds_src[kSrcFloats];
    __shared__ __attribute__((aligned(16))) float lds_w1[kW1TableFloats];
    __shared__ __attribute__((aligned(16))) float lds_q[8 * kQuarterFloats];
    __shared__ TeamSync lds_team;  // 84 of the 512 bytes the three images leave
    __shared__ NonFinite lds_nf;   // a NaN / inf among the voxels or head weights being staged (ahv_exact.h)
    // this workgroup's start: 100-MHz real time and shader clock for the diagnostic stamps (parked in LDS: registers that
    // live across the hypothesis loop are what this kernel has none to spare of)
    __shared__ unsigned long long lds_t_entry[2];
    if (clk != nullptr && threadIdx.x == 0) {
        lds_t_entry[0] = __builtin_amdgcn_s_memrealtime();
        lds_t_entry[1] = __builtin_amdgcn_s_memtime();
    }
#ifdef AHV_DIAG_CODE_SHIFT  // diagnostic builds only (tools/first_launch.cpp): moves all the code below by 4 bytes per unit,
    // i.e. to another position inside the 64-byte instruction-fetch lines
    asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(AHV_DIAG_CODE_SHIFT));
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* srcT = lds_src;
    float* buf = lds_q + wave * kQuarterFloats;

    static_assert(kSplitTableBytes == sizeof(float) * kW1TableFloats && kSplitImageBytes == sizeof(float) * kQuarterFloats, "LDS budget");

#ifdef AHV_STAMPS
    const unsigned long long wg_t_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long wg_t_loop = 0;
#endif
    if (!SPLIT && tid < (int)(sizeof(TeamSync) / 4)) reinterpret_cast<unsigned*>(&lds_team)[tid] = 0u;
    if (tid < (int)(sizeof(NonFinite) / 4)) reinterpret_cast<unsigned*>(&lds_nf)[tid] = 0u;
    // fp32 instances: EVERY global load of the prologue -- the W1 rows, the first sample's source volume, team 1's quarter
    // of the target volume (or the target features), W2 / b2 -- is issued here, before the first LDS store, so that the
    // workgroup pays ONE first-touch round trip instead of three in a row (W1 -> barrier -> volume -> barrier: 4.8 us of
    // prologue in round 4, profiles/r04z_small_launch_stamps.txt).  The registers are free: nothing else is live yet.
    SrcRegs sv;        // the NEXT sample's source voxels: loaded here for the first sample, at the bottom of the sample loop
    TgtRegs tv;        // for the following ones (never live across the hypothesis loop)
    W1Regs w1r;
    const bool w1_rows = (reinterpret_cast<unsigned long long>(W1) & 15ull) == 0;
    if (!SPLIT) {
        if (w1_rows) w1_regs_load(w1r, W1, tid);
        src_regs_load(sv, vol_src + (long)blockIdx.y * (16 * 512), tid);  // (blockIdx.y < B by construction of the grid)
        tgt_regs_load<TGT>(tv, tgt, (int)blockIdx.y, tid);
    }
    int w1_exp = 0, w2_exp = 0;
    if (SPLIT) {
        float m = 0.0f;
        bool bad = non_finite(W2[tid]) || non_finite(W2[tid + 512]) || non_finite(b2[tid & 31]);
        for (int i = tid; i < 32 * 384; i += kDualThreads) {
            m = fmaxf(m, fabsf(W1[i]));
            bad = bad || non_finite(W1[i]);
        }
        w1_exp = split_prescale_exp(block_absmax(m, lds_q, tid));  // (block_absmax: two barriers behind the zeroing above)
        if (bad) lds_nf.weights = 1u;
        stage_w1_split(reinterpret_cast<f16x8*>(lds_w1), W1, ldexpf(1.0f, w1_exp), tid, kDualThreads);
        w2_exp = split_prescale_exp(block_absmax(fmaxf(fabsf(W2[tid]), fabsf(W2[tid + 512])), lds_q, tid));
    }
    DualFrags f0;
    bool w_bad = false;  // fp32 instances: this thread saw a non-finite head weight
    if (!SPLIT) {
        load_dual_frags(f0, W2, b2, lane);
        bool bad = false;
        if (w1_rows) {
            w1_regs_store(lds_w1, w1r, tid);
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad = bad || non_finite(w1r.w[k][e]);
        } else {  // a view that starts off a 16-byte boundary: float by float
            stage_w1_table(lds_w1, W1, tid, kDualThreads);
            for (int i = tid; i < 32 * 384; i += kDualThreads) bad = bad || non_finite(W1[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) bad = bad || non_finite((&f0.a2[0][0][0])[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) bad = bad || non_finite(f0.bias[i >> 2][i & 3]);
        w_bad = bad;  // published behind the first staging barrier (the zeroing of lds_nf above is another wave's store)
    }
    const int team = wave >> 2, member = wave & 3;
    unsigned team_rounds = 0;  // exchanges this wave's team has completed (wave-uniform; the counters of TeamSync only grow)
    unsigned samples_done = 0;
    const GatherLane glane = gather_lane(lane);
    const GatherDst gdst = gather_dst_swizzled(lane);
    const SplitDst sdst = split_dst(lane);
    SplitResident w1res;
    if (SPLIT) {
        __syncthreads();  // the table is staged
        split_load_resident(w1res, reinterpret_cast<const f16x8*>(lds_w1), lane);
    }
    const long hstep = (long)gridDim.x * 8;
    // Diagnostic entry point (clk != NULL): shader-clock and 100 MHz real-time stamps around this workgroup's whole run
    // (lds_t_entry above, the end stamps below).  The stamps go to `clk` alone; no output depends on them.

    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        // The thread / lane numbers of this block go through an empty asm, so that the per-thread global and LDS addresses of
        // the staging code are built HERE, once per sample: hoisted out of the sample loop they live across the hypothesis
        // loop, i.e. in scratch (2-4 MB of spill stores per launch showed up as WRITE_SIZE, profiles/r04c_pmc_summary.json).
        const int lane_s = opaque(lane), tid_s = wave * 64 + lane_s;  // (not from threadIdx.x: that register may die here)
        const unsigned gen = samples_done + 1u;  // generation of this sample's non-finite flags (ahv_exact.h)
        if (w_bad) lds_nf.weights = 1u;
        DualFrags f = f0;
        Split2Frags w2s;
        int exp_sum = 0;  // of the three prescales of this sample: they leave again behind GEMM2 (gemm2_split)
        if (SPLIT) {
            load_dual_frags(f, W2, b2, lane_s);
            split_w2_frags(w2s, f, ldexpf(1.0f, w2_exp));
            bool bad;
            exp_sum = w2_exp + w1_exp + stage_src_volume_scaled(lds_src, vol_src + (long)b * (16 * 512), lds_q, tid_s, bad);
            if (bad) lds_nf.src_gen = gen;
        } else {
            src_regs_store(lds_src, sv, tid_s);
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 16; ++c) bad = bad || non_finite(sv.x[c]);
            if (bad) lds_nf.src_gen = gen;
            if constexpr (TGT) {
                bool tbad = false;
#pragma unroll
                for (int c = 0; c < 16; ++c) tbad = tbad || non_finite(tv.q[c].x) || non_finite(tv.q[c].y);
                if (tbad) lds_nf.tgt_gen = gen;  // (threads of team 0 hold zeros)
            }
            // TGT: team 1's un-rotated quarters of the TARGET volume go into its members' private images here, ahead of
            // the staging barrier (the images are free between the two barriers).  Else: the target features as the
            // per-lane fragments the score needs, parked in the 16-byte pad of source rows 0..511 (row (2t + m2)*64 +
            // lane): 32 registers less per wave, and the 80-byte row stride makes the eight ds_read_b128 of the
            // epilogue conflict-free.
            tgt_regs_store<TGT>(lds_src, buf, tv, tid_s);
        }
        if (SPLIT) {
            const float* ft = tgt + (long)b * (32 * 64);
            const int l = tid_s & 63, t = tid_s >> 7, m2 = (tid_s >> 6) & 1;
            f32x4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = ft[(16 * m2 + 4 * (l >> 4) + r) * 64 + 16 * t + (l & 15)];
            *reinterpret_cast<f32x4*>(lds_src + tid_s * kSrcStride + 16) = x;
        }
        __syncthreads();
        // a sample with a NaN / inf voxel or head weight leaves the hot loop for the exact path (ahv_exact.h); uniform
        const bool nf_w = __builtin_amdgcn_readfirstlane((int)lds_nf.weights) != 0;
#ifdef AHV_DIAG_NO_EXACT  // diagnostic builds (tools/kbench): without the exact path's call, to price its mere presence
        const bool exact = false;
#else
        const bool exact = nf_w || (unsigned)__builtin_amdgcn_readfirstlane((int)lds_nf.src_gen) == gen;
#endif
        if constexpr (SPLIT) {
            if (nf_w) {  // non-finite WEIGHTS: the exact path reads them unscaled, so the volume must be unscaled too
                restage_src_volume_unscaled(lds_src, vol_src + (long)b * (16 * 512), wave * 64 + opaque(lane));
                __syncthreads();
            }
        }
        bool tg_ready = !TGT;
        if constexpr (TGT) {
            // forward_3d2d(vol_tgt[b]) (test_co3d.py:141, modules/modules.py:112-124) by team 1, one quarter per wave, while
            // team 0 is on its first hypotheses: each member stages its un-rotated quarter straight from global memory
            // (R = I needs no gather), contracts it, the members exchange the z partials and member t finishes position
            // tile t and parks ITS eight rows of the fragment layout above.  Everybody else meets the result at the first
            // use (tg_wait below); the ~0.2 hypotheses' worth of work lands on the four waves that have one hypothesis
            // less than their SIMD partners whenever the last round is partial.
            if (team == 1) {
                f32x4 u[2], v[2];
                team_round(u, lds_w1, lds_q + 4 * kQuarterFloats, lds_team, 1, member, team_rounds, lane_s);
                ++team_rounds;
                if (nf_w || (unsigned)__builtin_amdgcn_readfirstlane((int)lds_nf.tgt_gen) == gen) team_head_exact(v, u, f0);
                else team_head(v, u, f0);
                const float inv = 1.0f / fmaxf(sqrtf(team_sumsq(v)), 1e-12f);  // F.normalize(dim=1), eps 1e-12
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const f32x4 x = v[m2] * inv;
                    *reinterpret_cast<f32x4*>(lds_src + ((2 * member + m2) * 64 + lane_s) * kSrcStride + 16) = x;
                    if (feat_tgt_out != nullptr && blockIdx.x == 0) {
                        const int ol = lane_s;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            feat_tgt_out[(long)b * (32 * 64) + (16 * m2 + 4 * (ol >> 4) + r) * 64 + 16 * member + (ol & 15)] = x[r];
                    }
                }
                team_signal(&lds_team.tgt_ready, lane_s);
                team_wait(&lds_team.done[1], 4u * team_rounds);  // my image is free again (the teammates have read it)
            }
        }
        auto tg_wait = [&]() {
            if (TGT && !tg_ready) {
                team_wait(&lds_team.tgt_ready, 4u * (samples_done + 1u));
                tg_ready = true;
            }
        };
#ifdef AHV_STAMPS
        wg_t_loop = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wg_c_loop = __builtin_amdgcn_s_memtime();
#endif
        const float* Rb0 = R + (long)b * r_batch_stride;
        const int residue = xcd_residue(blockIdx.x, gridDim.x, gridDim.y);
        float Rstar[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // C2F stage 1: the coarse winner's rotation (uniform)
        // hypothesis rotation from the nine values a wave has just broadcast: stage 1 composes R* d (compose_rotations_kernel)
        auto hyp_rotation = [&](float (&Rm)[9], bool compose) {
            if (C2F && compose) {
                float d[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) d[i] = Rm[i];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Rm[a * 3 + c] = Rstar[a * 3] * d[c] + Rstar[a * 3 + 1] * d[3 + c] + Rstar[a * 3 + 2] * d[6 + c];
            }
        };
      for (int stage = 0; stage < (C2F ? 2 : 1); ++stage) {
        // per-stage views (stage 1 exists in the C2F instance only)
        const bool st1 = C2F && stage != 0;
        const float* Rb = st1 ? cf.D : Rb0;
        const long N_s = st1 ? cf.N2 : N, n_main_s = st1 ? cf.n_main2 : n_main, n_offset_s = st1 ? 0l : n_offset;
        float* scores_s = st1 ? cf.scores2 : scores;
        key_t* key_s = st1 ? cf.best_key2 : best_key;
        if (st1) {
            // Meeting point of the sample's gridDim.x workgroups: every atomicMax of stage 0 is ahead of the add (workgroup
            // barrier, then a release add at agent scope); the poll is an acquire load at agent scope (sc1: the XCDs' L2s
            // are not coherent with each other for plain loads).  All workgroups of the grid are resident -- at most one
            // per CU by construction of the plan -- so the wait ends; if something else holds CUs for a second it gives up.
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cf.sync + 2 * b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned polls = 0;
                while (__hip_atomic_load(cf.sync + 2 * b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++polls > kC2fSpinLimit) {
                        __hip_atomic_fetch_or(cf.sync + 2 * B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
            const key_t kc = __hip_atomic_load(best_key + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long idx = key_index(kc) - n_offset;
            idx = (kc == kKeyEmpty || idx < 0 || idx >= N) ? 0 : idx;  // as compose_rotations_kernel: stay in bounds
#pragma unroll
            for (int i = 0; i < 9; ++i)
                Rstar[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Rb0[idx * 9 + i])));
        }
        key_t best = kKeyEmpty;
        const int rl = lane < 9 ? lane : 8;  // lane i < 9 fetches element i of a rotation
        auto score_remainder_by_teams = [&]() {
            // The remainder [n_main, N): one hypothesis per TEAM (ahv_team.h), scored FIRST -- at the head of a launch the
            // workgroup's other team is busy too (its own remainder hypothesis, the target features, or the first main-round
            // hypotheses), so the team's ~1/4-hypothesis pieces share the SIMDs with useful work; behind the main rounds a
            // team round is pure latency on an emptying CU.  Team slot g = team * gridDim.x + residue, so the remainder
            // spreads over all workgroups first and over their second team next.  Every member runs the same rounds;
            // member 0 emits the score of round r after the arrive point of round r + 1 (or of the flush below), where the
            // four partial sums of round r are known to be in place.  Scores are a lone wave's bit for bit.
            const long tstep = 2l * gridDim.x;
            long ht = n_main_s + (long)team * gridDim.x + residue;
            if (ht < N_s) {
                long h_prev = -1;
                float Rt = Rb[ht * 9 + rl];
                auto team_emit = [&]() {  // member 0, behind an arrive point: the partials of the round before are complete
                    const float* pp = lds_team.part[team][(team_rounds + 1u) & 1u];
#ifdef AHV_DIAG_STAGE
                    const float sc = __uint_as_float(__float_as_uint(pp[0]) ^ __float_as_uint(pp[1]) ^ __float_as_uint(pp[2]) ^ __float_as_uint(pp[3]));
#else
                    const float sc = ((pp[3] + pp[2]) + (pp[1] + pp[0])) * (1.0f / 64.0f);  // wave_sum_dpp's last two steps
#endif
                    if (scores_s != nullptr && lane == 0) scores_s[(long)b * N_s + h_prev] = sc;
                    const key_t key = pack_key(sc, (unsigned)(n_offset_s + h_prev));
                    best = key > best ? key : best;
                };
                for (; ht < N_s; ht += tstep) {
                    float Rm[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Rt), i));
                    hyp_rotation(Rm, st1);
                    if (ht + tstep < N_s) Rt = Rb[(ht + tstep) * 9 + rl];
                    GatherHyp gh;
                    gather_hyp(gh, Rm, glane);
                    team_gather(buf, srcT, gh, gdst, member, &lds_team.done[team], 4u * team_rounds);
                    wave_lds_fence();
                    f32x4 u[2], v[2];
                    team_round(u, lds_w1, lds_q + 4 * team * kQuarterFloats, lds_team, team, member, team_rounds, lane);
                    if (member == 0 && h_prev >= 0) team_emit();
                    tg_wait();
                    team_head(v, u, f);
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(lds_src + ((2 * member) * 64 + lane) * kSrcStride + 16);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(lds_src + ((2 * member + 1) * 64 + lane) * kSrcStride + 16);
                    float tot = team_tile_score<kFp32LowHalf>(v, g0, g1, lane);
#ifdef AHV_DIAG_STAGE
                    if (AHV_DIAG_STAGE == 1) tot = __uint_as_float(diag_wave_xor(diag_bits4(u[0]) ^ (diag_bits4(u[1]) * 11u)));
                    if (AHV_DIAG_STAGE == 2) tot = __uint_as_float(diag_wave_xor(diag_bits4(v[0]) ^ (diag_bits4(v[1]) * 11u)));
#endif
                    if (lane == 63) lds_team.part[team][team_rounds & 1u][member] = tot;
                    ++team_rounds;
                    h_prev = ht;
                }
                // flush: one more meeting point, then member 0 emits the last round's score
                team_signal(&lds_team.arrive[team], lane);
                team_wait(&lds_team.arrive[team], 4u * (team_rounds + 1u));
                if (member == 0) team_emit();
                team_signal(&lds_team.done[team], lane);  // keeps arrive and done in step: one of each per meeting point
                ++team_rounds;
                team_wait(&lds_team.done[team], 4u * team_rounds);  // (the main rounds below store into the image unasked)
            }
        };
        if (exact) {
            // A NaN / inf among this sample's voxels or the head weights: every hypothesis of the workgroup's share, main
            // rounds and remainder alike, one per wave through the exact path (ahv_exact.h) -- grid_sample's per-corner
            // zeros padding and F.relu's NaN propagation restated literally.  Finite samples never get here.  A CALL, not
            // inlined: inlined, the slow path's live ranges pushed the allocator into spilling inside the hot loop (which
            // sits at 256 registers); behind a call boundary it cannot touch the hot loop's allocation.
            if constexpr (TGT) tg_wait();
            const long h0 = (long)wave * gridDim.x + residue;
            float* sc_b = scores_s != nullptr ? scores_s + (long)b * N_s : nullptr;
            if constexpr (C2F) {
                ExactCompose ec;
#pragma unroll
                for (int i = 0; i < 9; ++i) ec.r[i] = Rstar[i];
                ec.on = st1;
                best = score_share_exact<SPLIT>(lds_src, lds_w1, buf, W1, W2, b2, Rb, ec, N_s, h0, hstep, sc_b, n_offset_s, best, lane);
            } else {
                best = score_share_exact<SPLIT>(lds_src, lds_w1, buf, W1, W2, b2, Rb, NoCompose{}, N_s, h0, hstep, sc_b, n_offset_s, best, lane);
            }
            // (re-read rather than kept alive across the call: kept alive, the resident W1 fragments were saved to scratch at
            // their definition -- once per wave and launch, 5 MB of stores -- for a path finite data never takes)
            if constexpr (SPLIT) split_load_resident(w1res, reinterpret_cast<const f16x8*>(lds_w1), lane);
            if constexpr (SAVE_U) {
                // the exact path keeps no accumulators: a sample with a non-finite voxel or weight hands NaN pre-activations
                // to the backward, which turns them into NaN gradients for that sample (like autograd would)
                const f32x4 qn = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                for (long he = h0; he < N_s; he += hstep) {
                    float* uo = feat_tgt_out + ((long)b * N_s + he) * 2048 + lane * 4;
#pragma unroll
                    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(uo + i * 256) = qn;
                }
            }
        } else {
#ifndef AHV_DIAG_TEAMS_LAST
        if constexpr (!SPLIT && !SAVE_U) score_remainder_by_teams();   // (the training forward runs single waves only: n_main = N)
#endif
        // Hypothesis h -> (workgroup h % gridDim.x, wave slot (h / gridDim.x) % 8): a partial last round of the persistent
        // grid spreads over ALL CUs with few waves each instead of filling some CUs completely and leaving the rest idle.
        // The deal inside a workgroup is STATIC.  The older wave of a SIMD issues first and finishes its share ~110 us before
        // its partner (in-kernel stamps); letting every wave CLAIM its next slot from an LDS counter keeps the pairs together
        // to the end -- and is not faster: 0.6902 against 0.6838 ms at N = 50 000, 0.358 against 0.348 at 25 000, equal at
        // 6 250, 1 % faster only at 200 000 (profiles/r04j_wave_claim.txt).  The workgroup's time is set by what its CU gets
        // through, not by how its waves share it.
        long h = (long)wave * gridDim.x + residue;
        // The rotation of the NEXT hypothesis travels as ONE vector load (lane i < 9 fetches element i) and is
        // broadcast with v_readlane at the top of the next iteration.  Not as scalar loads: SMEM shares lgkmcnt
        // with the LDS and returns out of order, so the first LDS wait behind an s_load has to wait for the s_load
        // too -- a first-touch read of R from HBM (~2 us) in front of every hypothesis' first gather step.
        float Rn = h < n_main_s ? Rb[h * 9 + rl] : 0.0f;  // nothing of R is touched when this wave has no hypothesis (N = 0: R may be null)
#ifdef AHV_STAMPS
        unsigned long long tsum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (; h < n_main_s; h += hstep) {
            float Rm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Rn), i));
            hyp_rotation(Rm, st1);
            {
                const long hn = (h + hstep < n_main_s) ? h + hstep : h;
                Rn = Rb[hn * 9 + rl];
            }
            f32x4 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef AHV_STAMPS
            unsigned long long ts[11];
#define AHV_TS(i) AHV_STAMP(ts[i])
#else
#define AHV_TS(i)
#endif
            AHV_TS(0)
            if constexpr (SPLIT) {
                char* cbuf = reinterpret_cast<char*>(buf);
                const f16x8* w1s = reinterpret_cast<const f16x8*>(lds_w1);
                GatherHyp gh;
                gather_hyp(gh, Rm, glane);
                HatState st;
                // quarters in the order 0, 3, 1, 2 (point mirror, as in the fp32 path); the prologues run between the GEMMs:
                // the request ring does not fit beside the GEMM's operands
                hat_prologue<0, true, kSplitHatDepth>(st, srcT, gh);
                hat_body_split(st, cbuf, sdst); wave_lds_fence(); AHV_TS(1)
                gemm1_quarter_split<0>(acc, w1res, w1s, cbuf, lane, [] {}); wave_lds_fence(); AHV_TS(2)
                hat_prologue_mirror<kSplitHatDepth>(st, srcT);
                hat_body_split<true>(st, cbuf, sdst); wave_lds_fence(); AHV_TS(3)
                gemm1_quarter_split<3>(acc, w1res, w1s, cbuf, lane, [] {}); wave_lds_fence(); AHV_TS(4)
                hat_prologue<1, true, kSplitHatDepth>(st, srcT, gh);
                hat_body_split(st, cbuf, sdst); wave_lds_fence(); AHV_TS(5)
                gemm1_quarter_split<1>(acc, w1res, w1s, cbuf, lane, [] {}); wave_lds_fence(); AHV_TS(6)
                hat_prologue_mirror<kSplitHatDepth>(st, srcT);
                hat_body_split<true>(st, cbuf, sdst); wave_lds_fence(); AHV_TS(7)
                gemm1_quarter_split<2>(acc, w1res, w1s, cbuf, lane, [] {}); wave_lds_fence(); AHV_TS(8)
            } else {
                // gather quarter q (16 blend steps) -> GEMM1 on it; the head of quarter q+1's gather (coordinates,
                // weights, first row requests) is issued from inside GEMM q, ahead of its last MFMA chunks
                GatherHyp gh;
                gather_hyp(gh, Rm, glane);
                HatState st;
                // quarters in the order 0, 3, 1, 2: quarter 3 - Q is the point mirror of quarter Q and reuses its set-up
                hat_prologue<0>(st, srcT, gh);
                hat_body(st, buf, gdst); wave_lds_fence(); AHV_TS(1)
                gemm1_quarter_pipe<0>(acc, lds_w1, buf, lane, [&] { hat_prologue_mirror(st, srcT); }); wave_lds_fence(); AHV_TS(2)
                hat_body<true>(st, buf, gdst); wave_lds_fence(); AHV_TS(3)
                gemm1_quarter_pipe<3>(acc, lds_w1, buf, lane, [&] { hat_prologue<1>(st, srcT, gh); }); wave_lds_fence(); AHV_TS(4)
                hat_body(st, buf, gdst); wave_lds_fence(); AHV_TS(5)
                gemm1_quarter_pipe<1>(acc, lds_w1, buf, lane, [&] { hat_prologue_mirror(st, srcT); }); wave_lds_fence(); AHV_TS(6)
                hat_body<true>(st, buf, gdst); wave_lds_fence(); AHV_TS(7)
                gemm1_quarter_pipe<2>(acc, lds_w1, buf, lane, [] {}); wave_lds_fence(); AHV_TS(8)
            }

            tg_wait();
            float s;
            if constexpr (SPLIT) {
                // tile by tile: the 8 outputs of a tile go straight into its sums (32 registers of v never exist at once)
                const Gemm2Scale sc = gemm2_split_scale(acc, exp_sum);
                float ss[4], dt[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(lds_src + ((2 * t) * 64 + lane) * kSrcStride + 16);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(lds_src + ((2 * t + 1) * 64 + lane) * kSrcStride + 16);
                    f32x4 v0, v1;
                    gemm2_split_tile(v0, v1, acc[0][t], acc[1][t], w2s, f.bias, sc);
                    hyp_tile_sums<true>(ss[t], dt[t], v0, v1, g0, g1);
                }
                AHV_TS(9)
                s = hyp_score_tail(ss, dt);
            } else {
                if constexpr (SAVE_U) {
                    float* uo = feat_tgt_out + ((long)b * N_s + h) * 2048 + lane * 4;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(uo + (t * 2 + m) * 256) = acc[m][t];
                }
                f32x4 v[2][4];
                gemm2_dual(v, acc, f);
                AHV_TS(9)
                f32x4 tg[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
                        tg[t][m2] = *reinterpret_cast<const f32x4*>(lds_src + ((2 * t + m2) * 64 + lane) * kSrcStride + 16);
                s = hyp_score_rs<kFp32LowHalf>(v, tg, lane);
#ifdef AHV_DIAG_STAGE
                {
                    unsigned x = 0;
                    if (AHV_DIAG_STAGE == 1) for (int t = 0; t < 4; ++t) x ^= diag_bits4(acc[0][t]) ^ (diag_bits4(acc[1][t]) * 11u);
                    if (AHV_DIAG_STAGE == 2) for (int t = 0; t < 4; ++t) x ^= diag_bits4(v[0][t]) ^ (diag_bits4(v[1][t]) * 11u);
                    if (AHV_DIAG_STAGE >= 3) {
                        float ss[4], dt[4];
                        for (int t = 0; t < 4; ++t) hyp_tile_sums<kFp32LowHalf>(ss[t], dt[t], v[0][t], v[1][t], tg[t][0], tg[t][1]);
                        const float s1 = swap_add16(swap_add32(ss[0], ss[2]), swap_add32(ss[1], ss[3]));
                        const float d1 = swap_add16(swap_add32(dt[0], dt[2]), swap_add32(dt[1], dt[3]));
                        const float c = d1 * __builtin_amdgcn_rsqf(fmaxf(s1, 1e-24f));
                        x = AHV_DIAG_STAGE == 3 ? (__float_as_uint(s1) ^ (__float_as_uint(d1) * 3u)) : __float_as_uint(c);
                    }
                    s = __uint_as_float(diag_wave_xor(x));
                }
#endif
            }
            if (scores_s != nullptr && lane == 0) scores_s[(long)b * N_s + h] = s;
            const key_t key = pack_key(s, (unsigned)(n_offset_s + h));
            best = key > best ? key : best;
            AHV_TS(10)
#ifdef AHV_STAMPS
            for (int i = 0; i < 10; ++i) tsum[i] += ts[i + 1] - ts[i];
            tsum[10] += 1;
#endif
        }
#ifdef AHV_DIAG_TEAMS_LAST  // diagnostic builds (tools/kbench): the remainder behind the main rounds, as in round 4
        if constexpr (!SPLIT) score_remainder_by_teams();
#endif
#ifdef AHV_STAMPS
        if (lane == 0) {
            const int gw = (blockIdx.x * 8 + wave) & 2047;
            for (int i = 0; i < 11; ++i) g_stamps[gw * 16 + i] = tsum[i];
        }
#endif
        }  // finite sample
        if (key_s != nullptr && lane == 0 && best != kKeyEmpty) atomicMax(key_s + b, best);
      }  // stage
        ++samples_done;
        if (!SPLIT) {
            // The next sample's loads travel while the workgroup's last waves finish this one.  UNCONDITIONAL (the last sample
            // re-reads itself): a conditional definition would keep the old values alive across the hypothesis loop.
            const int bn = b + (int)gridDim.y < B ? b + (int)gridDim.y : b;
            const int tid_e = wave * 64 + opaque(lane);  // (a fresh copy: tid_s must not stay alive across the hypothesis loop)
            src_regs_load(sv, vol_src + (long)bn * (16 * 512), tid_e);
            tgt_regs_load<TGT>(tv, tgt, bn, tid_e);
        }
        if constexpr (C2F) {
            // The workgroup that finishes the sample last decodes both keys: what select_rotation_kernel did, twice.
            __syncthreads();
            if (tid == 0) {
                const unsigned before = __hip_atomic_fetch_add(cf.sync + 2 * b + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (before == gridDim.x - 1) {
                    const key_t kc = __hip_atomic_load(best_key + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const key_t kf = __hip_atomic_load(cf.best_key2 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // A workgroup that gave the meeting point up (something else held CUs for ~1 s) has scored stage 1 against
                    // an incomplete coarse key: the sticky error word is set, and the results are POISONED -- NaN scores,
                    // index -1, NaN rotation -- so that the failure cannot pass for a result even if nobody reads the word.
                    const bool gave_up = __hip_atomic_load(cf.sync + 2 * B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    const float nanv = __builtin_nanf("");
                    if (cf.coarse_score) cf.coarse_score[b] = gave_up ? nanv : (kc == kKeyEmpty) ? -INFINITY : key_score(kc);
                    if (cf.coarse_idx) cf.coarse_idx[b] = (gave_up || kc == kKeyEmpty) ? -1l : key_index(kc);
                    if (cf.fine_score) cf.fine_score[b] = gave_up ? nanv : (kf == kKeyEmpty) ? -INFINITY : key_score(kf);
                    const long fi = (gave_up || kf == kKeyEmpty) ? -1l : key_index(kf);
                    if (cf.fine_idx) cf.fine_idx[b] = fi;
                    if (cf.R_pred) {
                        float Rm[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) Rm[i] = (fi >= 0 && fi < cf.N2) ? cf.D[fi * 9 + i] : (gave_up ? nanv : 0.0f);
                        if (fi >= 0 && fi < cf.N2) hyp_rotation(Rm, true);
#pragma unroll
                        for (int i = 0; i < 9; ++i) cf.R_pred[b * 9 + i] = Rm[i];
                    }
                    // keys and counters back empty for the next launch (everybody is past the meeting point: they all counted)
                    __hip_atomic_store(best_key + b, kKeyEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cf.best_key2 + b, kKeyEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cf.sync + 2 * b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cf.sync + 2 * b + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
#ifdef AHV_STAMPS
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = g_wgstamps + 4 * (blockIdx.x & 1023);
            o[0] = wg_t_entry;
            o[1] = wg_t_loop;
            o[2] = __builtin_amdgcn_s_memrealtime();
            o[3] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);  // HW_REG_XCC_ID, bits 3:0
            g_wgclk[2 * (blockIdx.x & 1023)] = wg_c_loop;
            g_wgclk[2 * (blockIdx.x & 1023) + 1] = __builtin_amdgcn_s_memtime();
        }
#endif
    }
    if (clk != nullptr) {
        __syncthreads();
        if (wave == 0 && lane == 0) {
            unsigned long long* o = clk + 4 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = lds_t_entry[1];
            o[1] = lds_t_entry[0];
            o[2] = __builtin_amdgcn_s_memtime();
            o[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

template <bool SPLIT, bool TGT>
__global__ __launch_bounds__(kDualThreads, 2) void score_hypotheses_dual_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, long n_main, float* __restrict__ scores,
    key_t* __restrict__ best_key, float* __restrict__ feat_tgt_out, unsigned long long* __restrict__ clk)
{
    score_hypotheses_body<SPLIT, TGT, false>(vol_src, tgt, R, r_batch_stride, n_offset, W1, W2, b2, B, N, n_main, scores, best_key,
                                             feat_tgt_out, clk, C2fArgs{});
}

// the training forward: scores + the pre-activations of every hypothesis (SAVE_U above); single waves only
__global__ __launch_bounds__(kDualThreads, 2) void score_hypotheses_train_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ tgt, const float* __restrict__ R,
    long r_batch_stride, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, float* __restrict__ scores, float* __restrict__ u_out)
{
    score_hypotheses_body<false, false, false, true>(vol_src, tgt, R, r_batch_stride, 0l, W1, W2, b2, B, N, N, scores, nullptr, u_out,
                                                     nullptr, C2fArgs{});
}

// the two-stage launch (C2fArgs above): `tgt` is the target volume, R the coarse set
__global__ __launch_bounds__(kDualThreads, 2) void coarse_to_fine_kernel(
    const float* __restrict__ vol_src, const float* __restrict__ tgt, const float* __restrict__ R,
    long r_batch_stride, long n_offset, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ b2, int B, long N, long n_main, float* __restrict__ scores,
    key_t* __restrict__ best_key, float* __restrict__ feat_tgt_out, C2fArgs cf)
{
    score_hypotheses_body<false, true, true>(vol_src, tgt, R, r_batch_stride, n_offset, W1, W2, b2, B, N, n_main, scores, best_key,
                                             feat_tgt_out, nullptr, cf);
}

__global__ void unpack_best_kernel(const key_t* __restrict__ best_key, int B, float* __restrict__ best_score,
                                   long* __restrict__ best_idx)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const key_t k = best_key[b];
    if (best_score) best_score[b] = (k == kKeyEmpty) ? -INFINITY : key_score(k);
    if (best_idx) best_idx[b] = (k == kKeyEmpty) ? -1l : key_index(k);
}

}  // namespace ahv

// ---- host-side launchers (called by the C ABI in ahv_abi.hip) ----------------------
namespace ahv {

// How a launch covers N hypotheses with gx workgroups of 8 waves (per sample).  Hypotheses [0, n_main) go to single
// waves, round after round of 8 * gx; a remainder of at most two per workgroup goes to teams of four waves (ahv_team.h),
// one hypothesis per team, scored AHEAD of the main rounds.  A team's score is the lone wave's bit for bit, so the rule is
// pure scheduling -- no result depends on N or on how a set is sharded.
// Round 4 ran the team rounds BEHIND the main rounds (tools/kbench, profiles/r04d_team_tail.txt; kernel time in us,
// teams / single waves): there a team round is latency on an emptying CU and only paid behind >= 4 full rounds,
//     N = 2 154 (1 round + 106)  53.6 / 56.8      N = 12 500 (6 rounds + 212)  181.5 / 185.7
//     N = 6 250 (3 rounds + 106) 103.8 / 103.5    N = 25 000 (12 rounds + 424) 349.9 / 353.7
//     N = 6 400 (3 rounds + 256) 112.4 / 104.3
// while a remainder left to single waves costs the wave that owns it a whole extra hypothesis at the end of a short launch.
// At the head of a launch the other team of the workgroup is busy too (its own remainder hypothesis, the target features
// of ahv_verify_pair_f32, or its first main-round hypotheses): profiles/r05_team_head.txt.
ScorePlan plan_score_launch(int B, int64_t N, int num_cu, int spare_cu, bool teams)
{
    ScorePlan p;
    int avail = num_cu - spare_cu;
    if (avail < 1) avail = 1;
    p.gy = B < avail ? B : avail;
    if (p.gy < 1) p.gy = 1;
    int64_t gx = avail / p.gy;
    // small N: spread over as many CUs as there is work for -- two hypotheses per workgroup when teams can take them
    // (one per team), else four (one wave per SIMD runs a hypothesis ~1.6x faster than two waves sharing the SIMD)
    const int64_t want = teams ? (N + 1) / 2 : (N + 3) / 4;
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    p.gx = (int)gx;
    const int64_t slots = 8 * gx, full = N / slots, rem = N - full * slots;
    p.n_main = (teams && rem > 0 && rem <= 2 * gx) ? full * slots : N;
    return p;
}

hipError_t launch_score_hypotheses(const ScoreLaunch& a, hipStream_t stream)
{
    static_assert(sizeof(key_t) == sizeof(int64_t), "key width");
    const bool teams = !a.split_f16 && !a.no_teams;
    const ScorePlan p = plan_score_launch(a.B, a.N, a.num_cu, a.spare_cu, teams);
    const dim3 grid(p.gx, p.gy), block(kDualThreads);
    auto* key = reinterpret_cast<key_t*>(a.best_key);
    auto* clk = reinterpret_cast<unsigned long long*>(a.clock_stamps);
    if (a.split_f16) {
        if (a.tgt_is_volume) return hipErrorInvalidValue;  // the ABI routes this case through forward_3d2d first
        hipLaunchKernelGGL((score_hypotheses_dual_kernel<true, false>), grid, block, 0, stream, a.vol_src, a.tgt, a.R,
                           (long)a.r_batch_stride, (long)a.n_offset, a.W1, a.W2, a.b2, a.B, (long)a.N, (long)p.n_main, a.scores,
                           key, a.feat_tgt_out, clk);
    } else if (a.tgt_is_volume) {
        hipLaunchKernelGGL((score_hypotheses_dual_kernel<false, true>), grid, block, 0, stream, a.vol_src, a.tgt, a.R,
                           (long)a.r_batch_stride, (long)a.n_offset, a.W1, a.W2, a.b2, a.B, (long)a.N, (long)p.n_main, a.scores,
                           key, a.feat_tgt_out, clk);
    } else {
        hipLaunchKernelGGL((score_hypotheses_dual_kernel<false, false>), grid, block, 0, stream, a.vol_src, a.tgt, a.R,
                           (long)a.r_batch_stride, (long)a.n_offset, a.W1, a.W2, a.b2, a.B, (long)a.N, (long)p.n_main, a.scores,
                           key, a.feat_tgt_out, clk);
    }
    return hipGetLastError();
}

hipError_t launch_score_hypotheses_train(const ScoreLaunch& a, float* u_out, hipStream_t stream)
{
    if (a.split_f16 || a.tgt_is_volume || a.scores == nullptr || u_out == nullptr) return hipErrorInvalidValue;
    const ScorePlan p = plan_score_launch(a.B, a.N, a.num_cu, a.spare_cu, /*teams=*/false);
    hipLaunchKernelGGL(score_hypotheses_train_kernel, dim3(p.gx, p.gy), dim3(kDualThreads), 0, stream, a.vol_src, a.tgt, a.R,
                       (long)a.r_batch_stride, a.W1, a.W2, a.b2, a.B, (long)a.N, a.scores, u_out);
    return hipGetLastError();
}

hipError_t launch_coarse_to_fine(const CoarseToFineLaunch& a, hipStream_t stream)
{
    const ScoreLaunch& c = a.coarse;
    const bool teams = !c.no_teams;
    const ScorePlan p = plan_score_launch(c.B, c.N, c.num_cu, c.spare_cu, teams);
    // stage 1 runs on the SAME grid: its remainder rule with stage 0's gx
    const int64_t slots = 8 * (int64_t)p.gx, full2 = a.N2 / slots, rem2 = a.N2 - full2 * slots;
    const int64_t n_main2 = (teams && rem2 > 0 && rem2 <= 2 * (int64_t)p.gx) ? full2 * slots : a.N2;
    // every workgroup must be resident for the meeting point: at most one per CU (159.5 KiB of LDS each)
    if ((int64_t)p.gx * p.gy > c.num_cu) return hipErrorInvalidConfiguration;
    C2fArgs cf;
    cf.D = a.D;
    cf.N2 = (long)a.N2;
    cf.n_main2 = (long)n_main2;
    cf.best_key2 = reinterpret_cast<key_t*>(a.best_key2);
    cf.scores2 = a.scores2;
    cf.sync = a.sync;
    cf.R_pred = a.R_pred;
    cf.fine_score = a.fine_score;
    cf.coarse_score = a.coarse_score;
    cf.fine_idx = reinterpret_cast<long*>(a.fine_idx);
    cf.coarse_idx = reinterpret_cast<long*>(a.coarse_idx);
    hipLaunchKernelGGL(coarse_to_fine_kernel, dim3(p.gx, p.gy), dim3(kDualThreads), 0, stream, c.vol_src, c.tgt, c.R,
                       (long)c.r_batch_stride, (long)c.n_offset, c.W1, c.W2, c.b2, c.B, (long)c.N, (long)p.n_main, c.scores,
                       reinterpret_cast<key_t*>(c.best_key), c.feat_tgt_out, cf);
    return hipGetLastError();
}

hipError_t launch_unpack_best(const int64_t* best_key, int B, float* best_score, int64_t* best_idx,
                              hipStream_t stream)
{
    const int threads = 64;
    hipLaunchKernelGGL(unpack_best_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, stream,
                       reinterpret_cast<const key_t*>(best_key), B, best_score, reinterpret_cast<long*>(best_idx));
    return hipGetLastError();
}

}  // namespace ahv
