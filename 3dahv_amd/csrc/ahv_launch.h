// ahv_launch.h -- host-side launch descriptors shared by the C ABI (ahv_abi.hip), the scorer's launcher
// (ahv_score.hip) and the developer benchmarks that include the kernel source (tools/kbench.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ahv {

struct ScoreLaunch {
    const float* vol_src;   // [B][16][8][8][8]
    const float* tgt;       // target features [B][32][64], or the target volume [B][16][8][8][8] if tgt_is_volume
    bool tgt_is_volume;     // ahv_verify_pair_f32: forward_3d2d(vol_tgt) is computed inside the launch
    const float* R;
    int64_t r_batch_stride, n_offset;
    const float *W1, *W2, *b2;
    int B;
    int64_t N;
    float* scores;          // [B][N] or NULL
    int64_t* best_key;      // [B] or NULL
    float* feat_tgt_out;    // [B][32][64] or NULL (tgt_is_volume only): the target features the launch computed
    int num_cu, spare_cu;   // compute units of the device; how many of them to leave without a workgroup
    bool split_f16;         // AHV_SCORE_SPLIT_F16
    bool no_teams;          // AHV_SCORE_NO_TEAMS
    uint64_t* clock_stamps; // diagnostic entry point, else NULL
};

struct ScorePlan {
    int gx, gy;       // grid: x strides the hypotheses, y the batch
    int64_t n_main;   // hypotheses [0, n_main) by single waves, [n_main, N) by teams of four
};

ScorePlan plan_score_launch(int B, int64_t N, int num_cu, int spare_cu, bool teams);
hipError_t launch_score_hypotheses(const ScoreLaunch& a, hipStream_t stream);
// the training forward: scores [B][N] and, in u_out, 8 KB of pre-activations per hypothesis for the backward
hipError_t launch_score_hypotheses_train(const ScoreLaunch& a, float* u_out, hipStream_t stream);

// ahv_coarse_to_fine_f32: the verify step on the coarse set (a.coarse, tgt = the target volume, n_offset 0) and, in the
// same launch, on the refinement set R* D[n] of its winner (coarse_to_fine_kernel, ahv_score.hip)
struct CoarseToFineLaunch {
    ScoreLaunch coarse;
    const float* D;            // [N2][3][3]
    int64_t N2;
    float* scores2;            // [B][N2] or NULL
    int64_t* best_key2;        // [B], EMPTY on entry and exit
    uint32_t* sync;            // [2 B + 1], zero on entry and exit (last word: error flag, sticky)
    float* R_pred;             // [B][3][3]
    float *fine_score, *coarse_score;
    int64_t *fine_idx, *coarse_idx;
};
hipError_t launch_coarse_to_fine(const CoarseToFineLaunch& a, hipStream_t stream);

}  // namespace ahv
