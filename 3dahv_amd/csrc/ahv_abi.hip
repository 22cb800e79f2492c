// ahv_abi.hip -- the C ABI declared in include/ahv.h: argument validation, error
// strings, launches.  No allocation, no host synchronisation, no exceptions.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ahv.h"
#include "../../include/ahv_diag.h"
#include "ahv_launch.h"

namespace ahv {
hipError_t launch_unpack_best(const int64_t*, int, float*, int64_t*, hipStream_t);
hipError_t launch_rotate_volume(const float*, int64_t, const float*, int64_t, int, int, int, int, float*, int,
                                hipStream_t);
hipError_t launch_rotate_volume_backward(const float*, int64_t, const float*, int64_t, int, int, int, int, float*, int,
                                         hipStream_t);
hipError_t launch_forward_3d2d(const float*, const float*, const float*, const float*, int64_t, float*, int,
                               hipStream_t);
hipError_t launch_score_features(const float*, const float*, int, int64_t, float*, int, hipStream_t);
hipError_t launch_argmax(const float*, int, int64_t, int64_t, int64_t*, int, hipStream_t);
hipError_t launch_compose_rotations(const int64_t*, const float*, int64_t, int64_t, int64_t, const float*, int64_t,
                                    int, float*, hipStream_t);
hipError_t launch_select_rotation(int64_t*, const float*, int64_t, int64_t, int64_t, int, float*, float*, int64_t*, bool,
                                  hipStream_t);
hipError_t launch_fill_keys(int64_t*, int, hipStream_t);
hipError_t launch_random_rotations(uint64_t, uint64_t, int64_t, float*, hipStream_t);
hipError_t launch_so3_grid(int64_t, int64_t, int64_t, float*, hipStream_t);
hipError_t launch_score_backward(const float*, const float*, const float*, int64_t, const float*, const float*,
                                 const float*, int, int64_t, const float*, float*, unsigned*, float*, float*, float*, float*,
                                 float*, float*, int, hipStream_t, bool);
hipError_t launch_zero_fill(void* const* ptrs, const size_t* bytes, int count, hipStream_t stream);
size_t transformer_workspace_floats(int B);
int transformer_blocks(const ahv_block_weights*, int, float*, float*, int, float*, hipStream_t, const char**);
size_t forward_2d3d_workspace_floats(int B);
int forward_2d3d(const ahv_aligner_weights*, const float*, const float*, int, float*, float*, float*, hipStream_t,
                 const char**);
}  // namespace ahv

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(const char* what, hipError_t e)
{
    return fail(AHV_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// CU count of the current device, cached per device id.
int cu_count()
{
    static thread_local int cached_dev = -1, cached_cu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev != cached_dev) {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
            return -1;
        cached_dev = dev;
        cached_cu = cu;
    }
    return cached_cu;
}

}  // namespace

extern "C" {

int ahv_abi_version(void) { return (2 << 16) | 3; }

const char* ahv_last_error(void) { return g_err; }

int ahv_device_cu_count(void)
{
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    return cu;
}

// hipMemsetAsync is not used anywhere in the library: see zero_fill_kernel (ahv_ops.hip).
static hipError_t zero_span(void* p, size_t bytes, hipStream_t s)
{
    void* const ptrs[1] = {p};
    const size_t n[1] = {bytes};
    return ahv::launch_zero_fill(ptrs, n, 1, s);
}

// tgt = target features (tgt_is_volume = false) or the target volume (ahv_verify_pair_f32)
static int score_common(const char* who, const float* vol_src, const float* tgt, bool tgt_is_volume, const float* R,
                        int64_t r_batch_stride, int64_t n_offset, const float* W1, const float* W2, const float* b2, int B,
                        int64_t N, float* scores, int64_t* best_key, float* feat_tgt_out, unsigned flags,
                        uint64_t* clock_stamps, void* stream)
{
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "%s: negative size (B=%d, N=%lld)", who, B, (long long)N);
    if (B > 0 && N > 0 && (!vol_src || !tgt || !R || !W1 || !W2 || !b2))
        return fail(AHV_EINVAL, "%s: null input pointer", who);
    if (r_batch_stride != 0 && r_batch_stride < N * 9)
        return fail(AHV_EINVAL, "%s: r_batch_stride %lld must be 0 or >= N*9", who, (long long)r_batch_stride);
    if (n_offset < 0 || n_offset + N > 4294967296ll)
        return fail(AHV_EINVAL, "%s: n_offset + N must fit in 32 bits", who);
    if (flags & ~(AHV_SCORE_RESET_BEST | AHV_SCORE_SPLIT_F16 | AHV_SCORE_NO_TEAMS | AHV_SCORE_SPARE_CUS_MASK))
        return fail(AHV_EINVAL, "%s: unknown flags 0x%x", who, flags);
    const bool split = (flags & AHV_SCORE_SPLIT_F16) != 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (best_key && (flags & AHV_SCORE_RESET_BEST) && B > 0) {
        hipError_t e = ahv::launch_fill_keys(best_key, B, s);
        if (e != hipSuccess) return hip_fail("score: key reset", e);
    }
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    if (tgt_is_volume && split) {
        // the split-f16 kernel takes ready-made target features: two launches, through the caller's buffer
        if (B > 0 && !feat_tgt_out)
            return fail(AHV_EINVAL, "%s: AHV_SCORE_SPLIT_F16 needs feat_tgt_out (the target features go through it)", who);
        if (B > 0) {
            hipError_t e = ahv::launch_forward_3d2d(tgt, W1, W2, b2, B, feat_tgt_out, cu, s);
            if (e != hipSuccess) return hip_fail("verify_pair: forward_3d2d launch", e);
        }
        tgt = feat_tgt_out;
        tgt_is_volume = false;
        feat_tgt_out = nullptr;
    }
    if (B == 0) return AHV_OK;
    if (N == 0 && !(tgt_is_volume && feat_tgt_out)) return AHV_OK;  // nothing to score (verify_pair still owes the features)
    ahv::ScoreLaunch a;
    a.vol_src = vol_src; a.tgt = tgt; a.tgt_is_volume = tgt_is_volume; a.R = R;
    a.r_batch_stride = r_batch_stride; a.n_offset = n_offset; a.W1 = W1; a.W2 = W2; a.b2 = b2; a.B = B; a.N = N;
    a.scores = scores; a.best_key = best_key; a.feat_tgt_out = feat_tgt_out; a.num_cu = cu;
    a.spare_cu = (int)((flags & AHV_SCORE_SPARE_CUS_MASK) >> AHV_SCORE_SPARE_CUS_SHIFT);
    a.split_f16 = split; a.no_teams = (flags & AHV_SCORE_NO_TEAMS) != 0; a.clock_stamps = clock_stamps;
    hipError_t e = ahv::launch_score_hypotheses(a, s);
    if (e != hipSuccess) return hip_fail("score: launch", e);
    return AHV_OK;
}

int ahv_score_hypotheses_f32(const float* vol_src, const float* feat_tgt, const float* R,
                             int64_t r_batch_stride, int64_t n_offset, const float* W1, const float* W2,
                             const float* b2, int B, int64_t N, float* scores, int64_t* best_key,
                             unsigned flags, void* stream)
{
    return score_common("score", vol_src, feat_tgt, false, R, r_batch_stride, n_offset, W1, W2, b2, B, N, scores, best_key,
                        nullptr, flags, nullptr, stream);
}

int ahv_score_hypotheses_clocked_f32(const float* vol_src, const float* feat_tgt, const float* R,
                                     int64_t r_batch_stride, int64_t n_offset, const float* W1, const float* W2,
                                     const float* b2, int B, int64_t N, float* scores, int64_t* best_key,
                                     unsigned flags, uint64_t* clock_stamps, void* stream)
{
    if (!clock_stamps) return fail(AHV_EINVAL, "score_clocked: null clock_stamps");
    return score_common("score_clocked", vol_src, feat_tgt, false, R, r_batch_stride, n_offset, W1, W2, b2, B, N, scores,
                        best_key, nullptr, flags, clock_stamps, stream);
}

int ahv_diag_score_plan(int B, int64_t N, unsigned flags, int* gx, int* gy, int64_t* n_main)
{
    if (B < 1 || N < 0) return fail(AHV_EINVAL, "score_plan: bad shape (B=%d, N=%lld)", B, (long long)N);
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    const bool teams = !(flags & AHV_SCORE_SPLIT_F16) && !(flags & AHV_SCORE_NO_TEAMS);
    const ahv::ScorePlan p = ahv::plan_score_launch(B, N, cu, (int)((flags & AHV_SCORE_SPARE_CUS_MASK) >> AHV_SCORE_SPARE_CUS_SHIFT), teams);
    if (gx) *gx = p.gx;
    if (gy) *gy = p.gy;
    if (n_main) *n_main = p.n_main;
    return AHV_OK;
}

int ahv_verify_pair_f32(const float* vol_src, const float* vol_tgt, const float* R, int64_t r_batch_stride,
                        int64_t n_offset, const float* W1, const float* W2, const float* b2, int B, int64_t N,
                        float* scores, int64_t* best_key, float* feat_tgt_out, unsigned flags, uint64_t* clock_stamps,
                        void* stream)
{
    return score_common("verify_pair", vol_src, vol_tgt, true, R, r_batch_stride, n_offset, W1, W2, b2, B, N, scores,
                        best_key, feat_tgt_out, flags, clock_stamps, stream);
}

int ahv_unpack_best(const int64_t* best_key, int B, float* best_score, int64_t* best_idx, void* stream)
{
    if (!best_key) return fail(AHV_EINVAL, "unpack: null best_key");
    if (B < 0) return fail(AHV_EINVAL, "unpack: negative B");
    if (B == 0) return AHV_OK;
    hipError_t e = ahv::launch_unpack_best(best_key, B, best_score, best_idx, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("unpack: launch", e);
    return AHV_OK;
}

int ahv_rotate_volume_f32(const float* vol, int64_t vol_batch_stride, const float* R, int64_t N, int C, int D,
                          int H, int W, float* out, void* stream)
{
    if (N < 0 || C < 1 || D < 1 || H < 1 || W < 1)
        return fail(AHV_EINVAL, "rotate_volume: bad shape N=%lld C=%d D=%d H=%d W=%d", (long long)N, C, D, H, W);
    if (vol_batch_stride < 0) return fail(AHV_EINVAL, "rotate_volume: negative batch stride");
    if (N == 0) return AHV_OK; /* empty batch: nothing to read or write, pointers may be null */
    if (!vol || !R || !out) return fail(AHV_EINVAL, "rotate_volume: null pointer");
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    hipError_t e = ahv::launch_rotate_volume(vol, vol_batch_stride, R, N, C, D, H, W, out, cu,
                                             static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("rotate_volume: launch", e);
    return AHV_OK;
}

int ahv_rotate_volume_backward_f32(const float* grad_out, int64_t vol_batch_stride, const float* R, int64_t N, int C,
                                   int D, int H, int W, float* grad_vol, void* stream)
{
    if (N < 0 || C < 1 || D < 1 || H < 1 || W < 1)
        return fail(AHV_EINVAL, "rotate_volume_backward: bad shape N=%lld C=%d D=%d H=%d W=%d", (long long)N, C, D, H, W);
    const int64_t vol_elems = (int64_t)C * D * H * W;
    if (vol_batch_stride != 0 && vol_batch_stride < vol_elems)
        return fail(AHV_EINVAL, "rotate_volume_backward: batch stride must be 0 or >= C*D*H*W");
    if (!grad_vol) return fail(AHV_EINVAL, "rotate_volume_backward: null grad_vol");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n_vol = vol_batch_stride == 0 ? 1 : N;
    if (n_vol > 0) {
        const size_t bytes = sizeof(float) * (size_t)(vol_batch_stride == 0 ? vol_elems : (N - 1) * vol_batch_stride + vol_elems);
        hipError_t e = zero_span(grad_vol, bytes, s);
        if (e != hipSuccess) return hip_fail("rotate_volume_backward: zero fill", e);
    }
    if (N == 0) return AHV_OK;
    if (!grad_out || !R) return fail(AHV_EINVAL, "rotate_volume_backward: null pointer");
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    hipError_t e = ahv::launch_rotate_volume_backward(grad_out, vol_batch_stride, R, N, C, D, H, W, grad_vol, cu, s);
    if (e != hipSuccess) return hip_fail("rotate_volume_backward: launch", e);
    return AHV_OK;
}

int ahv_forward_3d2d_f32(const float* vol, const float* W1, const float* W2, const float* b2, int64_t M,
                         float* out, void* stream)
{
    if (M < 0) return fail(AHV_EINVAL, "forward_3d2d: negative M");
    if (M == 0) return AHV_OK;
    if (!vol || !W1 || !W2 || !b2 || !out) return fail(AHV_EINVAL, "forward_3d2d: null pointer");
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    hipError_t e = ahv::launch_forward_3d2d(vol, W1, W2, b2, M, out, cu, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("forward_3d2d: launch", e);
    return AHV_OK;
}

int ahv_score_features_f32(const float* f_src, const float* f_tgt, int B, int64_t N, float* scores, void* stream)
{
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "score_features: negative size");
    if (B == 0 || N == 0) return AHV_OK;
    if (!f_src || !f_tgt || !scores) return fail(AHV_EINVAL, "score_features: null pointer");
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    hipError_t e = ahv::launch_score_features(f_src, f_tgt, B, N, scores, cu, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("score_features: launch", e);
    return AHV_OK;
}

int ahv_compose_rotations_f32(const int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset,
                                 int64_t N, const float* D, int64_t N2, int B, float* out, void* stream)
{
    if (B < 0 || N < 0 || N2 < 0) return fail(AHV_EINVAL, "compose_rotations: negative size");
    if (B == 0 || N2 == 0) return AHV_OK;
    if (!best_key || !R || !D || !out) return fail(AHV_EINVAL, "compose_rotations: null pointer");
    if (N == 0) return fail(AHV_EINVAL, "compose_rotations: empty rotation set");
    if (r_batch_stride != 0 && r_batch_stride < N * 9) return fail(AHV_EINVAL, "compose_rotations: bad r_batch_stride");
    hipError_t e = ahv::launch_compose_rotations(best_key, R, r_batch_stride, n_offset, N, D, N2, B, out,
                                                 static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("compose_rotations: launch", e);
    return AHV_OK;
}

int ahv_coarse_to_fine_f32(const float* vol_src, const float* vol_tgt, const float* R, int64_t r_batch_stride, int64_t N,
                           const float* D, int64_t N2, const float* W1, const float* W2, const float* b2, int B,
                           float* scores_coarse, float* scores_fine, int64_t* keys, uint32_t* sync, float* feat_tgt_out,
                           float* R_pred, float* fine_score, int64_t* fine_idx, float* coarse_score, int64_t* coarse_idx,
                           unsigned flags, void* stream)
{
    if (B < 0 || N < 0 || N2 < 0) return fail(AHV_EINVAL, "coarse_to_fine: negative size (B=%d, N=%lld, N2=%lld)", B, (long long)N, (long long)N2);
    if (B == 0) return AHV_OK;
    if (N == 0 || N2 == 0) return fail(AHV_EINVAL, "coarse_to_fine: empty hypothesis set (N=%lld, N2=%lld)", (long long)N, (long long)N2);
    if (!vol_src || !vol_tgt || !R || !D || !W1 || !W2 || !b2) return fail(AHV_EINVAL, "coarse_to_fine: null input pointer");
    if (!keys || !sync) return fail(AHV_EINVAL, "coarse_to_fine: keys and sync are required (2 B keys, 2 B + 1 counters)");
    if (r_batch_stride != 0 && r_batch_stride < N * 9)
        return fail(AHV_EINVAL, "coarse_to_fine: r_batch_stride %lld must be 0 or >= N*9", (long long)r_batch_stride);
    if (N > 4294967296ll || N2 > 4294967296ll) return fail(AHV_EINVAL, "coarse_to_fine: indices must fit in 32 bits");
    if (flags & ~(AHV_SCORE_NO_TEAMS | AHV_SCORE_SPARE_CUS_MASK)) return fail(AHV_EINVAL, "coarse_to_fine: unknown flags 0x%x", flags);
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    ahv::CoarseToFineLaunch a;
    ahv::ScoreLaunch& c = a.coarse;
    c.vol_src = vol_src; c.tgt = vol_tgt; c.tgt_is_volume = true; c.R = R; c.r_batch_stride = r_batch_stride; c.n_offset = 0;
    c.W1 = W1; c.W2 = W2; c.b2 = b2; c.B = B; c.N = N; c.scores = scores_coarse; c.best_key = keys;
    c.feat_tgt_out = feat_tgt_out; c.num_cu = cu;
    c.spare_cu = (int)((flags & AHV_SCORE_SPARE_CUS_MASK) >> AHV_SCORE_SPARE_CUS_SHIFT);
    c.split_f16 = false; c.no_teams = (flags & AHV_SCORE_NO_TEAMS) != 0; c.clock_stamps = nullptr;
    a.D = D; a.N2 = N2; a.scores2 = scores_fine; a.best_key2 = keys + B; a.sync = sync; a.R_pred = R_pred;
    a.fine_score = fine_score; a.coarse_score = coarse_score; a.fine_idx = fine_idx; a.coarse_idx = coarse_idx;
    hipError_t e = ahv::launch_coarse_to_fine(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("coarse_to_fine: launch", e);
    return AHV_OK;
}

int ahv_reset_best(int64_t* best_key, int B, void* stream)
{
    if (B < 0) return fail(AHV_EINVAL, "reset_best: negative B");
    if (B == 0) return AHV_OK;
    if (!best_key) return fail(AHV_EINVAL, "reset_best: null best_key");
    hipError_t e = ahv::launch_fill_keys(best_key, B, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("reset_best: launch", e);
    return AHV_OK;
}

int ahv_select_rotation_f32(int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset, int64_t N,
                            int B, float* R_out, float* best_score, int64_t* best_idx, unsigned flags, void* stream)
{
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "select_rotation: negative size");
    if (flags & ~AHV_SELECT_RESET_KEY) return fail(AHV_EINVAL, "select_rotation: unknown flags 0x%x", flags);
    if (B == 0) return AHV_OK;
    if (!best_key) return fail(AHV_EINVAL, "select_rotation: null best_key");
    if (R_out && (!R || N == 0)) return fail(AHV_EINVAL, "select_rotation: R_out needs a rotation set");
    if (r_batch_stride != 0 && r_batch_stride < N * 9) return fail(AHV_EINVAL, "select_rotation: bad r_batch_stride");
    hipError_t e = ahv::launch_select_rotation(best_key, R, r_batch_stride, n_offset, N, B, R_out, best_score, best_idx,
                                               (flags & AHV_SELECT_RESET_KEY) != 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("select_rotation: launch", e);
    return AHV_OK;
}

size_t ahv_transformer_workspace_bytes(int B) { return sizeof(float) * ahv::transformer_workspace_floats(B); }

int ahv_transformer_blocks_f32(const ahv_block_weights* blocks, int depth, float* x_src, float* x_tgt, int B,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    if (depth < 0 || B < 0) return fail(AHV_EINVAL, "transformer_blocks: negative size");
    if (depth == 0 || B == 0) return AHV_OK;
    if (!blocks || !x_src || !x_tgt || !workspace) return fail(AHV_EINVAL, "transformer_blocks: null pointer");
    if (workspace_bytes < ahv_transformer_workspace_bytes(B))
        return fail(AHV_EINVAL, "transformer_blocks: workspace of %zu bytes, need %zu", workspace_bytes,
                    ahv_transformer_workspace_bytes(B));
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(x_src) & 15) ||
        (reinterpret_cast<uintptr_t>(x_tgt) & 15))
        return fail(AHV_EINVAL, "transformer_blocks: buffers must be 16-byte aligned");
    for (int i = 0; i < 4 * depth; ++i) {
        const ahv_block_weights& w = blocks[i];
        if (!w.w_qkv || !w.w_out || !w.b_out || !w.ln1_g || !w.ln1_b || !w.w_ff1 || !w.b_ff1 || !w.w_ff2 || !w.b_ff2 ||
            !w.ln2_g || !w.ln2_b)
            return fail(AHV_EINVAL, "transformer_blocks: null weight pointer in block %d", i);
    }
    const char* what = "";
    const int rc = ahv::transformer_blocks(blocks, depth, x_src, x_tgt, B, static_cast<float*>(workspace),
                                           static_cast<hipStream_t>(stream), &what);
    if (rc != 0) return fail(AHV_ELAUNCH, "transformer_blocks: %s: %s", what, hipGetErrorString((hipError_t)rc));
    return AHV_OK;
}

size_t ahv_forward_2d3d_workspace_bytes(int B) { return sizeof(float) * ahv::forward_2d3d_workspace_floats(B); }

int ahv_forward_2d3d_f32(const ahv_aligner_weights* w, const float* layer4_src, const float* layer4_tgt, int B,
                         void* workspace, size_t workspace_bytes, float* vol_src, float* vol_tgt, void* stream)
{
    if (B < 0) return fail(AHV_EINVAL, "forward_2d3d: negative B");
    if (B == 0) return AHV_OK;
    if (!w || !layer4_src || !layer4_tgt || !workspace || !vol_src || !vol_tgt)
        return fail(AHV_EINVAL, "forward_2d3d: null pointer");
    if (!w->w_emb || !w->w_conv1 || !w->w_conv2 || !w->posemb || !w->gn_g || !w->gn_b || !w->w_in[0] || !w->w_in[1] ||
        !w->b_in[0] || !w->b_in[1] || !w->w_out[0] || !w->w_out[1] || !w->b_out[0] || !w->b_out[1] || !w->w3d_1 ||
        !w->w3d_2 || (w->depth > 0 && !w->blocks))
        return fail(AHV_EINVAL, "forward_2d3d: null weight pointer");
    if (w->depth < 0) return fail(AHV_EINVAL, "forward_2d3d: negative depth");
    if (workspace_bytes < ahv_forward_2d3d_workspace_bytes(B))
        return fail(AHV_EINVAL, "forward_2d3d: workspace of %zu bytes, need %zu", workspace_bytes,
                    ahv_forward_2d3d_workspace_bytes(B));
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(AHV_EINVAL, "forward_2d3d: workspace must be 16-byte aligned");
    const char* what = "";
    const int rc = ahv::forward_2d3d(w, layer4_src, layer4_tgt, B, static_cast<float*>(workspace), vol_src, vol_tgt,
                                     static_cast<hipStream_t>(stream), &what);
    if (rc != 0) return fail(AHV_ELAUNCH, "forward_2d3d: %s: %s", what, hipGetErrorString((hipError_t)rc));
    return AHV_OK;
}

size_t ahv_score_hypotheses_backward_workspace_bytes(int B, int64_t N)
{
    if (B <= 0 || N <= 0) return 0;
    // dL/du per hypothesis + one max|du| word per sample (rounded up to 16 bytes) + one partial dW1 (32 x 384) per
    // workgroup of the persistent grid (at most one per CU; 1024 covers any device if the query fails)
    const int cu = cu_count();
    const size_t wgs = cu > 0 ? (size_t)cu : 1024;
    return sizeof(float) * (2048 * (size_t)B * (size_t)N + (((size_t)B + 3) & ~(size_t)3) + wgs * 32 * 384);
}

static int score_backward_common(const float* vol_src, const float* feat_tgt, const float* R,
                                 int64_t r_batch_stride, const float* W1, const float* W2, const float* b2,
                                 int B, int64_t N, const float* grad_scores, void* workspace,
                                 size_t workspace_bytes, float* grad_vol_src, float* grad_feat_tgt,
                                 float* grad_W1, float* grad_W2, float* grad_b2, void* stream, bool saved_u);

int ahv_score_hypotheses_backward_f32(const float* vol_src, const float* feat_tgt, const float* R,
                                      int64_t r_batch_stride, const float* W1, const float* W2, const float* b2,
                                      int B, int64_t N, const float* grad_scores, void* workspace,
                                      size_t workspace_bytes, float* grad_vol_src, float* grad_feat_tgt,
                                      float* grad_W1, float* grad_W2, float* grad_b2, void* stream)
{
    return score_backward_common(vol_src, feat_tgt, R, r_batch_stride, W1, W2, b2, B, N, grad_scores, workspace, workspace_bytes,
                                 grad_vol_src, grad_feat_tgt, grad_W1, grad_W2, grad_b2, stream, false);
}

int ahv_score_hypotheses_backward_saved_f32(const float* vol_src, const float* feat_tgt, const float* R,
                                            int64_t r_batch_stride, const float* W1, const float* W2, const float* b2,
                                            int B, int64_t N, const float* grad_scores, void* workspace,
                                            size_t workspace_bytes, float* grad_vol_src, float* grad_feat_tgt,
                                            float* grad_W1, float* grad_W2, float* grad_b2, void* stream)
{
    return score_backward_common(vol_src, feat_tgt, R, r_batch_stride, W1, W2, b2, B, N, grad_scores, workspace, workspace_bytes,
                                 grad_vol_src, grad_feat_tgt, grad_W1, grad_W2, grad_b2, stream, true);
}

int ahv_score_hypotheses_train_f32(const float* vol_src, const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                   const float* W1, const float* W2, const float* b2, int B, int64_t N, float* scores,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "score_train: negative size");
    if (B > 65535) return fail(AHV_EINVAL, "score_train: B > 65535");
    if (B == 0 || N == 0) return AHV_OK;
    if (!vol_src || !feat_tgt || !R || !W1 || !W2 || !b2 || !scores || !workspace) return fail(AHV_EINVAL, "score_train: null pointer");
    if (r_batch_stride != 0 && r_batch_stride != N * 9) return fail(AHV_EINVAL, "score_train: r_batch_stride must be 0 or N*9");
    if (workspace_bytes < ahv_score_hypotheses_backward_workspace_bytes(B, N))
        return fail(AHV_EINVAL, "score_train: workspace of %zu bytes, need %zu (ahv_score_hypotheses_backward_workspace_bytes)",
                    workspace_bytes, ahv_score_hypotheses_backward_workspace_bytes(B, N));
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(AHV_EINVAL, "score_train: workspace must be 16-byte aligned");
    const int cu = cu_count();
    if (cu <= 0) return fail(AHV_EDEVICE, "score_train: no usable HIP device");
    ahv::ScoreLaunch a{};
    a.vol_src = vol_src; a.tgt = feat_tgt; a.tgt_is_volume = false; a.R = R; a.r_batch_stride = r_batch_stride; a.n_offset = 0;
    a.W1 = W1; a.W2 = W2; a.b2 = b2; a.B = B; a.N = N; a.scores = scores; a.best_key = nullptr; a.feat_tgt_out = nullptr;
    a.num_cu = cu; a.spare_cu = 0; a.split_f16 = false; a.no_teams = true; a.clock_stamps = nullptr;
    hipError_t e = ahv::launch_score_hypotheses_train(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("score_train: launch", e);
    return AHV_OK;
}

static int score_backward_common(const float* vol_src, const float* feat_tgt, const float* R,
                                 int64_t r_batch_stride, const float* W1, const float* W2, const float* b2,
                                 int B, int64_t N, const float* grad_scores, void* workspace,
                                 size_t workspace_bytes, float* grad_vol_src, float* grad_feat_tgt,
                                 float* grad_W1, float* grad_W2, float* grad_b2, void* stream, bool saved_u)
{
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "score_backward: negative size");
    if (B > 65535) return fail(AHV_EINVAL, "score_backward: B > 65535");
    if (!grad_W1 || !grad_W2 || !grad_b2) return fail(AHV_EINVAL, "score_backward: null weight-gradient pointer");
    if (B > 0 && (!grad_vol_src || !grad_feat_tgt)) return fail(AHV_EINVAL, "score_backward: null gradient pointer");
    if (B > 0 && N > 0) {
        if (!vol_src || !feat_tgt || !R || !W1 || !W2 || !b2 || !grad_scores || !workspace)
            return fail(AHV_EINVAL, "score_backward: null pointer");
        if (r_batch_stride != 0 && r_batch_stride != N * 9)
            return fail(AHV_EINVAL, "score_backward: r_batch_stride must be 0 or N*9");
        if (workspace_bytes < ahv_score_hypotheses_backward_workspace_bytes(B, N))
            return fail(AHV_EINVAL, "score_backward: workspace of %zu bytes, need %zu", workspace_bytes,
                        ahv_score_hypotheses_backward_workspace_bytes(B, N));
        if (reinterpret_cast<uintptr_t>(workspace) & 15)
            return fail(AHV_EINVAL, "score_backward: workspace must be 16-byte aligned");
    }
    const int cu = cu_count();
    if (cu <= 0) return fail(AHV_EDEVICE, "score_backward: no usable HIP device");
    hipError_t e = ahv::launch_score_backward(vol_src, feat_tgt, R, r_batch_stride, W1, W2, b2, B, N, grad_scores,
                                              static_cast<float*>(workspace),
                                              reinterpret_cast<unsigned*>(static_cast<float*>(workspace) + 2048 * (size_t)B * (size_t)N),
                                              static_cast<float*>(workspace) + 2048 * (size_t)B * (size_t)N + (((size_t)B + 3) & ~(size_t)3),
                                              grad_vol_src, grad_feat_tgt, grad_W1, grad_W2, grad_b2, cu,
                                              static_cast<hipStream_t>(stream), saved_u);
    if (e != hipSuccess) return hip_fail("score_backward: launch", e);
    return AHV_OK;
}

int ahv_random_rotations_f32(uint64_t seed, uint64_t offset, int64_t N, float* out, void* stream)
{
    if (N < 0) return fail(AHV_EINVAL, "random_rotations: negative N");
    if (N == 0) return AHV_OK;
    if (!out) return fail(AHV_EINVAL, "random_rotations: null pointer");
    hipError_t e = ahv::launch_random_rotations(seed, offset, N, out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("random_rotations: launch", e);
    return AHV_OK;
}

int ahv_so3_grid_f32(int64_t n_total, int64_t offset, int64_t N, float* out, void* stream)
{
    if (N < 0 || offset < 0 || n_total <= 0 || offset + N > n_total)
        return fail(AHV_EINVAL, "so3_grid: need 0 <= offset, 0 <= N, offset + N <= n_total");
    if (N == 0) return AHV_OK;
    if (!out) return fail(AHV_EINVAL, "so3_grid: null pointer");
    hipError_t e = ahv::launch_so3_grid(n_total, offset, N, out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail("so3_grid: launch", e);
    return AHV_OK;
}

int ahv_argmax_f32(const float* scores, int B, int64_t N, int64_t n_offset, int64_t* best_key, unsigned flags,
                   void* stream)
{
    if (!scores || !best_key) return fail(AHV_EINVAL, "argmax: null pointer");
    if (B < 0 || N < 0) return fail(AHV_EINVAL, "argmax: negative size");
    if (B > 65535) return fail(AHV_EINVAL, "argmax: B > 65535");
    if (n_offset < 0 || n_offset + N > 4294967296ll) return fail(AHV_EINVAL, "argmax: n_offset + N must fit in 32 bits");
    if (flags & ~AHV_SCORE_RESET_BEST) return fail(AHV_EINVAL, "argmax: unknown flags 0x%x", flags);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((flags & AHV_SCORE_RESET_BEST) && B > 0) {
        hipError_t e = ahv::launch_fill_keys(best_key, B, s);
        if (e != hipSuccess) return hip_fail("argmax: key reset", e);
    }
    if (B == 0 || N == 0) return AHV_OK;
    const int cu = cu_count();
    if (cu < 0) return fail(AHV_EDEVICE, "no usable HIP device");
    hipError_t e = ahv::launch_argmax(scores, B, N, n_offset, best_key, cu, s);
    if (e != hipSuccess) return hip_fail("argmax: launch", e);
    return AHV_OK;
}

}  // extern "C"
