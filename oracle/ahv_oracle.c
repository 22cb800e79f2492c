/*
 * ahv_oracle.c -- CPU restatement of 3DAHV's rotation-hypothesis verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (3dahv_amd/) never calls into it and fails loudly when the
 * HIP library is missing.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle.py)
 * against golden vectors produced by running the reference's own Python code in
 * the authoring container (tools/gen_golden.py -> tests/golden/ npz files).
 *
 * The reference's arithmetic for this path lives in PyTorch ATen operators
 * (third-party, not under /root/reference; reference pins pytorch=1.13.0,
 * install.sh:6; the golden vectors were generated with torch 2.10 CPU):
 *   F.affine_grid(align_corners=False)  -> base grid (2i+1)/S-1, grid = theta @ [x y z 1]
 *   F.grid_sample(5-D, bilinear, zeros, align_corners=False) -> trilinear gather
 *   nn.Conv2d 1x1, F.relu, F.normalize(p=2, dim=1, eps=1e-12)
 * Their published algorithms are restated here in plain C, fp32 like the
 * reference (torch.set_float32_matmul_precision("highest"), modules/model_co3d.py:24).
 *
 * Each function cites the reference call site it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define AHV_CV 16   /* volume channels           modules/modules.py:64  */
#define AHV_S 8     /* volume side D=H=W         modules/modules.py:97  */
#define AHV_K 384   /* 3*Cv*S                    modules/modules.py:67  */
#define AHV_O 32    /* head channels             modules/model_co3d.py:33 */
#define AHV_P 64    /* S*S output positions      modules/modules.py:122 */

int ahv_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* grid_sampler_unnormalize, align_corners=False: ((coord+1)*size-1)/2 */
static inline float unnormalize(float coord, int size)
{
    return ((coord + 1.0f) * (float)size - 1.0f) / 2.0f;
}

/*
 * utils.rotate_volume  (utils.py:113-131)
 *   theta = [R | 0]                          utils.py:123
 *   grid  = F.affine_grid(theta, size, align_corners=False)        utils.py:126
 *   out   = F.grid_sample(volume, grid, 'bilinear', 'zeros', align_corners=False)  utils.py:129
 * vol: [N or 1][C][D][H][W], vol_batch_stride in floats (0 = broadcast, the
 * stride-0 expand of test_co3d.py:137).  R: [N][3][3] row-major.
 * out: [N][C][D][H][W] contiguous.
 */
void ahv_oracle_rotate_volume_f32(const float* vol, int64_t vol_batch_stride, const float* R,
                                  int64_t N, int C, int D, int H, int W, float* out)
{
    const int64_t plane = (int64_t)D * H * W;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* v = vol + n * vol_batch_stride;
        const float* r = R + n * 9;
        float* o = out + n * C * plane;
        for (int d = 0; d < D; ++d)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    /* affine_grid base grid, align_corners=False: (2i+1)/size - 1 */
                    const float x = (2.0f * w + 1.0f) / (float)W - 1.0f;
                    const float y = (2.0f * h + 1.0f) / (float)H - 1.0f;
                    const float z = (2.0f * d + 1.0f) / (float)D - 1.0f;
                    /* grid = theta @ [x y z 1]^T with zero translation */
                    const float gx = r[0] * x + r[1] * y + r[2] * z;
                    const float gy = r[3] * x + r[4] * y + r[5] * z;
                    const float gz = r[6] * x + r[7] * y + r[8] * z;
                    /* grid[...,0] -> W axis, [1] -> H, [2] -> D */
                    const float ix = unnormalize(gx, W);
                    const float iy = unnormalize(gy, H);
                    const float iz = unnormalize(gz, D);
                    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
                    const float tx = ix - fx0, ty = iy - fy0, tz = iz - fz0;
                    const int64_t obase = ((int64_t)d * H + h) * W + w;
                    /* Coordinates with every corner out of range (incl. non-rotation R,
                     * inf and NaN: ATen casts floor() to int64, which lands out of
                     * bounds, so no corner is accumulated) give exact zeros. */
                    if (!(fx0 >= -1.0f && fx0 <= (float)W && fy0 >= -1.0f && fy0 <= (float)H &&
                          fz0 >= -1.0f && fz0 <= (float)D)) {
                        for (int c = 0; c < C; ++c) o[c * plane + obase] = 0.0f;
                        continue;
                    }
                    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
                    float wgt[8];
                    int64_t off[8];
                    int k = 0;
                    for (int dz = 0; dz < 2; ++dz)
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx, ++k) {
                                const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                                const float wx = dx ? tx : 1.0f - tx;
                                const float wy = dy ? ty : 1.0f - ty;
                                const float wz = dz ? tz : 1.0f - tz;
                                const int in = xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D;
                                /* zeros padding is per corner: ATen SKIPS a corner that is out of bounds (it does not
                                 * multiply a voxel by 0 -- that would turn a non-finite voxel into NaN); an in-bounds
                                 * corner is accumulated even when its weight is exactly 0 */
                                wgt[k] = wx * wy * wz;
                                off[k] = in ? ((int64_t)zz * H + yy) * W + xx : -1;
                            }
                    for (int c = 0; c < C; ++c) {
                        const float* vc = v + c * plane;
                        float acc = 0.0f;
                        for (k = 0; k < 8; ++k)
                            if (off[k] >= 0) acc += wgt[k] * vc[off[k]];
                        o[c * plane + obase] = acc;
                    }
                }
    }
}

/*
 * Feature_Aligner.forward_3d2d  (modules/modules.py:112-124, weights :66-70)
 *   z = 'b c d h w -> b (c d) h w'   :115
 *   y = 'b c d h w -> b (c h) d w'   :116
 *   x = 'b c d h w -> b (c w) d h'   :117
 *   cat [x, y, z] on channels        :118
 *   conv1x1 384->32 (no bias), ReLU, conv1x1 32->32 (+bias)   :66-70, :120
 *   F.normalize(p=2, dim=1).flatten(2)                         :122
 * vol: [M][16][8][8][8]; W1: [32][384]; W2: [32][32]; b2: [32]; out: [M][32][64].
 */
void ahv_oracle_forward_3d2d_f32(const float* vol, const float* W1, const float* W2, const float* b2,
                                 int64_t M, float* out)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float* V = vol + m * (AHV_CV * AHV_S * AHV_S * AHV_S);
        float* f = out + m * (AHV_O * AHV_P);
        for (int i = 0; i < AHV_S; ++i)
            for (int j = 0; j < AHV_S; ++j) {
                float u[AHV_O], v[AHV_O];
                for (int o = 0; o < AHV_O; ++o) {
                    const float* w = W1 + o * AHV_K;
                    float acc = 0.0f;
                    for (int c = 0; c < AHV_CV; ++c)
                        for (int k = 0; k < AHV_S; ++k) {
                            /* V[c][d][h][w] index = ((c*8+d)*8+h)*8+w */
                            const float vx = V[((c * 8 + i) * 8 + j) * 8 + k]; /* x slab: (d,h)=(i,j), w=k */
                            const float vy = V[((c * 8 + i) * 8 + k) * 8 + j]; /* y slab: (d,w)=(i,j), h=k */
                            const float vz = V[((c * 8 + k) * 8 + i) * 8 + j]; /* z slab: (h,w)=(i,j), d=k */
                            acc += w[c * 8 + k] * vx;
                            acc += w[128 + c * 8 + k] * vy;
                            acc += w[256 + c * 8 + k] * vz;
                        }
                    u[o] = acc < 0.0f ? 0.0f : acc; /* F.relu: a NaN stays a NaN (acc > 0 ? acc : 0 would drop it) */
                }
                float ss = 0.0f;
                for (int o = 0; o < AHV_O; ++o) {
                    float acc = b2[o];
                    for (int q = 0; q < AHV_O; ++q) acc += W2[o * AHV_O + q] * u[q];
                    v[o] = acc;
                    ss += acc * acc;
                }
                float nrm = sqrtf(ss);
                if (nrm < 1e-12f) nrm = 1e-12f; /* F.normalize eps clamp_min */
                for (int o = 0; o < AHV_O; ++o) f[o * AHV_P + i * 8 + j] = v[o] / nrm;
            }
    }
}

/*
 * score (test_co3d.py:143, modules/model.py:193, test_linemod.py:59):
 *   pred_sim = (f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)
 * f_src: [B][N][32][64]; f_tgt: [B][32][64]; scores: [B][N].
 */
void ahv_oracle_score_features_f32(const float* f_src, const float* f_tgt, int B, int64_t N, float* scores)
{
    for (int b = 0; b < B; ++b) {
        const float* t = f_tgt + (int64_t)b * AHV_O * AHV_P;
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const float* s = f_src + ((int64_t)b * N + n) * AHV_O * AHV_P;
            float tot = 0.0f;
            for (int p = 0; p < AHV_P; ++p) {
                float d = 0.0f;
                for (int o = 0; o < AHV_O; ++o) d += s[o * AHV_P + p] * t[o * AHV_P + p];
                tot += d;
            }
            scores[(int64_t)b * N + n] = tot / (float)AHV_P;
        }
    }
}

/*
 * arg-max (test_co3d.py:145-146): torch.max(pred_sim, dim=1) -> (value, first
 * maximal index); a NaN wins, first NaN index (torch semantics).
 */
void ahv_oracle_argmax_f32(const float* scores, int B, int64_t N, float* best, int64_t* best_idx)
{
    for (int b = 0; b < B; ++b) {
        const float* s = scores + (int64_t)b * N;
        float bv = -INFINITY;
        int64_t bi = 0;
        int have = 0;
        for (int64_t n = 0; n < N; ++n) {
            const float x = s[n];
            if (x != x) { bv = x; bi = n; have = 1; break; }
            if (!have || x > bv) { bv = x; bi = n; have = 1; }
        }
        best[b] = bv;
        best_idx[b] = bi;
    }
}

/*
 * Fused composition of the four steps, chunked so that temporaries stay small:
 * the full hot loop of test_co3d.py:135-146 for B volume pairs.
 * vol_src/vol_tgt: [B][16][8][8][8]; R: [N][3][3] shared (r_batch_stride 0) or
 * [B][N][3][3] (r_batch_stride N*9, modules/model.py:51).
 */
int ahv_oracle_score_hypotheses_f32(const float* vol_src, const float* vol_tgt, const float* R,
                                    int64_t r_batch_stride, const float* W1, const float* W2,
                                    const float* b2, int B, int64_t N, float* scores, float* best,
                                    int64_t* best_idx)
{
    const int64_t VOL = AHV_CV * AHV_S * AHV_S * AHV_S, FEAT = AHV_O * AHV_P;
    const int64_t CH = 256;
    float* rot = (float*)malloc(sizeof(float) * CH * VOL);
    float* fs = (float*)malloc(sizeof(float) * CH * FEAT);
    float* ft = (float*)malloc(sizeof(float) * FEAT);
    if (!rot || !fs || !ft) { free(rot); free(fs); free(ft); return -1; }
    for (int b = 0; b < B; ++b) {
        ahv_oracle_forward_3d2d_f32(vol_tgt + b * VOL, W1, W2, b2, 1, ft);
        for (int64_t n0 = 0; n0 < N; n0 += CH) {
            const int64_t n = (N - n0 < CH) ? (N - n0) : CH;
            ahv_oracle_rotate_volume_f32(vol_src + b * VOL, 0, R + b * r_batch_stride + n0 * 9, n,
                                         AHV_CV, AHV_S, AHV_S, AHV_S, rot);
            ahv_oracle_forward_3d2d_f32(rot, W1, W2, b2, n, fs);
            ahv_oracle_score_features_f32(fs, ft, 1, n, scores + (int64_t)b * N + n0);
        }
    }
    if (best && best_idx) ahv_oracle_argmax_f32(scores, B, N, best, best_idx);
    free(rot); free(fs); free(ft);
    return 0;
}

/*
 * geodesic error in degrees (test_co3d.py:149-150, modules/model.py:199-200):
 *   sim = (sum(R_pred * R_gt).clamp(-1, 3) - 1) / 2 ; err = arccos(sim) * 180 / pi
 */
void ahv_oracle_geodesic_deg_f32(const float* R_pred, const float* R_gt, int64_t n, float* err_deg)
{
    for (int64_t i = 0; i < n; ++i) {
        float t = 0.0f;
        for (int k = 0; k < 9; ++k) t += R_pred[i * 9 + k] * R_gt[i * 9 + k];
        if (t < -1.0f) t = -1.0f;
        if (t > 3.0f) t = 3.0f;
        const float sim = (t - 1.0f) / 2.0f;
        err_deg[i] = acosf(sim) * 180.0f / 3.14159265358979323846f;
    }
}

/* =====================================================================================================
 * "_cpu" twins of the C ABI ("identical signatures with suffix _cpu operating on host
 * pointers = the build's CPU restatement, used for ABI tests").  TEST INFRASTRUCTURE like the rest of this file:
 * they live in libahv_oracle.so, never in libahv_hip.so, and exist so that a GPU-free test can drive the product's
 * own ctypes signature table (3dahv_amd/_lib.py SIGNATURES) through a real computation -- argument order, integer
 * widths, strides, the packed-key convention and its merge / reset flag -- against the golden vectors.
 * Signatures: include/ahv.h, declaration by declaration; `stream` is ignored; return 0 or -1.
 * ===================================================================================================== */
#include <string.h>

#define AHV_TWIN_RESET_BEST 1u
#define AHV_TWIN_SELECT_RESET_KEY 1u
#define AHV_TWIN_KEY_EMPTY INT64_MIN

/* include/ahv.h "Packed keys": key = ordered_i32(score) << 32 | (0xFFFFFFFF - idx), SIGNED order; NaN ranks above
 * +inf; -0 == +0; INT64_MIN = nothing scored */
static int64_t twin_pack_key(float s, uint32_t idx)
{
    s += 0.0f;
    uint32_t u;
    memcpy(&u, &s, 4);
    if (s != s) u = 0x7FC00000u;
    u = (u & 0x80000000u) ? (u ^ 0x7FFFFFFFu) : u;
    return (int64_t)(((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - idx));
}

static void twin_merge(int64_t* best_key, int B, int64_t N, int64_t n_offset, const float* scores, unsigned flags)
{
    for (int b = 0; b < B; ++b) {
        int64_t k = (flags & AHV_TWIN_RESET_BEST) ? AHV_TWIN_KEY_EMPTY : best_key[b];
        for (int64_t n = 0; n < N; ++n) {
            const int64_t c = twin_pack_key(scores[(int64_t)b * N + n], (uint32_t)(n_offset + n));
            if (c > k) k = c;
        }
        best_key[b] = k;
    }
}

int ahv_rotate_volume_f32_cpu(const float* vol, int64_t vol_batch_stride, const float* R, int64_t N, int C, int D,
                              int H, int W, float* out, void* stream)
{
    (void)stream;
    if (N < 0 || C < 1 || D < 1 || H < 1 || W < 1) return -1;
    if (N == 0) return 0;
    if (!vol || !R || !out) return -1;
    ahv_oracle_rotate_volume_f32(vol, vol_batch_stride, R, N, C, D, H, W, out);
    return 0;
}

int ahv_forward_3d2d_f32_cpu(const float* vol, const float* W1, const float* W2, const float* b2, int64_t M,
                             float* out, void* stream)
{
    (void)stream;
    if (M < 0) return -1;
    if (M == 0) return 0;
    if (!vol || !W1 || !W2 || !b2 || !out) return -1;
    ahv_oracle_forward_3d2d_f32(vol, W1, W2, b2, M, out);
    return 0;
}

int ahv_score_features_f32_cpu(const float* f_src, const float* f_tgt, int B, int64_t N, float* scores, void* stream)
{
    (void)stream;
    if (B < 0 || N < 0) return -1;
    if (B == 0 || N == 0) return 0;
    if (!f_src || !f_tgt || !scores) return -1;
    ahv_oracle_score_features_f32(f_src, f_tgt, B, N, scores);
    return 0;
}

int ahv_argmax_f32_cpu(const float* scores, int B, int64_t N, int64_t n_offset, int64_t* best_key, unsigned flags,
                       void* stream)
{
    (void)stream;
    if (!scores || !best_key || B < 0 || N < 0 || n_offset < 0 || n_offset + N > 4294967296ll) return -1;
    twin_merge(best_key, B, N, n_offset, scores, flags);
    return 0;
}

/* feat_tgt [B][32][64] is GIVEN here (forward_3d2d of the target volume), as in the product's entry point */
int ahv_score_hypotheses_f32_cpu(const float* vol_src, const float* feat_tgt, const float* R, int64_t r_batch_stride,
                                 int64_t n_offset, const float* W1, const float* W2, const float* b2, int B, int64_t N,
                                 float* scores, int64_t* best_key, unsigned flags, void* stream)
{
    (void)stream;
    if (B < 0 || N < 0 || (!scores && !best_key)) return -1;
    if (r_batch_stride != 0 && r_batch_stride < N * 9) return -1;
    if (n_offset < 0 || n_offset + N > 4294967296ll) return -1;
    if (best_key && (flags & AHV_TWIN_RESET_BEST))
        for (int b = 0; b < B; ++b) best_key[b] = AHV_TWIN_KEY_EMPTY;
    if (B == 0 || N == 0) return 0;
    if (!vol_src || !feat_tgt || !R || !W1 || !W2 || !b2) return -1;
    const int64_t VOL = AHV_CV * AHV_S * AHV_S * AHV_S, FEAT = AHV_O * AHV_P, CH = 256;
    float* rot = (float*)malloc(sizeof(float) * CH * VOL);
    float* fs = (float*)malloc(sizeof(float) * CH * FEAT);
    float* sc = (float*)malloc(sizeof(float) * CH);
    if (!rot || !fs || !sc) { free(rot); free(fs); free(sc); return -1; }
    for (int b = 0; b < B; ++b)
        for (int64_t n0 = 0; n0 < N; n0 += CH) {
            const int64_t n = (N - n0 < CH) ? (N - n0) : CH;
            ahv_oracle_rotate_volume_f32(vol_src + b * VOL, 0, R + b * r_batch_stride + n0 * 9, n, AHV_CV, AHV_S, AHV_S,
                                         AHV_S, rot);
            ahv_oracle_forward_3d2d_f32(rot, W1, W2, b2, n, fs);
            ahv_oracle_score_features_f32(fs, feat_tgt + b * FEAT, 1, n, sc);
            if (scores) memcpy(scores + (int64_t)b * N + n0, sc, sizeof(float) * n);
            if (best_key) twin_merge(best_key + b, 1, n, n_offset + n0, sc, 0);
        }
    free(rot); free(fs); free(sc);
    return 0;
}

/* the whole verify step behind one entry point (include/ahv.h ahv_verify_pair_f32): forward_3d2d of the target volume
 * (test_co3d.py:141), then the fused loop; feat_tgt_out optional; clock_stamps ignored */
int ahv_verify_pair_f32_cpu(const float* vol_src, const float* vol_tgt, const float* R, int64_t r_batch_stride,
                            int64_t n_offset, const float* W1, const float* W2, const float* b2, int B, int64_t N,
                            float* scores, int64_t* best_key, float* feat_tgt_out, unsigned flags, uint64_t* clock_stamps,
                            void* stream)
{
    (void)clock_stamps;
    if (B < 0 || N < 0) return -1;
    if (B > 0 && (!vol_tgt || !W1 || !W2 || !b2)) return -1;
    float* ft = feat_tgt_out ? feat_tgt_out : (float*)calloc((size_t)(B > 0 ? B : 1) * AHV_O * AHV_P, sizeof(float));
    if (!ft) return -1;
    if (B > 0) ahv_oracle_forward_3d2d_f32(vol_tgt, W1, W2, b2, B, ft);
    const int rc = ahv_score_hypotheses_f32_cpu(vol_src, ft, R, r_batch_stride, n_offset, W1, W2, b2, B, N, scores, best_key,
                                                flags & 3u, stream);
    if (!feat_tgt_out) free(ft);
    return rc;
}

/* ahv_device.h key_score / key_index; the empty key (nothing scored) decodes to -inf, -1 */
int ahv_unpack_best_cpu(const int64_t* best_key, int B, float* best_score, int64_t* best_idx, void* stream)
{
    (void)stream;
    if (!best_key || B < 0) return -1;
    for (int b = 0; b < B; ++b) {
        const int64_t k = best_key[b];
        uint32_t u = (uint32_t)((uint64_t)k >> 32);
        float s;
        if (k == AHV_TWIN_KEY_EMPTY) {
            s = -INFINITY;
        } else {
            u = (u & 0x80000000u) ? (u ^ 0x7FFFFFFFu) : u;
            memcpy(&s, &u, 4);
        }
        if (best_score) best_score[b] = s;
        if (best_idx) best_idx[b] = k == AHV_TWIN_KEY_EMPTY ? -1 : (int64_t)(0xFFFFFFFFu - (uint32_t)((uint64_t)k & 0xFFFFFFFFu));
    }
    return 0;
}

int ahv_reset_best_cpu(int64_t* best_key, int B, void* stream)
{
    (void)stream;
    if (B < 0 || (B > 0 && !best_key)) return -1;
    for (int b = 0; b < B; ++b) best_key[b] = AHV_TWIN_KEY_EMPTY;
    return 0;
}

int ahv_select_rotation_f32_cpu(int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset, int64_t N,
                                int B, float* R_out, float* best_score, int64_t* best_idx, unsigned flags, void* stream)
{
    if (flags & ~AHV_TWIN_SELECT_RESET_KEY) return -1;
    if (ahv_unpack_best_cpu(best_key, B, best_score, NULL, stream)) return -1;
    for (int b = 0; b < B; ++b) {
        const int64_t k = best_key[b];
        const int64_t idx = k == AHV_TWIN_KEY_EMPTY ? -1 : (int64_t)(0xFFFFFFFFu - (uint32_t)((uint64_t)k & 0xFFFFFFFFu));
        if (best_idx) best_idx[b] = idx;
        if (R_out) {
            const int64_t local = idx - n_offset;
            for (int i = 0; i < 9; ++i)
                R_out[b * 9 + i] = (idx >= 0 && local >= 0 && local < N) ? R[b * r_batch_stride + local * 9 + i] : 0.0f;
        }
        if (flags & AHV_TWIN_SELECT_RESET_KEY) best_key[b] = AHV_TWIN_KEY_EMPTY;
    }
    return 0;
}
