/*
 * ahv.h -- C ABI of libahv_hip.so: MI355X (gfx950) kernels for 3DAHV's
 * rotation-hypothesis verification hot path.
 *
 * The reference (sailor-z/3DAHV) has no FFI layer: its boundary for this path is
 * three Python callables plus three inline tensor expressions.  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer adds on the
 * reference side.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers (HBM), fp32 unless stated, borrowed
 *    from the caller (e.g. torch tensors): the caller keeps them alive until the
 *    stream has executed the call.  Nothing is allocated or freed inside.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    Calls are asynchronous w.r.t. the host and ordered on that stream; they
 *    contain no host synchronisation, so they can be captured in a hipGraph.
 *  - Return value: AHV_OK (0) or a negative AHV_E* code; ahv_last_error()
 *    returns a thread-local description of the last failure on this thread.
 *    No C++ exception crosses the ABI.  Entry points are re-entrant and thread-safe:
 *    the library has no mutable process-wide state (kernel choices are per-call flags).
 *  - Non-finite inputs.  Rotation matrices may hold ANY value (non-orthonormal, scaled, zero, inf, NaN): a sample point
 *    that leaves the volume -- or is not a number -- contributes zeros, exactly as F.grid_sample's zeros padding does
 *    (ATen casts floor() of the coordinate to an integer that lands out of bounds), and the scores equal the reference's
 *    (tests/test_gpu_verify.py::test_any_rotation_matrix_matches_the_reference).  Non-finite VOXELS or head weights are
 *    outside the contract: the scorers clamp the 2x2x2 footprint of a sample INTO the volume and give the corners that
 *    zeros padding drops a weight of exactly 0 (csrc/ahv_dual.h, hat weights) -- for a sample coordinate i < 0 along an
 *    axis they read rows 0 and 1 with weights (1 + i, 0) where the reference reads row 0 alone, for i >= 7 rows 6 and 7
 *    with (0, 8 - i), beyond [-1, 8) both with 0 -- and 0 * NaN = NaN, 0 * inf = NaN: a non-finite voxel can make a
 *    score NaN that the reference keeps finite (never the other way round for the NaNs torch produces); ReLU is an
 *    integer max that maps a NaN with the sign bit set to +0 where torch.relu propagates it.  No hang, no fault, and the
 *    packed key stays torch.max of the launch's own scores (NaN first): tests/test_gpu_verify.py pins all of this.
 *  - Fixed geometry of the reference: volume channels Cv=16, side S=8
 *    (modules/modules.py:64,97), head width O=32, K=3*Cv*S=384
 *    (modules/modules.py:66-70), P=S*S=64 output positions.
 */
#ifndef AHV_H
#define AHV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AHV_OK 0
#define AHV_EINVAL (-1)  /* bad argument (null pointer, negative size, bad stride) */
#define AHV_ELAUNCH (-2) /* HIP launch / runtime error */
#define AHV_EDEVICE (-3) /* no usable gfx950 device */

#define AHV_CV 16
#define AHV_S 8
#define AHV_K 384
#define AHV_O 32
#define AHV_P 64
#define AHV_VOL_ELEMS (AHV_CV * AHV_S * AHV_S * AHV_S) /* 8192 floats = 32 KiB */
#define AHV_FEAT_ELEMS (AHV_O * AHV_P)                 /* 2048 floats = 8 KiB  */

/*
 * Packed (score, index) keys.  best_key[b] is a SIGNED 64-bit integer
 *   key = (ordered_i32(score) << 32) | (0xFFFFFFFF - global_idx)
 * where ordered_i32 maps the fp32 score onto int32 in score order (bits for s >= 0, bits ^ 0x7FFFFFFF for s < 0;
 * -0 counts as +0; every NaN as the quiet NaN 0x7FC00000, which orders above +inf because torch.max propagates NaN).
 * A signed max over keys therefore yields the largest score and, among equal scores, the LOWEST index -- the
 * (value, index) pair of torch.max (test_co3d.py:145) -- and is exactly what an int64 MAX all-reduce (RCCL / gloo)
 * or torch.max computes: keys of different launches, chunks or GPUs merge without any re-encoding.
 * AHV_KEY_EMPTY (INT64_MIN) is below every real key: "nothing scored"; it decodes to score -inf, index -1.
 * (ABI 1.x packed the same fields in UNSIGNED order with 0 as the empty key; 2.0 = that key ^ 0x8000000000000000.)
 */
#define AHV_KEY_EMPTY INT64_MIN

/* flags of ahv_score_hypotheses_f32 */
#define AHV_SCORE_RESET_BEST 1u /* set best_key[0..B) to AHV_KEY_EMPTY on the stream before scoring */
#define AHV_SCORE_SPLIT_F16 2u  /* opt-in kernel for THIS call: GEMM1 (the 384 -> 32 projection) and GEMM2 (32 -> 32) as
                                 * three f16 MFMA products of hi/lo-split operands with fp32 accumulation (the dropped
                                 * lo*lo term is 2^-22 relative; power-of-two prescales chosen on the device -- per launch
                                 * for the weights, per sample for the volume, per hypothesis for the activations -- keep
                                 * every finite fp32 input in range).  Scores are as close to the fp64 truth as the
                                 * default kernel's (~7e-8), the arithmetic is not IEEE fp32 operation by operation,
                                 * hence opt-in.  The blend, normalisation and score stay fp32.  2.1x faster.  The
                                 * selector is per call: there is no process-wide kernel switch. */

#define AHV_SCORE_NO_TEAMS 4u   /* every hypothesis by ONE wave.  Default: a remainder of the launch (at most two hypotheses
                                 * per workgroup beyond its full rounds of eight) is scored by teams of four waves, a quarter
                                 * of the volume each, ahead of the main rounds (csrc/ahv_team.h).  A team runs the lone
                                 * wave's accumulation chains tile by tile and associates the score's sums the same way: its
                                 * score is the lone wave's BIT FOR BIT.  The flag is therefore a scheduling knob only -- with
                                 * or without it a score is a function of (volumes, weights, R_n) alone, independent of N, of
                                 * n_offset and of how a set is sharded (tests/test_gpu_parity.py::
                                 * test_team_scores_are_bit_identical; ABI <= 2.1 exchanged partial sums: 1e-7 apart). */
/* (bits 8-15: AHV_SCORE_SPARE_CUS(k), a measurement knob declared in ahv_diag.h) */

/* flags of ahv_select_rotation_f32 */
#define AHV_SELECT_RESET_KEY 1u /* hand best_key[0..B) back as AHV_KEY_EMPTY after decoding it: the next verify step
                                 * then needs no launch of its own to clear the key */

/* ABI version (major<<16 | minor).  2.0: signed-order keys (above), flags argument of ahv_select_rotation_f32,
 * ahv_reset_best, ahv_verify_pair_f32.  2.1: ahv_coarse_to_fine_f32. */
int ahv_abi_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char* ahv_last_error(void);

/* Number of compute units of the current device (for sizing / reporting); <0 on error. */
int ahv_device_cu_count(void);

/*
 * Fused scorer: rotate + project + verification head + cosine score (+ running
 * arg-max) for N hypotheses and B volume pairs in ONE launch, O(1) HBM traffic
 * per hypothesis (36 B of R in, 4 B of score out).
 *
 * Replaces, for every b < B and n < N, the reference's inline hot loop
 *   rotate_volume(vol_src[b].expand(N,...), R)          test_co3d.py:137   (utils.py:113-131)
 *   feature_aligner.forward_3d2d(...)                   test_co3d.py:140   (modules/modules.py:112-124)
 *   (f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)    test_co3d.py:143
 *   torch.max(pred_sim, dim=1)                          test_co3d.py:145
 * (same code at modules/model.py:186-195, :133-146, test_linemod.py:49-62; per-sample
 *  R form at modules/model.py:51-54).
 *
 *  vol_src   [B][16][8][8][8]   source feature volumes (output of forward_2d3d)
 *  feat_tgt  [B][32][64]        forward_3d2d(vol_tgt): unit-norm target features
 *  R         hypotheses, row-major 3x3; hypothesis n of sample b is at
 *            R + b*r_batch_stride + n*9  (r_batch_stride = 0: shared across the
 *            batch, modules/model.py:184; = N*9: per sample, modules/model.py:51).
 *            R need not be orthonormal (the reference never assumes it).
 *  n_offset  global index of local hypothesis 0 (sharding N across GPUs); it is
 *            added to the index packed into best_key.
 *  W1 [32][384], W2 [32][32], b2 [32]   feature_embedding_2d.{0.weight, 2.weight, 2.bias}
 *  scores    [B][N] or NULL     per-hypothesis mean cosine similarity
 *  best_key  [B] or NULL        packed running maximum (see "Packed keys" above), merged
 *            with a signed atomic max.  Decode with ahv_unpack_best / ahv_select_rotation_f32.
 *            Requires n_offset + N <= 2^32.
 *  flags     bit-or of AHV_SCORE_RESET_BEST (else: merge into the existing keys, e.g. chunked N),
 *            AHV_SCORE_SPLIT_F16 (else: the all-fp32 kernel), AHV_SCORE_NO_TEAMS
 */
int ahv_score_hypotheses_f32(const float* vol_src, const float* feat_tgt, const float* R,
                             int64_t r_batch_stride, int64_t n_offset, const float* W1,
                             const float* W2, const float* b2, int B, int64_t N, float* scores,
                             int64_t* best_key, unsigned flags, void* stream);

/*
 * One-launch verify step: everything the reference does per image pair between the encoder and the arg-max,
 *   forward_3d2d(img_feat_tgt)                          test_co3d.py:141   (modules/modules.py:112-124)
 *   rotate_volume + forward_3d2d + score + running max  test_co3d.py:137-145
 * behind one entry point and, for the fp32 kernel, in ONE launch: the target features are built inside the scoring
 * kernel (four waves of every workgroup share the target volume by quarter while the others start on their first
 * hypotheses) instead of in a launch of their own.  Arguments as ahv_score_hypotheses_f32 except
 *  vol_tgt       [B][16][8][8][8]  target volumes (the second output of forward_2d3d) in place of feat_tgt
 *  feat_tgt_out  [B][32][64] or NULL: if given, forward_3d2d(vol_tgt) as the launch computed it (validation_step's
 *                gt_sim reuses it, modules/model.py:137-143).  REQUIRED with AHV_SCORE_SPLIT_F16, whose kernel
 *                takes ready-made features: that case runs forward_3d2d into it and then the scorer (two launches).
 *  clock_stamps  NULL, or as in ahv_score_hypotheses_clocked_f32 (ahv_diag.h; measurement only)
 * With N = 0 and feat_tgt_out given only the target features are produced.  The in-launch features are summed in
 * another order than ahv_forward_3d2d_f32's (both fp32, both within 1e-6 of the reference).
 */
int ahv_verify_pair_f32(const float* vol_src, const float* vol_tgt, const float* R, int64_t r_batch_stride,
                        int64_t n_offset, const float* W1, const float* W2, const float* b2, int B, int64_t N,
                        float* scores, int64_t* best_key, float* feat_tgt_out, unsigned flags, uint64_t* clock_stamps,
                        void* stream);

/*
 * Decode packed keys: best_score[b], best_idx[b] (int64, global hypothesis index).
 * Replaces the (value, index) pair of torch.max (test_co3d.py:145).  Either output may be NULL.
 * AHV_KEY_EMPTY (nothing scored: N = 0) decodes to best_score = -inf and best_idx = -1.
 */
int ahv_unpack_best(const int64_t* best_key, int B, float* best_score, int64_t* best_idx, void* stream);

/* best_key[0..B) = AHV_KEY_EMPTY on the stream (what AHV_SCORE_RESET_BEST does in front of a scoring launch). */
int ahv_reset_best(int64_t* best_key, int B, void* stream);

/*
 * Op-level drop-in for utils.rotate_volume (utils.py:113-131): F.affine_grid +
 * F.grid_sample (trilinear, zeros padding, align_corners=False), materialising
 * out [N][C][D][H][W].  vol_batch_stride (in floats) is 0 for the stride-0
 * expand the reference passes (test_co3d.py:137) or C*D*H*W.  Any C,D,H,W >= 1;
 * (16,8,8,8) with stride 0 takes the LDS-staged fast path.
 */
int ahv_rotate_volume_f32(const float* vol, int64_t vol_batch_stride, const float* R, int64_t N,
                          int C, int D, int H, int W, float* out, void* stream);

/*
 * Adjoint of ahv_rotate_volume_f32 w.r.t. the volume -- what autograd needs when the reference's differentiable
 * rotate_volume (utils.py:113-131) sits in a training graph (infoNCE_loss, modules/model_co3d.py:49-54).
 * grad_out [N][C][D][H][W] = dL/d(rotated volume); grad_vol is WRITTEN (zeroed on the stream first): one
 * [C][D][H][W] volume when vol_batch_stride = 0 (all hypotheses accumulate into it: the stride-0 expand the
 * reference passes), else volume n at grad_vol + n*vol_batch_stride.  R carries no gradient (the reference samples
 * it).  Sums use float atomics: reproducible to rounding, not bitwise.
 */
int ahv_rotate_volume_backward_f32(const float* grad_out, int64_t vol_batch_stride, const float* R, int64_t N,
                                   int C, int D, int H, int W, float* grad_vol, void* stream);

/*
 * Op-level drop-in for Feature_Aligner.forward_3d2d (modules/modules.py:112-124):
 * vol [M][16][8][8][8] -> out [M][32][64], unit L2 norm over the 32 channels.
 */
int ahv_forward_3d2d_f32(const float* vol, const float* W1, const float* W2, const float* b2,
                         int64_t M, float* out, void* stream);

/*
 * Op-level drop-in for the inline score expression (test_co3d.py:143):
 * scores[b][n] = mean_pos sum_ch f_src[b][n][ch][pos] * f_tgt[b][ch][pos].
 * f_src [B][N][32][64], f_tgt [B][32][64], scores [B][N].
 */
int ahv_score_features_f32(const float* f_src, const float* f_tgt, int B, int64_t N, float* scores,
                           void* stream);

/*
 * Arg-max over materialised scores (test_co3d.py:145): merges scores[b][0..N)
 * into best_key[b] with the same packing as the fused scorer.
 */
int ahv_argmax_f32(const float* scores, int B, int64_t N, int64_t n_offset, int64_t* best_key,
                   unsigned flags, void* stream);

/*
 * Unpack + gather in one launch: best_score[b], best_idx[b] (global index) and
 * R_out[b] = R[b*r_batch_stride + (idx - n_offset)*9 ...], i.e. the pair
 * `pred_sim, pred_index = torch.max(...)`; `proposals[pred_index]` of test_co3d.py:145-146.
 * With N sharded across GPUs only the rank whose slice [n_offset, n_offset+N) holds the
 * winner writes the row; the others write zeros (sum over ranks = R_pred).  Any output may be NULL.
 * flags: AHV_SELECT_RESET_KEY (best_key is then written: EMPTY for the next step) or 0 (best_key is only read).
 */
int ahv_select_rotation_f32(int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset,
                            int64_t N, int B, float* R_out, float* best_score, int64_t* best_idx, unsigned flags,
                            void* stream);

/*
 * The whole coarse-to-fine step of BASELINE.json configs[4] in ONE launch (one rank; with the hypothesis sets sharded
 * over ranks the stages are separate launches with a key all-reduce between them: ahv_verify_pair_f32,
 * ahv_compose_rotations_f32, ahv_score_hypotheses_f32, ahv_select_rotation_f32):
 *   stage 0  ahv_verify_pair_f32 on the coarse set R [N] (target features built in the launch);
 *   the workgroups meet at a device-wide counter, every one reads the winner R* = R[idx*];
 *   stage 1  scores the refinement set R* D[n], n < N2, composed per hypothesis as ahv_compose_rotations_f32 would;
 *   the workgroup that finishes last decodes both keys as ahv_select_rotation_f32 would.
 *  R              [N][3][3] (r_batch_stride 0) or [B][N][3][3];  D [N2][3][3]
 *  scores_coarse  [B][N] or NULL;  scores_fine [B][N2] or NULL;  feat_tgt_out [B][32][64] or NULL
 *  keys           [2][B] int64 scratch, AHV_KEY_EMPTY on entry; handed back AHV_KEY_EMPTY
 *  sync           [2 B + 1] uint32 scratch, zero on entry; words 0 .. 2B-1 are handed back zero.  Word 2B is an error
 *                 flag the launch sets (and nobody clears) if a workgroup gave the meeting point up after ~1 s -- something
 *                 else held compute units for that long; the outputs of that step are then not to be used.
 *  R_pred         [B][3][3] = R* D[n*];  fine_score / fine_idx (index into D) / coarse_score / coarse_idx [B]; any may be NULL
 *  flags          AHV_SCORE_NO_TEAMS, AHV_SCORE_SPARE_CUS(k)
 * The grid is one workgroup per compute unit at most, so all of it is resident and the meeting point is reached;
 * two such launches on one device at a time (two processes) may make each other wait.  Graph-capturable (a plain launch).
 */
int ahv_coarse_to_fine_f32(const float* vol_src, const float* vol_tgt, const float* R, int64_t r_batch_stride, int64_t N,
                           const float* D, int64_t N2, const float* W1, const float* W2, const float* b2, int B,
                           float* scores_coarse, float* scores_fine, int64_t* keys, uint32_t* sync, float* feat_tgt_out,
                           float* R_pred, float* fine_score, int64_t* fine_idx, float* coarse_score, int64_t* coarse_idx,
                           unsigned flags, void* stream);

/*
 * Coarse-to-fine refinement set, composed on the device (no host round trip, graph-capturable):
 * out[b][n] = R[idx_b] * D[n] for n < N2, where idx_b is decoded from best_key[b] (minus n_offset)
 * and D [N2][3][3] is a fixed set of small rotations.  Not in the reference (it scores one flat
 * set, test_objaverse.py:17); BASELINE.json configs[4].  out [B][N2][3][3].
 */
int ahv_compose_rotations_f32(const int64_t* best_key, const float* R, int64_t r_batch_stride, int64_t n_offset,
                              int64_t N, const float* D, int64_t N2, int B, float* out, void* stream);

/*
 * N Haar-uniform rotation matrices out [N][3][3], generated on the device; replaces the host call
 * pytorch3d.transforms.random_rotations(N) (test_co3d.py:106, modules/model.py:184).  Counter-based
 * (Philox-4x32-10): rotation n is a function of (seed, offset + n) only, so shards of one hypothesis set can
 * be generated independently on different GPUs (offset = first global index of the shard).
 */
int ahv_random_rotations_f32(uint64_t seed, uint64_t offset, int64_t N, float* out, void* stream);

/*
 * Rows [offset, offset + N) of the deterministic, nearly uniform n_total-point SO(3) grid (super-Fibonacci
 * spiral), out [N][3][3].  Build-defined: the reference has no grid (test_linemod.py:43 samples randomly);
 * BASELINE.json configs[2] asks for a dense one.  Shards can be generated independently (offset).
 */
int ahv_so3_grid_f32(int64_t n_total, int64_t offset, int64_t N, float* out, void* stream);

/*
 * Weights of one BasicTransformerBlock of the reference's encoder (transformer/attention.py:240-258),
 * device pointers, fp32, torch layouts (Linear weight = [out][in]).  State-dict names in comments.
 */
typedef struct ahv_block_weights {
    const float* w_qkv;  /* [768][256] = cat(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight) (no bias) */
    const float* w_out;  /* [256][256]  attn.to_out.0.weight */
    const float* b_out;  /* [256]       attn.to_out.0.bias   */
    const float* ln1_g;  /* [256]       norm1.weight */
    const float* ln1_b;  /* [256]       norm1.bias   */
    const float* w_ff1;  /* [4096][512] ff.net.0.proj.weight (GEGLU: value half, then gate half) */
    const float* b_ff1;  /* [4096]      ff.net.0.proj.bias   */
    const float* w_ff2;  /* [256][2048] ff.net.2.weight */
    const float* b_ff2;  /* [256]       ff.net.2.bias   */
    const float* ln2_g;  /* [256]       norm2.weight */
    const float* ln2_b;  /* [256]       norm2.bias   */
} ahv_block_weights;

/* Workspace (bytes, device memory) that ahv_transformer_blocks_f32 needs for B sample pairs. */
size_t ahv_transformer_workspace_bytes(int B);

/*
 * The token stage of the reference's "3D-aware encoder": `depth` BidirectionTransformerBlocks
 * (transformer/attention.py:260-274, the loop at :386-387) applied IN PLACE to the two token streams
 * x_src, x_tgt [B][64][256] (already GroupNorm'ed + proj_in'ed, :378-384).  blocks is a HOST array of
 * 4*depth entries in the order attn_self_1, attn_self_2, attn_cross_1, attn_cross_2 per layer.
 * Per layer: x = self_1(x); ctx = self_2(ctx); x, ctx = cross_1(x, ctx), cross_2(ctx, x), where a block is
 *   m = LN1(attn(x, ctx)); m = LN2(FF(cat[x, m])); x + m,   attn = 4 heads x 64, softmax(QK^T/8)V,
 *   FF = Linear(512 -> 2*2048) GEGLU (exact erf GELU) Linear(2048 -> 256).
 * Built for dim 256 / 4 heads / 64 tokens (the reference's only configuration).
 */
int ahv_transformer_blocks_f32(const ahv_block_weights* blocks, int depth, float* x_src, float* x_tgt, int B,
                               void* workspace, size_t workspace_bytes, void* stream);

/*
 * All weights of Feature_Aligner.forward_2d3d (modules/modules.py:49-110), device pointers, fp32.
 * Convolution weights are repacked tap-major so that a convolution is an implicit GEMM over K = (tap, ci):
 *   w_conv1/2 [256][2304] = feature_embedding.1.conv{1,2}.weight.permute(0,2,3,1)          k = (ky*3+kx)*256 + ci
 *   w3d_1     [32][1024]  rows 0..15  = feature_embedding_3d.conv1.weight.permute(0,2,3,4,1), 27 taps padded to 32
 *                         rows 16..31 = feature_embedding_3d.downsample.0.weight at tap 13 (the centre), else 0
 *   w3d_2     [16][512]   = feature_embedding_3d.conv2.weight.permute(0,2,3,4,1), 27 taps padded to 32
 * (bn_down exists in the reference state dict but is never applied, modules/modules.py:28-47.)
 */
typedef struct ahv_aligner_weights {
    const float* w_emb;     /* [256][768]  feature_embedding.0.weight (1x1, no bias) */
    const float* w_conv1;
    const float* w_conv2;
    const float* posemb;    /* [64][256]   posemb_sincos_2d(channel=256) token-major: [h*8+w][c] */
    const float* gn_g;      /* [256] att.norm.weight  (one GroupNorm shared by both streams) */
    const float* gn_b;      /* [256] att.norm.bias */
    const float* w_in[2];   /* [256][256] att.proj_in.weight, att.proj_context_in.weight */
    const float* b_in[2];   /* [256] */
    const float* w_out[2];  /* [256][256] att.proj_out.weight, att.proj_context_out.weight */
    const float* b_out[2];  /* [256] */
    const float* w3d_1;
    const float* w3d_2;
    const ahv_block_weights* blocks; /* HOST array of 4*depth entries (see ahv_transformer_blocks_f32) */
    int depth;
} ahv_aligner_weights;

size_t ahv_forward_2d3d_workspace_bytes(int B);

/*
 * Feature_Aligner.forward_2d3d(img_feat_src, img_feat_tgt, random_mask=False) (modules/modules.py:86-101):
 * layer_4 features [B][768][8][8] x2 -> volumes [B][16][8][8][8] x2 (conv embedding + res-block, sin/cos
 * position code, GroupNorm + proj_in, `depth` bidirectional transformer blocks, proj_out + residual, the
 * reshape to (32,8,8,8), 3-D res-block).  Built for the reference's only configuration (768 -> 256, 4 heads).
 */
int ahv_forward_2d3d_f32(const ahv_aligner_weights* w, const float* layer4_src, const float* layer4_tgt, int B,
                         void* workspace, size_t workspace_bytes, float* vol_src, float* vol_tgt, void* stream);

/*
 * Backward of ahv_score_hypotheses_f32 -- what autograd needs for Estimator.infoNCE_loss / training_step
 * (modules/model_co3d.py:41-61,71-91; modules/model.py:43-63).  Given
 * grad_scores[B][N] = dL/dscore it writes (not accumulates)
 *   grad_vol_src [B][16][8][8][8], grad_feat_tgt [B][32][64], grad_W1 [32][384], grad_W2 [32][32], grad_b2 [32].
 * R carries no gradient (the reference samples it).  r_batch_stride as in the forward (0 = shared, N*9 = the
 * per-sample rotations of training).  The workspace holds dL/du (8 KiB per hypothesis), one word per sample and one
 * partial dW1 (48 KiB) per workgroup of the persistent grid; ask ahv_score_hypotheses_backward_workspace_bytes.
 * Sums across hypotheses use float atomics (d vol: exact fixed-point sums per workgroup, then float atomics), so
 * results are reproducible to rounding, not bitwise.  Graph-capturable: the accumulation targets are zeroed by a
 * kernel of the library, not by hipMemsetAsync.
 */
size_t ahv_score_hypotheses_backward_workspace_bytes(int B, int64_t N);

int ahv_score_hypotheses_backward_f32(const float* vol_src, const float* feat_tgt, const float* R,
                                      int64_t r_batch_stride, const float* W1, const float* W2, const float* b2,
                                      int B, int64_t N, const float* grad_scores, void* workspace,
                                      size_t workspace_bytes, float* grad_vol_src, float* grad_feat_tgt,
                                      float* grad_W1, float* grad_W2, float* grad_b2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AHV_H */
