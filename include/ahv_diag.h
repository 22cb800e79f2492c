/*
 * ahv_diag.h -- measurement and developer entry points of libahv_hip.so.  NOT part of the drop-in boundary (include/ahv.h):
 * nothing here replaces a reference interface; bench.py, tools/ and tests/ use them to time launches from the inside and
 * to read back scheduling decisions.  Same conventions as ahv.h (device pointers, stream, status codes).
 */
#ifndef AHV_DIAG_H
#define AHV_DIAG_H

#include "ahv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Extra flag bits of ahv_score_hypotheses_f32 / ahv_verify_pair_f32 / ahv_coarse_to_fine_f32: leave k compute units
 * without a workgroup of the persistent grid (default 0: one workgroup per CU, 159.5 KiB of LDS each, so nothing else fits
 * on the device while the scorer runs).  Measured in round 4 to cost the scorer 1.4 % at k = 2 and to buy a concurrent
 * kernel nothing (it slots in when the grid drains either way): a knob for experiments, not for production callers. */
#define AHV_SCORE_SPARE_CUS_SHIFT 8
#define AHV_SCORE_SPARE_CUS_MASK 0xFF00u
#define AHV_SCORE_SPARE_CUS(k) (((unsigned)(k) << AHV_SCORE_SPARE_CUS_SHIFT) & AHV_SCORE_SPARE_CUS_MASK)

/*
 * The same launch as
 * ahv_score_hypotheses_f32, and additionally every workgroup w of the persistent grid writes
 *   clock_stamps[4w + 0..3] = { s_memtime, s_memrealtime (100 MHz) before its hypothesis loop, the same two after }
 * so that the shader clock the chip actually held during THIS kernel is
 *   (stamps[2] - stamps[0]) / (stamps[3] - stamps[1]) * 100 MHz   (median over workgroups).
 * clock_stamps: device memory, 4 * ahv_device_cu_count() words, zeroed by the caller (the grid never exceeds
 * one workgroup per CU; entries of unused slots stay zero).  bench.py reports roofline.shader_clock_ghz from it.
 */
int ahv_score_hypotheses_clocked_f32(const float* vol_src, const float* feat_tgt, const float* R,
                                     int64_t r_batch_stride, int64_t n_offset, const float* W1,
                                     const float* W2, const float* b2, int B, int64_t N, float* scores,
                                     int64_t* best_key, unsigned flags, uint64_t* clock_stamps, void* stream);

/*
 * How a launch of B samples x N hypotheses would be laid out on the current device with these flags: the persistent grid
 * (gx workgroups along the hypothesis axis x gy along the batch) and n_main -- hypotheses [0, n_main) of every sample go to
 * single waves, [n_main, N) to teams of four (csrc/ahv_team.h).  Pure host arithmetic, no launch.  Tests use it to know
 * that a case really exercises the team path (its scores being bit-identical, nothing else shows it).
 */
int ahv_diag_score_plan(int B, int64_t N, unsigned flags, int* gx, int* gy, int64_t* n_main);

#ifdef __cplusplus
}
#endif
#endif /* AHV_DIAG_H */
