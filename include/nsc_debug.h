/*
 * nsc_debug.h -- diagnostics of libnsc_hip.so that are NOT part of the product ABI (include/nsc.h): parity triage and
 * measurement aids.  Same conventions as nsc.h (device pointers, asynchronous on `stream`, status codes).
 */
#ifndef NSC_DEBUG_H
#define NSC_DEBUG_H

#include "nsc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Always built (the GPU parity tests use it: tests/test_encoder_gpu.py). */
/* Parity triage: per point, the pixel index row*360+col the scatter uses (-1 = dropped) and
 * whether the exact (float64 atan2) path decided it (bit0 azimuth, bit1 elevation). */
int nsc_debug_point_bins(const float *pts, int64_t n_points, int32_t stride_floats,
                         const NscEncParams *p, int32_t *out_idx, uint8_t *out_flags /*nullable*/,
                         void *stream);

/* Development builds only (NSC_DEV_BUILD=1 python neural-spectral-codec_amd/build.py -> -DNSC_DEV_TUNING); the product
 * library does not export it. */
/* Diagnostic co-runner (bench.py --gnn-burn): `workgroups` x 4 waves of the co-resident GNN kernels' footprint (0 B of LDS,
 * < 56 VGPRs), each wave issuing `per_wave` operations of ONE kind -- mode 0: v_mfma_f32_16x16x4_f32 on register operands
 * (4 independent accumulators), 1: v_fma_f32 (64 lanes), 2: 16-byte loads from a 1 MB L2-resident buffer (`scratch`, >= 1 MB),
 * 3: ds_bpermute_b32, 4: 16-byte loads that hit L1 (every wave the same 16 KB), 5: 16-byte loads of which the four waves of a
 * workgroup read the same addresses.  Answers what a given amount of one resource costs the kernel it runs beside.  scratch also takes the
 * (never read) results. */
int nsc_debug_burn(int32_t mode, int32_t workgroups, int32_t per_wave, float *scratch, size_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSC_DEBUG_H */
