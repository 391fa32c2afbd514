/*
 * nsc_oracle.h -- CPU restatement (plain C) of the Neural-Spectral-Codec descriptor encoder.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (neural-spectral-codec_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Checked against outputs of the reference itself
 * (/root/reference/src/encoding/{range_image,spectral_encoder}.py, imported in the build
 * container by oracle/gen_golden.py); the vectors are committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef NSC_ORACLE_H
#define NSC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct NscOracleParams {
    int32_t n_elevation;      /* projector rows E              (range_image.py:104,120)          */
    int32_t n_azimuth;        /* projector cols A (360)        (range_image.py:105,121)          */
    int32_t n_bins;           /* histogram bins per row (50)   (spectral_encoder.py:39,67)       */
    int32_t target_rows;      /* target_elevation_bins (16)    (spectral_encoder.py:43,69)       */
    double  elev_min_rad;     /* np.deg2rad(elevation_range[0]) as float64 (range_image.py:126)  */
    double  elev_max_rad;     /* np.deg2rad(elevation_range[1]) as float64 (range_image.py:127)  */
    float   min_range;        /* 1.0  (range_image.py:108,123)                                   */
    float   max_range;        /* 80.0 (range_image.py:107,122)                                   */
    float   epsilon;          /* 1e-8 (spectral_encoder.py:42,68)                                */
    int32_t interpolate;      /* interpolate_empty (spectral_encoder.py:44,70,220)               */
    int32_t elev_f64;         /* 1: numpy>=2 promotion (float64 row math); 0: numpy 1.24 float32 */
} NscOracleParams;

void nsc_oracle_default_params(NscOracleParams *p);

/* correctly rounded float32 atan2 (the definition both oracle and HIP path share) */
float nsc_oracle_atan2f(float y, float x);

/* range_image.py:129-214 -- returns number of points that survived the filters.
 * linear_idx (nullable): per input point, row*A+col or -1 if the point was dropped. */
int64_t nsc_oracle_project(const float *pts, int64_t n_pts, int32_t stride_floats,
                           const NscOracleParams *p, float *img /*E*A*/, int32_t *linear_idx);

/* range_image.py:15-89 (method='linear'), in place on an (E,A) float32 image */
void nsc_oracle_interpolate(float *img, int32_t E, int32_t A);

/* spectral_encoder.py:93-116,136-145 -- bin edges (n_bins+1 float32) and the n_freqs->bin LUT */
void nsc_oracle_bin_lut(float alpha, int32_t n_bins, int32_t n_freqs, float epsilon,
                        float *edges /*nullable*/, int32_t *lut);

/* torch adaptive_avg_pool2d((rows_out, A)) as used at spectral_encoder.py:171-176 */
void nsc_oracle_adaptive_rows(const float *img, int32_t rows_in, int32_t A, int32_t rows_out, float *out);

/* spectral_encoder.py:160-204 on a (rows,A) image (rows pooled to target_rows if different) */
void nsc_oracle_encode_range_image(const float *img, int32_t rows, const NscOracleParams *p,
                                   const int32_t *lut, float *desc /*target_rows*n_bins*/,
                                   float *mags /*nullable, target_rows*n_freqs*/);

/* spectral_encoder.py:206-229 for one cloud; img_raw / img_interp nullable (E*A each) */
void nsc_oracle_encode_points(const float *pts, int64_t n_pts, int32_t stride_floats,
                              const NscOracleParams *p, const int32_t *lut,
                              float *desc, float *img_raw, float *img_interp);

/* batch of clouds, cloud c = points [offsets[c], offsets[c+1]); n_threads worker threads */
void nsc_oracle_encode_clouds(const float *pts, const int64_t *offsets, int32_t n_clouds,
                              int32_t stride_floats, const NscOracleParams *p, const int32_t *lut,
                              float *desc, float *img_raw, float *img_interp, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
