/* CPU restatement of the keyframe-side helpers next to the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see nsc_oracle.h): linked into libnsc_oracle.so, called by tests/ only.
 *
 * nsc_oracle_voxel_overlap restates compute_overlap (reference src/data/pose_utils.py:323-389)
 * AFTER its random down-sampling (:340-347, unseeded np.random.choice -- the caller samples):
 *   cloud 1: transform_points (:107-122) = (T @ [x y z 1]^T) in float64.  numpy hands the 4x4 by 4xN
 *            product to OpenBLAS dgemm, whose micro-kernel accumulates over k with fused multiply-adds:
 *            v = fma(T3, 1, fma(T2, z, fma(T1, y, T0 * x)))  (checked bit-for-bit against numpy in
 *            oracle/gen_golden_keyframe.py); voxel = floor(clip(v, -1e6, 1e6) / voxel_size) in float64 (:361-363)
 *   cloud 2: never transformed, stays float32: floor(clip(p, -1e6, 1e6) / float32(voxel_size)) (:361-363)
 *   rows with a non-finite entry (any column, intensity included) are dropped (:352-353)
 *   IoU = |unique(v1) & unique(v2)| / |unique(v1) | unique(v2)|, 0 when the union is empty (:375-389)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t x, y, z; } Vox;

static int vox_cmp(const void *a, const void *b)
{
    const Vox *p = (const Vox *)a, *q = (const Vox *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->z != q->z) return p->z < q->z ? -1 : 1;
    return 0;
}

static int64_t vox_unique(Vox *v, int64_t n)
{
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof(Vox), vox_cmp);
    int64_t m = 1;
    for (int64_t i = 1; i < n; ++i)
        if (vox_cmp(&v[i], &v[m - 1]) != 0) v[m++] = v[i];
    return m;
}

static double clipd(double v) { return v < -1e6 ? -1e6 : (v > 1e6 ? 1e6 : v); }
static float clipf(float v) { return v < -1e6f ? -1e6f : (v > 1e6f ? 1e6f : v); }

/* counts[0..2] = |unique(v1)|, |unique(v2)|, |intersection|; returns the IoU. */
double nsc_oracle_voxel_overlap(const float *pts1, int64_t n1, const float *pts2, int64_t n2,
                                int32_t stride, const double *T /* 4x4 row-major */, double voxel_size,
                                int32_t *counts)
{
    Vox *a = (Vox *)malloc(sizeof(Vox) * (size_t)(n1 > 0 ? n1 : 1));
    Vox *b = (Vox *)malloc(sizeof(Vox) * (size_t)(n2 > 0 ? n2 : 1));
    int64_t na = 0, nb = 0;
    for (int64_t i = 0; i < n1; ++i) {
        const float *p = pts1 + i * stride;
        const double x = p[0], y = p[1], z = p[2];
        double v[3];
        int ok = 1;
        for (int c = 0; c < 3; ++c) {
            const double *r = T + 4 * c;
            v[c] = fma(r[3], 1.0, fma(r[2], z, fma(r[1], y, r[0] * x)));
            ok &= isfinite(v[c]) != 0;
        }
        for (int c = 3; c < stride; ++c) ok &= isfinite(p[c]) != 0;     /* intensity rides along (:131) */
        if (!ok) continue;
        a[na].x = (int32_t)floor(clipd(v[0]) / voxel_size);
        a[na].y = (int32_t)floor(clipd(v[1]) / voxel_size);
        a[na].z = (int32_t)floor(clipd(v[2]) / voxel_size);
        ++na;
    }
    const float vs = (float)voxel_size;
    for (int64_t i = 0; i < n2; ++i) {
        const float *p = pts2 + i * stride;
        int ok = 1;
        for (int c = 0; c < stride; ++c) ok &= isfinite(p[c]) != 0;
        if (!ok) continue;
        b[nb].x = (int32_t)floorf(clipf(p[0]) / vs);
        b[nb].y = (int32_t)floorf(clipf(p[1]) / vs);
        b[nb].z = (int32_t)floorf(clipf(p[2]) / vs);
        ++nb;
    }
    na = vox_unique(a, na);
    nb = vox_unique(b, nb);
    int64_t i = 0, j = 0, inter = 0;
    while (i < na && j < nb) {
        const int c = vox_cmp(&a[i], &b[j]);
        if (c == 0) { ++inter; ++i; ++j; }
        else if (c < 0) ++i;
        else ++j;
    }
    free(a);
    free(b);
    if (counts) { counts[0] = (int32_t)na; counts[1] = (int32_t)nb; counts[2] = (int32_t)inter; }
    const int64_t uni = na + nb - inter;
    return uni > 0 ? (double)inter / (double)uni : 0.0;
}
