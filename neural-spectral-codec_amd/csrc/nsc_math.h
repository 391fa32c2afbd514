// nsc_math.h -- per-point binning math shared by the HIP kernels and the host-side margin checks.
//
// The pixel a point lands in is DEFINED by the exact chain below (float32 ops of
// RangeImageProjector.project, reference src/encoding/range_image.py:157-198, with atan2f taken
// as the correctly rounded float32 result).  The kernels evaluate a cheap float32 estimate first
// and accept it only when it is provably far from every bin edge; otherwise the exact chain runs.
// Compile with -ffp-contract=off: nothing here may be contracted into an FMA behind our back.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NSC_HD __host__ __device__ inline
#else
#define NSC_HD inline
#endif

#define NSC_A 360                       /* azimuth columns (fixed: octant trick + 360-point FFT) */
#define NSC_F 181                       /* rfft bins                                               */
#define NSC_PI_F 3.1415927410125732f    /* float32(np.pi)    range_image.py:167                    */
#define NSC_2PI_F 6.2831854820251465f   /* float32(2*np.pi)  range_image.py:167,195                */
#define NSC_EMPTY_BITS 0xffffffffu      /* "no point yet" in the squared-range image               */

struct NscBinParams {
    double emin, espan;        // float64 elevation_min and (elevation_max - elevation_min)
    float emin_f, espan_f;     // their float32 casts (numpy 1.24 float32 row math)
    float el_scale;            // E / espan, float32, for the fast estimate
    float az_delta, el_delta;  // acceptance margins of the fast estimate, in bin units
    float s_lo, s_hi;          // range filter expressed on the squared range (see host code)
    float el_u_scale, el_u_bias; // narrow-FOV row estimate: u = atan(t) * scale + bias
    int E;                     // projector rows
    int elev_f64;              // row math in float64 (numpy >= 2) or float32 (numpy 1.24)
    int narrow_fov;            // both FOV edges within |tan| < 0.58: cheap row estimate is valid
    int simple_valid;          // s_hi < 1e10: the 1e10 clip and the isfinite tests are implied by the
                               // squared-range window (inf/NaN coordinates give s = inf/NaN -> dropped)
};

// ---- exact chain ----------------------------------------------------------------------------

NSC_HD float nsc_clip_sq(float v)
{
    float s = v * v;                    // range_image.py:159-161: np.clip(v**2, 0, 1e10), v finite
    return s > 1e10f ? 1e10f : s;
}

NSC_HD float nsc_atan2f_cr(float y, float x)
{
    return (float)atan2((double)y, (double)x);
}

NSC_HD int nsc_col_from_angle(float a)  // a = atan2f_cr(y, x)            range_image.py:167,194-198
{
    a = a + NSC_PI_F;
    if (a >= NSC_2PI_F) a = a - NSC_2PI_F;      // fmodf for 0 <= a <= 2*pi_f (exact subtraction)
    float cf = a / NSC_2PI_F;
    cf = cf * (float)NSC_A;
    int col = (int)floorf(cf);
    col = col < 0 ? 0 : col;
    return col > NSC_A - 1 ? NSC_A - 1 : col;
}

NSC_HD int nsc_row_from_elev(float e, const NscBinParams &bp)           // range_image.py:186-191
{
    int row;
    if (bp.elev_f64) {
        double en = ((double)e - bp.emin) / bp.espan;
        row = (int)floor(en * (double)bp.E);
    } else {
        float en = (e - bp.emin_f) / bp.espan_f;
        row = (int)floorf(en * (float)bp.E);
    }
    row = row < 0 ? 0 : row;
    return row > bp.E - 1 ? bp.E - 1 : row;
}

NSC_HD int nsc_col_exact(float y, float x) { return nsc_col_from_angle(nsc_atan2f_cr(y, x)); }

NSC_HD int nsc_row_exact(float z, float sxy, const NscBinParams &bp)
{
    return nsc_row_from_elev(nsc_atan2f_cr(z, sqrtf(sxy)), bp);        // range_image.py:170-171
}

// ---- fast estimate --------------------------------------------------------------------------

NSC_HD float nsc_rcp_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);    // v_rcp_f32, 1 ULP
#elif defined(NSC_TEST_APPROX_BIAS)  /* host margin tests: push the 1-ULP error to either side */
    return nextafterf(1.0f / x, NSC_TEST_APPROX_BIAS > 0 ? INFINITY : 0.0f);
#else
    return 1.0f / x;
#endif
}

NSC_HD float nsc_sqrt_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);   // v_sqrt_f32, 1 ULP
#elif defined(NSC_TEST_APPROX_BIAS)
    return nextafterf(sqrtf(x), NSC_TEST_APPROX_BIAS > 0 ? 0.0f : INFINITY);
#else
    return sqrtf(x);
#endif
}

NSC_HD float nsc_rsq_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);    // v_rsq_f32, 1 ULP
#elif defined(NSC_TEST_APPROX_BIAS)
    return nextafterf(1.0f / sqrtf(x), NSC_TEST_APPROX_BIAS > 0 ? INFINITY : 0.0f);
#else
    return 1.0f / sqrtf(x);
#endif
}

// atan(t) for t in [0,1]: t * P(t^2), degree-7 minimax (3.7e-8 rad exact, 1.4e-7 rad in float32)
NSC_HD float nsc_atan01(float t)
{
    const float w = t * t;
    float p = -4.0545659302e-03f;
    p = __builtin_fmaf(p, w, 2.1862953564e-02f);
    p = __builtin_fmaf(p, w, -5.5912321150e-02f);
    p = __builtin_fmaf(p, w, 9.6421969742e-02f);
    p = __builtin_fmaf(p, w, -1.3908629443e-01f);
    p = __builtin_fmaf(p, w, 1.9946565639e-01f);
    p = __builtin_fmaf(p, w, -3.3329860785e-01f);
    p = __builtin_fmaf(p, w, 9.9999933558e-01f);
    return p * t;
}

// atan(t) for |t| <= 0.62: degree-5 minimax in t^2 (1.3e-8 rad exact, 8e-8 rad in float32)
NSC_HD float nsc_atan_small(float t)
{
    const float w = t * t;
    float p = -3.6013321370e-02f;
    p = __builtin_fmaf(p, w, 8.9950942561e-02f);
    p = __builtin_fmaf(p, w, -1.3851101441e-01f);
    p = __builtin_fmaf(p, w, 1.9954741257e-01f);
    p = __builtin_fmaf(p, w, -3.3331274679e-01f);
    p = __builtin_fmaf(p, w, 9.9999973037e-01f);
    return p * t;
}

// Column estimate.  Returns true when `col` is certain (estimate farther than az_delta from every
// column edge); NaN estimates (x = y = 0) and exact octant boundaries come out uncertain.
NSC_HD bool nsc_col_fast(float y, float x, float az_delta, int &col)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * nsc_rcp_approx(mx);
    const float q = nsc_atan01(t) * 57.29577951308232f;   // columns inside the octant, [0,45]
    const float fi = floorf(q);
    const float f = q - fi;
    const int iq = (int)fi;
    int ib = (ax >= ay) ? iq : 89 - iq;                   // column of the first-quadrant angle
    ib = (x < 0.0f) ? 179 - ib : ib;                      // angle from +x, [0,180)
    col = (y < 0.0f) ? 179 - ib : 180 + ib;               // + pi shift of range_image.py:167
    return (f > az_delta) && (f < 1.0f - az_delta);
}

// Row estimate; same contract.  Rows beyond the FOV clamp (range_image.py:187-191), so estimates
// far outside [0,E] are certain too.
NSC_HD bool nsc_row_fast(float z, float sxy, const NscBinParams &bp, int &row)
{
    const float rxy = nsc_sqrt_approx(sxy);
    const float az = fabsf(z);
    const float mx = fmaxf(az, rxy), mn = fminf(az, rxy);
    const float t = mn * nsc_rcp_approx(mx);
    float e = nsc_atan01(t);
    e = (az > rxy) ? 1.5707963267948966f - e : e;
    e = (z < 0.0f) ? -e : e;
    const float u = (e - bp.emin_f) * bp.el_scale;
    const float fi = floorf(u);
    const float f = u - fi;
    int r = (int)fminf(fmaxf(fi, 0.0f), (float)(bp.E - 1));
    row = r;
    const float d = bp.el_delta;
    return ((f > d) && (f < 1.0f - d)) || (u < -d) || (u > (float)bp.E + d);
}

// Row estimate for sensors whose FOV edges lie within |tan(elevation)| < 0.58 (about +-30 deg, every
// FOV the reference configures): t = z / rxy straight from v_rsq_f32, no octant logic.  |t| is
// clamped to 0.62, beyond which the row is clamped to 0 / E-1 with a margin of > 0.4 rows.
NSC_HD bool nsc_row_fast_narrow(float z, float sxy, const NscBinParams &bp, int &row)
{
    float t = z * nsc_rsq_approx(sxy);
    t = fminf(fmaxf(t, -0.62f), 0.62f);       // z = 0 and sxy = 0 gives NaN: ruled uncertain below
    const float u = __builtin_fmaf(nsc_atan_small(t), bp.el_u_scale, bp.el_u_bias);
    const float fi = floorf(u);
    const float f = u - fi;
    row = (int)fminf(fmaxf(fi, 0.0f), (float)(bp.E - 1));
    const float d = bp.el_delta;
    return (((f > d) && (f < 1.0f - d)) || (u < -d) || (u > (float)bp.E + d)) && (sxy > 0.0f);
}

// One point -> (pixel, squared range).  Returns 0 if the point is dropped, else 1 | 2*(exact path
// decided the column) | 4*(exact path decided the row).            range_image.py:151-202
NSC_HD int nsc_point_pixel(float x, float y, float z, const NscBinParams &bp, int &pix, float &s)
{
    float sxy;
    if (bp.simple_valid) {
        sxy = x * x + y * y;
        s = sxy + z * z;
        if (!(s >= bp.s_lo && s <= bp.s_hi)) return 0;                // :151-155, :174-177
    } else {
        const bool fin = (fabsf(x) < INFINITY) && (fabsf(y) < INFINITY) && (fabsf(z) < INFINITY);
        sxy = nsc_clip_sq(x) + nsc_clip_sq(y);
        s = sxy + nsc_clip_sq(z);
        if (!(fin && s >= bp.s_lo && s <= bp.s_hi)) return 0;
    }
    int col, row, flags = 1;
    const bool cok = nsc_col_fast(y, x, bp.az_delta, col);
    const bool rok = bp.narrow_fov ? nsc_row_fast_narrow(z, sxy, bp, row) : nsc_row_fast(z, sxy, bp, row);
    if (__builtin_expect(!(cok && rok), 0)) {
        if (!cok) { col = nsc_col_exact(y, x); flags |= 2; }
        if (!rok) { row = nsc_row_exact(z, sxy, bp); flags |= 4; }
    }
    pix = row * NSC_A + col;                                          // :202
    return flags;
}

// ---- lean estimate (the streaming loop of encode_fast_kernel) ----------------------------------
// Same definition, same polynomials and margins as above, arranged for the fewest VALU instructions:
//   * the 180/pi of the column estimate is folded into the polynomial coefficients;
//   * "f in (d, 1-d)" is one compare, |f - 1/2| < 1/2 - d (f - 1/2 rounds by <= 2^-26: covered by NSC_LEAN_GUARD);
//   * rows beyond the FOV are made certain by clamping the row coordinate to [-1/2, E + 1/2] (f = 1/2 there);
//   * no NaN reaches the row estimate: x = y = z = 0 fails the range window, which requires s_lo > 0.
// Valid only for bp.simple_valid && bp.narrow_fov && bp.s_lo > 0 (nsc_lean_ok); everything else runs
// nsc_point_pixel.  Returns 0 = dropped, 1 = (pix, s) certain, 2 = uncertain: the caller resolves the point with
// nsc_point_exact (the kernel parks it in an LDS queue and resolves the queue after the stream).
#define NSC_LEAN_GUARD 2.0e-7f

NSC_HD float nsc_clampf(float v, float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi);          // v_med3_f32 (no NaN reaches it, see above)
#else
    return fminf(fmaxf(v, lo), hi);
#endif
}

// atan(t) * 180/pi for t in [0,1]: the coefficients of nsc_atan01 times 57.29577951308232, rounded to float32
NSC_HD float nsc_atan01_deg(float t)
{
    const float w = t * t;
    float p = -2.3230952024e-01f;
    p = __builtin_fmaf(p, w, 1.2526550293e+00f);
    p = __builtin_fmaf(p, w, -3.2035398483e+00f);
    p = __builtin_fmaf(p, w, 5.5245718956e+00f);
    p = __builtin_fmaf(p, w, -7.9690575600e+00f);
    p = __builtin_fmaf(p, w, 1.1428540230e+01f);
    p = __builtin_fmaf(p, w, -1.9096603394e+01f);
    p = __builtin_fmaf(p, w, 5.7295742035e+01f);
    return p * t;
}

NSC_HD bool nsc_lean_ok(const NscBinParams &bp) { return bp.simple_valid && bp.narrow_fov && bp.s_lo > 0.0f; }

NSC_HD float nsc_fractf(float v)          // v - floor(v), in [0, 1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(v);                  // v_fract_f32
#else
    const float f = v - floorf(v);
    return f < 1.0f ? f : 0x1.fffffep-1f;               // v_fract_f32 clamps below 1
#endif
}

NSC_HD int nsc_mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);                               // v_mul_i32_i24: full rate (v_mul_lo_u32 is quarter rate)
#else
    return a * b;
#endif
}

// max / min of |x| and |y|: on the device one VOP3 instruction each with |.| source modifiers (fmaxf(fabsf, fabsf)
// compiles to two extra canonicalising v_max_f32 under IEEE mode)
NSC_HD float nsc_absmax(float x, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return fmaxf(fabsf(x), fabsf(y));
#endif
}

NSC_HD float nsc_absmin(float x, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return fminf(fabsf(x), fabsf(y));
#endif
}

NSC_HD float nsc_absdiff(float x, float y)      // |x| - |y|
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_sub_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return fabsf(x) - fabsf(y);
#endif
}

// First-octant angle q = atan(min/max) in degrees, [0,45]  ->  column coordinate (atan2(y,x) + 180 deg), [0,360], with
// three sign transfers (v_bfi_b32) instead of three compare/subtract/select triples on the integer column:
//   a1 = |x| < |y| ? 90 - q : q;   a2 = x < 0 ? 180 - a1 : a1;   result = y < 0 ? 180 - a2 : 180 + a2.
// The four float additions round by at most 2^-19 + 2^-18 + 2^-17 + 2^-16 < 2.9e-5 columns in total, less than the
// 4e-5 (NSC_AZ_EST_ERR, part of az_delta) by which a CERTAIN q stays clear of every integer: truncating the result
// gives the same column as unfolding (int)q.  Zero coordinates of either sign give q = 0 or NaN: never certain.
NSC_HD float nsc_unfold_octant(float q, float x, float y)
{
    const float d = nsc_absdiff(x, y);                         // sign set <=> |x| < |y| (the two octants that swap)
    const float c = 45.0f - q;
    const float b = 45.0f + __builtin_copysignf(c, d);         // 90 - a1
    const float a2 = 90.0f - __builtin_copysignf(b, x);
    return 180.0f + __builtin_copysignf(a2, y);
}

// Returns false if the point is dropped; else `certain` says whether (pix, s) is final or pix only the estimate.
NSC_HD bool nsc_point_lean_flags(float x, float y, float z, const NscBinParams &bp, int &pix, float &s, bool &certain)
{
    const float sxy = x * x + y * y;
    s = sxy + z * z;
    certain = false;
    if (!(s >= bp.s_lo && s <= bp.s_hi)) return false;                // :151-155, :174-177 (inf/NaN fall out here)
    // column: octant reduction + degree-7 minimax atan, in columns
    const float q = nsc_atan01_deg(nsc_absmin(x, y) * nsc_rcp_approx(nsc_absmax(x, y)));   // [0,45]; NaN for x = y = 0
    const float gc = nsc_fractf(q) - 0.5f;
    const bool cok = fabsf(gc) < (0.5f - NSC_LEAN_GUARD) - bp.az_delta;
    const int col = (int)nsc_unfold_octant(q, x, y);                  // garbage when !cok (never used then)
    // row: t = z / rxy from v_rsq_f32, degree-5 atan, affine map to rows; clamped so that out-of-FOV rows are certain
    // (f = 1/2 at both ends) and the truncation below lands in [0, E - 1]
    const float t = nsc_clampf(z * nsc_rsq_approx(sxy), -0.62f, 0.62f);
    const float u = nsc_clampf(__builtin_fmaf(nsc_atan_small(t), bp.el_u_scale, bp.el_u_bias), -0.5f,
                               (float)bp.E - 0.5f);
    const float gr = nsc_fractf(u) - 0.5f;
    const int row = (int)u;                                           // u in [-0.5, E - 0.5]: (-0.5, 0) truncates to row 0
    const bool rok = fabsf(gr) < (0.5f - NSC_LEAN_GUARD) - bp.el_delta;
    pix = nsc_mul24(row, NSC_A) + col;
    certain = cok && rok;
    return true;
}

#if defined(__HIPCC__)
// Two points at once: the same operations in the same order as nsc_point_lean_flags on each point, with the
// multiplies, adds and FMAs of the two points in the two halves of gfx950's packed float32 instructions
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per lane and issue slot, each rounded exactly like
// its scalar form) -- ~26 of the ~55 VALU instructions per point are shared by the pair.  Nothing is skipped for a
// dropped point (its results are simply not used), so the range test moves from a branch to two flags.
typedef float nsc_f32x2 __attribute__((ext_vector_type(2)));

struct NscLeanPair {
    float s[2];          // squared range
    int pix[2];          // pixel (final when ok, the estimate otherwise)
    bool ok[2];          // inside the range window and certain
    bool park[2];        // inside the range window, not certain: resolve with nsc_point_exact
};

__device__ __forceinline__ nsc_f32x2 nsc_pk_fma(nsc_f32x2 a, nsc_f32x2 b, float c)
{
    return __builtin_elementwise_fma(a, b, nsc_f32x2{c, c});
}

__device__ __forceinline__ NscLeanPair nsc_point_lean_pair(float xa, float ya, float za, float xb, float yb, float zb,
                                                           const NscBinParams &bp)
{
    NscLeanPair r;
    // (x^2, y^2) of one point is one packed multiply on the register pair the load left them in; the empty asm
    // statements keep the SLP vectoriser from pairing the remaining scalar operations ACROSS the two points, which
    // costs a register move per operand
    float sxy0 = xa * xa + ya * ya, sxy1 = xb * xb + yb * yb;
    asm("" : "+v"(sxy0));
    asm("" : "+v"(sxy1));
    float zz0 = za * za, zz1 = zb * zb;
    asm("" : "+v"(zz0));
    asm("" : "+v"(zz1));
    r.s[0] = sxy0 + zz0;
    asm("" : "+v"(r.s[0]));
    r.s[1] = sxy1 + zz1;
    const bool keep0 = r.s[0] >= bp.s_lo && r.s[0] <= bp.s_hi, keep1 = r.s[1] >= bp.s_lo && r.s[1] <= bp.s_hi;
    // column
    const nsc_f32x2 mn = {nsc_absmin(xa, ya), nsc_absmin(xb, yb)};
    const nsc_f32x2 rc = {nsc_rcp_approx(nsc_absmax(xa, ya)), nsc_rcp_approx(nsc_absmax(xb, yb))};
    const nsc_f32x2 t = mn * rc;
    const nsc_f32x2 w = t * t;
    nsc_f32x2 p = nsc_pk_fma(nsc_f32x2{-2.3230952024e-01f, -2.3230952024e-01f}, w, 1.2526550293e+00f);
    p = nsc_pk_fma(p, w, -3.2035398483e+00f);
    p = nsc_pk_fma(p, w, 5.5245718956e+00f);
    p = nsc_pk_fma(p, w, -7.9690575600e+00f);
    p = nsc_pk_fma(p, w, 1.1428540230e+01f);
    p = nsc_pk_fma(p, w, -1.9096603394e+01f);
    p = nsc_pk_fma(p, w, 5.7295742035e+01f);
    const nsc_f32x2 q = p * t;
    const nsc_f32x2 gc = nsc_f32x2{nsc_fractf(q.x), nsc_fractf(q.y)} - nsc_f32x2{0.5f, 0.5f};
    const float cthr = (0.5f - NSC_LEAN_GUARD) - bp.az_delta;
    const bool cok0 = fabsf(gc.x) < cthr, cok1 = fabsf(gc.y) < cthr;
    // nsc_unfold_octant on both halves
    const nsc_f32x2 c = nsc_f32x2{45.0f, 45.0f} - q;
    const nsc_f32x2 b = nsc_f32x2{45.0f, 45.0f} + nsc_f32x2{__builtin_copysignf(c.x, nsc_absdiff(xa, ya)),
                                                           __builtin_copysignf(c.y, nsc_absdiff(xb, yb))};
    const nsc_f32x2 a2 = nsc_f32x2{90.0f, 90.0f} - nsc_f32x2{__builtin_copysignf(b.x, xa), __builtin_copysignf(b.y, xb)};
    const nsc_f32x2 colf = nsc_f32x2{180.0f, 180.0f} + nsc_f32x2{__builtin_copysignf(a2.x, ya), __builtin_copysignf(a2.y, yb)};
    // row
    const nsc_f32x2 e = {nsc_clampf(za * nsc_rsq_approx(sxy0), -0.62f, 0.62f), nsc_clampf(zb * nsc_rsq_approx(sxy1), -0.62f, 0.62f)};
    const nsc_f32x2 v = e * e;
    nsc_f32x2 a = nsc_pk_fma(nsc_f32x2{-3.6013321370e-02f, -3.6013321370e-02f}, v, 8.9950942561e-02f);
    a = nsc_pk_fma(a, v, -1.3851101441e-01f);
    a = nsc_pk_fma(a, v, 1.9954741257e-01f);
    a = nsc_pk_fma(a, v, -3.3331274679e-01f);
    a = nsc_pk_fma(a, v, 9.9999973037e-01f);
    a = a * e;
    const nsc_f32x2 uu = __builtin_elementwise_fma(a, nsc_f32x2{bp.el_u_scale, bp.el_u_scale},
                                                   nsc_f32x2{bp.el_u_bias, bp.el_u_bias});
    const float uhi = (float)bp.E - 0.5f;
    const nsc_f32x2 u = {nsc_clampf(uu.x, -0.5f, uhi), nsc_clampf(uu.y, -0.5f, uhi)};
    const nsc_f32x2 gr = nsc_f32x2{nsc_fractf(u.x), nsc_fractf(u.y)} - nsc_f32x2{0.5f, 0.5f};
    const float rthr = (0.5f - NSC_LEAN_GUARD) - bp.el_delta;
    const bool rok0 = fabsf(gr.x) < rthr, rok1 = fabsf(gr.y) < rthr;
    r.pix[0] = nsc_mul24((int)u.x, NSC_A) + (int)colf.x;
    r.pix[1] = nsc_mul24((int)u.y, NSC_A) + (int)colf.y;
    const bool c0 = cok0 && rok0, c1 = cok1 && rok1;
    r.ok[0] = keep0 && c0;
    r.ok[1] = keep1 && c1;
    r.park[0] = keep0 && !c0;
    r.park[1] = keep1 && !c1;
    return r;
}
#endif

// 0 = dropped, 1 = (pix, s) certain, 2 = uncertain (host-side checks)
NSC_HD int nsc_point_lean(float x, float y, float z, const NscBinParams &bp, int &pix, float &s)
{
    bool certain;
    if (!nsc_point_lean_flags(x, y, z, bp, pix, s, certain)) return 0;
    return certain ? 1 : 2;
}

// Exact chain for a point that passed the range window (the definition of the pixel).
NSC_HD int nsc_point_exact(float x, float y, float z, const NscBinParams &bp)
{
    const float sxy = nsc_clip_sq(x) + nsc_clip_sq(y);
    return nsc_row_exact(z, sxy, bp) * NSC_A + nsc_col_exact(y, x);
}

// ---- host-side setup (plain C++, also used by the CPU margin tests) ---------------------------
// (host functions: hipcc's device pass parses but never emits them)

// smallest float32 s >= 0 with pred(s) true, pred monotone false->true over non-negative floats
template <class P> inline float nsc_first_true(P pred)
{
    uint32_t lo = 0u, hi = 0x7f800000u;          // +0 .. +inf bit patterns are ordered like floats
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2u;
        float f;
        __builtin_memcpy(&f, &mid, 4);
        if (pred(f)) hi = mid; else lo = mid + 1u;
    }
    float f;
    __builtin_memcpy(&f, &lo, 4);
    return f;
}

// Largest distance (in columns) between an ideal column edge c*2pi/360 - pi and the float32
// threshold at which the exact chain actually switches to column c.
inline float nsc_az_edge_slack()
{
    static const float slack = [] {
        double worst = 0.0;
        for (int c = 1; c < NSC_A; ++c) {
            // bisect over the ordered float32 angles in [-pi_f, pi_f)
            auto key = [](float a) { uint32_t u; __builtin_memcpy(&u, &a, 4);
                                     return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
            auto unkey = [](uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
                                          float a; __builtin_memcpy(&a, &u, 4); return a; };
            uint32_t lo = key(-NSC_PI_F), hi = key(NSC_PI_F);
            while (lo < hi) {
                uint32_t mid = lo + (hi - lo) / 2u;
                if (nsc_col_from_angle(unkey(mid)) >= c) hi = mid; else lo = mid + 1u;
            }
            const double thr = (double)unkey(lo);
            const double ideal = -M_PI + (double)c * (2.0 * M_PI / NSC_A);
            // thr is the first float32 angle in column c; the last one in column c-1 is one ULP
            // below.  Both bracket where the exact chain switches.
            const double below = (double)nextafterf((float)thr, -10.0f);
            const double d1 = fabs(thr - ideal), d0 = fabs(below - ideal);
            const double d = (d1 > d0 ? d1 : d0) * (NSC_A / (2.0 * M_PI));
            if (d > worst) worst = d;
        }
        return (float)worst;
    }();
    return slack;
}

// Error budget of the float32 estimates (validated by tests/test_binning_margins.py on 1e8 points
// and by the -m gpu debug-bin tests): v_rcp 1 ULP + polynomial + roundings, in bin units.
#define NSC_AZ_EST_ERR 4.0e-5f
#define NSC_EL_EST_ERR 3.0e-5f

inline NscBinParams nsc_make_bin_params(int E, double emin, double emax, float rmin, float rmax,
                                        int elev_f64)
{
    NscBinParams bp;
    bp.emin = emin;
    bp.espan = emax - emin;
    bp.emin_f = (float)emin;
    bp.espan_f = (float)bp.espan;
    bp.el_scale = (float)((double)E / bp.espan);
    bp.az_delta = nsc_az_edge_slack() + NSC_AZ_EST_ERR;
    bp.el_delta = NSC_EL_EST_ERR;
    bp.E = E;
    bp.elev_f64 = elev_f64;
    // range filter r >= rmin && r <= rmax with r = sqrtf(s) correctly rounded (monotone in s)
    // (range_image.py:174): expressed on s so the kernels never take a per-point sqrt
    bp.s_lo = nsc_first_true([&](float s) { return sqrtf(s) >= rmin; });
    const float above = nsc_first_true([&](float s) { return sqrtf(s) > rmax; });
    bp.s_hi = nextafterf(above, 0.0f);
    bp.el_u_scale = bp.el_scale;
    bp.el_u_bias = (float)(-emin * ((double)E / bp.espan));
    bp.narrow_fov = (fabs(tan(emin)) < 0.58 && fabs(tan(emax)) < 0.58 && fabs(emin) < 1.0 && fabs(emax) < 1.0);
    bp.simple_valid = (bp.s_hi < 1e10f);
    return bp;
}
