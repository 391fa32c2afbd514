// nsc_encoder.hip -- gfx950 kernels + C ABI for the descriptor encoder (include/nsc.h).
//
// Path (reference file:line):
//   RangeImageProjector.project        src/encoding/range_image.py:129-214   -> scatter_*()
//   interpolate_range_image            src/encoding/range_image.py:15-89     -> interp_row(), row copy
//   SpectralEncoder.encode_range_image src/encoding/spectral_encoder.py:160-204 -> fft_row(), finish_image()
//
// Kernels
//   encode_fused_kernel   one workgroup per cloud: stream the cloud's points once from HBM
//                         (16 B/point, coalesced 1 KiB per wave-instruction), LDS-resident E x 360
//                         squared-range image updated with ds_min_u32, then interpolation, 360-point
//                         real FFT (one wavefront per row), histogram and normalisation -- the only
//                         HBM traffic is the points in and 3 200 B out.
//   scatter_split_kernel  small batches: a cloud is split over several workgroups, partial LDS
//                         images are merged with global atomicMin into a workspace image
//   finish_kernel         workspace image / caller range images -> descriptor (same finish_image())
//   point_bins_kernel     parity triage (per-point pixel index)
//
// Built with -ffp-contract=off: float32/float64 expressions round exactly as written; FMAs appear
// only where fma()/__builtin_fmaf is spelled out.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "../../include/nsc.h"
#include "../../include/nsc_debug.h"
#include "nsc_math.h"

namespace {

#include "nsc_fill.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int A = NSC_A;      // 360 columns
constexpr int F = NSC_F;      // 181 rfft bins
constexpr int NH = 180;       // complex length of the packed real FFT
constexpr int MAXR = 16;      // target rows the finish stage supports
constexpr int MAXE = 64;      // projector rows the LDS image supports

struct EncDev {
    NscBinParams bp;
    int E, R, B;
    float eps;
    int interp;
#ifdef NSC_DEV_TUNING
    int dev_skip;          // development builds only (NSC_TUNE_SKIP_FINISH): phase masks for tools/ab_enc.py
#endif
};

// Phase masks exist only in development builds (NSC_DEV_BUILD=1); the shipped kernels carry none of it.
#ifdef NSC_DEV_TUNING
#define NSC_DEV_SKIP(d, bit) (((d).dev_skip & (bit)) != 0)
#define NSC_DEV_MODE(d) ((d).dev_skip)
#else
#define NSC_DEV_SKIP(d, bit) false
#define NSC_DEV_MODE(d) 0
#endif

// exp(-2 pi i j / 360) = (cos, -sin): table holds (cos, sin)
__device__ const double2 g_tw360[A] = {
#include "nsc_twiddle360.inc"
};

// ---------------------------------------------------------------------------------------------
// LDS carve-up (bytes), identical on host and device
// ---------------------------------------------------------------------------------------------
struct LdsPlan {
    int img, pool, tw, fft, seg, misc, total;
};

constexpr int TW_N = 240;        // twiddle indices used: stages <= 2*2*59 = 236, unpack <= 180
constexpr int ROW_BYTES = A * 4; // one image row; once a row's FFT has read it, the row's bytes are
constexpr int HIST_OFF = 736;    // reused: |X| magnitudes at +0 (181 floats), histogram at +736
constexpr int MAXB = (ROW_BYTES - HIST_OFF) / 4;   // 176 bins fit behind the magnitudes

__host__ __device__ inline LdsPlan lds_plan(int E, int R, int B, int nw)
{
    LdsPlan p;
    int o = 0;
    p.img = o;  o += E * ROW_BYTES;
    p.pool = o; o += (E != R) ? R * ROW_BYTES : 0;
    p.tw = o;   o += TW_N * 16;
    // With row pooling the FFT reads the pooled rows, the E-row image is dead by then: the per-wave FFT scratch
    // goes into it when it fits (a 64-row image + its own scratch would not fit the 160 KB of a CU otherwise).
    if (E != R && E * ROW_BYTES >= nw * NH * 16) {
        p.fft = p.img;
    } else {
        p.fft = o;  o += nw * NH * 16;
    }
    p.seg = o;  o += ((2 * B * 4) + 15) & ~15;
    p.misc = o; o += MAXR * 8 + MAXE * 4 + MAXE * 4;   // rowsum f64[16], rowflag int[64], rowsrc int[64]
    p.total = o;
    return p;
}

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the point where lanes exchange data through LDS.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// scatter: one point into the LDS squared-range image          range_image.py:151-208
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void scatter_point(float x, float y, float z, const NscBinParams &bp,
                                              unsigned *img)
{
    int pix; float s;
    if (nsc_point_pixel(x, y, z, bp, pix, s))
        atomicMin(&img[pix], __float_as_uint(s));   // ds_min_u32: s >= 0, so uint order == float order (:208)
}

template <int NT, int U>
__device__ __forceinline__ void scatter_range(const float *__restrict__ pts, long long p0, long long p1,
                                              int stride, int tid, const NscBinParams &bp, unsigned *img)
{
    const int n = (int)(p1 - p0);              // a cloud holds < 2^31 points
    if (stride == 4) {
        const f32x4 *P = reinterpret_cast<const f32x4 *>(pts) + p0;
        for (int i = tid; i < n; i += NT * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = i + u * NT;
                if (j < n) v[u] = __builtin_nontemporal_load(&P[j]);
                else v[u] = f32x4{NAN, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) scatter_point(v[u].x, v[u].y, v[u].z, bp, img);
        }
    } else {
        const float *P = pts + p0 * 3;
        for (int i = tid; i < n; i += NT * U) {
            float v[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = i + u * NT;
                if (j < n) { v[u][0] = P[j * 3LL]; v[u][1] = P[j * 3LL + 1]; v[u][2] = P[j * 3LL + 2]; }
                else { v[u][0] = NAN; v[u][1] = 0.f; v[u][2] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) scatter_point(v[u][0], v[u][1], v[u][2], bp, img);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// interpolate one row in LDS (one wavefront)                    range_image.py:33-64
// ---------------------------------------------------------------------------------------------
// method: 0 = count the valid pixels only, 1 = circular linear (np.interp, :52-64), 2 = circular nearest (:66-75)
// v[j] = row[lane + 64 j] (0 beyond column 359), already in registers
__device__ __forceinline__ int interp_row_v(float *row, int lane, int method, const float (&v)[6])
{
    unsigned long long m[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = __ballot(v[j] > 0.0f);    // :35 valid_mask = row > 0
    int nv = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nv += __popcll(m[j]);
    if (method == 0 || nv == 0 || nv == A) return nv;            // :37-43

    const unsigned long long below_mask = (1ull << lane) - 1ull;         // lanes < lane
    const unsigned long long above_mask = ~(below_mask | (1ull << lane)); // lanes > lane
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = lane + 64 * j;
        if (c < A && !(v[j] > 0.0f)) {
            // circular previous valid column p < c (position may be negative = wrapped by -360)
            int p = 0; bool fp_found = false;
            {
                const unsigned long long b = m[j] & below_mask;
                if (b) { p = 64 * j + 63 - __clzll(b); fp_found = true; }
            }
#pragma unroll
            for (int jj = 5; jj >= 0; --jj)
                if (jj < j && !fp_found && m[jj]) { p = 64 * jj + 63 - __clzll(m[jj]); fp_found = true; }
#pragma unroll
            for (int jj = 5; jj >= 0; --jj)
                if (jj > j && !fp_found && m[jj]) { p = 64 * jj + 63 - __clzll(m[jj]) - A; fp_found = true; }
            if (!fp_found) { const unsigned long long b = m[j] & above_mask; p = 64 * j + 63 - __clzll(b) - A; }

            // circular next valid column q > c (position may exceed 359 = wrapped by +360)
            int q = 0; bool fq_found = false;
            {
                const unsigned long long b = m[j] & above_mask;
                if (b) { q = 64 * j + __ffsll((long long)b) - 1; fq_found = true; }
            }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
                if (jj > j && !fq_found && m[jj]) { q = 64 * jj + __ffsll((long long)m[jj]) - 1; fq_found = true; }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
                if (jj < j && !fq_found && m[jj]) { q = 64 * jj + __ffsll((long long)m[jj]) - 1 + A; fq_found = true; }
            if (!fq_found) { const unsigned long long b = m[j] & below_mask; q = 64 * j + __ffsll((long long)b) - 1 + A; }

            const int vp = p < 0 ? p + A : p, vq = q >= A ? q - A : q;   // the two pixels' own columns
            if (method == 2) {
                // nearest valid pixel by circular distance; np.argmin takes the first minimum of the ascending
                // valid_indices, i.e. the smaller column of two equally near pixels (:68-75)
                const int dp = c - p, dq = q - c;
                const int pick = (dp < dq) ? vp : (dq < dp) ? vq : min(vp, vq);
                row[c] = row[pick];                              // reads original pixels only (pick is valid)
            } else {
                // np.interp in float64: slope*(x - xp[j]) + fp[j]   (:63, numpy compiled_base.c)
                const double f0 = (double)row[vp];
                const double f1 = (double)row[vq];
                const double slope = (f1 - f0) / (double)(q - p);
                const double val = slope * (double)(c - p) + f0;
                row[c] = (float)val;                             // :64 store into the float32 image
            }
        }
    }
    return nv;
}

__device__ __forceinline__ int interp_row(float *row, int lane, int method)
{
    float v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = lane + 64 * j;
        v[j] = (c < A) ? row[c] : 0.0f;
    }
    return interp_row_v(row, lane, method, v);
}

// ---------------------------------------------------------------------------------------------
// 360-point real FFT magnitude of one row (one wavefront), float64
//   z[n] = x[2n] + i x[2n+1], n < 180;  180 = 4 x 5 x 3 x 3 mixed-radix Stockham autosort: stage with
//   radix R and Ns = product of the earlier radices, butterfly j < 180/R, k = j mod Ns:
//       v[r] = in[j + r*180/R] * W^(r k), W = exp(-2 pi i/(Ns R));  v = DFT_R(v);
//       out[(j div Ns) Ns R + k + q Ns] = v[q]
//   The small DFTs are register butterflies with literal constants; the only LDS twiddle reads are
//   the R-1 inter-stage factors, issued together.  One wave owns the row and LDS executes a wave's
//   operations in order, so every stage works in place (all reads precede all writes in program
//   order).  X[k] = E[k] + W360^k O[k] unpacking for k = 0..180.     spectral_encoder.py:180-186
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void twmul(double &re, double &im, const double2 w)   // *= (w.x - i w.y)
{
    const double r = fma(re, w.x, im * w.y);
    const double i = fma(im, w.x, -(re * w.y));
    re = r; im = i;
}

__device__ __forceinline__ void dft3(double (&re)[3], double (&im)[3])
{
    constexpr double S = 0.86602540378443864676;       // sin(2 pi / 3)
    const double t1r = re[1] + re[2], t1i = im[1] + im[2];
    const double t2r = fma(-0.5, t1r, re[0]), t2i = fma(-0.5, t1i, im[0]);
    const double t3r = S * (re[1] - re[2]), t3i = S * (im[1] - im[2]);
    re[0] = re[0] + t1r; im[0] = im[0] + t1i;
    re[1] = t2r + t3i;   im[1] = t2i - t3r;            // t2 - i t3
    re[2] = t2r - t3i;   im[2] = t2i + t3r;            // t2 + i t3
}

__device__ __forceinline__ void dft4(double (&re)[4], double (&im)[4])
{
    const double t0r = re[0] + re[2], t0i = im[0] + im[2];
    const double t1r = re[0] - re[2], t1i = im[0] - im[2];
    const double t2r = re[1] + re[3], t2i = im[1] + im[3];
    const double t3r = re[1] - re[3], t3i = im[1] - im[3];
    re[0] = t0r + t2r; im[0] = t0i + t2i;
    re[2] = t0r - t2r; im[2] = t0i - t2i;
    re[1] = t1r + t3i; im[1] = t1i - t3r;              // t1 - i t3
    re[3] = t1r - t3i; im[3] = t1i + t3r;              // t1 + i t3
}

__device__ __forceinline__ void dft5(double (&re)[5], double (&im)[5])
{
    constexpr double C1 = 0.30901699437494742410;      // cos(2 pi / 5)
    constexpr double C2 = -0.80901699437494742410;     // cos(4 pi / 5)
    constexpr double S1 = 0.95105651629515357212;      // sin(2 pi / 5)
    constexpr double S2 = 0.58778525229247312917;      // sin(4 pi / 5)
    const double t1r = re[1] + re[4], t1i = im[1] + im[4];
    const double t2r = re[2] + re[3], t2i = im[2] + im[3];
    const double t3r = re[1] - re[4], t3i = im[1] - im[4];
    const double t4r = re[2] - re[3], t4i = im[2] - im[3];
    const double m1r = fma(C2, t2r, fma(C1, t1r, re[0])), m1i = fma(C2, t2i, fma(C1, t1i, im[0]));
    const double m2r = fma(C1, t2r, fma(C2, t1r, re[0])), m2i = fma(C1, t2i, fma(C2, t1i, im[0]));
    const double n1r = fma(S2, t4r, S1 * t3r), n1i = fma(S2, t4i, S1 * t3i);
    const double n2r = fma(-S1, t4r, S2 * t3r), n2i = fma(-S1, t4i, S2 * t3i);
    re[0] = re[0] + t1r + t2r; im[0] = im[0] + t1i + t2i;
    re[1] = m1r + n1i; im[1] = m1i - n1r;              // m1 - i n1
    re[4] = m1r - n1i; im[4] = m1i + n1r;              // m1 + i n1
    re[2] = m2r + n2i; im[2] = m2i - n2r;              // m2 - i n2
    re[3] = m2r - n2i; im[3] = m2i + n2r;              // m2 + i n2
}

// stage-1 operands of one row: lane j < 45 holds z[j + 45 r] = (x[2(j+45r)], x[2(j+45r)+1]), r = 0..3
__device__ __forceinline__ void fft_load_row(const float *x, int lane, f32x2 (&in)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
        in[r] = (lane < 45) ? *reinterpret_cast<const f32x2 *>(&x[2 * (lane + 45 * r)]) : f32x2{0.f, 0.f};
}

// NR rows at once, each in its own scratch buffer: the same four stages, with the NR independent butterflies of a lane
// interleaved by the compiler (one row alone is a chain of LDS round trips with nothing to overlap them).  The
// inter-stage twiddles depend on the lane only and are read once for all rows.  Returns |X[k]| of row p for
// k = lane + 64 jj in mg[p][jj] (0 beyond k = 180); every read of buf precedes the return in program order, so the
// caller may store the magnitudes over the scratch.
// LEAN = true keeps the same arithmetic (same operations on the same operands: bit-identical results) but runs stages
// 2-4 one row after the other -- the caller of the first pair of a wave holds the second pair's 16 stage-1 registers
// through it, and with them the interleaved stages 3/4 take 82 VGPRs, the sequential ones 72.  The radix-5 stage is
// sequential in both forms and reads its four twiddles one at a time (all four at once: 88 VGPRs).  The register
// budget is what lets five encoder workgroups AND two waves of the co-resident GNN kernels share a SIMD
// (5 x 80 + 2 x 56 = 512; tests/test_abi_cpu.py::test_coresident_register_budget).
template <int NR, bool LEAN = false>
__device__ __forceinline__ void fft_rows(const f32x2 (*in)[4], double2 *const (&buf)[NR], const double2 *tw,
                                         float (&mg)[NR][3], int lane)
{
    {   // stage 1: R = 4, Ns = 1, 45 butterflies, input = the float32 image row (held in registers)
        if (lane < 45) {
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                double re[4], im[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { re[r] = (double)in[p][r].x; im[r] = (double)in[p][r].y; }
                dft4(re, im);
#pragma unroll
                for (int q = 0; q < 4; ++q) { double2 o; o.x = re[q]; o.y = im[q]; buf[p][4 * lane + q] = o; }
            }
        }
    }
    wave_sync();
    {   // stage 2: R = 5, Ns = 4, 36 butterflies, W = W360^(18 r k).  One row after the other (each in place in its
        // own buffer), the twiddles read one at a time: this stage is the register peak of the whole kernel.
        const int k = lane & 3;
        const int j0 = (lane >> 2) * 20 + k;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            double re[5], im[5];
            if (lane < 36) {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const double2 v = buf[p][lane + 36 * r];
                    re[r] = v.x; im[r] = v.y;
                }
#pragma unroll
                for (int r = 1; r < 5; ++r) {
                    const double2 w = tw[18 * r * k];
                    twmul(re[r], im[r], w);
                    asm volatile("" ::: "memory");     // keeps the four reads from being hoisted together (16 registers)
                }
                dft5(re, im);
            }
            wave_sync();
            if (lane < 36) {
#pragma unroll
                for (int q = 0; q < 5; ++q) { double2 o; o.x = re[q]; o.y = im[q]; buf[p][j0 + 4 * q] = o; }
            }
        }
    }
    wave_sync();
    // stages 3 and 4: R = 3, 60 butterflies each; stage 3: Ns = 20, W = W360^(6 r k), k = lane mod 20, output index
    // (lane div 20) 60 + k + 20 q; stage 4: Ns = 60, W = W360^(2 r lane), output index lane + 60 q
#pragma unroll
    for (int st = 3; st <= 4; ++st) {
        const int k = lane % 20;
        const int j0 = (st == 3) ? (lane / 20) * 60 + k : lane;
        const int qs = (st == 3) ? 20 : 60;
        double2 w1 = {0.0, 0.0}, w2 = {0.0, 0.0};
        if (lane < 60) {
            w1 = tw[(st == 3) ? 6 * k : 2 * lane];
            w2 = tw[(st == 3) ? 12 * k : 4 * lane];
        }
        if (LEAN) {
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                double re[3], im[3];
                if (lane < 60) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double2 v = buf[p][lane + 60 * r]; re[r] = v.x; im[r] = v.y; }
                    twmul(re[1], im[1], w1);
                    twmul(re[2], im[2], w2);
                    dft3(re, im);
                }
                wave_sync();
                if (lane < 60) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) { double2 o; o.x = re[q]; o.y = im[q]; buf[p][j0 + qs * q] = o; }
                }
            }
        } else {
            double re[NR][3], im[NR][3];
            if (lane < 60) {
#pragma unroll
                for (int p = 0; p < NR; ++p)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double2 v = buf[p][lane + 60 * r];
                        re[p][r] = v.x; im[p][r] = v.y;
                    }
#pragma unroll
                for (int p = 0; p < NR; ++p) {
                    twmul(re[p][1], im[p][1], w1);
                    twmul(re[p][2], im[p][2], w2);
                    dft3(re[p], im[p]);
                }
            }
            wave_sync();
            if (lane < 60) {
#pragma unroll
                for (int p = 0; p < NR; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) { double2 o; o.x = re[p][q]; o.y = im[p][q]; buf[p][j0 + qs * q] = o; }
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {                   // unpack the real spectrum, |X[k]| -> float32
        const int k = lane + 64 * jj;
#pragma unroll
        for (int p = 0; p < NR; ++p) mg[p][jj] = 0.0f;
        if (k <= NH) {
            const double2 w = tw[k];
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const double2 zk = buf[p][k == NH ? 0 : k];
                const double2 zn = buf[p][(k == 0 || k == NH) ? 0 : NH - k];
                const double sp = zk.x + zn.x, sm = zk.x - zn.x;   // a+c, a-c
                const double tp = zk.y + zn.y, tm = zk.y - zn.y;   // b+d, b-d
                const double xr = 0.5 * (sp + tp * w.x - sm * w.y);
                const double xi = 0.5 * (tm - sm * w.x - tp * w.y);
                // :183 (the x sqrt(360) of :186 undoes 'ortho').  |X|^2 in float64, one float32 rounding,
                // then a correctly rounded float32 sqrt: within 1 ULP of (float)sqrt(double)
                mg[p][jj] = sqrtf((float)(xr * xr + xi * xi));
            }
        }
        if (NR > 1) wave_sync();            // keeps the three passes' operands from being live at once (register budget)
    }
    wave_sync();
}

__device__ __forceinline__ void fft_row_regs(const f32x2 (&in)[4], double2 *buf, const double2 *tw, float *mags,
                                             int lane)
{
    float mg[1][3];
    double2 *const bufs[1] = {buf};
    fft_rows<1>(&in, bufs, tw, mg, lane);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int k = lane + 64 * jj;
        if (k <= NH) mags[k] = mg[0][jj];
    }
    wave_sync();
}

__device__ __forceinline__ void fft_row(const float *x, double2 *buf, const double2 *tw, float *mags, int lane)
{
    f32x2 in[4];
    fft_load_row(x, lane, in);
    fft_row_regs(in, buf, tw, mags, lane);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Twiddles and histogram segments -> LDS, by `nthr` cooperating threads (index t).  In the fused
// kernel one wave does this while the others already stream points, so none of its global-load
// latency is exposed.  The LUT is monotone: bin b owns frequencies [seg[b], seg[B+b]) with
// seg[b] = first k with lut[k] >= b and seg[B+b] = first k with lut[k] >= b+1 (equal for bins no
// frequency maps to).  A workgroup barrier must separate this from finish_image().
template <int NW>
__device__ __forceinline__ void setup_tables(unsigned char *lds, const EncDev &d, const int *__restrict__ lut,
                                             int t, int nthr)
{
    const LdsPlan lp = lds_plan(d.E, d.R, d.B, NW);
    double2 *tw = reinterpret_cast<double2 *>(lds + lp.tw);
    int *seg = reinterpret_cast<int *>(lds + lp.seg);
    const int B = d.B;
    for (int i = t; i < TW_N; i += nthr) tw[i] = g_tw360[i];
    for (int b = t; b < 2 * B; b += nthr) {
        const int key = (b < B) ? b : b - B + 1;
        int l = 0, h = F;
        while (l < h) { const int m = (l + h) >> 1; if (lut[m] >= key) h = m; else l = m + 1; }
        seg[b] = l;
    }
}

// ---------------------------------------------------------------------------------------------
// everything after the image exists in LDS
//   mode 0: img holds squared-range bits (scatter output)  -> sqrt, interpolate, row copy
//   mode 1: img holds float32 range images from the caller -> no interpolation (forward(), :231)
//   mode 2: img holds float32 range images from the caller -> interpolate + row copy only
//           (interpolate_range_image(), range_image.py:15-89), result to out_interp
template <int NW>
__device__ __forceinline__ void finish_image(unsigned char *lds, const EncDev &d, int mode,
                                             float *__restrict__ out_desc,
                                             float *__restrict__ out_raw, float *__restrict__ out_interp)
{
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = d.E, R = d.R, B = d.B;
    const LdsPlan lp = lds_plan(E, R, B, NW);
    float *img = reinterpret_cast<float *>(lds + lp.img);
    float *pool = reinterpret_cast<float *>(lds + lp.pool);
    const double2 *tw = reinterpret_cast<const double2 *>(lds + lp.tw);
    double2 *fftbuf = reinterpret_cast<double2 *>(lds + lp.fft) + wave * NH;
    const int *seg = reinterpret_cast<const int *>(lds + lp.seg);
    double *rowsum = reinterpret_cast<double *>(lds + lp.misc);
    int *rowflag = reinterpret_cast<int *>(lds + lp.misc + MAXR * 8);
    int *rowsrc = rowflag + MAXE;

    if (mode == 0 || mode == 2) {
        // each wave converts and interpolates the rows it owns: no workgroup barrier in between
        for (int r = wave; r < E; r += NW) {
            float *row = img + r * A;
            const unsigned *raw = reinterpret_cast<const unsigned *>(row);
            float *graw = out_raw ? out_raw + r * A : nullptr;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int c = lane + 64 * j;
                if (c < A && mode == 0) {
                    const unsigned v = raw[c];
                    // min over sqrtf(s_i) == sqrtf(min s_i): sqrtf is correctly rounded, hence monotone
                    const float rr = (v == NSC_EMPTY_BITS) ? 0.0f : sqrtf(__uint_as_float(v));   // :162,:214
                    row[c] = rr;
                    if (graw) graw[c] = rr;
                }
            }
            wave_sync();
            const int nv = interp_row(row, lane, mode == 2 ? (d.interp ? d.interp : 1) : d.interp);
            if (lane == 0) rowflag[r] = (nv > 0);
        }
        __syncthreads();
        if (d.interp || mode == 2) {                              // range_image.py:77-87
            unsigned long long ne = 0ull;
            for (int r = 0; r < E; ++r) ne |= (unsigned long long)(rowflag[r] != 0) << r;
            const unsigned long long all = (E == 64) ? ~0ull : ((1ull << E) - 1ull);
            if (ne != all && ne != 0ull) {                        // some (not all) rows are empty: rare
                if (tid == 0) {
                    unsigned long long m = ne;
                    for (int r = 0; r < E; ++r) rowsrc[r] = r;
                    for (int r = 0; r < E; ++r) {
                        if ((m >> r) & 1ull) continue;
                        for (int k = 1; k < E; ++k) {
                            if (r - k >= 0 && ((m >> (r - k)) & 1ull)) { rowsrc[r] = rowsrc[r - k]; m |= 1ull << r; break; }
                            if (r + k < E && ((m >> (r + k)) & 1ull)) { rowsrc[r] = r + k; m |= 1ull << r; break; }
                        }
                    }
                }
                __syncthreads();
                for (int r = wave; r < E; r += NW) {
                    const int sr = rowsrc[r];                     // always an original (never copied) row
                    if (sr != r)
                        for (int c = lane; c < A; c += 64) img[r * A + c] = img[sr * A + c];
                }
                __syncthreads();
            }
        }
        if (out_interp)
            for (int r = wave; r < E; r += NW)                    // rows this wave owns (or just copied)
                for (int c = lane; c < A; c += 64) out_interp[r * A + c] = img[r * A + c];
        if (mode == 2) return;
    }

    float *rows = img;
    if (E != R) {                                                 // adaptive_avg_pool2d, :171-176
        for (int i = tid; i < R * A; i += NT) {
            const int pr = i / A, c = i - pr * A;
            const int r0 = (pr * E) / R;
            const int r1 = ((pr + 1) * E + R - 1) / R;
            float sacc = 0.0f;
            for (int r = r0; r < r1; ++r) sacc += img[r * A + c];
            pool[i] = sacc / (float)(r1 - r0);
        }
        rows = pool;
        __syncthreads();
    }
    for (int r = wave; r < R; r += NW) {
        float *row = rows + r * A;
        float *mags = row;                                        // the row is dead after FFT stage 1
        float *hist = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(row) + HIST_OFF);
        fft_row(row, fftbuf, tw, mags, lane);
        double part = 0.0;
        for (int b = lane; b < B; b += 64) {
            float h = 0.0f;
            const int k1 = seg[B + b];
            for (int k = seg[b]; k < k1; k += 8) {               // scatter_add_, ascending k (:152-155)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (k + u < k1) ? mags[k + u] : 0.0f;   // loads issued together
#pragma unroll
                for (int u = 0; u < 8; ++u) h += v[u];            // h + 0.0f == h: order and rounding unchanged
            }
            hist[b] = h;
            part += (double)h;
        }
        part = wave_sum(part);
        if (lane == 0) rowsum[r] = part;
    }
    __syncthreads();

    double tot = 0.0;
    for (int r = 0; r < R; ++r) tot += rowsum[r];
    const float s = (float)tot;                                   // :197
    const int D = R * B;
    if (s > d.eps) {
        const float den = s + d.eps;                              // :199
        for (int i = tid; i < D; i += NT) {
            const int r = i / B, b = i - r * B;
            const float h = *reinterpret_cast<const float *>(
                reinterpret_cast<const unsigned char *>(rows + r * A) + HIST_OFF + 4 * b);
            out_desc[i] = h / den;
        }
    } else {
        const float u = 1.0f / (float)D;                          // :202
        for (int i = tid; i < D; i += NT) out_desc[i] = u;
    }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <int NW, int U, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void encode_fused_kernel(
    const float *__restrict__ pts, const long long *__restrict__ off, int stride, EncDev d,
    const int *__restrict__ lut, float *__restrict__ out_desc, float *__restrict__ out_raw,
    float *__restrict__ out_interp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = NW * 64;
    const int c = blockIdx.x, tid = threadIdx.x;
    unsigned *img = reinterpret_cast<unsigned *>(lds);
    const int npix = d.E * A;
    for (int i = tid; i < npix; i += NT) img[i] = NSC_EMPTY_BITS;  // :205 full(inf)
    __syncthreads();
    if (tid >= NT - 64) setup_tables<NW>(lds, d, lut, tid - (NT - 64), 64);   // last wave; joins the stream late
    scatter_range<NT, U>(pts, off[c], off[c + 1], stride, tid, d.bp, img);
    __syncthreads();
    const long long D = (long long)d.R * d.B;
    finish_image<NW>(lds, d, 0, out_desc + c * D,
                     out_raw ? out_raw + (long long)c * npix : nullptr,
                     out_interp ? out_interp + (long long)c * npix : nullptr);
}

// ---------------------------------------------------------------------------------------------
// encode_fast_kernel -- the configuration every caller of the reference uses (16 x 360 image, no
// pooling, (N,4) points, default FOV and range window), one 4-wave workgroup per cloud.
//   stream   each lane keeps U 16-byte non-temporal loads in flight as a ROLLING window (a slot is refilled the
//            moment its point has been copied out), bins with nsc_point_lean (~50 VALU instructions per point)
//            and updates the LDS image with ds_min_u32.  A point whose estimate sits within the margin of a bin
//            edge (~2e-4 of them) is not resolved in the loop -- that would drag all 64 lanes through two
//            float64 atan2 -- but parked in an LDS queue;
//   drain    the queue is resolved with the exact chain, compacted (a few dozen points per cloud);
//   finish   as finish_image(), with the rows owned in contiguous blocks of four per wave so that the FFT
//            scratch, the magnitudes and the histograms live in the wave's own (by then consumed) image rows.
// LDS: image 23 040 + queue/twiddles 3 840 + segments + misc = 27.9 KB (the generic kernel: 39.4 KB) -- a resident
// 1 024-cloud grid leaves 48 KB of every CU to whatever runs beside it.
// ---------------------------------------------------------------------------------------------
constexpr int FQ_CAP = TW_N;               // 240 queue entries of 16 B in the bytes the twiddle table occupies later
constexpr int FAST_HSTRIDE = 64;           // floats between the histograms of a wave's four rows (n_bins <= 64)
constexpr long long FAST_MAX_POINTS = 1LL << 27;   // per cloud: byte offsets stay below 2^31 (tested in the kernel)

struct FastLds {
    int img, aux, seg, misc, total;
};

__host__ __device__ inline FastLds fast_lds(int B)
{
    FastLds p;
    int o = 0;
    p.img = o;  o += 16 * ROW_BYTES;
    p.aux = o;  o += TW_N * 16;                         // stream: uncertain-point queue; finish: twiddles
    p.seg = o;  o += ((2 * B * 4) + 15) & ~15;
    p.misc = o; o += MAXR * 8 + 16 * 4 + 16 * 4 + 16;   // rowsum f64[16], rowflag int[16], rowsrc int[16], qcount
    p.total = o;
    return p;
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// The stream.  P = first point of the cloud, n > 0 points.  Each lane owns the points tid, tid + NT, ... and keeps
// U 16-byte non-temporal loads in flight as a ROLLING window: slot u of round r holds point tid + (r U + u) NT; at
// its turn the lane waits for exactly that load (s_waitcnt vmcnt(U - 1): loads return in order), bins the point and
// refills the slot with the point U NT further on -- U - 1 loads are in flight while a point is binned, across
// rounds too.
//
// The loads and their waits are inline asm: hipcc's own s_waitcnt insertion puts vmcnt(0) at the head of such a loop
// (every round would start with the full latency of the load issued just before the back edge) and hoists the
// squares of all U points to the top of a round.  Rules kept here (cdna_hip_programming.md 5.7): a slot is one
// "+v" operand of BOTH statements, so the compiler holds it in one register quad and never reads it between the load
// and the wait; every statement is volatile with a memory clobber (program order is kept); the last round issues no
// loads and counts its waits down to vmcnt(0), so nothing is in flight when the registers are reused.
template <int NT, int U>
__device__ __forceinline__ void stream_fast(const f32x4 *__restrict__ P, int n, int tid, const NscBinParams &bp,
                                            unsigned *img, f32x4 *queue, unsigned *qcount, int dev_mode = 0)
{
#ifdef NSC_DEV_TUNING
    float dev_acc = 0.f;
#endif
    auto park = [&](const f32x4 &v, float s) {                       // ~2e-4 of the points
        const unsigned slot = atomicAdd(qcount, 1u);                 // counts past FQ_CAP: the kernel then re-streams
        if (slot < (unsigned)FQ_CAP) queue[slot] = f32x4{v.x, v.y, v.z, s};
    };
    // two slots at a time: the lean estimate of both points in packed float32 instructions (nsc_point_lean_pair)
    auto process2 = [&](const f32x4 &va, const f32x4 &vb) {
#ifdef NSC_DEV_TUNING
        if (dev_mode & 64) { dev_acc += (va.x + va.y + va.z) + (vb.x + vb.y + vb.z); return; }   // loads only
#endif
        const NscLeanPair p = nsc_point_lean_pair(va.x, va.y, va.z, vb.x, vb.y, vb.z, bp);
#ifdef NSC_DEV_TUNING
        if (dev_mode & 128) {                                                    // loads + binning, no LDS atomics
            dev_acc += (float)(p.pix[0] + (int)p.ok[0] + p.pix[1] + (int)p.ok[1]) + p.s[0] + p.s[1] + (float)(p.park[0] | p.park[1]);
            return;
        }
#endif
        if (p.ok[0]) atomicMin(&img[p.pix[0]], __float_as_uint(p.s[0]));   // ds_min_u32 (s >= 0: uint order == float order)
        if (p.ok[1]) atomicMin(&img[p.pix[1]], __float_as_uint(p.s[1]));
        if (p.park[0] | p.park[1]) {
            if (p.park[0]) park(va, p.s[0]);
            if (p.park[1]) park(vb, p.s[1]);
        }
    };
    // wave-uniform base address in SGPRs + a 32-bit byte offset per lane (clouds hold < 2^27 points)
    const unsigned long long pa = reinterpret_cast<unsigned long long>(P);
    unsigned long long Pb = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa);
    // callers guarantee 0 < n < FAST_MAX_POINTS (encode_fast_kernel tests both): n = 0 would make `last` wrap and
    // every clamp below a no-op, n >= 2^27 would wrap the 32-bit byte offsets
    const unsigned last = (unsigned)(n - 1) * 16u;                  // loads past the cloud re-read its last point
    const int T = (n + NT * U - 1) / (NT * U);                      // rounds, the last one possibly partial
    f32x4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the operands are local references: clang does not capture a variable that only an asm operand names)
#define NSC_SLOT_LOAD(slot, ofs)                                                                                  \
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(slot) : "v"(ofs), "s"(base_) : "memory")
#define NSC_SLOT_WAIT2(s0, s1, cnt) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s0), "+v"(s1) : "i"(cnt) : "memory")

    // v_readfirstlane writes the SGPR pair; a VMEM instruction that reads it as its scalar base needs 5 wait states
    // after a VALU write, and the hazard recogniser does not look inside inline asm.  The s_nop takes the base as an
    // in/out operand: it cannot be scheduled before the SGPRs are written, and every load below reads ITS result, so
    // none can be scheduled before it.  (A dangling `asm volatile("s_nop 4")` orders nothing against the
    // readfirstlane -- a development variant with a per-wave base faulted on address 0 in round 2, the stale-SGPR
    // signature; in the shipped kernel the compiler proves the base uniform and computes it on the SALU.)
    asm volatile("s_nop 4" : "+s"(Pb) : : "memory");
    unsigned o = (unsigned)tid * 16u;                               // byte offset of slot 0 of the current round
    static_for<U>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        f32x4 &slot = buf[u];
        const unsigned long long base_ = Pb;
        const unsigned a = min(o + (unsigned)(u * NT * 16), last);
        NSC_SLOT_LOAD(slot, a);
    });
    constexpr unsigned RB = (unsigned)(U * NT * 16);                 // bytes per round
    static_assert(U % 2 == 0, "slots are binned in pairs");
    for (int r = 0; r + 2 < T; ++r) {            // rounds whose refill (round r + 1) is complete: no clamping
        static_for<U / 2>([&](auto uc) {
            constexpr int u = 2 * decltype(uc)::value;
            f32x4 &slot = buf[u], &slot1 = buf[u + 1];
            const unsigned long long base_ = Pb;
            NSC_SLOT_WAIT2(slot, slot1, U - 2);
            process2(slot, slot1);
            const unsigned a = o + RB + (unsigned)(u * NT * 16), a1 = a + (unsigned)(NT * 16);
            NSC_SLOT_LOAD(slot, a);
            NSC_SLOT_LOAD(slot1, a1);
        });
        o += RB;
        // Keep the four waves of a cloud in step: one s_barrier per round (it waits for no memory counter, the loads
        // stay in flight across it).  Waves that drift apart turn the workgroup's 32 KB-per-round sequential sweep into
        // four unrelated streams; in step, the same kernel measured 2.2-2.7 % faster (interleaved A/B, round 2) and the
        // four waves also reach the finish together.  Every wave runs the same number of rounds (T depends on n only).
        __builtin_amdgcn_s_barrier();
    }
    if (T >= 2) {                                // round T - 2: its refill is the last, possibly partial, round
        static_for<U / 2>([&](auto uc) {
            constexpr int u = 2 * decltype(uc)::value;
            f32x4 &slot = buf[u], &slot1 = buf[u + 1];
            const unsigned long long base_ = Pb;
            NSC_SLOT_WAIT2(slot, slot1, U - 2);
            process2(slot, slot1);
            const unsigned a = min(o + RB + (unsigned)(u * NT * 16), last);
            const unsigned a1 = min(o + RB + (unsigned)((u + 1) * NT * 16), last);
            NSC_SLOT_LOAD(slot, a);
            NSC_SLOT_LOAD(slot1, a1);
        });
        o += RB;
    }
    static_for<U / 2>([&](auto uc) {             // last round: no refills, waits count down to vmcnt(0)
        constexpr int u = 2 * decltype(uc)::value;
        f32x4 &slot = buf[u], &slot1 = buf[u + 1];
        NSC_SLOT_WAIT2(slot, slot1, U - 2 - u);
        f32x4 v = slot, v1 = slot1;
        if (o + (unsigned)(u * NT * 16) > last) v.x = NAN;          // past the cloud: fails the range window
        if (o + (unsigned)((u + 1) * NT * 16) > last) v1.x = NAN;
        process2(v, v1);
    });
#undef NSC_SLOT_WAIT2
#undef NSC_SLOT_LOAD
#ifdef NSC_DEV_TUNING
    if (dev_mode & (64 | 128)) atomicMin(&img[tid & 63], __float_as_uint(fabsf(dev_acc)));
#endif
}

// Cold path of encode_fast_kernel: the whole cloud through the lean estimate, one point per thread and iteration,
// uncertain points resolved in place with the exact chain.  Runs for (a) a cloud that parked more uncertain points than
// the queue holds (adversarial clouds on bin edges) -- min is idempotent, so the pixels the stream already wrote stay
// right -- and (b) a cloud of >= FAST_MAX_POINTS points, whose byte offsets do not fit the 32 bits stream_fast uses
// (64-bit indexing here).  Deliberately not unrolled: four exact chains in flight (scatter_range<NT, 4>, round 2) were
// the kernel's register peak, 92 VGPRs.
__device__ __forceinline__ void restream_cold(const f32x4 *__restrict__ P, long long n, int tid, int nt,
                                              const NscBinParams &bp, unsigned *img)
{
    for (long long i = tid; i < n; i += nt) {
        const f32x4 v = __builtin_nontemporal_load(&P[i]);
        int pix; float s; bool certain;
        if (!nsc_point_lean_flags(v.x, v.y, v.z, bp, pix, s, certain)) continue;
        if (!certain) pix = nsc_point_exact(v.x, v.y, v.z, bp);
        atomicMin(&img[pix], __float_as_uint(s));
    }
}

// The tables the finish needs from global memory, one element per thread; their latency lands under the square roots
// (loading them before the stream and carrying them through it measured the same).
struct FinishTables {
    double2 twv;                  // g_tw360[tid]
    int lut_cur, lut_prev;        // lut[tid], lut[tid - 1]
};

__device__ __forceinline__ FinishTables load_finish_tables(const int *__restrict__ lut, int tid)
{
    FinishTables t;
    t.twv = double2{0.0, 0.0};
    t.lut_cur = t.lut_prev = -1;
    if (tid < TW_N) t.twv = g_tw360[tid];
    if (tid < F) { t.lut_cur = lut[tid]; t.lut_prev = tid ? lut[tid - 1] : -1; }
    return t;
}

__device__ __forceinline__ void finish_fast(unsigned char *lds, const EncDev &d, const FinishTables &ft,
                                            float *__restrict__ out_desc, float *__restrict__ out_raw,
                                            float *__restrict__ out_interp)
{
    constexpr int E = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = d.B;
    const FastLds lp = fast_lds(B);
    float *img = reinterpret_cast<float *>(lds + lp.img);
    double2 *tw = reinterpret_cast<double2 *>(lds + lp.aux);
    int *seg = reinterpret_cast<int *>(lds + lp.seg);
    double *rowsum = reinterpret_cast<double *>(lds + lp.misc);
    int *rowflag = reinterpret_cast<int *>(lds + lp.misc + MAXR * 8);
    int *rowsrc = rowflag + 16;

    // the twiddle table replaces the (drained) queue.  Histogram segments (see setup_tables): bin b owns the
    // frequencies [seg[b], seg[B + b]); no dependent search: thread k sees where the monotone LUT steps at frequency
    // k and writes the bins that start there.
    const double2 twv = ft.twv;
    int lut_cur = ft.lut_cur;
    const int lut_prev = ft.lut_prev;

#ifdef NSC_DEV_TUNING
    unsigned long long st[7];
#define NSC_STAMP(i) st[i] = wall_clock64()
#else
#define NSC_STAMP(i)
#endif
    NSC_STAMP(0);
    const int r0 = 4 * wave;                                      // this wave owns rows r0 .. r0 + 3
    {
        float v[4][6];                                            // the four rows' square roots are independent
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *row = img + (r0 + q) * A;
            const unsigned *raw = reinterpret_cast<const unsigned *>(row);
            float *graw = out_raw ? out_raw + (r0 + q) * A : nullptr;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int c = lane + 64 * j;
                v[q][j] = 0.0f;
                if (c < A) {
                    const unsigned b = raw[c];
                    const float rr = (b == NSC_EMPTY_BITS) ? 0.0f : sqrtf(__uint_as_float(b));   // :162,:214
                    v[q][j] = rr;
                    row[c] = rr;
                    if (graw) graw[c] = rr;
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nv = interp_row_v(img + (r0 + q) * A, lane, d.interp, v[q]);
            if (lane == 0) rowflag[r0 + q] = (nv > 0);
        }
    }
    NSC_STAMP(1);
    if (tid < TW_N) tw[tid] = twv;
    if (tid < F) {
        lut_cur = min(lut_cur, B - 1);
        for (int b = lut_prev + 1; b <= lut_cur; ++b) { seg[b] = tid; if (b > 0) seg[B + b - 1] = tid; }
        if (tid == F - 1)
            for (int b = lut_cur; b < B; ++b) { seg[B + b] = F; if (b > lut_cur) seg[b] = F; }
    }
    __syncthreads();
    NSC_STAMP(2);
    if (d.interp) {                                               // range_image.py:77-87
        unsigned ne = 0u;
        for (int r = 0; r < E; ++r) ne |= (unsigned)(rowflag[r] != 0) << r;
        if (ne != 0xffffu && ne != 0u) {                          // some (not all) rows are empty: rare
            if (tid == 0) {
                unsigned m = ne;
                for (int r = 0; r < E; ++r) rowsrc[r] = r;
                for (int r = 0; r < E; ++r) {
                    if ((m >> r) & 1u) continue;
                    for (int k = 1; k < E; ++k) {
                        if (r - k >= 0 && ((m >> (r - k)) & 1u)) { rowsrc[r] = rowsrc[r - k]; m |= 1u << r; break; }
                        if (r + k < E && ((m >> (r + k)) & 1u)) { rowsrc[r] = r + k; m |= 1u << r; break; }
                    }
                }
            }
            __syncthreads();
            for (int r = r0; r < r0 + 4; ++r) {
                const int sr = rowsrc[r];                         // always an original (never copied) row
                if (sr != r)
                    for (int c = lane; c < A; c += 64) img[r * A + c] = img[sr * A + c];
            }
            __syncthreads();
        }
    }
    if (out_interp)
        for (int r = r0; r < r0 + 4; ++r)
            for (int c = lane; c < A; c += 64) out_interp[r * A + c] = img[r * A + c];
    if (NSC_DEV_SKIP(d, 16)) return;

    // spectrum + histogram.  All four rows' stage-1 operands go to registers first; from then on the wave's 5 760
    // image bytes are two FFT scratch buffers (180 double2 each) and the rows run as two PAIRS, the two FFTs of a pair
    // interleaved instruction by instruction.  The magnitudes of a row land on its own scratch once the unpack has
    // read it, the histogram value of bin `lane` of each of the four rows stays in a register until the normalisation.
    NSC_STAMP(3);
    f32x2 in[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fft_load_row(img + (r0 + q) * A, lane, in[q]);
    wave_sync();
    double2 *const bufs[2] = {reinterpret_cast<double2 *>(img + r0 * A), reinterpret_cast<double2 *>(img + (r0 + 2) * A)};
    float h[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    double part = 0.0;
    const int k0 = lane < B ? seg[lane] : 0, k1 = lane < B ? seg[B + lane] : 0;     // B <= 64: one bin per lane
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        float mg[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (!NSC_DEV_SKIP(d, 4)) {            // pair 0 carries pair 1's stage-1 operands: the lean form (see fft_rows)
            if (pr == 0) fft_rows<2, true>(in, bufs, tw, mg, lane);
            else fft_rows<2, false>(in + 2, bufs, tw, mg, lane);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float *mags = reinterpret_cast<float *>(bufs[p]);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int k = lane + 64 * jj;
                if (k <= NH) mags[k] = mg[p][jj];
            }
        }
        wave_sync();
        if (!NSC_DEV_SKIP(d, 8)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float *mags = reinterpret_cast<const float *>(bufs[p]);
                float acc = 0.0f;
                for (int k = k0; k < k1; k += 8) {               // scatter_add_, ascending k (:152-155)
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (k + u < k1) ? mags[k + u] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u];      // acc + 0.0f == acc: order and rounding unchanged
                }
                h[2 * pr + p] = acc;
                part += (double)acc;
            }
        }
        wave_sync();                                              // the magnitudes are overwritten by the next pair
    }
    NSC_STAMP(4);
    part = wave_sum(part);
    if (lane == 0) rowsum[wave] = part;
    __syncthreads();
    NSC_STAMP(5);

    const double tot = (rowsum[0] + rowsum[1]) + (rowsum[2] + rowsum[3]);
    const float s = (float)tot;                                   // :197
    if (lane < B) {
        if (s > d.eps) {
            const float den = s + d.eps;                          // :199
#pragma unroll
            for (int q = 0; q < 4; ++q) out_desc[(r0 + q) * B + lane] = h[q] / den;
        } else {
            const float u = 1.0f / (float)(E * B);                // :202
#pragma unroll
            for (int q = 0; q < 4; ++q) out_desc[(r0 + q) * B + lane] = u;
        }
    }
#ifdef NSC_DEV_TUNING
    if (NSC_DEV_SKIP(d, 512)) {
        NSC_STAMP(6);
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < 7; ++i) out_desc[3 + i] = __uint_as_float((unsigned)st[i]);
    }
#endif
#undef NSC_STAMP
}

template <int U>
__global__ __launch_bounds__(256, 6) void encode_fast_kernel(
    const float *__restrict__ pts, const long long *__restrict__ off, EncDev d, const int *__restrict__ lut,
    float *__restrict__ out_desc, float *__restrict__ out_raw, float *__restrict__ out_interp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = 256;
    const int c = blockIdx.x, tid = threadIdx.x;
    const FastLds lp = fast_lds(d.B);
    unsigned *img = reinterpret_cast<unsigned *>(lds + lp.img);
    f32x4 *queue = reinterpret_cast<f32x4 *>(lds + lp.aux);
    unsigned *qcount = reinterpret_cast<unsigned *>(lds + lp.misc + MAXR * 8 + 32 * 4);
    {
        const uint4 e4 = {NSC_EMPTY_BITS, NSC_EMPTY_BITS, NSC_EMPTY_BITS, NSC_EMPTY_BITS};   // :205 full(inf)
        uint4 *i4 = reinterpret_cast<uint4 *>(img);
        for (int i = tid; i < 16 * A / 4; i += NT) i4[i] = e4;
        if (tid == 0) *qcount = 0u;
    }
    __syncthreads();
    const long long p0 = off[c];
    const long long n64 = off[c + 1] - p0;
    const f32x4 *P = reinterpret_cast<const f32x4 *>(pts) + p0;
    // the size limit of the streaming loop is PER CLOUD (32-bit byte offsets relative to the cloud's own base); a
    // cloud beyond it takes the cold loop, the rest of the batch is unaffected
    const bool streamable = n64 > 0 && n64 < FAST_MAX_POINTS;
    const int n = streamable ? (int)n64 : 0;
#ifdef NSC_DEV_TUNING
    const unsigned long long dev_t0 = wall_clock64();
    unsigned long long dev_t1 = 0;
#endif
    if (n > 0 && !NSC_DEV_SKIP(d, 2)) stream_fast<NT, U>(P, n, tid, d.bp, img, queue, qcount, NSC_DEV_MODE(d));
    __syncthreads();
    {   // drain the uncertain-point queue with the exact chain (the definition of the pixel), compacted
        const unsigned qn = *qcount;
        if (qn <= (unsigned)FQ_CAP && (streamable || n64 <= 0)) {
            for (unsigned i = tid; i < qn; i += NT) {
                const f32x4 e = queue[i];
                atomicMin(&img[nsc_point_exact(e.x, e.y, e.z, d.bp)], __float_as_uint(e.w));
            }
        } else {
            // queue overflow (some uncertain points were not recorded) or a cloud too large for the streaming loop
            restream_cold(P, n64, tid, NT, d.bp, img);
        }
    }
    __syncthreads();
#ifdef NSC_DEV_TUNING
    dev_t1 = wall_clock64();
#endif
    if (NSC_DEV_SKIP(d, 1)) {
        if (tid == 0) out_desc[(long long)c * 16 * d.B] = __uint_as_float(img[0]);
        return;
    }
    const long long D = 16LL * d.B;
#ifdef NSC_DEV_TUNING
    if (NSC_DEV_SKIP(d, 512)) {         // per-workgroup timeline probe (tools/wg_timeline.py): 100 MHz wall clock
        finish_fast(lds, d, load_finish_tables(lut, tid), out_desc + c * D, nullptr, nullptr);
        __syncthreads();
        if (tid == 0) {
            out_desc[c * D + 0] = __uint_as_float((unsigned)dev_t0);
            out_desc[c * D + 1] = __uint_as_float((unsigned)dev_t1);
            out_desc[c * D + 2] = __uint_as_float((unsigned)wall_clock64());
        }
        return;
    }
#endif
    finish_fast(lds, d, load_finish_tables(lut, tid), out_desc + c * D, out_raw ? out_raw + (long long)c * 16 * A : nullptr,
                out_interp ? out_interp + (long long)c * 16 * A : nullptr);
}

template <int NW, int U>
__global__ __launch_bounds__(NW * 64) void scatter_split_kernel(
    const float *__restrict__ pts, const long long *__restrict__ off, int stride, int parts,
    EncDev d, unsigned *__restrict__ ws)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = NW * 64;
    const int c = blockIdx.x / parts, part = blockIdx.x - c * parts, tid = threadIdx.x;
    unsigned *img = reinterpret_cast<unsigned *>(lds);
    const int npix = d.E * A;
    for (int i = tid; i < npix; i += NT) img[i] = NSC_EMPTY_BITS;
    __syncthreads();
    const long long p0 = off[c], n = off[c + 1] - p0;
    const long long chunk = (n + parts - 1) / parts;
    const long long a = p0 + (long long)part * chunk;
    long long b = a + chunk;
    if (b > p0 + n) b = p0 + n;
    if (a < b) scatter_range<NT, U>(pts, a, b, stride, tid, d.bp, img);
    __syncthreads();
    unsigned *g = ws + (long long)c * npix;
    if (parts == 1) {                       // sole owner of the cloud: plain 16-byte stores, no pre-fill
        const uint4 *s4 = reinterpret_cast<const uint4 *>(img);
        uint4 *g4 = reinterpret_cast<uint4 *>(g);
        for (int i = tid; i < npix / 4; i += NT) g4[i] = s4[i];
        return;
    }
    for (int i = tid; i < npix; i += NT) {
        const unsigned v = img[i];
        if (v != NSC_EMPTY_BITS) atomicMin(&g[i], v);
    }
}

// src_u32 != null: squared-range workspace images (mode 0); else src_f32 caller images
// (mode 1: descriptor only; interp_only: mode 2)
template <int NW>
__global__ __launch_bounds__(NW * 64) void finish_kernel(
    const unsigned *__restrict__ src_u32, const float *__restrict__ src_f32, int interp_only, EncDev d,
    const int *__restrict__ lut, float *__restrict__ out_desc, float *__restrict__ out_raw,
    float *__restrict__ out_interp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = NW * 64;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int npix = d.E * A;
    setup_tables<NW>(lds, d, lut, tid, NT);
    {
        // 16-byte loads, all issued before the LDS stores (rows are 1 440 B, images 16-B aligned)
        const f32x4 *g = src_u32 ? reinterpret_cast<const f32x4 *>(src_u32 + (long long)c * npix)
                                 : reinterpret_cast<const f32x4 *>(src_f32 + (long long)c * npix);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        const int nvec = npix / 4;
        for (int base = 0; base < nvec; base += NT * 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * NT + tid;
                if (i < nvec) v[u] = g[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * NT + tid;
                if (i < nvec) dst[i] = v[u];
            }
        }
    }
    __syncthreads();
    const long long D = (long long)d.R * d.B;
    finish_image<NW>(lds, d, src_u32 ? 0 : (interp_only ? 2 : 1), out_desc ? out_desc + c * D : nullptr,
                     out_raw ? out_raw + (long long)c * npix : nullptr,
                     out_interp ? out_interp + (long long)c * npix : nullptr);
}

// Intensity image of RangeImageProjector.project(keep_intensity=True), range_image.py:216-228: per pixel the
// MAXIMUM intensity over the points whose range equals the pixel's minimum range, starting from 0 (np.maximum.at
// into a zero image).  `range` is the raw range image of the same cloud (sqrtf of the min squared range; sqrtf
// is correctly rounded, so r == range[pix] is the reference's float32 comparison).  Positive floats order like
// their bit patterns, so the maximum is an integer atomicMax; intensities <= 0 never raise the 0 the image
// starts from; a NaN intensity among a pixel's closest points makes the pixel NaN, as np.maximum.at does (:225) -- the
// quiet NaN 0x7fc00000 orders above every positive float and +inf as an integer (payloads are not preserved).
__global__ __launch_bounds__(256) void intensity_kernel(const float *__restrict__ pts, const long long *__restrict__ off,
                                                        int parts, NscBinParams bp, int npix,
                                                        const float *__restrict__ range, int *__restrict__ out)
{
    const int c = blockIdx.x / parts, part = blockIdx.x - c * parts;
    const long long p0 = off[c], n = off[c + 1] - p0;
    const long long chunk = (n + parts - 1) / parts;
    const long long a = p0 + (long long)part * chunk;
    long long b = a + chunk;
    if (b > p0 + n) b = p0 + n;
    const f32x4 *P = reinterpret_cast<const f32x4 *>(pts);
    const float *rng = range + (long long)c * npix;
    int *o = out + (long long)c * npix;
    for (long long i = a + threadIdx.x; i < b; i += 256) {
        const f32x4 v = P[i];
        int pix; float s;
        if (!nsc_point_pixel(v.x, v.y, v.z, bp, pix, s)) continue;
        if (sqrtf(s) != rng[pix]) continue;
        if (v.w > 0.0f) atomicMax(&o[pix], __float_as_int(v.w));
        else if (v.w != v.w) atomicMax(&o[pix], 0x7fc00000);     // np.maximum propagates NaN: above every number, +inf included
    }
}

__global__ __launch_bounds__(256) void point_bins_kernel(
    const float *__restrict__ pts, long long n, int stride, NscBinParams bp, int lean,
    int *__restrict__ out_idx, unsigned char *__restrict__ out_flags)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float x = pts[i * stride], y = pts[i * stride + 1], z = pts[i * stride + 2];
        int pix; float s;
        if (lean) {            // what encode_fast_kernel does with a point: lean estimate (the packed two-point form,
                               // here with the point in both halves swapped by parity), exact chain when uncertain
            const float xn = pts[(i ^ 1) < n ? (i ^ 1) * stride : i * stride], yn = pts[((i ^ 1) < n ? (i ^ 1) : i) * stride + 1],
                        zn = pts[((i ^ 1) < n ? (i ^ 1) : i) * stride + 2];
            const int h = (int)(i & 1);
            const NscLeanPair p = h ? nsc_point_lean_pair(xn, yn, zn, x, y, z, bp) : nsc_point_lean_pair(x, y, z, xn, yn, zn, bp);
            const int st = p.ok[h] ? 1 : p.park[h] ? 2 : 0;
            pix = p.pix[h];
            if (st == 2) pix = nsc_point_exact(x, y, z, bp);
            out_idx[i] = st ? pix : -1;
            if (out_flags) out_flags[i] = (unsigned char)(st == 2 ? 3 : 0);
            continue;
        }
        const int fl = nsc_point_pixel(x, y, z, bp, pix, s);
        out_idx[i] = fl ? pix : -1;
        if (out_flags) out_flags[i] = (unsigned char)(fl >> 1);
    }
}

// ---------------------------------------------------------------------------------------------
// host side of the ABI
// ---------------------------------------------------------------------------------------------
constexpr int FUSED_NW = 8;          // split / finish kernels: 512 threads per workgroup
constexpr int FUSED_U = 4;           // float4 loads in flight per thread (split scatter)
constexpr int SPLIT_MIN_PTS = 16384; // a part must amortise its 5 760-pixel LDS init + flush
constexpr int SPLIT_TARGET_WGS = 512;

// Development knobs (tools/ab_enc.py, tools/sweep_enc.py) exist only in builds made with
// NSC_DEV_BUILD=1 (-DNSC_DEV_TUNING); the shipped library reads no environment variables.
int tune_env(const char *name, int def)
{
#ifdef NSC_DEV_TUNING
    const char *v = getenv(name);
    return v ? atoi(v) : def;
#else
    (void)name;
    return def;
#endif
}

int check_params(const NscEncParams *p)
{
    if (!p) return NSC_EINVAL;
    if (p->n_azimuth != A) return NSC_EUNSUPPORTED;
    if (p->n_elevation < 1 || p->n_elevation > MAXE) return NSC_EUNSUPPORTED;
    if (p->target_rows < 1 || p->target_rows > MAXR) return NSC_EUNSUPPORTED;
    if (p->n_bins < 1 || p->n_bins > MAXB) return NSC_EUNSUPPORTED;
    if (!(p->elev_max_rad > p->elev_min_rad)) return NSC_EINVAL;
    return NSC_OK;
}

EncDev make_dev(const NscEncParams *p, int rows_in)
{
    EncDev d;
    d.bp = nsc_make_bin_params(rows_in, p->elev_min_rad, p->elev_max_rad, p->min_range, p->max_range,
                               p->elev_f64);
    d.E = rows_in;
    d.R = p->target_rows;
    d.B = p->n_bins;
    d.eps = p->epsilon;
    d.interp = p->interpolate;
#ifdef NSC_DEV_TUNING
    d.dev_skip = tune_env("NSC_TUNE_SKIP_FINISH", 0);
#endif
    return d;
}

int split_parts(int32_t n_clouds, int64_t total_points);

// Which kernel set nsc_encode_clouds launches for a batch: the one decision function the launcher and
// nsc_encode_clouds_path() share.
int encode_path(const EncDev &d, int32_t n_clouds, int64_t total_points, int32_t stride, int variant)
{
    if (split_parts(n_clouds, total_points) > 1) return NSC_ENC_PATH_SPLIT;
    // the configuration of every reference caller: the lean streaming kernel.  No limit on the batch: the kernel's
    // 32-bit byte offsets are relative to each cloud's own base, and a single cloud of >= 2^27 points takes its cold
    // loop (round 2 tested total_points here and sent batches of more than 1 118 x 120 000 points to the generic kernel).
    if (variant <= 0 && stride == 4 && d.E == 16 && d.R == 16 && d.B <= FAST_HSTRIDE && nsc_lean_ok(d.bp))
        return NSC_ENC_PATH_FAST;
    return NSC_ENC_PATH_FUSED;
}

int split_parts(int32_t n_clouds, int64_t total_points)
{
    const int force = tune_env("NSC_TUNE_SPLIT", 0);
    if (force > 0) return force;
    if (n_clouds <= 0 || n_clouds >= SPLIT_TARGET_WGS) return 1;
    const int64_t avg = total_points / n_clouds;
    int64_t by_size = avg / SPLIT_MIN_PTS;
    int64_t want = (SPLIT_TARGET_WGS + n_clouds - 1) / n_clouds;
    int64_t s = want < by_size ? want : by_size;
    return s < 2 ? 1 : (int)s;
}

// Dynamic LDS above 64 KiB (E = 64 images) must be opted into once per kernel; the call is
// idempotent and not a stream operation, so it is legal under graph capture as well.
template <class K> int set_lds(K kernel, int bytes)
{
    if (bytes <= 64 * 1024) return NSC_OK;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
               ? NSC_OK : NSC_ELAUNCH;
}

}  // namespace

extern "C" {

int nsc_abi_version(void) { return NSC_ABI_VERSION; }

const char *nsc_status_string(int s)
{
    switch (s) {
    case NSC_OK: return "ok";
    case NSC_EINVAL: return "invalid argument";
    case NSC_EUNSUPPORTED: return "unsupported shape or parameter for this entry point (limits: include/nsc.h; encoder: n_azimuth 360, rows <= 64, target_rows <= 16, n_bins <= 176; W1 retrieval: width <= 1024; wire format: width <= 4096)";
    case NSC_EWORKSPACE: return "workspace too small";
    case NSC_ELAUNCH: return "kernel launch failed";
    default: return "unknown status";
    }
}

void nsc_enc_default_params(NscEncParams *p)
{
    // configs/training_multi_dataset.yaml:40-53 + class defaults never overridden by callers
    p->n_elevation = 16;
    p->n_azimuth = 360;
    p->n_bins = 50;
    p->target_rows = 16;
    p->elev_min_rad = -24.8 * (M_PI / 180.0);
    p->elev_max_rad = 2.0 * (M_PI / 180.0);
    p->min_range = 1.0f;
    p->max_range = 80.0f;
    p->epsilon = 1e-8f;
    p->interpolate = 1;
    p->elev_f64 = 1;
}

size_t nsc_encode_clouds_workspace_bytes(int32_t n_clouds, int64_t total_points, const NscEncParams *p)
{
    if (check_params(p) != NSC_OK || n_clouds <= 0) return 0;
    if (split_parts(n_clouds, total_points) <= 1) return 0;
    return (size_t)n_clouds * p->n_elevation * A * sizeof(unsigned);
}

int nsc_encode_clouds_path(int32_t n_clouds, int64_t total_points, int32_t stride, const NscEncParams *p)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_clouds < 0 || total_points < 0 || (stride != 3 && stride != 4)) return NSC_EINVAL;
    return encode_path(make_dev(p, p->n_elevation), n_clouds, total_points, stride, tune_env("NSC_TUNE_VARIANT", 0));
}

int nsc_encode_clouds(const float *pts, const int64_t *cloud_offsets, int32_t n_clouds,
                      int64_t total_points, int32_t stride, const NscEncParams *p, const int32_t *lut,
                      float *out_desc, float *out_raw, float *out_interp, void *ws, size_t ws_bytes,
                      void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_clouds < 0 || total_points < 0 || (stride != 3 && stride != 4)) return NSC_EINVAL;
    if (n_clouds == 0) return NSC_OK;
    if (!cloud_offsets || !lut || !out_desc || (!pts && total_points > 0)) return NSC_EINVAL;
    if (stride == 4 && (reinterpret_cast<uintptr_t>(pts) & 15u)) return NSC_EINVAL;   // 16-B loads
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(p, p->n_elevation);
    const LdsPlan lp = lds_plan(d.E, d.R, d.B, FUSED_NW);
    const int parts = split_parts(n_clouds, total_points);
    const long long *off = reinterpret_cast<const long long *>(cloud_offsets);
    const int variant = tune_env("NSC_TUNE_VARIANT", 0);
    const int path = encode_path(d, n_clouds, total_points, stride, variant);

    if (path != NSC_ENC_PATH_SPLIT) {
        if (path == NSC_ENC_PATH_FAST) {
            const FastLds fl = fast_lds(d.B);
                // two loads per lane: measured against 4, 6, 8 and 12, alone 1-2 % faster on uniform and 3-4 % on
                // ring-ordered clouds, identical in the two-stream step (round 2, interleaved A/B)
                hipLaunchKernelGGL(encode_fast_kernel<2>, dim3(n_clouds), dim3(256), fl.total, stream, pts, off, d, lut,
                                   out_desc, out_raw, out_interp);
            return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
        }
#define NSC_LAUNCH_FUSED(NW_, U_, MINW_)                                                              \
    {                                                                                                 \
        auto k = encode_fused_kernel<NW_, U_, MINW_>;                                                 \
        const LdsPlan lpv = lds_plan(d.E, d.R, d.B, NW_);                                             \
        if ((st = set_lds(k, lpv.total)) != NSC_OK) return st;                                        \
        hipLaunchKernelGGL(k, dim3(n_clouds), dim3(NW_ * 64), lpv.total, stream, pts, off, stride, d, \
                           lut, out_desc, out_raw, out_interp);                                       \
    }
        // 16 waves per CU in every shape: the LDS image decides how many workgroups share a CU, the workgroup
        // brings the waves (E <= 16: 4 x 4 waves, <= 32 rows: 2 x 8, up to 64 rows: 1 x 16)
        // (the wave / load-depth shapes measured against these in rounds 1-2 -- 8 loads per lane, 8 or 16 waves on 16 rows --
        // are DESIGN.md section 7, experiments 1-12; the library instantiates only what it launches)
        if (d.E > 32) NSC_LAUNCH_FUSED(16, 4, 1)
        else if (d.E > 16) NSC_LAUNCH_FUSED(8, 4, 2)
        // 4 waves x 4 float4 loads in flight per lane (92 VGPRs), 39.4 KB LDS -> 4 workgroups per CU: a 1 024-cloud batch is
        // exactly one resident round.  Interleaved A/B on three boxes: 1-1.5 % faster than 8 loads per lane (108 VGPRs), and
        // it leaves 128 VGPRs per SIMD lane to co-resident kernels.
        else NSC_LAUNCH_FUSED(4, 4, 4)
    } else {
        const size_t need = (size_t)n_clouds * d.E * A * sizeof(unsigned);
        if (!ws || ws_bytes < need) return NSC_EWORKSPACE;
        nsc_fill_u32(stream, ws, 0xffffffffu, (long long)(need / 4));
        auto ks = scatter_split_kernel<FUSED_NW, FUSED_U>;
        const int img_bytes = d.E * A * 4;
        if ((st = set_lds(ks, img_bytes)) != NSC_OK) return st;
        hipLaunchKernelGGL(ks, dim3(n_clouds * parts), dim3(FUSED_NW * 64), img_bytes, stream, pts, off,
                           stride, parts, d, static_cast<unsigned *>(ws));
        auto kf = finish_kernel<FUSED_NW>;
        if ((st = set_lds(kf, lp.total)) != NSC_OK) return st;
        hipLaunchKernelGGL(kf, dim3(n_clouds), dim3(FUSED_NW * 64), lp.total, stream,
                           static_cast<const unsigned *>(ws), static_cast<const float *>(nullptr), 0, d, lut,
                           out_desc, out_raw, out_interp);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_project_intensity(const float *pts, const int64_t *cloud_offsets, int32_t n_clouds, int64_t total_points,
                          const NscEncParams *p, const float *range_raw, float *out_intensity, void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_clouds < 0 || total_points < 0) return NSC_EINVAL;
    if (n_clouds == 0) return NSC_OK;
    if (!cloud_offsets || !range_raw || !out_intensity || (!pts && total_points > 0)) return NSC_EINVAL;
    if (reinterpret_cast<uintptr_t>(pts) & 15u) return NSC_EINVAL;                     // (N,4) rows, 16-byte loads
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(p, p->n_elevation);
    const int npix = d.E * A;
    nsc_fill_u32(stream, out_intensity, 0u, (long long)n_clouds * npix);
    int parts = 1;
    if (n_clouds < 1024) {                       // fill the chip when the batch is small
        const long long avg = total_points / n_clouds;
        long long want = (2048 + n_clouds - 1) / n_clouds, by_size = avg / 4096;
        parts = (int)(want < by_size ? want : by_size);
        if (parts < 1) parts = 1;
    }
    hipLaunchKernelGGL(intensity_kernel, dim3((unsigned)(n_clouds * parts)), dim3(256), 0, stream, pts,
                       reinterpret_cast<const long long *>(cloud_offsets), parts, d.bp, npix, range_raw,
                       reinterpret_cast<int *>(out_intensity));
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_scatter_clouds(const float *pts, const int64_t *cloud_offsets, int32_t n_clouds, int64_t total_points,
                       int32_t stride, const NscEncParams *p, uint32_t *out_sqr, void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_clouds < 0 || total_points < 0 || (stride != 3 && stride != 4)) return NSC_EINVAL;
    if (n_clouds == 0) return NSC_OK;
    if (!cloud_offsets || !out_sqr || (!pts && total_points > 0)) return NSC_EINVAL;
    if (stride == 4 && (reinterpret_cast<uintptr_t>(pts) & 15u)) return NSC_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(p, p->n_elevation);
    const int parts = split_parts(n_clouds, total_points);
    const long long *off = reinterpret_cast<const long long *>(cloud_offsets);
    const int img_bytes = d.E * A * 4;
    if (parts > 1) {
        nsc_fill_u32(stream, out_sqr, 0xffffffffu, (long long)n_clouds * (img_bytes / 4));
        auto ks = scatter_split_kernel<FUSED_NW, FUSED_U>;
        if ((st = set_lds(ks, img_bytes)) != NSC_OK) return st;
        hipLaunchKernelGGL(ks, dim3(n_clouds * parts), dim3(FUSED_NW * 64), img_bytes, stream, pts, off, stride, parts,
                           d, out_sqr);
    } else {
        auto ks = scatter_split_kernel<4, 8>;          // one 4-wave workgroup per cloud, 23 KB LDS
        if ((st = set_lds(ks, img_bytes)) != NSC_OK) return st;
        hipLaunchKernelGGL(ks, dim3(n_clouds), dim3(256), img_bytes, stream, pts, off, stride, 1, d, out_sqr);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_finish_images(const uint32_t *sqr, int32_t n_images, const NscEncParams *p, const int32_t *lut,
                      float *out_desc, float *out_raw, float *out_interp, void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_images < 0) return NSC_EINVAL;
    if (n_images == 0) return NSC_OK;
    if (!sqr || !lut || !out_desc) return NSC_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(p, p->n_elevation);
    const LdsPlan lp = lds_plan(d.E, d.R, d.B, FUSED_NW);
    auto kf = finish_kernel<FUSED_NW>;
    if ((st = set_lds(kf, lp.total)) != NSC_OK) return st;
    hipLaunchKernelGGL(kf, dim3(n_images), dim3(FUSED_NW * 64), lp.total, stream, sqr,
                       static_cast<const float *>(nullptr), 0, d, lut, out_desc, out_raw, out_interp);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_encode_range_images(const float *imgs, int32_t n_images, int32_t rows, const NscEncParams *p,
                            const int32_t *lut, float *out_desc, void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_images < 0) return NSC_EINVAL;
    if (rows < 1 || rows > MAXE) return NSC_EUNSUPPORTED;
    if (n_images == 0) return NSC_OK;
    if (!imgs || !lut || !out_desc) return NSC_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(p, rows);
    const LdsPlan lp = lds_plan(d.E, d.R, d.B, FUSED_NW);
    auto kf = finish_kernel<FUSED_NW>;
    if ((st = set_lds(kf, lp.total)) != NSC_OK) return st;
    hipLaunchKernelGGL(kf, dim3(n_images), dim3(FUSED_NW * 64), lp.total, stream,
                       static_cast<const unsigned *>(nullptr), imgs, 0, d, lut, out_desc,
                       static_cast<float *>(nullptr), static_cast<float *>(nullptr));
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_interpolate_range_images(const float *imgs, int32_t n_images, int32_t rows, const int32_t *lut,
                                 float *out, void *stream_)
{
    return nsc_interpolate_range_images_ex(imgs, n_images, rows, lut, NSC_INTERP_LINEAR, out, stream_);
}

int nsc_interpolate_range_images_ex(const float *imgs, int32_t n_images, int32_t rows, const int32_t *lut,
                                    int32_t method, float *out, void *stream_)
{
    if (method != NSC_INTERP_LINEAR && method != NSC_INTERP_NEAREST) return NSC_EINVAL;
    if (n_images < 0) return NSC_EINVAL;
    if (rows < 1 || rows > MAXE) return NSC_EUNSUPPORTED;
    if (n_images == 0) return NSC_OK;
    if (!imgs || !out || !lut) return NSC_EINVAL;
    NscEncParams p;
    nsc_enc_default_params(&p);
    p.n_elevation = rows;
    p.target_rows = rows < MAXR ? rows : MAXR;
    p.interpolate = method;
    int st;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const EncDev d = make_dev(&p, rows);
    const LdsPlan lp = lds_plan(d.E, d.R, d.B, FUSED_NW);
    auto kf = finish_kernel<FUSED_NW>;
    if ((st = set_lds(kf, lp.total)) != NSC_OK) return st;
    hipLaunchKernelGGL(kf, dim3(n_images), dim3(FUSED_NW * 64), lp.total, stream,
                       static_cast<const unsigned *>(nullptr), imgs, 1, d, lut, static_cast<float *>(nullptr),
                       static_cast<float *>(nullptr), out);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_debug_point_bins(const float *pts, int64_t n_points, int32_t stride, const NscEncParams *p,
                         int32_t *out_idx, uint8_t *out_flags, void *stream_)
{
    int st = check_params(p);
    if (st != NSC_OK) return st;
    if (n_points < 0 || (stride != 3 && stride != 4)) return NSC_EINVAL;
    if (n_points == 0) return NSC_OK;
    if (!pts || !out_idx) return NSC_EINVAL;
    const EncDev d = make_dev(p, p->n_elevation);
    long long blocks = (n_points + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(point_bins_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), pts, (long long)n_points, stride, d.bp,
                       (int)(stride == 4 && d.E == 16 && nsc_lean_ok(d.bp)), out_idx, out_flags);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

}  // extern "C"
