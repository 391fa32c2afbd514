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

#include "../../include/nsc.h"
#include "nsc_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int A = NSC_A;      // 360 columns
constexpr int F = NSC_F;      // 181 rfft bins
constexpr int NH = 180;       // complex length of the packed real FFT
constexpr int MAXR = 16;      // target rows the finish stage supports
constexpr int MAXE = 64;      // projector rows the LDS image supports
constexpr int MAGS_STRIDE = 184;

struct EncDev {
    NscBinParams bp;
    int E, R, B;
    float eps;
    int interp;
};

// exp(-2 pi i j / 360) = (cos, -sin): table holds (cos, sin)
__device__ const double2 g_tw360[A] = {
#include "nsc_twiddle360.inc"
};

// ---------------------------------------------------------------------------------------------
// LDS carve-up (bytes), identical on host and device
// ---------------------------------------------------------------------------------------------
struct LdsPlan {
    int img, pool, tw, fft, mags, hist, seg, misc, total;
};

__host__ __device__ inline LdsPlan lds_plan(int E, int R, int B, int nw)
{
    LdsPlan p;
    int o = 0;
    p.img = o;  o += E * A * 4;
    p.pool = o; o += (E != R) ? R * A * 4 : 0;
    o = (o + 15) & ~15;
    p.tw = o;   o += A * 16;
    p.fft = o;  o += nw * NH * 16;
    p.mags = o; o += nw * MAGS_STRIDE * 4;
    p.hist = o; o += ((R * B * 4) + 15) & ~15;
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
__device__ __forceinline__ bool point_pixel(float x, float y, float z, const NscBinParams &bp,
                                            int &pix, float &s, unsigned &flags)
{
    const bool fin = (fabsf(x) < INFINITY) && (fabsf(y) < INFINITY) && (fabsf(z) < INFINITY);
    const float xs = nsc_clip_sq(x), ys = nsc_clip_sq(y), zs = nsc_clip_sq(z);
    const float sxy = xs + ys;
    s = sxy + zs;
    if (!(fin && s >= bp.s_lo && s <= bp.s_hi)) return false;    // :151-155, :174-177
    int col, row;
    flags = 0;
    if (!nsc_col_fast(y, x, bp.az_delta, col)) { col = nsc_col_exact(y, x); flags |= 1u; }
    if (!nsc_row_fast(z, sxy, bp, row)) { row = nsc_row_exact(z, sxy, bp); flags |= 2u; }
    pix = row * A + col;                                         // :202
    return true;
}

__device__ __forceinline__ void scatter_point(float x, float y, float z, const NscBinParams &bp,
                                              unsigned *img)
{
    int pix; float s; unsigned fl;
    if (point_pixel(x, y, z, bp, pix, s, fl))
        atomicMin(&img[pix], __float_as_uint(s));   // ds_min_u32: s >= 0, so uint order == float order (:208)
}

template <int NT, int U>
__device__ __forceinline__ void scatter_range(const float *__restrict__ pts, long long p0, long long p1,
                                              int stride, int tid, const NscBinParams &bp, unsigned *img)
{
    const long long n = p1 - p0;
    if (stride == 4) {
        const f32x4 *P = reinterpret_cast<const f32x4 *>(pts) + p0;
        for (long long i = tid; i < n; i += (long long)NT * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = i + (long long)u * NT;
                if (j < n) v[u] = __builtin_nontemporal_load(&P[j]);
                else v[u] = f32x4{NAN, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) scatter_point(v[u].x, v[u].y, v[u].z, bp, img);
        }
    } else {
        const float *P = pts + p0 * 3;
        for (long long i = tid; i < n; i += (long long)NT * U) {
            float v[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = i + (long long)u * NT;
                if (j < n) { v[u][0] = P[j * 3]; v[u][1] = P[j * 3 + 1]; v[u][2] = P[j * 3 + 2]; }
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
__device__ __forceinline__ int interp_row(float *row, int lane, bool do_interp)
{
    float v[6];
    unsigned long long m[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = lane + 64 * j;
        v[j] = (c < A) ? row[c] : 0.0f;
        m[j] = __ballot(v[j] > 0.0f);                            // :35 valid_mask = row > 0
    }
    int nv = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nv += __popcll(m[j]);
    if (!do_interp || nv == 0 || nv == A) return nv;             // :37-43

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

            // np.interp in float64: slope*(x - xp[j]) + fp[j]   (:63, numpy compiled_base.c)
            const double f0 = (double)row[p < 0 ? p + A : p];
            const double f1 = (double)row[q >= A ? q - A : q];
            const double slope = (f1 - f0) / (double)(q - p);
            const double val = slope * (double)(c - p) + f0;
            row[c] = (float)val;                                 // :64 store into the float32 image
        }
    }
    return nv;
}

// ---------------------------------------------------------------------------------------------
// 360-point real FFT magnitude of one row (one wavefront), float64
//   z[n] = x[2n] + i x[2n+1], n < 180;  180 = 12 x 15 Cooley-Tukey (n = 15 n1 + n2, k = k1 + 12 k2);
//   X[k] = E[k] + W360^k O[k] unpacking for k = 0..180.          spectral_encoder.py:180-186
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fft_row(const float *x, double2 *buf, const double2 *tw, float *mags,
                                        int lane)
{
    if (lane < 60) {                                   // stage 1: 15 DFTs of length 12 over n1
        const int n2 = lane % 15, g = lane / 15;
        double zr[12], zi[12];
#pragma unroll
        for (int n1 = 0; n1 < 12; ++n1) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(&x[2 * (15 * n1 + n2)]);
            zr[n1] = (double)v.x;
            zi[n1] = (double)v.y;
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int k1 = 3 * g + o;
            const int step = 30 * k1;                  // W12^(n1 k1) = W360^(30 n1 k1)
            double re = 0.0, im = 0.0;
            int j = 0;
#pragma unroll
            for (int n1 = 0; n1 < 12; ++n1) {
                const double2 w = tw[j];
                re = fma(zr[n1], w.x, re); re = fma(zi[n1], w.y, re);
                im = fma(zi[n1], w.x, im); im = fma(-zr[n1], w.y, im);
                j += step; j -= (j >= A) ? A : 0;
            }
            const double2 w = tw[2 * n2 * k1];         // W180^(n2 k1), 2*14*11 < 360
            double2 y;
            y.x = fma(re, w.x, im * w.y);
            y.y = fma(im, w.x, -(re * w.y));
            buf[k1 * 15 + n2] = y;
        }
    }
    wave_sync();
    double yr[15], yi[15];
    if (lane < 60) {                                   // stage 2: 12 DFTs of length 15 over n2
        const int k1 = lane % 12;
#pragma unroll
        for (int n2 = 0; n2 < 15; ++n2) {
            const double2 v = buf[k1 * 15 + n2];
            yr[n2] = v.x;
            yi[n2] = v.y;
        }
    }
    wave_sync();                                       // every lane has its inputs: safe to overwrite buf
    if (lane < 60) {
        const int k1 = lane % 12, g = lane / 12;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int k2 = 3 * g + o;
            const int step = 24 * k2;                  // W15^(n2 k2) = W360^(24 n2 k2)
            double re = 0.0, im = 0.0;
            int j = 0;
#pragma unroll
            for (int n2 = 0; n2 < 15; ++n2) {
                const double2 w = tw[j];
                re = fma(yr[n2], w.x, re); re = fma(yi[n2], w.y, re);
                im = fma(yi[n2], w.x, im); im = fma(-yr[n2], w.y, im);
                j += step; j -= (j >= A) ? A : 0;
            }
            double2 z; z.x = re; z.y = im;
            buf[k1 + 12 * k2] = z;
        }
    }
    wave_sync();
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {                   // unpack the real spectrum, |X[k]| -> float32
        const int k = lane + 64 * jj;
        if (k <= NH) {
            const double2 zk = buf[k == NH ? 0 : k];
            const double2 zn = buf[(k == 0 || k == NH) ? 0 : NH - k];
            const double2 w = tw[k];
            const double sp = zk.x + zn.x, sm = zk.x - zn.x;   // a+c, a-c
            const double tp = zk.y + zn.y, tm = zk.y - zn.y;   // b+d, b-d
            const double xr = 0.5 * (sp + tp * w.x - sm * w.y);
            const double xi = 0.5 * (tm - sm * w.x - tp * w.y);
            mags[k] = (float)sqrt(xr * xr + xi * xi);          // :183 (x sqrt(360) of :186 undoes 'ortho')
        }
    }
    wave_sync();
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// everything after the image exists in LDS
//   mode 0: img holds squared-range bits (scatter output)  -> sqrt, interpolate, row copy
//   mode 1: img holds float32 range images from the caller -> no interpolation (forward(), :231)
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void finish_image(unsigned char *lds, const EncDev &d, int mode,
                                             const int *__restrict__ lut, float *__restrict__ out_desc,
                                             float *__restrict__ out_raw, float *__restrict__ out_interp)
{
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = d.E, R = d.R, B = d.B;
    const LdsPlan lp = lds_plan(E, R, B, NW);
    float *img = reinterpret_cast<float *>(lds + lp.img);
    float *pool = reinterpret_cast<float *>(lds + lp.pool);
    double2 *tw = reinterpret_cast<double2 *>(lds + lp.tw);
    double2 *fftbuf = reinterpret_cast<double2 *>(lds + lp.fft) + wave * NH;
    float *mags = reinterpret_cast<float *>(lds + lp.mags) + wave * MAGS_STRIDE;
    float *hist = reinterpret_cast<float *>(lds + lp.hist);
    int *seg = reinterpret_cast<int *>(lds + lp.seg);
    double *rowsum = reinterpret_cast<double *>(lds + lp.misc);
    int *rowflag = reinterpret_cast<int *>(lds + lp.misc + MAXR * 8);
    int *rowsrc = rowflag + MAXE;

    // twiddles and histogram segments (LUT is monotone: bin b owns frequencies [seg[b], seg[B+b]))
    for (int i = tid; i < A; i += NT) tw[i] = g_tw360[i];
    for (int b = tid; b < B; b += NT) {
        int lo = 0, hi = 0;
        for (int k = 0; k < F; ++k) { const int v = lut[k]; lo += (v < b); hi += (v <= b); }
        seg[b] = lo;
        seg[B + b] = hi;
    }

    if (mode == 0) {
        unsigned *raw = reinterpret_cast<unsigned *>(img);
        for (int i = tid; i < E * A; i += NT) {
            const unsigned v = raw[i];
            // min over sqrtf(s_i) == sqrtf(min s_i): sqrtf is correctly rounded, hence monotone.
            const float r = (v == NSC_EMPTY_BITS) ? 0.0f : sqrtf(__uint_as_float(v));   // :162,:214
            img[i] = r;
            if (out_raw) out_raw[i] = r;
        }
        __syncthreads();
        for (int r = wave; r < E; r += NW) {
            const int nv = interp_row(img + r * A, lane, d.interp != 0);
            if (lane == 0) rowflag[r] = (nv > 0);
        }
        __syncthreads();
        if (d.interp) {                                           // range_image.py:77-87
            if (tid == 0) {
                unsigned long long ne = 0ull;
                for (int r = 0; r < E; ++r) { ne |= (unsigned long long)(rowflag[r] != 0) << r; rowsrc[r] = r; }
                for (int r = 0; r < E; ++r) {
                    if ((ne >> r) & 1ull) continue;
                    for (int k = 1; k < E; ++k) {
                        if (r - k >= 0 && ((ne >> (r - k)) & 1ull)) { rowsrc[r] = rowsrc[r - k]; ne |= 1ull << r; break; }
                        if (r + k < E && ((ne >> (r + k)) & 1ull)) { rowsrc[r] = r + k; ne |= 1ull << r; break; }
                    }
                }
            }
            __syncthreads();
            for (int r = wave; r < E; r += NW) {
                const int s = rowsrc[r];                          // always an original (never copied) row
                if (s != r)
                    for (int c = lane; c < A; c += 64) img[r * A + c] = img[s * A + c];
            }
            __syncthreads();
        }
        if (out_interp)
            for (int i = tid; i < E * A; i += NT) out_interp[i] = img[i];
    } else {
        __syncthreads();
    }

    const float *rows = img;
    if (E != R) {                                                 // adaptive_avg_pool2d, :171-176
        for (int i = tid; i < R * A; i += NT) {
            const int pr = i / A, c = i - pr * A;
            const int r0 = (pr * E) / R;
            const int r1 = ((pr + 1) * E + R - 1) / R;
            float s = 0.0f;
            for (int r = r0; r < r1; ++r) s += img[r * A + c];
            pool[i] = s / (float)(r1 - r0);
        }
        rows = pool;
        __syncthreads();
    }

    for (int r = wave; r < R; r += NW) {
        fft_row(rows + r * A, fftbuf, tw, mags, lane);
        double part = 0.0;
        for (int b = lane; b < B; b += 64) {
            float h = 0.0f;
            const int k1 = seg[B + b];
            for (int k = seg[b]; k < k1; ++k) h += mags[k];      // scatter_add_, ascending k (:152-155)
            hist[r * B + b] = h;
            part += (double)h;
        }
        part = wave_sum(part);
        if (lane == 0) rowsum[r] = part;
        wave_sync();
    }
    __syncthreads();

    double tot = 0.0;
    for (int r = 0; r < R; ++r) tot += rowsum[r];
    const float s = (float)tot;                                   // :197
    const int D = R * B;
    if (s > d.eps) {
        const float den = s + d.eps;                              // :199
        for (int i = tid; i < D; i += NT) out_desc[i] = hist[i] / den;
    } else {
        const float u = 1.0f / (float)D;                          // :202
        for (int i = tid; i < D; i += NT) out_desc[i] = u;
    }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <int NW, int U>
__global__ __launch_bounds__(NW * 64) void encode_fused_kernel(
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
    scatter_range<NT, U>(pts, off[c], off[c + 1], stride, tid, d.bp, img);
    __syncthreads();
    const long long D = (long long)d.R * d.B;
    finish_image<NW>(lds, d, 0, lut, out_desc + c * D,
                     out_raw ? out_raw + (long long)c * npix : nullptr,
                     out_interp ? out_interp + (long long)c * npix : nullptr);
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
    for (int i = tid; i < npix; i += NT) {
        const unsigned v = img[i];
        if (v != NSC_EMPTY_BITS) atomicMin(&g[i], v);
    }
}

// src_u32 != null: squared-range workspace images (mode 0); else src_f32 caller images (mode 1)
template <int NW>
__global__ __launch_bounds__(NW * 64) void finish_kernel(
    const unsigned *__restrict__ src_u32, const float *__restrict__ src_f32, EncDev d,
    const int *__restrict__ lut, float *__restrict__ out_desc, float *__restrict__ out_raw,
    float *__restrict__ out_interp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = NW * 64;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int npix = d.E * A;
    unsigned *img = reinterpret_cast<unsigned *>(lds);
    if (src_u32) {
        const unsigned *g = src_u32 + (long long)c * npix;
        for (int i = tid; i < npix; i += NT) img[i] = g[i];
    } else {
        const float *g = src_f32 + (long long)c * npix;
        for (int i = tid; i < npix; i += NT) img[i] = __float_as_uint(g[i]);
    }
    __syncthreads();
    const long long D = (long long)d.R * d.B;
    finish_image<NW>(lds, d, src_u32 ? 0 : 1, lut, out_desc + c * D,
                     out_raw ? out_raw + (long long)c * npix : nullptr,
                     out_interp ? out_interp + (long long)c * npix : nullptr);
}

__global__ __launch_bounds__(256) void point_bins_kernel(
    const float *__restrict__ pts, long long n, int stride, NscBinParams bp,
    int *__restrict__ out_idx, unsigned char *__restrict__ out_flags)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float x = pts[i * stride], y = pts[i * stride + 1], z = pts[i * stride + 2];
        int pix; float s; unsigned fl = 0;
        const bool ok = point_pixel(x, y, z, bp, pix, s, fl);
        out_idx[i] = ok ? pix : -1;
        if (out_flags) out_flags[i] = ok ? (unsigned char)fl : (unsigned char)0;
    }
}

// ---------------------------------------------------------------------------------------------
// host side of the ABI
// ---------------------------------------------------------------------------------------------
constexpr int FUSED_NW = 8;          // 512 threads per workgroup
constexpr int FUSED_U = 4;           // float4 loads in flight per thread
constexpr int SPLIT_MIN_PTS = 16384; // a part must amortise its 5 760-pixel LDS init + flush
constexpr int SPLIT_TARGET_WGS = 512;

int check_params(const NscEncParams *p)
{
    if (!p) return NSC_EINVAL;
    if (p->n_azimuth != A) return NSC_EUNSUPPORTED;
    if (p->n_elevation < 1 || p->n_elevation > MAXE) return NSC_EUNSUPPORTED;
    if (p->target_rows < 1 || p->target_rows > MAXR) return NSC_EUNSUPPORTED;
    if (p->n_bins < 1 || p->n_bins > F) return NSC_EUNSUPPORTED;
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
    return d;
}

int split_parts(int32_t n_clouds, int64_t total_points)
{
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
    case NSC_EUNSUPPORTED: return "unsupported shape (n_azimuth must be 360, rows <= 64, target_rows <= 16, n_bins <= 181)";
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

    if (parts <= 1) {
        auto k = encode_fused_kernel<FUSED_NW, FUSED_U>;
        if ((st = set_lds(k, lp.total)) != NSC_OK) return st;
        hipLaunchKernelGGL(k, dim3(n_clouds), dim3(FUSED_NW * 64), lp.total, stream, pts, off, stride, d,
                           lut, out_desc, out_raw, out_interp);
    } else {
        const size_t need = (size_t)n_clouds * d.E * A * sizeof(unsigned);
        if (!ws || ws_bytes < need) return NSC_EWORKSPACE;
        if (hipMemsetAsync(ws, 0xff, need, stream) != hipSuccess) return NSC_ELAUNCH;
        auto ks = scatter_split_kernel<FUSED_NW, FUSED_U>;
        const int img_bytes = d.E * A * 4;
        if ((st = set_lds(ks, img_bytes)) != NSC_OK) return st;
        hipLaunchKernelGGL(ks, dim3(n_clouds * parts), dim3(FUSED_NW * 64), img_bytes, stream, pts, off,
                           stride, parts, d, static_cast<unsigned *>(ws));
        auto kf = finish_kernel<FUSED_NW>;
        if ((st = set_lds(kf, lp.total)) != NSC_OK) return st;
        hipLaunchKernelGGL(kf, dim3(n_clouds), dim3(FUSED_NW * 64), lp.total, stream,
                           static_cast<const unsigned *>(ws), static_cast<const float *>(nullptr), d, lut,
                           out_desc, out_raw, out_interp);
    }
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
                       static_cast<const unsigned *>(nullptr), imgs, d, lut, out_desc,
                       static_cast<float *>(nullptr), static_cast<float *>(nullptr));
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
                       static_cast<hipStream_t>(stream_), pts, (long long)n_points, stride, d.bp, out_idx,
                       out_flags);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

}  // extern "C"
