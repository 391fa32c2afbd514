// nsc_keyframe.hip -- keyframe-side helpers either side of the hot path.
//
// Path (reference file:line):
//   build_graph_from_keyframes_batch   src/keyframe/graph_manager.py:515-596   chain edges + [log1p(d)/5, theta/pi]
//   HistogramQuantizer.quantize        src/encoding/quantization.py:131-168    float32 -> uint16, sum forced to 65535
//   HistogramQuantizer.dequantize      src/encoding/quantization.py:170-191
//   CompressedDescriptor.to/from_bytes src/encoding/quantization.py:41-110     (2*n_bins + 120)-byte records
//   compute_overlap                    src/data/pose_utils.py:323-389          voxel IoU of two clouds
//
// All of it is byte / integer / index work and is bit-exact against the oracle, except the two float32
// edge features (2 ulp: device log1p / acos vs numpy's SVML/libm).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nsc.h"

namespace {

// ---------------------------------------------------------------------------------------------
// numpy's float32 add.reduce order: pairwise summation with 8 strided accumulators per leaf of at
// most 128 elements (numpy/_core/src/umath/loops_utils.h.src; oracle/keyframe_oracle.py
// pairwise_sum_f32).  The split only depends on the length, so the host lays the tree out once:
// leaves left to right + a post-order program (0 = push next leaf, 1 = add the two on top).
// ---------------------------------------------------------------------------------------------
constexpr int PW_MAX_LEAVES = 64;      // dim <= 4096: every leaf of a split node has >= 64 elements
constexpr int PW_MAX_DIM = 4096;

struct PwPlan {
    uint16_t start[PW_MAX_LEAVES];
    uint8_t len[PW_MAX_LEAVES];
    uint8_t prog[2 * PW_MAX_LEAVES];
    int32_t n_leaves, n_ops;
};

void pw_build(PwPlan &p, int start, int n)
{
    if (n <= 128) {
        p.start[p.n_leaves] = (uint16_t)start;
        p.len[p.n_leaves] = (uint8_t)n;
        ++p.n_leaves;
        p.prog[p.n_ops++] = 0;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_build(p, start, n2);
    pw_build(p, start + n2, n - n2);
    p.prog[p.n_ops++] = 1;
}

// All waves of the workgroup call this together (it synchronises with __syncthreads); every wave sums
// its own row.  leaf / stack: per-wave LDS scratch of PW_MAX_LEAVES and 16 floats.
__device__ float wave_pairwise_sum(const float *row, const PwPlan &plan, float *leaf, float *stack, int lane)
{
    for (int l0 = 0; l0 < plan.n_leaves; l0 += 8) {
        const int l = l0 + (lane >> 3), k = lane & 7;
        if (l < plan.n_leaves) {
            const int st = plan.start[l], len = plan.len[l];
            float res;
            if (len < 8) {
                res = 0.0f;
                for (int i = 0; i < len; ++i) res += row[st + i];
            } else {
                const int m = len - (len & 7);
                float r = row[st + k];
                for (int i = 8; i < m; i += 8) r += row[st + i + k];
                // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)): float add commutes, so the xor butterfly is that tree
                r += __shfl_xor(r, 1);
                r += __shfl_xor(r, 2);
                r += __shfl_xor(r, 4);
                res = r;
                for (int i = m; i < len; ++i) res += row[st + i];
            }
            if (k == 0) leaf[l] = res;
        }
    }
    __syncthreads();
    if (lane == 0) {
        int sp = 0, nl = 0;
        for (int o = 0; o < plan.n_ops; ++o) {
            if (plan.prog[o] == 0) stack[sp++] = leaf[nl++];
            else { const float b = stack[--sp], a = stack[--sp]; stack[sp++] = a + b; }
        }
    }
    __syncthreads();
    return stack[0];
}

// quantization.py:131-168 (DEQ = false) and :170-191 (DEQ = true); one wavefront per histogram
template <bool DEQ>
__global__ __launch_bounds__(256) void quantize_kernel(const void *__restrict__ in, int n, int dim, float eps,
                                                       void *__restrict__ out, PwPlan plan)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *row = smem + (size_t)w * (dim + PW_MAX_LEAVES + 16);
    float *leaf = row + dim, *stack = leaf + PW_MAX_LEAVES;
    const int d = blockIdx.x * 4 + w;
    const bool live = d < n;
    const long long base = (long long)(live ? d : n - 1) * dim;

    for (int j = lane; j < dim; j += 64)
        row[j] = DEQ ? (float)static_cast<const uint16_t *>(in)[base + j] : static_cast<const float *>(in)[base + j];
    __syncthreads();
    const float s = wave_pairwise_sum(row, plan, leaf, stack, lane);
    const bool norm = s > eps;
    const float den = s + eps;

    if (DEQ) {
        float *o = static_cast<float *>(out);
        const float uni = 1.0f / (float)dim;
        if (live)
            for (int j = lane; j < dim; j += 64) o[base + j] = norm ? row[j] / den : uni;
        return;
    }

    int *qrow = reinterpret_cast<int *>(row);
    int tot = 0, best = -1, best_j = 0x7fffffff;
    for (int j = lane; j < dim; j += 64) {
        float h = row[j];
        if (norm) h = h / den;
        float r = rintf(h * 65535.0f);                                   // np.round = half to even
        r = r > 0.0f ? (r < 65535.0f ? r : 65535.0f) : 0.0f;            // (NaN -> 0)
        const int qv = (int)r;
        qrow[j] = qv;
        tot += qv;
        if (qv > best) { best = qv; best_j = j; }                        // first maximum of this lane's bins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o);
        const int ob = __shfl_xor(best, o), oj = __shfl_xor(best_j, o);
        if (ob > best || (ob == best && oj < best_j)) { best = ob; best_j = oj; }
    }
    __syncthreads();
    if (lane == 0 && tot > 0 && tot != 65535) {                          // :154-166 put the rounding error on the first largest bin
        const int v = qrow[best_j] + (65535 - tot);
        qrow[best_j] = v < 0 ? 0 : (v > 65535 ? 65535 : v);
    }
    __syncthreads();
    uint16_t *o = static_cast<uint16_t *>(out);
    if (live)
        for (int j = lane; j < dim; j += 64) o[base + j] = (uint16_t)qrow[j];
}

// ---------------------------------------------------------------------------------------------
// records: [n_bins x u16][7 x f32 pose][f64 timestamp][u32 id][20 B hash][60 B zero]   quantization.py:41-72
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int meta_field(int b, int &off)
{
    // byte b of the 120-byte metadata block -> field id (0 pose, 1 timestamp, 2 id, 3 hash, 4 reserved)
    if (b < 28) { off = b; return 0; }
    if (b < 36) { off = b - 28; return 1; }
    if (b < 40) { off = b - 36; return 2; }
    if (b < 60) { off = b - 40; return 3; }
    off = b - 60;
    return 4;
}

__global__ __launch_bounds__(128) void pack_kernel(const uint16_t *__restrict__ q, const uint8_t *__restrict__ pose7,
                                                   const uint8_t *__restrict__ ts, const uint8_t *__restrict__ ids,
                                                   const uint8_t *__restrict__ hashes, int dim,
                                                   uint8_t *__restrict__ rec)
{
    const long long r = blockIdx.x;
    const int rb = 2 * dim + 120;
    uint8_t *o = rec + r * rb;
    for (int j = threadIdx.x; j < dim; j += 128)                          // records start on even addresses
        reinterpret_cast<uint16_t *>(o)[j] = q[r * dim + j];
    if (threadIdx.x < 120) {
        int off;
        const int f = meta_field(threadIdx.x, off);
        uint8_t v = 0;
        if (f == 0) v = pose7[r * 28 + off];
        else if (f == 1) v = ts[r * 8 + off];
        else if (f == 2) v = ids[r * 4 + off];
        else if (f == 3) v = hashes[r * 20 + off];
        o[2 * dim + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(128) void unpack_kernel(const uint8_t *__restrict__ rec, int dim, uint16_t *__restrict__ q,
                                                     uint8_t *__restrict__ pose7, uint8_t *__restrict__ ts,
                                                     uint8_t *__restrict__ ids, uint8_t *__restrict__ hashes)
{
    const long long r = blockIdx.x;
    const int rb = 2 * dim + 120;
    const uint8_t *in = rec + r * rb;
    for (int j = threadIdx.x; j < dim; j += 128) q[r * dim + j] = reinterpret_cast<const uint16_t *>(in)[j];
    if (threadIdx.x < 60) {
        int off;
        const int f = meta_field(threadIdx.x, off);
        const uint8_t v = in[2 * dim + threadIdx.x];
        if (f == 0) pose7[r * 28 + off] = v;
        else if (f == 1) ts[r * 8 + off] = v;
        else if (f == 2) ids[r * 4 + off] = v;
        else hashes[r * 20 + off] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// chain graph: graph_manager.py:520-596
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline long long s1(long long m, long long h)        // sum_{k < m} min(k, h)
{
    return m <= h + 1 ? m * (m - 1) / 2 : h * (h + 1) / 2 + (m - h - 1) * h;
}
// number of chain edges leaving nodes 0..i-1
__host__ __device__ inline long long chain_prefix(long long i, long long n, long long h)
{
    return s1(i, h) + (s1(n, h) - s1(n - i, h));
}

__device__ __forceinline__ void edge_feature(const double *__restrict__ poses, long long i, long long j,
                                             float *__restrict__ attr)
{
    const double *pi = poses + 16 * i, *pj = poses + 16 * j;
    const double dx = pi[3] - pj[3], dy = pi[7] - pj[7], dz = pi[11] - pj[11];
    const double d = sqrt((dx * dx + dy * dy) + dz * dz);                 // :537-539
    double tr = 0.0;                                                      // trace(R_j R_i^T) :546-548
#pragma unroll
    for (int a = 0; a < 3; ++a)
        tr += (pj[4 * a] * pi[4 * a] + pj[4 * a + 1] * pi[4 * a + 1]) + pj[4 * a + 2] * pi[4 * a + 2];
    tr = fmin(fmax(tr, -1.0), 3.0);
    const double c = fmin(fmax((tr - 1.0) / 2.0, -1.0), 1.0);
    const float ang = (float)acos(c);                                    // :549, float32 at :585
    const float d32 = (float)d;                                          // :584
    attr[0] = (float)log1p((double)d32) / 5.0f;                          // :588
    attr[1] = ang / 3.14159274101257324f;                                // :591 (float32 / weak python float)
}

__global__ __launch_bounds__(256) void chain_graph_kernel(const double *__restrict__ poses, int n, int half,
                                                          const long long *__restrict__ loops, int n_loops,
                                                          long long n_chain, long long E,
                                                          long long *__restrict__ edge_index,
                                                          float *__restrict__ edge_attr)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long slots = (long long)n * 2 * half;
    long long i, j, e;
    if (t < slots) {
        i = t / (2 * half);
        const int s = (int)(t - i * 2 * half);
        const int off = s < half ? s - half : s - half + 1;
        j = i + off;
        if (j < 0 || j >= n) return;
        const long long cneg = i < half ? i : half;
        e = chain_prefix(i, n, half) + (off < 0 ? off + cneg : cneg + off - 1);
    } else if (t < slots + 2LL * n_loops) {
        const long long l = (t - slots) >> 1;
        const int back = (int)((t - slots) & 1);
        const long long q = loops[2 * l], m = loops[2 * l + 1];
        i = back ? m : q;
        j = back ? q : m;
        e = n_chain + (t - slots);
    } else {
        return;
    }
    edge_index[e] = i;
    edge_index[E + e] = j;
    if (poses && edge_attr) {
        // the reference computes BOTH directions of a loop closure from (query, match) :565-572
        if (t >= slots && ((t - slots) & 1)) edge_feature(poses, j, i, edge_attr + 2 * e);
        else edge_feature(poses, i, j, edge_attr + 2 * e);
    }
}

// ---------------------------------------------------------------------------------------------
// voxel IoU: pose_utils.py:349-389 (after the caller's down-sampling).  One workgroup per pair; an
// LDS hash set over the voxel coordinates, which themselves sit in the (L2-resident) workspace.
// ---------------------------------------------------------------------------------------------
constexpr int VOX_TABLE = 16384;                  // 64 KB of LDS
constexpr int VOX_MAX_POINTS = 12288;             // load factor <= 0.75
constexpr unsigned VOX_FLAG = 0x80000000u;

__device__ __forceinline__ unsigned vox_hash(int x, int y, int z)
{
    unsigned h = (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(512) void voxel_overlap_kernel(const float *__restrict__ pts1, const long long *__restrict__ off1,
                                                            const float *__restrict__ pts2, const long long *__restrict__ off2,
                                                            int stride, const double *__restrict__ Tm, double voxel,
                                                            int4 *__restrict__ vox, long long total1,
                                                            int *__restrict__ counts, double *__restrict__ iou)
{
    __shared__ unsigned table[VOX_TABLE];
    __shared__ int cnt[3];                         // unique in 1, unique only in 2, intersection
    const int p = blockIdx.x, tid = threadIdx.x;
    const long long b1 = off1[p], b2 = off2[p];
    const long long n1l = off1[p + 1] - b1, n2l = off2[p + 1] - b2;
    if (n1l < 0 || n2l < 0 || n1l + n2l > VOX_MAX_POINTS) {               // host validates its bound; never overrun the table
        if (tid == 0) { counts[3 * p] = counts[3 * p + 1] = counts[3 * p + 2] = -1; iou[p] = -1.0; }
        return;
    }
    const int n1 = (int)n1l, n2 = (int)n2l;
    for (int i = tid; i < VOX_TABLE; i += 512) table[i] = 0u;
    if (tid < 3) cnt[tid] = 0;
    int4 *v1 = vox + b1, *v2 = vox + total1 + b2;
    const double *T = Tm + 16 * (long long)p;

    // voxel coordinates; .w = 1 for a row that survives the finite filter (:352-353)
    for (int i = tid; i < n1; i += 512) {
        const float *q = pts1 + (b1 + i) * stride;
        const double x = q[0], y = q[1], z = q[2];
        double c[3];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {                                     // dgemm order: fused multiply-adds over k
            const double *r = T + 4 * a;
            c[a] = __builtin_fma(r[3], 1.0, __builtin_fma(r[2], z, __builtin_fma(r[1], y, r[0] * x)));
            ok = ok && isfinite(c[a]);
        }
        for (int a = 3; a < stride; ++a) ok = ok && isfinite(q[a]);
        int4 o = {0, 0, 0, 0};
        if (ok) {
            o.x = (int)floor(fmin(fmax(c[0], -1e6), 1e6) / voxel);
            o.y = (int)floor(fmin(fmax(c[1], -1e6), 1e6) / voxel);
            o.z = (int)floor(fmin(fmax(c[2], -1e6), 1e6) / voxel);
            o.w = 1;
        }
        v1[i] = o;
    }
    const float vs = (float)voxel;
    for (int i = tid; i < n2; i += 512) {
        const float *q = pts2 + (b2 + i) * stride;
        bool ok = true;
        for (int a = 0; a < stride; ++a) ok = ok && isfinite(q[a]);
        int4 o = {0, 0, 0, 0};
        if (ok) {                                                         // cloud 2 stays float32 (:361-363)
            o.x = (int)floorf(fminf(fmaxf(q[0], -1e6f), 1e6f) / vs);
            o.y = (int)floorf(fminf(fmaxf(q[1], -1e6f), 1e6f) / vs);
            o.z = (int)floorf(fminf(fmaxf(q[2], -1e6f), 1e6f) / vs);
            o.w = 1;
        }
        v2[i] = o;
    }
    __threadfence_block();
    __syncthreads();

    auto coords_of = [&](unsigned e) -> int4 {                            // e = 1 + point index (cloud 2 after cloud 1)
        const int idx = (int)(e & ~VOX_FLAG) - 1;
        return idx < n1 ? v1[idx] : v2[idx - n1];
    };

    // set 1
    for (int i = tid; i < n1; i += 512) {
        const int4 c = v1[i];
        if (!c.w) continue;
        unsigned h = vox_hash(c.x, c.y, c.z) & (VOX_TABLE - 1);
        for (;;) {
            unsigned e = table[h];
            if (e == 0u) {
                e = atomicCAS(&table[h], 0u, (unsigned)(i + 1));
                if (e == 0u) { atomicAdd(&cnt[0], 1); break; }
            }
            const int4 o = coords_of(e);
            if (o.x == c.x && o.y == c.y && o.z == c.z) break;            // voxel already present
            h = (h + 1) & (VOX_TABLE - 1);
        }
    }
    __syncthreads();
    // set 2: a voxel of set 1 is counted into the intersection by the first point that reaches it
    for (int i = tid; i < n2; i += 512) {
        const int4 c = v2[i];
        if (!c.w) continue;
        unsigned h = vox_hash(c.x, c.y, c.z) & (VOX_TABLE - 1);
        for (;;) {
            unsigned e = table[h];
            if (e == 0u) {
                e = atomicCAS(&table[h], 0u, (unsigned)(n1 + i + 1));
                if (e == 0u) { atomicAdd(&cnt[1], 1); break; }
            }
            const int4 o = coords_of(e);
            if (o.x == c.x && o.y == c.y && o.z == c.z) {
                if ((int)(e & ~VOX_FLAG) - 1 < n1) {
                    const unsigned old = atomicOr(&table[h], VOX_FLAG);
                    if (!(old & VOX_FLAG)) atomicAdd(&cnt[2], 1);
                }
                break;
            }
            h = (h + 1) & (VOX_TABLE - 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int u1 = cnt[0], only2 = cnt[1], inter = cnt[2];
        counts[3 * p] = u1;
        counts[3 * p + 1] = only2 + inter;
        counts[3 * p + 2] = inter;
        const int uni = u1 + only2;
        iou[p] = uni > 0 ? (double)inter / (double)uni : 0.0;            // :386-389
    }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH; }

int quantize_common(bool deq, const void *in, int32_t n, int32_t dim, float eps, void *out, void *stream)
{
    if (n < 0 || dim < 1 || (n > 0 && (!in || !out))) return NSC_EINVAL;
    if (dim > PW_MAX_DIM) return NSC_EUNSUPPORTED;
    if (n == 0) return NSC_OK;
    PwPlan plan;
    plan.n_leaves = plan.n_ops = 0;
    pw_build(plan, 0, dim);
    const size_t lds = 4 * (size_t)(dim + PW_MAX_LEAVES + 16) * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((n + 3) / 4), block(256);
    if (deq) {
        if (lds > 65536 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(quantize_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return NSC_ELAUNCH;
        hipLaunchKernelGGL(quantize_kernel<true>, grid, block, lds, st, in, n, dim, eps, out, plan);
    } else {
        if (lds > 65536 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(quantize_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return NSC_ELAUNCH;
        hipLaunchKernelGGL(quantize_kernel<false>, grid, block, lds, st, in, n, dim, eps, out, plan);
    }
    return launch_status();
}

}  // namespace

extern "C" {

int nsc_quantize_descriptors(const float *hist, int32_t n, int32_t dim, float eps, uint16_t *quantized, void *stream)
{
    return quantize_common(false, hist, n, dim, eps, quantized, stream);
}

int nsc_dequantize_descriptors(const uint16_t *quantized, int32_t n, int32_t dim, float eps, float *hist, void *stream)
{
    return quantize_common(true, quantized, n, dim, eps, hist, stream);
}

size_t nsc_record_bytes(int32_t dim) { return dim > 0 ? 2 * (size_t)dim + 120 : 0; }

int nsc_pack_records(const uint16_t *quantized, const float *pose7, const double *timestamps,
                     const uint32_t *keyframe_ids, const uint8_t *hashes, int32_t n, int32_t dim,
                     uint8_t *records, void *stream)
{
    if (n < 0 || dim < 1) return NSC_EINVAL;
    if (n == 0) return NSC_OK;
    if (!quantized || !pose7 || !timestamps || !keyframe_ids || !hashes || !records) return NSC_EINVAL;
    hipLaunchKernelGGL(pack_kernel, dim3(n), dim3(128), 0, static_cast<hipStream_t>(stream), quantized,
                       reinterpret_cast<const uint8_t *>(pose7), reinterpret_cast<const uint8_t *>(timestamps),
                       reinterpret_cast<const uint8_t *>(keyframe_ids), hashes, dim, records);
    return launch_status();
}

int nsc_unpack_records(const uint8_t *records, int32_t n, int32_t dim, uint16_t *quantized, float *pose7,
                       double *timestamps, uint32_t *keyframe_ids, uint8_t *hashes, void *stream)
{
    if (n < 0 || dim < 1) return NSC_EINVAL;
    if (n == 0) return NSC_OK;
    if (!quantized || !pose7 || !timestamps || !keyframe_ids || !hashes || !records) return NSC_EINVAL;
    hipLaunchKernelGGL(unpack_kernel, dim3(n), dim3(128), 0, static_cast<hipStream_t>(stream), records, dim,
                       quantized, reinterpret_cast<uint8_t *>(pose7), reinterpret_cast<uint8_t *>(timestamps),
                       reinterpret_cast<uint8_t *>(keyframe_ids), hashes);
    return launch_status();
}

int64_t nsc_chain_graph_num_edges(int32_t n_nodes, int32_t temporal_neighbors, int32_t n_loops)
{
    if (n_nodes < 0 || temporal_neighbors < 0 || n_loops < 0) return -1;
    return chain_prefix(n_nodes, n_nodes, temporal_neighbors / 2) + 2LL * n_loops;
}

int nsc_build_chain_graph(const double *poses, int32_t n_nodes, int32_t temporal_neighbors, const int64_t *loops,
                          int32_t n_loops, int64_t *edge_index, float *edge_attr, void *stream)
{
    if (n_nodes < 0 || temporal_neighbors < 0 || n_loops < 0 || (n_loops > 0 && !loops)) return NSC_EINVAL;
    const int half = temporal_neighbors / 2;
    const long long n_chain = chain_prefix(n_nodes, n_nodes, half);
    const long long E = n_chain + 2LL * n_loops;
    if (E == 0) return NSC_OK;
    if (!edge_index) return NSC_EINVAL;
    const long long threads = (long long)n_nodes * 2 * half + 2LL * n_loops;
    if (threads > 0x7fffffffLL * 256) return NSC_EUNSUPPORTED;
    hipLaunchKernelGGL(chain_graph_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), poses, n_nodes, half,
                       reinterpret_cast<const long long *>(loops), n_loops, n_chain, E,
                       reinterpret_cast<long long *>(edge_index), edge_attr);
    return launch_status();
}

size_t nsc_voxel_overlap_workspace_bytes(int64_t total_points1, int64_t total_points2)
{
    if (total_points1 < 0 || total_points2 < 0) return 0;
    return (size_t)(total_points1 + total_points2) * sizeof(int4);
}

int nsc_voxel_overlap(const float *points1, const int64_t *offsets1, const float *points2, const int64_t *offsets2,
                      int32_t n_pairs, int64_t total_points1, int64_t total_points2, int64_t max_pair_points,
                      int32_t stride_floats, const double *transforms, double voxel_size, int32_t *counts,
                      double *iou, void *ws, size_t ws_bytes, void *stream)
{
    if (n_pairs < 0 || total_points1 < 0 || total_points2 < 0 || max_pair_points < 0) return NSC_EINVAL;
    if (stride_floats != 3 && stride_floats != 4) return NSC_EINVAL;
    if (!(voxel_size > 0.0)) return NSC_EINVAL;
    if (n_pairs == 0) return NSC_OK;
    if (!offsets1 || !offsets2 || !transforms || !counts || !iou) return NSC_EINVAL;
    if ((total_points1 > 0 && !points1) || (total_points2 > 0 && !points2)) return NSC_EINVAL;
    if (max_pair_points > VOX_MAX_POINTS) return NSC_EUNSUPPORTED;
    const size_t need = nsc_voxel_overlap_workspace_bytes(total_points1, total_points2);
    if (need > 0 && (!ws || ws_bytes < need)) return NSC_EWORKSPACE;
    hipLaunchKernelGGL(voxel_overlap_kernel, dim3(n_pairs), dim3(512), 0, static_cast<hipStream_t>(stream), points1,
                       reinterpret_cast<const long long *>(offsets1), points2,
                       reinterpret_cast<const long long *>(offsets2), stride_floats, transforms, voxel_size,
                       static_cast<int4 *>(ws), (long long)total_points1, counts, iou);
    return launch_status();
}

}  // extern "C"
