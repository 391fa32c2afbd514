// nsc_gat_train.hip -- gfx950 kernels + C ABI for the training step of the GNN enhancer
// (BASELINE.json configs[4]: triplet loss + GAT backward).
//
// Path (reference file:line):
//   SpectralGNN.forward in train() mode          src/gnn/model.py:96-153 (BatchNorm batch statistics
//                                                :117,:132, feature dropout :137, GATConv attention dropout)
//   TripletLoss.forward                          src/gnn/trainer.py:44-68
//   loss.backward() through the full graph       src/gnn/trainer.py:205-213
//
// Structure: the forward saves z (pre-BatchNorm activations), the per-layer transformed features g,
// attention logits parts and softmax weights in a caller workspace; the backward walks the layers in
// reverse with (a) column reductions for BatchNorm / bias / attention-vector gradients (two-pass,
// float64 partials, deterministic), (b) a per-target kernel for the softmax / leaky-ReLU backward,
// (c) a per-source kernel (transposed CSR, no atomics) for the feature gradient, (d) f32-MFMA GEMMs in
// NT / NN / TN form with split-K slabs for the weight gradients (deterministic reduce).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "../../include/nsc.h"

namespace {

#include "nsc_gemm_glds.h"
#include "nsc_fill.h"

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------
// counter-based dropout mask: keep iff u01(hash(seed, stream, idx)) >= p
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned hash3(unsigned long long seed, unsigned stream, unsigned long long idx)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((unsigned long long)stream << 48);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;       // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 40);                        // 24 random bits
}
// The seed of a launch: a value baked into the kernel arguments, or -- when the caller gives NscGatTrainCfg.seed_dev -- a
// device word read at run time, so that a captured hipGraph replays with a fresh mask per step.
struct SeedRef {
    unsigned long long v;
    const unsigned long long *p;
    __device__ __forceinline__ unsigned long long get() const { return p ? *p : v; }
};

__device__ __forceinline__ float keep_scale(float p, unsigned long long seed, unsigned stream,
                                            unsigned long long idx)
{
    if (p <= 0.0f) return 1.0f;
    const float u = (float)hash3(seed, stream, idx) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// general f32-MFMA GEMM:  C[M,N] = sum_k A(m,k) * B(n,k)
//   AKM = false: A(m,k) = A[m*lda + k] (k contiguous);  AKM = true: A(m,k) = A[k*lda + m] (m contiguous)
//   same for B.  blockIdx.z = split-K slice; slices write their own (M,N) slab.
// ---------------------------------------------------------------------------------------------
// (BM = 64 -- four accumulators per wave sharing every B operand -- was measured in round 4 on the split-K weight gradients:
// 34.2 us against 25.8 us per product at 4 541 keyframes for the 32-row form, which stays.)
template <bool AKM, bool BKM, int BM = 32>
__global__ __launch_bounds__(256) void gemm_gen_kernel(const float *__restrict__ A, int lda,
                                                       const float *__restrict__ B, int ldb, int M, int N,
                                                       int K, int kchunk, float *__restrict__ C, int ldc,
                                                       long long slab, const float *__restrict__ bias,
                                                       int accumulate)
{
    constexpr int BN = 64, BK = 64, LD = BK + 4, NH = BM / 16;
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k0 = blockIdx.z * kchunk, k1 = min(K, k0 + kchunk);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = zero;

    // One 64-deep chunk in registers ahead of the one being multiplied (round 3: the first form fetched, staged, multiplied
    // and only then fetched again -- with 4-5 chunks per split-K slab its time was the sum of their L2 round trips).
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;
    f32x4 ra[NA], rb[NB];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i;
            f32x4 v = zero;
            if (!AKM) {
                const int row = f >> 4, c4 = f & 15, gm = m0 + row, gk = kb + 4 * c4;
                if (gm < M && gk + 3 < k1) v = *reinterpret_cast<const f32x4 *>(A + (long long)gm * lda + gk);
                else if (gm < M)
                    for (int j = 0; j < 4; ++j) if (gk + j < k1) v[j] = A[(long long)gm * lda + gk + j];
            } else {
                const int kl = f / (BM / 4), m4 = f % (BM / 4), gk = kb + kl, gm = m0 + 4 * m4;
                if (gk < k1 && gm + 3 < M) v = *reinterpret_cast<const f32x4 *>(A + (long long)gk * lda + gm);
                else if (gk < k1)
                    for (int j = 0; j < 4; ++j) if (gm + j < M) v[j] = A[(long long)gk * lda + gm + j];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            f32x4 v = zero;
            if (!BKM) {
                const int row = f >> 4, c4 = f & 15, gn = n0 + row, gk = kb + 4 * c4;
                if (gn < N && gk + 3 < k1) v = *reinterpret_cast<const f32x4 *>(B + (long long)gn * ldb + gk);
                else if (gn < N)
                    for (int j = 0; j < 4; ++j) if (gk + j < k1) v[j] = B[(long long)gn * ldb + gk + j];
            } else {
                const int kl = f / (BN / 4), n4 = f % (BN / 4), gk = kb + kl, gn = n0 + 4 * n4;
                if (gk < k1 && gn + 3 < N) v = *reinterpret_cast<const f32x4 *>(B + (long long)gk * ldb + gn);
                else if (gk < k1)
                    for (int j = 0; j < 4; ++j) if (gn + j < N) v[j] = B[(long long)gk * ldb + gn + j];
            }
            rb[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i;
            if (!AKM) {
                *reinterpret_cast<f32x4 *>(&As[(f >> 4) * LD + 4 * (f & 15)]) = ra[i];
            } else {
                const int kl = f / (BM / 4), m4 = f % (BM / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(4 * m4 + j) * LD + kl] = ra[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            if (!BKM) {
                *reinterpret_cast<f32x4 *>(&Bs[(f >> 4) * LD + 4 * (f & 15)]) = rb[i];
            } else {
                const int kl = f / (BN / 4), n4 = f % (BN / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(4 * n4 + j) * LD + kl] = rb[i][j];
            }
        }
    };
    if (k0 < k1) fetch(k0);
    for (int kb = k0; kb < k1; kb += BK) {
        stage();
        __syncthreads();
        if (kb + BK < k1) fetch(kb + BK);
#pragma unroll
        for (int d = 0; d < BK / 16; ++d) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(&Bs[(wave * 16 + r) * LD + 16 * d + 4 * q]);
            f32x4 av[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) av[h] = *reinterpret_cast<const f32x4 *>(&As[(16 * h + r) * LD + 16 * d + 4 * q]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][t], bv[t], acc[h], 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wave * 16 + r;
    if (col >= N) return;
    float *Cz = C + (long long)blockIdx.z * slab;
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = m0 + 16 * h + 4 * q + reg;
            if (row >= M) continue;
            float v = acc[h][reg] + bv;
            float *dst = Cz + (long long)row * ldc + col;
            if (accumulate) v += *dst;
            *dst = v;
        }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ part, int S, long long MN,
                                                          float *__restrict__ out, int accumulate)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.0f;
    for (int z = 0; z < S; ++z) s += part[(long long)z * MN + i];     // fixed order: deterministic
    out[i] = accumulate ? out[i] + s : s;
}

// The slab sums of ALL weight-gradient products of a backward together, in the backward's last launch (round 4): nothing inside the backward reads a weight
// gradient, so every product keeps its slabs in a region of its own and the sums -- same fixed order, same (out + s) -- run once at
// the end: five 5 us launches per batch become one.
struct SlabJob {
    const float *part;
    float *out;
    long long MN;
    int S, accumulate;
    unsigned blk0;                                                  // first workgroup of this job
};
constexpr int SLAB_JOBS = 8;
struct SlabBatch {
    SlabJob j[SLAB_JOBS];
    int n;
    unsigned blocks;
};
struct SlabDefer {                                                  // host side: the regions handed out so far
    SlabBatch b;
    float *base;
    size_t cap, used;                                               // floats
};

inline float *slab_defer_take(SlabDefer *d, long long MN, int S)
{
    if (!d || !d->base || d->b.n >= SLAB_JOBS) return nullptr;
    const size_t need = ((size_t)MN * (size_t)S + 63) / 64 * 64;
    if (d->used + need > d->cap) return nullptr;
    float *p = d->base + d->used;
    d->used += need;
    return p;
}

inline void slab_defer_push(SlabDefer *d, const float *part, float *out, long long MN, int S, int accumulate)
{
    SlabJob &J = d->b.j[d->b.n++];
    J.part = part; J.out = out; J.MN = MN; J.S = S; J.accumulate = accumulate; J.blk0 = d->b.blocks;
    d->b.blocks += (unsigned)((MN + 255) / 256);
}

__device__ __forceinline__ void slab_reduce_multi_body(const SlabBatch &b, unsigned bid)
{
    int j = 0;
#pragma unroll
    for (int t = 1; t < SLAB_JOBS; ++t)
        if (t < b.n && bid >= b.j[t].blk0) j = t;
    const float *__restrict__ part = b.j[j].part;
    float *__restrict__ out = b.j[j].out;
    const long long MN = b.j[j].MN;
    const int S = b.j[j].S;
    const long long i = (long long)(bid - b.j[j].blk0) * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.0f;
    for (int z = 0; z < S; ++z) s += part[(long long)z * MN + i];     // fixed order: deterministic
    out[i] = b.j[j].accumulate ? out[i] + s : s;
}



// ---------------------------------------------------------------------------------------------
// Weight gradients  dW[M,N] = sum_k A[k][m] * B[k][n]  with BOTH operands k-major (dW = dY^T X: the rows of dY and X are the
// k index), round 4.  The k-major operands need no transpose when the MFMA operand is read element by element: the tiles go
// to LDS as they lie in memory, [k][64 m] and [k][64 n], by LDS-DMA (global_load_lds_dwordx4: 16 bytes = four consecutive m of
// one k; no register ring, no scalar LDS stores -- gemm_gen_kernel<true,true> staged every float4 through four ds_write_b32 and
// kept one LDS buffer between two barriers per chunk: 18-26 us per product), and lane (r, q) of a 16x16x4 MFMA reads A[k = 4 s + q]
// [m = .. + r] as one ds_read_b32.  Bank conflicts are avoided on the SOURCE side like in gemm_glds_kernel: the 16-byte slot of
// m-quad mq in row k sits at slot mq ^ (4 (k mod 4)), so the four k rows a wave instruction reads land in four different bank
// groups.  64 x 64 tiles of 2 x 2 waves (a 32 x 32 wave tile = four accumulators share two A and two B reads per k-step),
// four computing + four staging waves, three stages, one raw barrier per 64-deep chunk with the younger chunk's pieces in flight
// across it, split over K slabs
// (blockIdx.z) whose sums slab_reduce_kernel adds in fixed order: deterministic.  k order of every output element: ascending.
// ---------------------------------------------------------------------------------------------
template <int TM>   // tile (64 TM) x 64: TM = 2 for the wide products (a computing wave owns a 64 x 32 tile: eight accumulators)
__global__ __launch_bounds__(512) void gemm_tn_glds_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                           int ldb, int M, int N, int K, int kslab,
                                                           float *__restrict__ slabs, long long slab)
{
    constexpr int BM = 64 * TM, NST = 3, STAGE = 64 * BM + 64 * 64;   // floats per stage: A tile [64 k][BM m] | B tile [64 k][64 n]
    constexpr int SA = 16 * TM;                                    // 16-byte slots per A row; a 1 KiB piece = 64 / SA rows
    constexpr int PA = 64 * SA / 64, NPW = (PA + 16) / 4;          // A pieces per chunk (+ 16 B pieces), pieces per staging wave
    extern __shared__ __attribute__((aligned(1024))) float tn_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * 64;
    const int k0 = blockIdx.z * kslab, k1 = min(K, k0 + kslab);
    const int nchunks = (k1 - k0 + 63) >> 6;
    if (nchunks <= 0) return;                                       // (cannot happen: the host trims empty slabs)

    if (wave8 >= 4) {
        // ---- staging waves (wave specialisation as in gemm_glds_kernel: an LDS-DMA instruction costs its wave 60-185 cycles of
        // issue; with them in the computing waves' own stream the first form of this kernel took 17.6 us per product): piece p of a
        // chunk = 1 KiB, lane-linear: 64 / SA rows of the A tile (p < PA) or 4 rows of the B tile; wave w takes pieces w, w + 4, ...
        auto issue = [&](int ch, int stage) {
            float *dst = tn_lds + stage * STAGE + wave * 256;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int p = wave + 4 * j;                        // uniform
                const float *g;
                if (p < PA) {
                    const int row = (64 / SA) * p + lane / SA, slot = lane % SA;
                    int ma = m0 + 4 * (slot ^ ((row & 3) << 2));    // the slot holds quad slot ^ 4 (k mod 4)
                    ma = ma + 3 < M ? ma : M - 4;                   // quads past the matrix re-read valid columns (never stored)
                    int k = k0 + (ch << 6) + row;
                    k = k < K ? k : K - 1;                          // rows past K re-read the last row: masked out of the products
                    g = A + (long long)k * lda + ma;
                } else {
                    const int row = 4 * (p - PA) + (lane >> 4), slot = lane & 15;
                    int nb = n0 + 4 * (slot ^ ((row & 3) << 2));
                    nb = nb + 3 < N ? nb : N - 4;
                    int k = k0 + (ch << 6) + row;
                    k = k < K ? k : K - 1;
                    g = B + (long long)k * ldb + nb;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
            }
        };
        issue(0, 0);
        if (1 < nchunks) issue(1, 1);
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) glds_wait_barrier<NPW>();         // chunk c landed (this wave's pieces of chunk c + 1 may fly on)
            else glds_wait_barrier<0>();
            if (c + 2 < nchunks) issue(c + 2, (c + 2) % NST);       // its stage held chunk c - 1: every computing wave is past that
        }
        return;
    }

    // ---- computing waves: 2 x 2, a (32 TM) x 32 tile each
    constexpr int NI = 2 * TM;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i][0] = acc[i][1] = zero;
    const int wm = wave >> 1, wn = wave & 1;
    // operand addresses inside a stage (floats): row k = 4 s + q -> + (4 s + q) * row length; quad (m / 4) ^ (4 q), element r % 4
    int offA[NI], offB[2];
#pragma unroll
    for (int i = 0; i < NI; ++i) offA[i] = q * BM + 4 * ((8 * TM * wm + 4 * i + (r >> 2)) ^ (q << 2)) + (r & 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) offB[i] = 64 * BM + q * 64 + 4 * ((8 * wn + 4 * i + (r >> 2)) ^ (q << 2)) + (r & 3);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_barrier" ::: "memory");                    // barrier c: chunk c has landed, chunk c - 1 is read
        const float *st = tn_lds + (c % NST) * STAGE;
        const int kleft = k1 - (k0 + (c << 6));                    // valid k rows in this chunk
        if (kleft >= 64) {
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                float a[NI], b[2];
#pragma unroll
                for (int i = 0; i < NI; ++i) a[i] = st[offA[i] + s4 * 4 * BM];
#pragma unroll
                for (int i = 0; i < 2; ++i) b[i] = st[offB[i] + s4 * 256];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            for (int s4 = 0; 4 * s4 < kleft; ++s4) {                // the slab's last, short chunk: k rows past the end count as zero
                const bool in = 4 * s4 + q < kleft;
                float a[NI], b[2];
#pragma unroll
                for (int i = 0; i < NI; ++i) a[i] = in ? st[offA[i] + s4 * 4 * BM] : 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i) b[i] = in ? st[offB[i] + s4 * 256] : 0.0f;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float *Cz = slabs + (long long)blockIdx.z * slab;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 32 * wn + 16 * j + r;
            if (col >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {                    // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
                const int row = m0 + 32 * TM * wm + 16 * i + 4 * q + reg;
                if (row < M) Cz[(long long)row * N + col] = acc[i][j][reg];
            }
        }
}

// dW = A^T B over K slabs on gemm_tn_glds_kernel; false when the operands do not fit it (the caller takes gemm_gen_kernel)
template <int TM>
bool launch_tn_glds_cfg(hipStream_t st, const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C,
                        int accumulate, float *slabs, int max_slabs, SlabDefer *defer)
{
    constexpr unsigned lds = 3 * (64 * 64 * TM + 64 * 64) * 4;     // 96 / 144 KB: above 64 KB a kernel is opted in, per device
    static std::atomic<int> opted[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    int o = opted[dev].load(std::memory_order_acquire);
    if (o == 0) {
        o = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_glds_kernel<TM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) == hipSuccess ? 1 : 2;
        opted[dev].store(o, std::memory_order_release);
    }
    if (o != 1) return false;
    const int tiles = ((M + 64 * TM - 1) / (64 * TM)) * ((N + 63) / 64);
    int splits = 256 / tiles;                                       // about one round of workgroups on the 256 CUs
    splits = splits < 1 ? 1 : (splits > max_slabs ? max_slabs : splits);
    int kslab = (K + splits - 1) / splits;
    kslab = (kslab + 63) / 64 * 64;
    splits = (K + kslab - 1) / kslab;                               // no empty slab
    const long long MN = (long long)M * N;
    float *own = slab_defer_take(defer, MN, splits);                // a region of this product's own: its sums wait for the batched launch
    float *dst = own ? own : slabs;
    hipLaunchKernelGGL(gemm_tn_glds_kernel<TM>, dim3((N + 63) / 64, (M + 64 * TM - 1) / (64 * TM), splits), dim3(512), lds, st, A, lda, B,
                       ldb, M, N, K, kslab, dst, MN);
    if (own) slab_defer_push(defer, own, C, MN, splits, accumulate);
    else hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, slabs, splits, MN, C, accumulate);
    return true;
}

bool launch_tn_glds(hipStream_t st, const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C,
                    int accumulate, float *slabs, int max_slabs, SlabDefer *defer = nullptr)
{
    if ((lda & 3) || (ldb & 3) || M < 4 || N < 4 || (M & 3) || (N & 3) || K < 1 || !slabs ||
        (reinterpret_cast<unsigned long long>(A) & 15) || (reinterpret_cast<unsigned long long>(B) & 15))
        return false;
    // the wide products (800 x 256 / 256 x 800: 52 tiles of 64 x 64 x 4 slabs would leave 48 CUs idle) take 128 x 64 tiles:
    // 28 / 26 tiles x 9 slabs = one round of 252 / 234 workgroups
    if (((M + 63) / 64) * ((N + 63) / 64) >= 40 && M >= 128 &&
        launch_tn_glds_cfg<2>(st, A, lda, B, ldb, M, N, K, C, accumulate, slabs, max_slabs, defer))
        return true;
    return launch_tn_glds_cfg<1>(st, A, lda, B, ldb, M, N, K, C, accumulate, slabs, max_slabs, defer);
}

// ---------------------------------------------------------------------------------------------
// column reductions over the N rows of an (N, C) matrix, float64 partials, two passes:
//   out_a[c] = sum_n P[n][c] * (w ? w[n] : 1)
//   out_b[c] = sum_n P[n][c] * Q'[n][c],  Q' = Q or (Q - qm[c]) * qs[c]        (Q nullable)
// ---------------------------------------------------------------------------------------------
// mode 0: out_a = A, out_b = B (plain sums, nullable outputs)
// mode 1: BatchNorm statistics of P (Q = P): mean = A/N, var = B/N - mean^2 -> out_a = mean,
//         out_b = 1/sqrt(var+eps); optional running-stat update (momentum, unbiased variance)
struct ColFinal {
    int mode, N;
    float eps, momentum;
    float *out_a, *out_b, *run_mean, *run_var;
    int acc_a, acc_b;          // mode 0: add to what out_a / out_b hold (gradient accumulation) instead of overwriting
    float *copy_a, *copy_b;    // mode 0, nullable: the same sums also stored to (acc_copy: added into) a second vector -- the
    int acc_copy;              // BatchNorm gradients are both an input of the next kernel and a parameter gradient
};

__device__ __forceinline__ void colreduce_finish(const double *__restrict__ part, int R, int C, int c, int cx, int ry,
                                                 double *sa, double *sb, const ColFinal &f)
{
    // 64 columns per workgroup, the R partial rows split over the 4 waves (fixed order -> deterministic)
    // (loads in groups of 8 ahead of the adds: one partial per L2 round trip was a 5 us latency chain; the order of the
    // additions is unchanged)
    double a = 0.0, b = 0.0;
    if (c < C) {
        int r = ry;
        for (; r + 28 < R; r += 32) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2 *>(&part[((long long)(r + 4 * u) * C + c) * 2]);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
        }
        for (; r < R; r += 4) { a += part[((long long)r * C + c) * 2]; b += part[((long long)r * C + c) * 2 + 1]; }
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (ry != 0 || c >= C) return;
    a = (sa[cx] + sa[64 + cx]) + (sa[128 + cx] + sa[192 + cx]);
    b = (sb[cx] + sb[64 + cx]) + (sb[128 + cx] + sb[192 + cx]);
    if (f.mode == 0) {
        if (f.out_a) f.out_a[c] = f.acc_a ? f.out_a[c] + (float)a : (float)a;
        if (f.out_b) f.out_b[c] = f.acc_b ? f.out_b[c] + (float)b : (float)b;
        if (f.copy_a) f.copy_a[c] = f.acc_copy ? f.copy_a[c] + (float)a : (float)a;
        if (f.copy_b) f.copy_b[c] = f.acc_copy ? f.copy_b[c] + (float)b : (float)b;
    } else {
        const double mean = a / f.N;
        double var = b / f.N - mean * mean;
        if (var < 0.0) var = 0.0;
        f.out_a[c] = (float)mean;
        f.out_b[c] = (float)(1.0 / sqrt(var + (double)f.eps));
        if (f.run_mean) {                                 // nn.BatchNorm1d: momentum 0.1, unbiased running_var
            const double unb = f.N > 1 ? var * f.N / (f.N - 1) : var;
            f.run_mean[c] = (1.0f - f.momentum) * f.run_mean[c] + f.momentum * (float)mean;
            f.run_var[c] = (1.0f - f.momentum) * f.run_var[c] + f.momentum * (float)unb;
        }
    }
}

// The sum of the R partial rows for the 64 columns of a column block, by the workgroup that CONSUMES it (round 4: the forward's
// BatchNorm apply and the backward's apply pass finish the reduction of the pass before them themselves -- 64 x 64 x 16 bytes of
// L2 hits per workgroup instead of a 5 us launch per reduction).  Same order of additions as colreduce_finish: wave ry sums rows
// ry, ry + 4, ..., then (w0 + w1) + (w2 + w3).  All 256 threads call it; the sums of column cx are returned to every thread.
__device__ __forceinline__ void colreduce_local(const double *__restrict__ part, int R, int C, int c, int cx, int ry,
                                                double *sa, double *sb, double &a_out, double &b_out)
{
    double a = 0.0, b = 0.0;
    if (c < C) {
        int r = ry;
        for (; r + 28 < R; r += 32) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2 *>(&part[((long long)(r + 4 * u) * C + c) * 2]);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
        }
        for (; r < R; r += 4) { a += part[((long long)r * C + c) * 2]; b += part[((long long)r * C + c) * 2 + 1]; }
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    a_out = (sa[cx] + sa[64 + cx]) + (sa[128 + cx] + sa[192 + cx]);
    b_out = (sb[cx] + sb[64 + cx]) + (sb[128 + cx] + sb[192 + cx]);
    __syncthreads();
}

// Pass 1: every workgroup reduces its rows of a 64-column block to float64 partials.
// (Round 3 measured both passes fused into ONE launch -- the last workgroup of a column block to arrive, by a device-scope
// ticket, finishing the sum: the release / acquire fences it needs are an L2 write-back + invalidate on an 8-XCD part, and the
// fused kernel took 26.5 us where the two launches take 8 + 5.5 -- the whole training step got slower by a third of its kernel
// time.  Two launches it stays; inside a captured hipGraph the second launch costs about a microsecond.)
__device__ __forceinline__ void colreduce_partial_body(const float *__restrict__ P, const float *__restrict__ w,
                                                       const float *__restrict__ Q, const float *__restrict__ qm,
                                                       const float *__restrict__ qs, int N, int C,
                                                       int rows_per_block, double *__restrict__ part,
                                                       const float *__restrict__ w2, int bx, int by)
{
    // w2 (with Q null): out_b[c] = sum_n P[n][c] * w2[n] -- the two attention-vector gradients of a layer read the same
    // matrix with two weight vectors: one pass instead of two.  (bx, by): column block and row block of this workgroup
    __shared__ double sa[256], sb[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = bx * 64 + cx;
    const int n0 = by * rows_per_block, n1 = min(N, n0 + rows_per_block);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float m = qm ? qm[c] : 0.0f, s = qs ? qs[c] : 1.0f;
        auto step = [&](float p, float wv, float qv) {
            a += (double)(w ? p * wv : p);
            if (Q) {
                if (qm) qv = (qv - m) * s;
                b += (double)p * (double)qv;
            } else if (w2) {
                b += (double)(p * qv);                          // qv carries w2[n] here
            }
        };
        // 8 rows' loads go out before the first add (one row per round trip was an 8 us latency chain for 16 rows per
        // thread); the additions keep their order
        int n = n0 + ry;
        for (; n + 28 < n1; n += 32) {
            float pv[8], wv[8], qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pv[u] = P[(long long)(n + 4 * u) * C + c];
                wv[u] = w ? w[n + 4 * u] : 1.0f;
                qv[u] = Q ? Q[(long long)(n + 4 * u) * C + c] : (w2 ? w2[n + 4 * u] : 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) step(pv[u], wv[u], qv[u]);
        }
        for (; n < n1; n += 4) step(P[(long long)n * C + c], w ? w[n] : 1.0f, Q ? Q[(long long)n * C + c] : (w2 ? w2[n] : 0.0f));
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (ry == 0 && c < C) {
        a = sa[cx] + sa[64 + cx] + sa[128 + cx] + sa[192 + cx];
        b = sb[cx] + sb[64 + cx] + sb[128 + cx] + sb[192 + cx];
        part[((long long)by * C + c) * 2] = a;
        part[((long long)by * C + c) * 2 + 1] = b;
    }
}

__global__ __launch_bounds__(256) void colreduce_partial_kernel(const float *__restrict__ P, const float *__restrict__ w,
                                                                const float *__restrict__ Q, const float *__restrict__ qm,
                                                                const float *__restrict__ qs, int N, int C,
                                                                int rows_per_block, double *__restrict__ part,
                                                                const float *__restrict__ w2)
{
    colreduce_partial_body(P, w, Q, qm, qs, N, C, rows_per_block, part, w2, blockIdx.x, blockIdx.y);
}

// The attention-vector gradients of ALL layers (datt_src = sum_j da_src[j] g_j, datt_dst likewise: two weighted column sums of
// the layer's saved G) at the end of the backward (backward_end_partials_kernel; round 4: nothing inside the backward reads
// them; every layer keeps its da_src / da_dst and its partials in buffers of its own).
struct ColPartJobs {
    const float *P[NSC_GAT_MAX_LAYERS], *w[NSC_GAT_MAX_LAYERS], *w2[NSC_GAT_MAX_LAYERS];
    double *part[NSC_GAT_MAX_LAYERS];
};

// A second, one-component sum that rides along a final pass (round 4): out[c] (+)= sum_r part[r][c], float64 partials of
// ANOTHER matrix written by an earlier kernel of the stream (the conv / input bias gradient, whose pass 1 is fused into
// bn_apply_colsum_kernel: one launch and one read of the (N, H) matrix less per BatchNorm).
struct ColExtra {
    const double *part;          // (R, C) or null
    int R, acc;
    float *out;
};

// Pass 2: the R partial rows of a 64-column block, summed in fixed order (deterministic), then the finish (ColFinal).
__device__ __forceinline__ void colreduce_final_body(const double *__restrict__ part, int R, int C, const ColFinal &f,
                                                     const ColExtra &x, int bx)
{
    __shared__ double sa[256], sb[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = bx * 64 + cx;
    if (part) colreduce_finish(part, R, C, c, cx, ry, sa, sb, f);
    if (x.part) {
        __syncthreads();                                   // (sa is reused; the threads that left colreduce_finish early are back)
        double a = 0.0;
        if (c < C)
            for (int r = ry; r < x.R; r += 4) a += x.part[(long long)r * C + c];
        sa[threadIdx.x] = a;
        __syncthreads();
        if (ry == 0 && c < C) {
            a = (sa[cx] + sa[64 + cx]) + (sa[128 + cx] + sa[192 + cx]);
            x.out[c] = x.acc ? x.out[c] + (float)a : (float)a;
        }
    }
}

__global__ __launch_bounds__(256) void colreduce_final_kernel(const double *__restrict__ part, int R, int C, ColFinal f,
                                                              ColExtra x)
{
    colreduce_final_body(part, R, C, f, x, blockIdx.x);
}

// Every final pass a backward has left for its end (backward_end_finals_kernel): the attention-vector gradients of the layers
// with the conv-bias finals riding along, the output bias, the input bias.
constexpr int COL_JOBS = NSC_GAT_MAX_LAYERS + 2;
struct ColFinalJobs {
    const double *part[COL_JOBS];
    int R[COL_JOBS], C[COL_JOBS];
    ColFinal f[COL_JOBS];
    ColExtra x[COL_JOBS];
    int n;
};

// BatchNorm backward, pass 1, with the activation backward fused in (round 4: bn_act_bwd_dv_kernel + colreduce_partial_kernel
// were two launches and two passes over the (N, C) matrix):  dV = dH * dropmask * relu'(v) is computed on the fly (v from z
// and the batch statistics, as the forward did), STORED for the apply pass, and reduced:
//   s1[c] = sum_n dV[n][c],  s2[c] = sum_n dV[n][c] * xhat[n][c],  xhat = (z - mean) * invstd
// Same expressions, same order of additions as the two kernels it replaces.
__global__ __launch_bounds__(256) void bn_bwd_colsum_kernel(const float *__restrict__ dh, const float *__restrict__ z,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            int relu, float p, SeedRef seed, unsigned stream, int N, int C,
                                                            int rows_per_block, float *__restrict__ dv,
                                                            double *__restrict__ part)
{
    __shared__ double sa[256], sb[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float m = mean[c], s = invstd[c], ga = gamma[c], be = beta[c];
        const unsigned long long sd = p > 0.0f ? seed.get() : 0ull;
        auto step = [&](int n, float dhv, float zv) {
            const long long i = (long long)n * C + c;
            float g = dhv * keep_scale(p, sd, stream, (unsigned long long)i);
            if (relu) {
                const float v = (zv - m) * s * ga + be;
                if (!(v > 0.0f)) g = 0.0f;
            }
            dv[i] = g;
            a += (double)g;
            b += (double)g * (double)((zv - m) * s);
        };
        int n = n0 + ry;
        for (; n + 28 < n1; n += 32) {
            float dv8[8], zv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dv8[u] = dh[(long long)(n + 4 * u) * C + c];
                zv8[u] = z[(long long)(n + 4 * u) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) step(n + 4 * u, dv8[u], zv8[u]);
        }
        for (; n < n1; n += 4) step(n, dh[(long long)n * C + c], z[(long long)n * C + c]);
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (ry == 0 && c < C) {
        a = sa[cx] + sa[64 + cx] + sa[128 + cx] + sa[192 + cx];
        b = sb[cx] + sb[64 + cx] + sb[128 + cx] + sb[192 + cx];
        part[((long long)blockIdx.y * C + c) * 2] = a;
        part[((long long)blockIdx.y * C + c) * 2 + 1] = b;
    }
}

// BatchNorm backward, apply pass, with pass 1 of the bias gradient fused in (round 4: bn_bwd_apply_kernel +
// colreduce_partial_kernel):  dZ = gamma * invstd * (dV - s1 / N - xhat * s2 / N)  in place, and the float64 partials of
// sum_n dZ[n][c] (the gradient of the bias in front of this BatchNorm; a final pass adds them up: ColExtra).
__global__ __launch_bounds__(256) void bn_apply_colsum_kernel(float *__restrict__ dv, const float *__restrict__ z,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma, const double *__restrict__ part_in,
                                                              int R, float *__restrict__ g_bn_b, float *__restrict__ g_bn_w,
                                                              int acc, int N, int C, int rows_per_block,
                                                              double *__restrict__ part)
{
    __shared__ double sa[256], sb[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
    // s1 = sum dV, s2 = sum dV xhat: the finish of bn_bwd_colsum_kernel's partials (their float values, as the separate final
    // pass stored them); the first row block also stores them as the BatchNorm parameter gradients
    double d1, d2;
    colreduce_local(part_in, R, C, c, cx, ry, sa, sb, d1, d2);
    const float t1 = (float)d1, t2 = (float)d2;
    if (blockIdx.y == 0 && ry == 0 && c < C) {
        g_bn_b[c] = acc ? g_bn_b[c] + t1 : t1;
        g_bn_w[c] = acc ? g_bn_w[c] + t2 : t2;
    }
    double a = 0.0;
    if (c < C) {
        const float m = mean[c], s = invstd[c], ga = gamma[c];
        const float invn = 1.0f / (float)N;
        auto step = [&](int n, float dvv, float zv) {
            const float xhat = (zv - m) * s;
            const float dz = ga * s * (dvv - t1 * invn - xhat * t2 * invn);
            dv[(long long)n * C + c] = dz;
            a += (double)dz;
        };
        int n = n0 + ry;
        for (; n + 28 < n1; n += 32) {
            float dv8[8], zv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dv8[u] = dv[(long long)(n + 4 * u) * C + c];
                zv8[u] = z[(long long)(n + 4 * u) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) step(n + 4 * u, dv8[u], zv8[u]);
        }
        for (; n < n1; n += 4) step(n, dv[(long long)n * C + c], z[(long long)n * C + c]);
    }
    sa[threadIdx.x] = a;
    __syncthreads();
    if (ry == 0 && c < C) part[(long long)blockIdx.y * C + c] = sa[cx] + sa[64 + cx] + sa[128 + cx] + sa[192 + cx];
}

// ---------------------------------------------------------------------------------------------
// BatchNorm(train) apply + ReLU + dropout + residual:  h = drop(relu(gamma*xhat + beta)) + resid
// ---------------------------------------------------------------------------------------------
// (round 4: column-blocked, and it finishes the batch statistics of the reduction before it itself -- mean = A / N,
// var = B / N - mean^2, invstd = 1 / sqrt(var + eps) in float64 from the float64 partials, exactly colreduce_finish's mode 1;
// the first row block stores mean / invstd for the backward and updates the running statistics)
__global__ __launch_bounds__(256) void bn_act_kernel(const float *__restrict__ z, const double *__restrict__ part_in, int R,
                                                     float eps, float momentum, float *__restrict__ mean_out,
                                                     float *__restrict__ invstd_out, float *__restrict__ run_mean,
                                                     float *__restrict__ run_var, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, int relu, float p,
                                                     SeedRef seed, unsigned stream, const float *__restrict__ resid, int N, int C,
                                                     int rows_per_block, float *__restrict__ out)
{
    __shared__ double sa[256], sb[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
    double a, b;
    colreduce_local(part_in, R, C, c, cx, ry, sa, sb, a, b);
    if (c >= C) return;
    const double dmean = a / N;
    double var = b / N - dmean * dmean;
    if (var < 0.0) var = 0.0;
    const float m = (float)dmean, s = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.y == 0 && ry == 0) {
        mean_out[c] = m;
        invstd_out[c] = s;
        if (run_mean) {                                   // nn.BatchNorm1d: momentum 0.1, unbiased running_var
            const double unb = N > 1 ? var * N / (N - 1) : var;
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)dmean;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
        }
    }
    const float ga = gamma[c], be = beta[c];
    const unsigned long long sd = p > 0.0f ? seed.get() : 0ull;
    auto step = [&](int n, float zv, float rv) {
        const long long i = (long long)n * C + c;
        float v = (zv - m) * s * ga + be;
        if (relu) v = fmaxf(v, 0.0f);
        v *= keep_scale(p, sd, stream, (unsigned long long)i);
        if (resid) v += rv;
        out[i] = v;
    };
    int n = n0 + ry;
    for (; n + 28 < n1; n += 32) {
        float z8[8], r8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            z8[u] = z[(long long)(n + 4 * u) * C + c];
            r8[u] = resid ? resid[(long long)(n + 4 * u) * C + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) step(n + 4 * u, z8[u], r8[u]);
    }
    for (; n < n1; n += 4) step(n, z[(long long)n * C + c], resid ? resid[(long long)n * C + c] : 0.0f);
}

// Zero fill / copy as KERNELS (round 4).  The training step is replayed as a captured hipGraph, and a hipMemsetAsync captured into
// a graph becomes a memset NODE: in replays on ROCm 7.2 the node that zeroes the triplet gradient was observed not to be ordered
// before the kernel that accumulates into it -- the embedding gradient then starts from whatever the graph's pool holds (sums of
// 1e25-1e32 in the parameter gradients, differently in every run; Adam's normalisation hid it from every +-lr parameter check:
// tests/test_a_multirank_gpu.py::test_data_parallel_captured_step_replays_under_a_process_group and
// test_captured_step_gradients_match_eager now compare the gradients themselves).  Kernel nodes keep their stream order.
__global__ __launch_bounds__(256) void fill_zero_kernel(float *__restrict__ p, long long n)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<f32x4 *>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    else
        for (long long j = i; j < n; ++j) p[j] = 0.0f;
}
__global__ __launch_bounds__(256) void copy_kernel(float *__restrict__ dst, const float *__restrict__ src, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
inline void fill_zero(hipStream_t st, float *p, long long n)
{
    if (n > 0) hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, p, n);
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ a, const float *__restrict__ b,
                                                          long long total)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) a[i] += b[i];
}

// ---------------------------------------------------------------------------------------------
// attention (training): logits from the true definition a_src = <g, att_src>, a_dst = <g, att_dst>
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sumf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_maxf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// a_src[i] = <g_i, att_src>, a_dst[i] = <g_i, att_dst>; one wave per node
// (round 4: the last workgroup of the grid computes v = W_edge^T att_edge instead -- edge_vec_kernel's body, which was a launch
// of its own per layer and batch)
__global__ __launch_bounds__(256) void att_dots_kernel(const float *__restrict__ G, const float *__restrict__ att_s,
                                                       const float *__restrict__ att_d, int N, int H,
                                                       float *__restrict__ a_src, float *__restrict__ a_dst,
                                                       const float *__restrict__ w_edge, const float *__restrict__ att_edge,
                                                       int edge_dim, float *__restrict__ v)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w_edge && blockIdx.x == gridDim.x - 1) {
        // one wavefront per component d: lane-strided partial dot products, then the xor-shuffle tree
        for (int d = threadIdx.x >> 6; d < edge_dim; d += 4) {
            float s = 0.0f;
            for (int c = lane; c < H; c += 64) s = __builtin_fmaf(w_edge[(long long)c * edge_dim + d], att_edge[c], s);
            s = wave_sumf(s);
            if (lane == 0) v[d] = s;
        }
        return;
    }
    if (i >= N) return;
    float s = 0.f, d = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float g = G[(long long)i * H + c];
        s = __builtin_fmaf(g, att_s[c], s);
        d = __builtin_fmaf(g, att_d[c], d);
    }
    s = wave_sumf(s); d = wave_sumf(d);
    if (lane == 0) { a_src[i] = s; a_dst[i] = d; }
}

struct TrainAgg {
    const int *row_ptr, *src, *eid;
    const float *loop_attr, *edge_attr, *v;   // v = W_edge^T att_edge (edge_dim), nullable
    const float *a_src, *a_dst, *G, *bias;
    float *alpha;                             // (nnz) softmax weights BEFORE dropout (saved)
    float *y;                                 // (N,H) aggregate + bias (pre-BatchNorm)
    float slope, p;
    SeedRef seed;
    unsigned stream;
    int N, H, edge_dim;
};

__device__ __forceinline__ float edge_raw(const TrainAgg &a, int i, int e, int &j)
{
    j = a.src[e];
    float l = a.a_src[j] + a.a_dst[i];
    if (a.edge_attr && a.v) {
        const int id = a.eid[e];
        const float *ea = id >= 0 ? a.edge_attr + (long long)id * a.edge_dim : a.loop_attr + (long long)i * a.edge_dim;
        float t = 0.0f;
        for (int d = 0; d < a.edge_dim; ++d) t = __builtin_fmaf(ea[d], a.v[d], t);
        l += t;
    }
    return l;
}

// CH = ceil(H / 256).  Round 3: in-degree <= 64 (every graph of the path) takes one logit per lane -- evaluated once, the first
// form evaluated it three times -- and gathers 8 neighbour rows at a time as float4; per column the same terms in the same
// order as the general form below.
template <int CH>
__global__ __launch_bounds__(256) void agg_train_kernel(TrainAgg a)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.N) return;
    const int beg = a.row_ptr[i], end = a.row_ptr[i + 1];
    if (end - beg <= 64) {
        const int deg = end - beg, e = beg + lane;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        int j = i;
        float l = -INFINITY;
        if (e < end) { l = edge_raw(a, i, e, j); l = l > 0.f ? l : a.slope * l; }
        const float m = wave_maxf(l);
        const float pe = (e < end) ? expf(l - m) : 0.0f;
        const float s = wave_sumf(pe) + 1e-16f;                      // PyG softmax
        float ald = 0.0f;
        if (e < end) {
            const float al = pe / s;
            a.alpha[e] = al;                                          // saved for the backward (pre-dropout)
            ald = al * keep_scale(a.p, a.seed.get(), a.stream, (unsigned long long)e);   // attention dropout
        }
        f32x4 acc[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = zero;
        for (int t0 = 0; t0 < deg; t0 += 8) {
            f32x4 g[8][CH];
            float at[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                at[u] = __shfl(ald, (t0 + u) & 63);
                const int jt = __shfl(j, (t0 + u) & 63);
                const float *row = a.G + (long long)((t0 + u < deg) ? jt : i) * a.H;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = 4 * lane + 256 * k;
                    g[u][k] = (c < a.H) ? *reinterpret_cast<const f32x4 *>(row + c) : zero;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u >= deg) break;                             // wave-uniform
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    acc[k].x = __builtin_fmaf(at[u], g[u][k].x, acc[k].x);
                    acc[k].y = __builtin_fmaf(at[u], g[u][k].y, acc[k].y);
                    acc[k].z = __builtin_fmaf(at[u], g[u][k].z, acc[k].z);
                    acc[k].w = __builtin_fmaf(at[u], g[u][k].w, acc[k].w);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = 4 * lane + 256 * k;
            if (c < a.H) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + c);
                *reinterpret_cast<f32x4 *>(a.y + (long long)i * a.H + c) =
                    f32x4{acc[k].x + b.x, acc[k].y + b.y, acc[k].z + b.z, acc[k].w + b.w};
            }
        }
        return;
    }
    float m = -INFINITY;
    for (int e = beg + lane; e < end; e += 64) { int j; float l = edge_raw(a, i, e, j); l = l > 0.f ? l : a.slope * l; m = fmaxf(m, l); }
    m = wave_maxf(m);
    float s = 0.0f;
    for (int e = beg + lane; e < end; e += 64) { int j; float l = edge_raw(a, i, e, j); l = l > 0.f ? l : a.slope * l; s += expf(l - m); }
    s = wave_sumf(s) + 1e-16f;                                   // PyG softmax
    // H <= 1024: up to 16 columns per lane; softmax weights travel between lanes by shuffle only
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    for (int c0 = beg; c0 < end; c0 += 64) {
        const int e = c0 + lane;
        int j = 0;
        float al = 0.0f, ald = 0.0f;
        if (e < end) {
            float l = edge_raw(a, i, e, j); l = l > 0.f ? l : a.slope * l;
            al = expf(l - m) / s;
            a.alpha[e] = al;                                      // saved for the backward (pre-dropout)
            ald = al * keep_scale(a.p, a.seed.get(), a.stream, (unsigned long long)e);   // attention dropout
        }
        const int cnt = min(64, end - c0);
        for (int t = 0; t < cnt; ++t) {
            const float at = __shfl(ald, t);
            const int jt = __shfl(j, t);
            const float *g = a.G + (long long)jt * a.H;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = lane + 64 * k;
                if (c < a.H) acc[k] = __builtin_fmaf(at, g[c], acc[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = lane + 64 * k;
        if (c < a.H) a.y[(long long)i * a.H + c] = acc[k] + a.bias[c];
    }
}

// backward, per target i: dalpha'_e = <dY_i, g_j>; softmax + leaky-relu backward -> draw[e]; da_dst[i]
struct AttBwdA {
    const int *row_ptr, *src, *eid;
    const float *loop_attr, *edge_attr, *v;
    const float *a_src, *a_dst, *G, *alpha, *dY;
    float *draw;        // (nnz) gradient of the pre-leaky-relu logit
    float *da_dst;      // (N)
    float slope, p;
    SeedRef seed;
    unsigned stream;
    int N, H, edge_dim;
};

// CH = ceil(H / 256): float4 chunks per lane.  Round 3: the common case (in-degree <= 64) keeps dY_i in registers, fetches
// 8 neighbour rows of G at a time as float4 (all in flight before the first reduction) and computes every <dY_i, g_j> ONCE
// (lane e keeps entry e's value); the first form walked the entries twice, one dependent round trip chain per entry
// (17.5 us at 4 541 keyframes).
template <int CH>
__global__ __launch_bounds__(256) void att_bwd_target_kernel(AttBwdA a)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.N) return;
    const int beg = a.row_ptr[i], end = a.row_ptr[i + 1], deg = end - beg;
    TrainAgg t;   // view for edge_raw()
    t.src = a.src; t.eid = a.eid; t.a_src = a.a_src; t.a_dst = a.a_dst; t.edge_attr = a.edge_attr;
    t.loop_attr = a.loop_attr; t.v = a.v; t.edge_dim = a.edge_dim;
    if (deg <= 64) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 dy[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = 4 * lane + 256 * k;
            dy[k] = (c < a.H) ? *reinterpret_cast<const f32x4 *>(a.dY + (long long)i * a.H + c) : zero;
        }
        const int e_l = beg + lane;
        const bool have = e_l < end;
        const int j_l = have ? a.src[e_l] : 0;
        const float al_l = have ? a.alpha[e_l] : 0.0f;
        const unsigned long long seed = a.seed.get();
        float da = 0.0f;                                          // <dY_i, g_j> of entry `lane`, through the dropout mask
        for (int t0 = 0; t0 < deg; t0 += 8) {
            f32x4 g[8][CH];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jt = __shfl(j_l, (t0 + u) & 63);
                const float *row = a.G + (long long)((t0 + u < deg) ? jt : i) * a.H;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = 4 * lane + 256 * k;
                    g[u][k] = (c < a.H) ? *reinterpret_cast<const f32x4 *>(row + c) : zero;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u >= deg) break;                         // wave-uniform
                float d = 0.0f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    d = __builtin_fmaf(dy[k].x, g[u][k].x, d);
                    d = __builtin_fmaf(dy[k].y, g[u][k].y, d);
                    d = __builtin_fmaf(dy[k].z, g[u][k].z, d);
                    d = __builtin_fmaf(dy[k].w, g[u][k].w, d);
                }
                const float sum = wave_sumf(d) * keep_scale(a.p, seed, a.stream, (unsigned long long)(beg + t0 + u));
                if (lane == t0 + u) da = sum;
            }
        }
        float inner = 0.0f;                                       // sum_e alpha_e dalpha_e, in entry order
        for (int tt = 0; tt < deg; ++tt) inner += __shfl(al_l, tt) * __shfl(da, tt);
        float dr = 0.0f;
        if (have) {
            int j;
            const float raw = edge_raw(t, i, e_l, j);
            const float dl = al_l * (da - inner);                 // softmax backward
            dr = raw > 0.0f ? dl : a.slope * dl;                  // leaky-relu backward
            a.draw[e_l] = dr;
        }
        float dd = 0.0f;
        for (int tt = 0; tt < deg; ++tt) dd += __shfl(dr, tt);
        if (lane == 0) a.da_dst[i] = dd;
        return;
    }
    auto dalpha = [&](int e) -> float {                       // <dY_i, g_j> through the dropout mask
        const int j = a.src[e];
        float d = 0.0f;
        for (int c = lane; c < a.H; c += 64)
            d = __builtin_fmaf(a.dY[(long long)i * a.H + c], a.G[(long long)j * a.H + c], d);
        return wave_sumf(d) * keep_scale(a.p, a.seed.get(), a.stream, (unsigned long long)e);
    };
    float inner = 0.0f;                                       // sum_e alpha_e dalpha_e (wave-uniform)
    for (int e = beg; e < end; ++e) inner += a.alpha[e] * dalpha(e);
    float dd = 0.0f;
    for (int e = beg; e < end; ++e) {
        int j;
        const float raw = edge_raw(t, i, e, j);
        const float dl = a.alpha[e] * (dalpha(e) - inner);    // softmax backward
        const float dr = raw > 0.0f ? dl : a.slope * dl;      // leaky-relu backward
        if (lane == 0) a.draw[e] = dr;
        dd += dr;
    }
    if (lane == 0) a.da_dst[i] = dd;
}

// backward, per source j (transposed CSR): dG_j = sum_e alpha'_e dY_tgt(e) + da_src[j] att_src + da_dst[j] att_dst
struct AttBwdB {
    const int *t_ptr, *t_entry, *tgt;
    const float *alpha, *dY, *draw, *da_dst, *att_src, *att_dst;
    float *dG;          // (N,H)
    float *da_src;      // (N)
    float p;
    SeedRef seed;
    unsigned stream;
    int N, H;
};

// CH = ceil(H / 256).  Round 3: a lane fetches ONE entry of the source's list (its edge index, draw, alpha through the
// dropout mask, target), the values travel by shuffle, and the dY rows come 8 at a time as float4; every column still adds its
// terms in list order (same bits as the first form, which re-read the list once per 64 columns, a dependent chain per entry).
template <int CH>
__global__ __launch_bounds__(256) void att_bwd_source_kernel(AttBwdB a)
{
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= a.N) return;
    const int beg = a.t_ptr[j], end = a.t_ptr[j + 1];
    const unsigned long long seed = a.seed.get();
    float das = 0.0f;
    for (int c0 = beg; c0 < end; c0 += 64) {                      // da_src[j] = sum of draw over the list, in list order
        const int tl = c0 + lane;
        const float dl = (tl < end) ? a.draw[a.t_entry[tl]] : 0.0f;
        const int cnt = min(64, end - c0);
        for (int tt = 0; tt < cnt; ++tt) das += __shfl(dl, tt);
    }
    const float dad = a.da_dst[j];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = 4 * lane + 256 * k;
        acc[k] = zero;
        if (c < a.H) {
            const f32x4 s1 = *reinterpret_cast<const f32x4 *>(a.att_src + c), s2 = *reinterpret_cast<const f32x4 *>(a.att_dst + c);
            acc[k].x = das * s1.x + dad * s2.x; acc[k].y = das * s1.y + dad * s2.y;
            acc[k].z = das * s1.z + dad * s2.z; acc[k].w = das * s1.w + dad * s2.w;
        }
    }
    for (int c0 = beg; c0 < end; c0 += 64) {
        const int tl = c0 + lane;
        float al_l = 0.0f;
        int tg_l = j;
        if (tl < end) {
            const int e = a.t_entry[tl];
            al_l = a.alpha[e] * keep_scale(a.p, seed, a.stream, (unsigned long long)e);
            tg_l = a.tgt[e];
        }
        const int cnt = min(64, end - c0);
        for (int t0 = 0; t0 < cnt; t0 += 8) {
            f32x4 r[8][CH];
            float al[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                al[u] = __shfl(al_l, (t0 + u) & 63);
                const int tg = __shfl(tg_l, (t0 + u) & 63);
                const float *row = a.dY + (long long)((t0 + u < cnt) ? tg : j) * a.H;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = 4 * lane + 256 * k;
                    r[u][k] = (c < a.H) ? *reinterpret_cast<const f32x4 *>(row + c) : zero;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u >= cnt) break;                         // wave-uniform
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    acc[k].x = __builtin_fmaf(al[u], r[u][k].x, acc[k].x);
                    acc[k].y = __builtin_fmaf(al[u], r[u][k].y, acc[k].y);
                    acc[k].z = __builtin_fmaf(al[u], r[u][k].z, acc[k].z);
                    acc[k].w = __builtin_fmaf(al[u], r[u][k].w, acc[k].w);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = 4 * lane + 256 * k;
        if (c < a.H) *reinterpret_cast<f32x4 *>(a.dG + (long long)j * a.H + c) = acc[k];
    }
    if (lane == 0) a.da_src[j] = das;
}

// dv[d] = sum_e draw[e] * ea_e[d]  (edge term gradient): EDGE_BWD_WGS workgroups write float64 partials, a second
// tiny kernel adds them in workgroup order -> deterministic
constexpr int EDGE_BWD_WGS = 64;

struct EdgeTermArgs {
    const int *row_ptr, *eid, *tgt;
    const float *loop_attr, *edge_attr, *draw_all;
    long long draw_stride;                                          // floats between two layers' d logit / d raw buffers
    int N, edge_dim;
    double *part_all;
};
// workgroup bx of EDGE_BWD_WGS, layer l (round 4: the edge terms of ALL layers at the end of the backward; every layer keeps its
// d logit / d raw in a buffer of its own)
__device__ __forceinline__ void edge_term_bwd_body(const EdgeTermArgs &a, int bx, int l)
{
    const int *__restrict__ row_ptr = a.row_ptr, *__restrict__ eid = a.eid, *__restrict__ tgt = a.tgt;
    const float *__restrict__ loop_attr = a.loop_attr, *__restrict__ edge_attr = a.edge_attr;
    const int N = a.N, edge_dim = a.edge_dim;
    const float *__restrict__ draw = a.draw_all + (long long)l * a.draw_stride;
    double *__restrict__ part = a.part_all + (long long)l * EDGE_BWD_WGS * NSC_GAT_MAX_EDGE_DIM;
    __shared__ double sh[256];
    const int nnz = row_ptr[N];
    double s[NSC_GAT_MAX_EDGE_DIM];
#pragma unroll
    for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d) s[d] = 0.0;
    for (int e = bx * 256 + threadIdx.x; e < nnz; e += EDGE_BWD_WGS * 256) {
        const int id = eid[e];
        const float *ea = id >= 0 ? edge_attr + (long long)id * edge_dim : loop_attr + (long long)tgt[e] * edge_dim;
        const double dr = (double)draw[e];
#pragma unroll
        for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d)
            if (d < edge_dim) s[d] += dr * (double)ea[d];
    }
#pragma unroll
    for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d) {
        if (d >= edge_dim) break;
        sh[threadIdx.x] = s[d];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) part[bx * NSC_GAT_MAX_EDGE_DIM + d] = sh[0];
        __syncthreads();
    }
}

// (one workgroup per layer)  dv[d] = sum of the workgroups' partials in a
// fixed shuffle tree (deterministic), then the backward of v = W_edge^T att_edge: dW_edge[c,d] = dv[d] att_edge[c],
// datt_edge[c] = sum_d dv[d] W_edge[c,d]
struct EdgeVecJobs {                                                // per layer (blockIdx.x): the edge projection and its gradient buffers
    const float *w_edge[NSC_GAT_MAX_LAYERS], *att_edge[NSC_GAT_MAX_LAYERS];
    float *dw_edge[NSC_GAT_MAX_LAYERS], *datt_edge[NSC_GAT_MAX_LAYERS];
};
__device__ __forceinline__ void edge_vec_bwd_body(const double *__restrict__ part_all, int nparts, const EdgeVecJobs &jobs,
                                                  int H, int edge_dim, int accumulate, int l)
{
    const double *__restrict__ part = part_all + (long long)l * EDGE_BWD_WGS * NSC_GAT_MAX_EDGE_DIM;
    const float *__restrict__ w_edge = jobs.w_edge[l], *__restrict__ att_edge = jobs.att_edge[l];
    float *__restrict__ dw_edge = jobs.dw_edge[l], *__restrict__ datt_edge = jobs.datt_edge[l];
    if (!dw_edge || !datt_edge) return;                             // (a layer without an edge projection)
    __shared__ float dv[NSC_GAT_MAX_EDGE_DIM];
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;                        // lane g holds partial g (nparts <= 64)
        for (int d = 0; d < edge_dim; ++d) {
            double s = lane < nparts ? part[lane * NSC_GAT_MAX_EDGE_DIM + d] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) dv[d] = (float)s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0.0f;
        for (int d = 0; d < edge_dim; ++d) {
            const float gw = dv[d] * att_edge[c];
            dw_edge[(long long)c * edge_dim + d] = accumulate ? dw_edge[(long long)c * edge_dim + d] + gw : gw;
            s = __builtin_fmaf(dv[d], w_edge[(long long)c * edge_dim + d], s);
        }
        datt_edge[c] = accumulate ? datt_edge[c] + s : s;
    }
}

// The end of a backward in TWO launches (round 4): what nothing inside the backward reads -- the attention-vector gradients of
// all layers, the edge-term gradients of all layers, every column sum's final pass, the slab sums of the five weight gradients
// -- used to be 4 + 2 + 2 launches per layer and five more; they are independent pieces of work of the same workgroup shape, so
// they share a grid: a workgroup finds its piece by its index range.  Every piece computes exactly what its own kernel did.
struct EndPartials {
    ColPartJobs att;                                                // workgroups [0, n_att): (column block, row block, layer)
    int N, C, rows, nbx, R;
    unsigned n_att;
    const float *ob_P;                                              // then the output bias: column sums of dOut (N, ob_C),
    double *ob_part;                                                //   workgroups (column block, row block)
    int ob_C, ob_nbx;
    unsigned n_ob;
    EdgeTermArgs edge;                                              // then EDGE_BWD_WGS per layer
};
__global__ __launch_bounds__(256) void backward_end_partials_kernel(EndPartials p)
{
    unsigned b = blockIdx.x;
    if (b < p.n_att) {
        const int bx = (int)(b % (unsigned)p.nbx), t = (int)(b / (unsigned)p.nbx), by = t % p.R, l = t / p.R;
        colreduce_partial_body(p.att.P[l], p.att.w[l], nullptr, nullptr, nullptr, p.N, p.C, p.rows, p.att.part[l], p.att.w2[l], bx, by);
    } else if (b < p.n_att + p.n_ob) {
        b -= p.n_att;
        colreduce_partial_body(p.ob_P, nullptr, nullptr, nullptr, nullptr, p.N, p.ob_C, p.rows, p.ob_part, nullptr,
                               (int)(b % (unsigned)p.ob_nbx), (int)(b / (unsigned)p.ob_nbx));
    } else {
        b -= p.n_att + p.n_ob;
        edge_term_bwd_body(p.edge, (int)(b % EDGE_BWD_WGS), (int)(b / EDGE_BWD_WGS));
    }
}

struct EndFinals {
    ColFinalJobs fin;                                               // workgroups [0, n_fin): (column block, job)
    int nbx;
    unsigned n_fin, n_edge;                                         // then one workgroup per layer with an edge projection,
    EdgeVecJobs ev;                                                 // then the slab sums
    const double *edge_part;
    int H, edge_dim, accumulate;
    SlabBatch slabs;
};
__global__ __launch_bounds__(256) void backward_end_finals_kernel(EndFinals p)
{
    unsigned b = blockIdx.x;
    if (b < p.n_fin) {
        const int bx = (int)(b % (unsigned)p.nbx), l = (int)(b / (unsigned)p.nbx);
        if (bx * 64 >= p.fin.C[l]) return;                          // (workgroup-uniform: ahead of every barrier)
        colreduce_final_body(p.fin.part[l], p.fin.R[l], p.fin.C[l], p.fin.f[l], p.fin.x[l], bx);
    } else if (b < p.n_fin + p.n_edge) {
        edge_vec_bwd_body(p.edge_part, EDGE_BWD_WGS, p.ev, p.H, p.edge_dim, p.accumulate, (int)(b - p.n_fin));
    } else {
        slab_reduce_multi_body(p.slabs, b - p.n_fin - p.n_edge);
    }
}

// transposed CSR (entries grouped by source) from the forward CSR
__global__ __launch_bounds__(256) void tcsr_count_kernel(const int *__restrict__ row_ptr, const int *__restrict__ src,
                                                         int N, int *__restrict__ cnt, int *__restrict__ tgt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) { atomicAdd(&cnt[src[e]], 1); tgt[e] = i; }
}
__global__ __launch_bounds__(1024) void tcsr_scan_kernel(const int *__restrict__ cnt, int N, int *__restrict__ t_ptr)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x, chunk = (N + 1023) / 1024;
    const int b = tid * chunk, e = min(b + chunk, N);
    int s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = b; i < e; ++i) { t_ptr[i] = run; run += cnt[i]; }
    if (tid == 1023) t_ptr[N] = part[1023];
}
__global__ __launch_bounds__(256) void tcsr_fill_kernel(const int *__restrict__ row_ptr, const int *__restrict__ src,
                                                        int N, const int *__restrict__ t_ptr, int *__restrict__ cursor,
                                                        int *__restrict__ t_entry)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
        const int j = src[e];
        t_entry[t_ptr[j] + atomicAdd(&cursor[j], 1)] = e;
    }
}
__global__ __launch_bounds__(256) void tcsr_sort_kernel(const int *__restrict__ t_ptr, int N, int *__restrict__ t_entry)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int b = t_ptr[j], e = t_ptr[j + 1];
    for (int a = b + 1; a < e; ++a) {
        const int v = t_entry[a];
        int c = a - 1;
        while (c >= b && t_entry[c] > v) { t_entry[c + 1] = t_entry[c]; --c; }
        t_entry[c + 1] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// triplet loss forward + backward (trainer.py:62-68): one wave per triplet
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void triplet_kernel(const float *__restrict__ emb, const long long *__restrict__ ia,
                                                      const long long *__restrict__ ip, const long long *__restrict__ in_,
                                                      int T, int N, int D, float margin, float scale,
                                                      float *__restrict__ per_triplet, float *__restrict__ grad)
{
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    // embeddings[idx] semantics (trainer.py:207-209): negative indices wrap once; anything still outside [0, N)
    // would raise IndexError in the reference -- here the triplet touches no memory and poisons the loss with NaN
    long long a_i = ia[t], p_i = ip[t], n_i = in_[t];
    a_i += (a_i < 0) ? N : 0; p_i += (p_i < 0) ? N : 0; n_i += (n_i < 0) ? N : 0;
    if (a_i < 0 || a_i >= N || p_i < 0 || p_i >= N || n_i < 0 || n_i >= N) {
        if (lane == 0) per_triplet[t] = NAN;
        return;
    }
    const float *a = emb + a_i * D, *p = emb + p_i * D, *n = emb + n_i * D;
    float dp = 0.f, dn = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float x = a[c] - p[c], y = a[c] - n[c];
        dp = __builtin_fmaf(x, x, dp);
        dn = __builtin_fmaf(y, y, dn);
    }
    dp = wave_sumf(dp); dn = wave_sumf(dn);
    const float l = dp - dn + margin;
    if (lane == 0) per_triplet[t] = l > 0.0f ? l : 0.0f;
    if (grad && l > 0.0f) {
        const float g = 2.0f * scale / (float)T;              // d(mean relu)/d(dist) * 2(a - .)
        float *ga = grad + a_i * D, *gp = grad + p_i * D, *gn = grad + n_i * D;
        for (int c = lane; c < D; c += 64) {
            const float av = a[c], pv = p[c], nv = n[c];
            atomicAdd(&ga[c], g * (nv - pv));
            atomicAdd(&gp[c], -g * (av - pv));
            atomicAdd(&gn[c], g * (av - nv));
        }
    }
}
__global__ __launch_bounds__(256) void triplet_reduce_kernel(const float *__restrict__ per_triplet, int T, float scale,
                                                             float *__restrict__ loss)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) s += (double)per_triplet[t];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / T) * scale;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
constexpr int SPLITK_SLABS = 16;
constexpr int COLRED_MAXR = 64;

struct TrainWs {
    // saved by the forward
    size_t z0, mean0, invstd0, h, g, a_src, a_dst, alpha, y, mean, invstd, vvec;
    // backward scratch
    size_t dh, dh2, dv, dg, draw, edgepart, da_src, da_dst, s1, s2, dvvec, slabs, slab_cap, colpart, colpart2, colpart2_stride, attpart, attpart_stride, colpart_ob, total;
    size_t nh, nn, hh, nz;
};

// floats of ONE slab of the largest weight-gradient product: input / output projection, a layer's H x H lin weight (round 4: this
// term was missing -- a model whose hidden width exceeds both its input and output widths wrote its lin slabs past the region;
// every configuration of the path has 800 > 256), residual_proj
inline size_t slab_region0_floats(const NscGatModel *m)
{
    const size_t H = (size_t)m->hidden;
    size_t big = std::max((size_t)std::max(m->in_dim, m->out_dim) * H, H * H);
    if (m->residual && m->in_dim != m->out_dim) big = std::max(big, (size_t)m->in_dim * m->out_dim);   // dW of residual_proj
    return big;
}

TrainWs train_ws(const NscGatModel *m, int N, int nnz)
{
    TrainWs w;
    const int H = m->hidden, L = m->n_layers;
    w.nh = align256((size_t)N * H * 4); w.nn = align256((size_t)N * 4); w.hh = align256((size_t)H * 4);
    w.nz = align256((size_t)nnz * 4);
    size_t o = 0;
    w.z0 = o; o += w.nh;
    w.mean0 = o; o += w.hh;
    w.invstd0 = o; o += w.hh;
    w.h = o; o += w.nh * (L + 1);
    w.g = o; o += w.nh * L;
    w.a_src = o; o += w.nn * L;
    w.a_dst = o; o += w.nn * L;
    w.alpha = o; o += w.nz * L;
    w.y = o; o += w.nh * L;
    w.mean = o; o += w.hh * L;
    w.invstd = o; o += w.hh * L;
    w.vvec = o; o += 256 * L;
    w.dh = o; o += w.nh;
    w.dh2 = o; o += w.nh;
    w.dv = o; o += w.nh;
    w.dg = o; o += w.nh;
    w.draw = o; o += w.nz * L;                                      // per layer: the edge terms of all layers are reduced in one launch at the end
    w.edgepart = o; o += align256((size_t)NSC_GAT_MAX_LAYERS * EDGE_BWD_WGS * NSC_GAT_MAX_EDGE_DIM * 8);
    w.da_src = o; o += w.nn * L;                                    // per layer: reduced against G at the end of the backward
    w.da_dst = o; o += w.nn * L;
    w.s1 = o; o += align256((size_t)std::max(H, m->out_dim) * 4);
    w.s2 = o; o += align256((size_t)std::max(H, m->out_dim) * 4);
    w.dvvec = o; o += 256;
    const size_t big = slab_region0_floats(m);
    w.slabs = o; o += align256(big * 4 * SPLITK_SLABS);            // region 0: a product that sums its slabs at once, transposed weights
    // ... and a region per weight-gradient product of a backward whose sums wait for the backward's last launch (backward_end_finals_kernel)
    size_t all_w = (size_t)m->in_dim * H + (size_t)m->out_dim * H + (size_t)L * H * H;
    if (m->residual && m->in_dim != m->out_dim) all_w += (size_t)m->in_dim * m->out_dim;
    w.slab_cap = all_w * SPLITK_SLABS + 64 * SLAB_JOBS;            // floats
    o += align256(w.slab_cap * 4);
    w.colpart = o; o += align256((size_t)COLRED_MAXR * std::max(std::max(H, m->out_dim), m->in_dim) * 2 * 8);
    w.colpart2_stride = align256((size_t)COLRED_MAXR * H * 8);      // partials of a bias gradient (bn_apply_colsum_kernel): one slot
    w.colpart2 = o; o += w.colpart2_stride * (L + 1);               // per BatchNorm, their finals run together at the end
    w.attpart_stride = align256((size_t)COLRED_MAXR * H * 2 * 8);   // partials of a layer's two attention-vector gradients
    w.attpart = o; o += w.attpart_stride * L;
    w.colpart_ob = o; o += align256((size_t)COLRED_MAXR * m->out_dim * 2 * 8);   // partials of the output bias gradient
    w.total = o;
    return w;
}

// out[c * rows + r] = in[r * ld + c]: the (K, N) weight of a dX = dY W product as the (N, K) operand the NT GEMM takes
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int rows, int cols, int ld,
                                                        float *__restrict__ out)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) t[ty + 8 * i][tx] = in[(long long)r * ld + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) out[(long long)c * rows + r] = t[tx][ty + 8 * i];
    }
}

// slabs: split-K slabs of the weight-gradient products; for a single-slice product with a k-major B (dX = dY W) the same
// buffer, free between two weight-gradient products of the stream, takes the transposed weight.
template <bool AKM, bool BKM>
void gemm(hipStream_t st, const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C, int ldc,
          const float *bias, int accumulate, int splits, float *slabs, SlabDefer *defer = nullptr, const float *resid = nullptr)
{
    // resid (non-accumulating single-slice products only, same leading dimension as C): C = product + resid, in the epilogue of
    // the LDS-DMA GEMM where that kernel takes the product (the backward's dh_{l} = dG W + dh_{l+1} residual path), by an
    // add_inplace_kernel behind any other form -- the same two roundings either way
    if (!AKM && splits <= 1 && (!BKM || (slabs && !(K & 15) && !(reinterpret_cast<unsigned long long>(slabs) & 15)))) {
        // the projections of the training forward, and the dX = dY W products of the backward through a transposed copy of
        // the weight (0.8 MB at most: a 3 us kernel): the inference forward's LDS-DMA GEMM -- same chain per output element
        // as gemm_gen_kernel, same (acc + bias) + C order; anything it cannot take (unaligned operands) falls through
        const float *Bn = B;
        int ldn = ldb;
        if (BKM) {
            // round 4: the k-major weight read element-wise from an untransposed LDS tile -- no transposed copy (a 5 us kernel per
            // product); anything that form cannot take goes through the copy as before
            GemmEpi e0 = {};
            e0.bias = bias;
            if (accumulate) { e0.resid = C; e0.ldr = ldc; }
            else if (resid) { e0.resid = resid; e0.ldr = ldc; }
            if (launch_glds_bkm(st, A, lda, B, ldb, M, N, K, C, ldc, e0)) return;
        }
        if (BKM) {
            hipLaunchKernelGGL(transpose_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, st, B, K, N, ldb, slabs);
            Bn = slabs;
            ldn = K;
        }
        GemmEpi ep = {};
        ep.bias = bias;
        if (accumulate) { ep.resid = C; ep.ldr = ldc; }
        else if (resid) { ep.resid = resid; ep.ldr = ldc; }
        if (launch_glds<2>(st, A, lda, Bn, ldn, nullptr, M, N, N, K, C, ldc, ep)) return;
    }
    if (AKM && BKM && splits > 1 && !bias && ldc == N && launch_tn_glds(st, A, lda, B, ldb, M, N, K, C, accumulate, slabs, SPLITK_SLABS, defer))
        return;
    dim3 grid((N + 63) / 64, (M + 31) / 32, splits);
    if (splits <= 1) {
        hipLaunchKernelGGL((gemm_gen_kernel<AKM, BKM>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, K, K, C, ldc,
                           0LL, bias, accumulate);
    } else {
        int kchunk = (K + splits - 1) / splits;
        kchunk = (kchunk + 63) / 64 * 64;
        const long long MN = (long long)M * N;      // requires ldc == N
        float *own = slab_defer_take(defer, MN, splits);
        hipLaunchKernelGGL((gemm_gen_kernel<AKM, BKM>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, K, kchunk, own ? own : slabs,
                           N, MN, static_cast<const float *>(nullptr), 0);
        if (own) slab_defer_push(defer, own, C, MN, splits, accumulate);
        else hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, slabs, splits, MN, C, accumulate);
    }
    if (resid && !accumulate && ldc == N)
        hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, C, resid, (long long)M * N);
}

inline int colred_rows(int N)
{
    int R = (N + 63) / 64;
    if (R > COLRED_MAXR) R = COLRED_MAXR;
    return R < 1 ? 1 : R;
}

// Forward BatchNorm (batch statistics) + activation in two launches (round 3: three): pass 1 of the statistics, then the apply
// kernel, which finishes them itself (bn_act_kernel).
void bn_forward(hipStream_t st, const float *z, int N, int C, double *part, float eps, float momentum, float *mean, float *invstd,
                float *run_mean, float *run_var, const float *gamma, const float *beta, int relu, float p, SeedRef seed,
                unsigned stream, const float *resid, float *out);

void colreduce(hipStream_t st, const float *P, const float *w, const float *Q, const float *qm, const float *qs,
               int N, int C, double *part, int mode, float eps, float momentum, float *out_a,
               float *out_b, float *run_mean, float *run_var, int acc_a = 0, int acc_b = 0, float *copy_a = nullptr,
               float *copy_b = nullptr, int acc_copy = 0, const float *w2 = nullptr, ColExtra extra = ColExtra{nullptr, 0, 0, nullptr})
{
    const int R = colred_rows(N);
    const int rows = (N + R - 1) / R;
    const ColFinal f = {mode, N, eps, momentum, out_a, out_b, run_mean, run_var, acc_a, acc_b, copy_a, copy_b, acc_copy};
    hipLaunchKernelGGL(colreduce_partial_kernel, dim3((C + 63) / 64, R), dim3(256), 0, st, P, w, Q, qm, qs, N, C, rows, part, w2);
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, part, R, C, f, extra);
}

void bn_forward(hipStream_t st, const float *z, int N, int C, double *part, float eps, float momentum, float *mean, float *invstd,
                float *run_mean, float *run_var, const float *gamma, const float *beta, int relu, float p, SeedRef seed,
                unsigned stream, const float *resid, float *out)
{
    const int R = colred_rows(N);
    const int rows = (N + R - 1) / R;
    const dim3 grid((C + 63) / 64, R);
    hipLaunchKernelGGL(colreduce_partial_kernel, grid, dim3(256), 0, st, z, static_cast<const float *>(nullptr), z,
                       static_cast<const float *>(nullptr), static_cast<const float *>(nullptr), N, C, rows, part,
                       static_cast<const float *>(nullptr));
    hipLaunchKernelGGL(bn_act_kernel, grid, dim3(256), 0, st, z, part, R, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta,
                       relu, p, seed, stream, resid, N, C, rows, out);
}

// BatchNorm backward of one layer in TWO launches (round 3: five): dV + the partials of its two column sums; then dZ in place (the
// kernel finishes those sums itself and stores the BatchNorm parameter gradients) + the partials of ITS column sum, returned as
// the ColExtra the caller hands to its next final pass (or to bias_final).
ColExtra bn_backward(hipStream_t st, const float *dh, const float *z, const float *mean, const float *invstd, const float *gamma,
                     const float *beta, int relu, float p, SeedRef seed, unsigned stream, int N, int C, float *dv, float *s1,
                     float *s2, float *g_bn_b, float *g_bn_w, int acc, double *part, double *part2, float *g_bias)
{
    const int R = colred_rows(N);
    const int rows = (N + R - 1) / R;
    const dim3 grid((C + 63) / 64, R);
    hipLaunchKernelGGL(bn_bwd_colsum_kernel, grid, dim3(256), 0, st, dh, z, mean, invstd, gamma, beta, relu, p, seed, stream, N, C,
                       rows, dv, part);
    // (round 4, second step: no final launch -- the apply pass finishes s1 / s2 itself and stores the BatchNorm parameter gradients)
    (void)s1; (void)s2;
    hipLaunchKernelGGL(bn_apply_colsum_kernel, grid, dim3(256), 0, st, dv, z, mean, invstd, gamma, part, R, g_bn_b, g_bn_w, acc, N, C,
                       rows, part2);
    return ColExtra{part2, R, acc, g_bias};
}

void bias_final(hipStream_t st, int C, const ColExtra &x)
{
    const ColFinal none = {0, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, static_cast<const double *>(nullptr), 0, C, none, x);
}

inline unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }

int check_train(const NscGatModel *m, const NscGraph *g)
{
    if (!m || !g) return NSC_EINVAL;
    if (m->n_layers < 1 || m->n_layers > NSC_GAT_MAX_LAYERS) return NSC_EUNSUPPORTED;
    if (m->hidden < 16 || m->hidden > 1024 || (m->hidden & 15)) return NSC_EUNSUPPORTED;
    if (m->in_dim < 16 || (m->in_dim & 15) || m->out_dim < 4 || (m->out_dim & 3)) return NSC_EUNSUPPORTED;
    if (m->edge_dim < 0 || m->edge_dim > NSC_GAT_MAX_EDGE_DIM) return NSC_EUNSUPPORTED;
    if (m->residual && m->in_dim != m->out_dim && (!m->res_w || !m->res_b)) return NSC_EINVAL;   // model.py:91-94
    if (!g->row_ptr || !g->src || !g->eid) return NSC_EINVAL;
    return NSC_OK;
}

}  // namespace

extern "C" {

size_t nsc_graph_transpose_workspace_bytes(int32_t n_nodes) { return n_nodes > 0 ? align256((size_t)n_nodes * 4) * 2 : 0; }

int nsc_graph_transpose(const NscGraph *g, int32_t *t_ptr, int32_t *t_entry, int32_t *tgt, void *ws, size_t ws_bytes,
                        void *stream_)
{
    if (!g || !t_ptr || !t_entry || !tgt || !g->row_ptr || !g->src) return NSC_EINVAL;
    const int N = g->n_nodes;
    if (N <= 0) return NSC_OK;
    const size_t need = nsc_graph_transpose_workspace_bytes(N);
    if (!ws || ws_bytes < need) return NSC_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    int *cnt = static_cast<int *>(ws);
    int *cursor = reinterpret_cast<int *>(static_cast<char *>(ws) + need / 2);
    nsc_fill_u32(st, ws, 0u, (long long)(need / 4));
    hipLaunchKernelGGL(tcsr_count_kernel, dim3(blocks(N)), dim3(256), 0, st, g->row_ptr, g->src, N, cnt, tgt);
    hipLaunchKernelGGL(tcsr_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, N, t_ptr);
    hipLaunchKernelGGL(tcsr_fill_kernel, dim3(blocks(N)), dim3(256), 0, st, g->row_ptr, g->src, N, t_ptr, cursor, t_entry);
    hipLaunchKernelGGL(tcsr_sort_kernel, dim3(blocks(N)), dim3(256), 0, st, t_ptr, N, t_entry);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_gat_train_workspace_bytes(const NscGatModel *m, const NscGraph *g)
{
    if (check_train(m, g) != NSC_OK || g->n_nodes <= 0) return 0;
    return train_ws(m, g->n_nodes, g->nnz).total;
}

int nsc_gat_forward_train(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                          const NscGatTrainCfg *cfg, float *out, void *ws, size_t ws_bytes, void *stream_)
{
    int stt = check_train(m, g);
    if (stt != NSC_OK) return stt;
    if (!cfg || !x || !out) return NSC_EINVAL;
    const int N = g->n_nodes, H = m->hidden, L = m->n_layers;
    if (N == 0) return NSC_OK;
    const TrainWs w = train_ws(m, N, g->nnz);
    if (!ws || ws_bytes < w.total) return NSC_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    char *b = static_cast<char *>(ws);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(b + off); };
    double *colpart = reinterpret_cast<double *>(b + w.colpart);
    const bool use_edge = m->edge_dim > 0 && edge_attr && g->loop_attr;
    const int upd = cfg->update_running_stats;

    // input_proj (bias) -> z0 ; BatchNorm(batch stats) ; ReLU                 model.py:116-118
    gemm<false, false>(st, x, m->in_dim, m->in_w, m->in_dim, N, H, m->in_dim, F(w.z0), H, m->in_b, 0, 1, nullptr);
    bn_forward(st, F(w.z0), N, H, colpart, m->bn_eps, cfg->bn_momentum, F(w.mean0), F(w.invstd0),
               upd ? const_cast<float *>(m->in_bn_mean) : nullptr, upd ? const_cast<float *>(m->in_bn_var) : nullptr, m->in_bn_w,
               m->in_bn_b, 1, 0.0f, SeedRef{0ull, nullptr}, 0u, nullptr, F(w.h));

    for (int l = 0; l < L; ++l) {
        const NscGatLayer &Ly = m->layers[l];
        float *hin = F(w.h + w.nh * l), *hout = F(w.h + w.nh * (l + 1));
        float *G = F(w.g + w.nh * l), *as = F(w.a_src + w.nn * l), *ad = F(w.a_dst + w.nn * l);
        float *alpha = F(w.alpha + w.nz * l), *y = F(w.y + w.nh * l);
        float *mean = F(w.mean + w.hh * l), *invstd = F(w.invstd + w.hh * l), *vv = F(w.vvec + 256 * l);
        gemm<false, false>(st, hin, H, Ly.lin_w, H, N, H, H, G, H, nullptr, 0, 1, nullptr);
        hipLaunchKernelGGL(att_dots_kernel, dim3((N + 3) / 4 + (use_edge ? 1 : 0)), dim3(256), 0, st, G, Ly.att_src, Ly.att_dst, N, H, as, ad,
                           use_edge ? Ly.lin_edge_w : nullptr, Ly.att_edge, m->edge_dim, vv);
        TrainAgg a;
        a.row_ptr = g->row_ptr; a.src = g->src; a.eid = g->eid;
        a.loop_attr = use_edge ? g->loop_attr : nullptr;
        a.edge_attr = use_edge ? edge_attr : nullptr;
        a.v = use_edge ? vv : nullptr;
        a.a_src = as; a.a_dst = ad; a.G = G; a.bias = Ly.bias; a.alpha = alpha; a.y = y;
        a.slope = m->negative_slope; a.p = cfg->dropout_p; a.seed = SeedRef{cfg->seed, reinterpret_cast<const unsigned long long *>(cfg->seed_dev)}; a.stream = 100u + l;
        a.N = N; a.H = H; a.edge_dim = m->edge_dim;
        switch ((H + 255) / 256) {
        case 1: hipLaunchKernelGGL(agg_train_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(agg_train_kernel<2>, dim3((N + 3) / 4), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(agg_train_kernel<3>, dim3((N + 3) / 4), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(agg_train_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, st, a); break;
        }
        const int act = (l < L - 1);                                           // model.py:135-137
        const float *resid = (m->residual && l > 0 && l < L - 1) ? hin : nullptr;   // model.py:140-141
        bn_forward(st, y, N, H, colpart, m->bn_eps, cfg->bn_momentum, mean, invstd, upd ? const_cast<float *>(Ly.bn_mean) : nullptr,
                   upd ? const_cast<float *>(Ly.bn_var) : nullptr, Ly.bn_w, Ly.bn_b, act, act ? cfg->dropout_p : 0.0f,
                   SeedRef{cfg->seed, reinterpret_cast<const unsigned long long *>(cfg->seed_dev)}, 200u + l, resid, hout);
    }
    // output_proj + input residual                                             model.py:144-151
    bool fused_res = false;
    if (m->residual && m->in_dim == m->out_dim) {
        // (acc + bias) + x in the GEMM's epilogue: the order of the GEMM followed by out += x, one launch and one pass over the
        // (N, 800) output less (round 4)
        GemmEpi ep = {};
        ep.bias = m->out_b;
        ep.resid = x; ep.ldr = m->in_dim;
        fused_res = launch_glds<2>(st, F(w.h + w.nh * L), H, m->out_w, H, nullptr, N, m->out_dim, m->out_dim, H, out, m->out_dim, ep);
    }
    if (!fused_res)
        gemm<false, false>(st, F(w.h + w.nh * L), H, m->out_w, H, N, m->out_dim, H, out, m->out_dim, m->out_b, 0, 1, nullptr);
    if (m->residual && m->in_dim == m->out_dim) {
        const long long tot = (long long)N * m->out_dim;
        if (!fused_res) hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks(tot)), dim3(256), 0, st, out, x, tot);
    } else if (m->residual) {                                                  // out += residual_proj(x)  model.py:147-149
        gemm<false, false>(st, x, m->in_dim, m->res_w, m->in_dim, N, m->out_dim, m->in_dim, out, m->out_dim, m->res_b, 1, 1,
                           nullptr);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_gat_backward(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                     const NscGatTrainCfg *cfg, const float *grad_out, const NscGatGrads *gr, void *ws,
                     size_t ws_bytes, void *stream_)
{
    int stt = check_train(m, g);
    if (stt != NSC_OK) return stt;
    if (!cfg || !x || !grad_out || !gr || !g->t_ptr || !g->t_entry || !g->tgt) return NSC_EINVAL;
    const int N = g->n_nodes, H = m->hidden, L = m->n_layers, Dout = m->out_dim, Din = m->in_dim;
    if (N == 0) return NSC_OK;
    const TrainWs w = train_ws(m, N, g->nnz);
    if (!ws || ws_bytes < w.total) return NSC_EWORKSPACE;
    if (m->residual && Din != Dout && (!gr->res_w || !gr->res_b)) return NSC_EINVAL;   // (before anything is enqueued)
    hipStream_t st = static_cast<hipStream_t>(stream_);
    char *b = static_cast<char *>(ws);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(b + off); };
    double *colpart = reinterpret_cast<double *>(b + w.colpart);
    auto colpart2 = [&](int slot) { return reinterpret_cast<double *>(b + w.colpart2 + w.colpart2_stride * slot); };
    float *slabs = F(w.slabs);
    // what nothing inside the backward reads is reduced at its END, in batched launches: the final passes of every column sum
    // (in backward_end_finals_kernel), the attention-vector sums of all layers (in backward_end_partials_kernel), the edge terms,
    // the slab sums of the weight gradients
    ColFinalJobs fin = {};
    ColPartJobs attp = {};
    const int Rn = colred_rows(N), rows_n = (N + Rn - 1) / Rn;
    auto push_final = [&](const double *part, int C, const ColFinal &f, const ColExtra &x) {
        const int k = fin.n++;
        fin.part[k] = part; fin.R[k] = Rn; fin.C[k] = C; fin.f[k] = f; fin.x[k] = x;
    };
    const ColExtra no_extra = {nullptr, 0, 0, nullptr};
    SlabDefer defer_ = {};
    {
        defer_.base = slabs + align256(slab_region0_floats(m) * 4 * SPLITK_SLABS) / 4;
        defer_.cap = w.slab_cap;
    }
    SlabDefer *defer = &defer_;
    EdgeVecJobs edge_jobs = {};
    bool any_edge = false;
    const bool use_edge = m->edge_dim > 0 && edge_attr && g->loop_attr;
    const int splits = N >= 512 ? SPLITK_SLABS : 1;
    // NscGatTrainCfg.accumulate_grads: every PARAMETER gradient is added to what its buffer holds (gradient accumulation
    // over the batches of an optimizer step without a pass of axpy kernels behind the backward); gr->x is always overwritten
    const int acc = cfg->accumulate_grads ? 1 : 0;

    // output_proj: out = h_L W_out^T + b (+ x)
    // (its column sums of dOut: with the other partial sums in the backward's second-to-last launch -- dOut is the caller's, nothing
    // overwrites it)
    double *ob = reinterpret_cast<double *>(b + w.colpart_ob);
    push_final(ob, Dout, ColFinal{0, N, 0.f, 0.f, gr->out_b, nullptr, nullptr, nullptr, acc, 0, nullptr, nullptr, 0}, no_extra);
    gemm<true, true>(st, grad_out, Dout, F(w.h + w.nh * L), H, Dout, H, N, gr->out_w, H, nullptr, acc, splits, slabs, defer);
    float *dh = F(w.dh), *dh_prev = F(w.dh2);
    gemm<false, true>(st, grad_out, Dout, m->out_w, H, N, H, Dout, dh, H, nullptr, 0, 1, slabs);   // dh_L = dOut W_out
    const bool res_id = m->residual && Din == Dout, res_proj = m->residual && Din != Dout;
    if (res_proj) {                // residual_proj: dW_res = dOut^T x, db_res = colsum dOut      model.py:147-149
        colreduce(st, grad_out, nullptr, nullptr, nullptr, nullptr, N, Dout, colpart, 0, 0.f, 0.f, gr->res_b, nullptr, nullptr, nullptr, acc);
        gemm<true, true>(st, grad_out, Dout, x, Din, Dout, Din, N, gr->res_w, Din, nullptr, acc, splits, slabs, defer);
    }
    if (gr->x) {
        // gradient wrt the input features through the residual connection: dOut itself (identity residual),
        // dOut W_res (residual_proj) or nothing (residual=False); dZ0 W_in is accumulated at the end
        if (res_id) {
            hipLaunchKernelGGL(copy_kernel, dim3(blocks((long long)N * Din)), dim3(256), 0, st, gr->x, grad_out, (long long)N * Din);
        } else if (res_proj) {
            gemm<false, true>(st, grad_out, Dout, m->res_w, Din, N, Din, Dout, gr->x, Din, nullptr, 0, 1, slabs);
        } else {
            fill_zero(st, gr->x, (long long)N * Din);
        }
    }

    for (int l = L - 1; l >= 0; --l) {
        const NscGatLayer &Ly = m->layers[l];
        const NscGatGradLayer &Gl = gr->layers[l];
        float *hin = F(w.h + w.nh * l);
        float *G = F(w.g + w.nh * l), *as = F(w.a_src + w.nn * l), *ad = F(w.a_dst + w.nn * l);
        float *alpha = F(w.alpha + w.nz * l), *y = F(w.y + w.nh * l);
        float *mean = F(w.mean + w.hh * l), *invstd = F(w.invstd + w.hh * l), *vv = F(w.vvec + 256 * l);
        float *dv = F(w.dv), *dG = F(w.dg), *s1 = F(w.s1), *s2 = F(w.s2);
        const int act = (l < L - 1);
        const bool has_res = (m->residual && l > 0 && l < L - 1);
        // h_{l+1} = drop(relu(bn(y))) [+ h_l]  ->  dV, BatchNorm backward -> dY (in place in dv).  s1 / s2 feed the apply
        // pass AND are the BatchNorm parameter gradients (the final stores both); the conv bias gradient = column sums of dY:
        // its partials come out of the apply pass, its final rides along the attention-vector reduction below
        const ColExtra bias_x = bn_backward(st, dh, y, mean, invstd, Ly.bn_w, Ly.bn_b, act, act ? cfg->dropout_p : 0.0f,
                                            SeedRef{cfg->seed, reinterpret_cast<const unsigned long long *>(cfg->seed_dev)}, 200u + l,
                                            N, H, dv, s1, s2, Gl.bn_b, Gl.bn_w, acc, colpart, colpart2(l), Gl.bias);
        float *dY = dv;
        // attention backward
        AttBwdA A;
        A.row_ptr = g->row_ptr; A.src = g->src; A.eid = g->eid;
        A.loop_attr = use_edge ? g->loop_attr : nullptr; A.edge_attr = use_edge ? edge_attr : nullptr;
        A.v = use_edge ? vv : nullptr;
        A.a_src = as; A.a_dst = ad; A.G = G; A.alpha = alpha; A.dY = dY;
        A.draw = F(w.draw + w.nz * l); A.da_dst = F(w.da_dst + w.nn * l);
        A.slope = m->negative_slope; A.p = cfg->dropout_p; A.seed = SeedRef{cfg->seed, reinterpret_cast<const unsigned long long *>(cfg->seed_dev)}; A.stream = 100u + l;
        A.N = N; A.H = H; A.edge_dim = m->edge_dim;
        switch ((H + 255) / 256) {
        case 1: hipLaunchKernelGGL(att_bwd_target_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, st, A); break;
        case 2: hipLaunchKernelGGL(att_bwd_target_kernel<2>, dim3((N + 3) / 4), dim3(256), 0, st, A); break;
        case 3: hipLaunchKernelGGL(att_bwd_target_kernel<3>, dim3((N + 3) / 4), dim3(256), 0, st, A); break;
        default: hipLaunchKernelGGL(att_bwd_target_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, st, A); break;
        }
        AttBwdB Bk;
        Bk.t_ptr = g->t_ptr; Bk.t_entry = g->t_entry; Bk.tgt = g->tgt;
        Bk.alpha = alpha; Bk.dY = dY; Bk.draw = F(w.draw + w.nz * l); Bk.da_dst = F(w.da_dst + w.nn * l);
        Bk.att_src = Ly.att_src; Bk.att_dst = Ly.att_dst; Bk.dG = dG; Bk.da_src = F(w.da_src + w.nn * l);
        Bk.p = cfg->dropout_p; Bk.seed = SeedRef{cfg->seed, reinterpret_cast<const unsigned long long *>(cfg->seed_dev)}; Bk.stream = 100u + l; Bk.N = N; Bk.H = H;
        switch ((H + 255) / 256) {
        case 1: hipLaunchKernelGGL(att_bwd_source_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, st, Bk); break;
        case 2: hipLaunchKernelGGL(att_bwd_source_kernel<2>, dim3((N + 3) / 4), dim3(256), 0, st, Bk); break;
        case 3: hipLaunchKernelGGL(att_bwd_source_kernel<3>, dim3((N + 3) / 4), dim3(256), 0, st, Bk); break;
        default: hipLaunchKernelGGL(att_bwd_source_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, st, Bk); break;
        }
        // datt_src = sum_j da_src[j] g_j ; datt_dst = sum_j da_dst[j] g_j
        // (at the end of the backward, with the other layers': the conv-bias final rides along this layer's final)
        {
            double *ap = reinterpret_cast<double *>(b + w.attpart + w.attpart_stride * l);
            attp.P[l] = G; attp.w[l] = F(w.da_src + w.nn * l); attp.w2[l] = F(w.da_dst + w.nn * l); attp.part[l] = ap;
            push_final(ap, H, ColFinal{0, N, 0.f, 0.f, Gl.att_src, Gl.att_dst, nullptr, nullptr, acc, acc, nullptr, nullptr, 0}, bias_x);
        }
        if (m->edge_dim > 0 && Gl.lin_edge_w && Gl.att_edge) {
            if (use_edge) {                         // reduced with the other layers' at the end (backward_end_partials_kernel)
                edge_jobs.w_edge[l] = Ly.lin_edge_w; edge_jobs.att_edge[l] = Ly.att_edge;
                edge_jobs.dw_edge[l] = Gl.lin_edge_w; edge_jobs.datt_edge[l] = Gl.att_edge;
                any_edge = true;
            } else if (!acc) {                      // no edge term in this forward: zero gradient (nothing to add when accumulating)
                fill_zero(st, Gl.lin_edge_w, (long long)H * m->edge_dim);
                fill_zero(st, Gl.att_edge, H);
            }
        }
        // g = h_l W^T :  dW = dG^T h_l ,  dh_l = dG W (+ residual path)
        gemm<true, true>(st, dG, H, hin, H, H, H, N, Gl.lin_w, H, nullptr, acc, splits, slabs, defer);
        gemm<false, true>(st, dG, H, Ly.lin_w, H, N, H, H, dh_prev, H, nullptr, 0, 1, slabs, nullptr, has_res ? dh : nullptr);
        float *t = dh; dh = dh_prev; dh_prev = t;
    }
    // h_0 = relu(bn(z0)),  z0 = x W_in^T + b_in
    float *dv = F(w.dv), *s1 = F(w.s1), *s2 = F(w.s2);
    push_final(nullptr, H, ColFinal{0, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0},
               bn_backward(st, dh, F(w.z0), F(w.mean0), F(w.invstd0), m->in_bn_w, m->in_bn_b, 1, 0.0f, SeedRef{0ull, nullptr}, 0u,
                           N, H, dv, s1, s2, gr->in_bn_b, gr->in_bn_w, acc, colpart, colpart2(L), gr->in_b));
    gemm<true, true>(st, dv, H, x, Din, H, Din, N, gr->in_w, Din, nullptr, acc, splits, slabs, defer);
    if (gr->x) {   // + dZ0 W_in
        gemm<false, true>(st, dv, H, m->in_w, Din, N, Din, H, gr->x, Din, nullptr, 1, 1, slabs);
    }
    {
        double *ep = reinterpret_cast<double *>(b + w.edgepart);
        EndPartials pa = {};
        pa.att = attp; pa.N = N; pa.C = H; pa.rows = rows_n; pa.nbx = (H + 63) / 64; pa.R = Rn;
        pa.n_att = (unsigned)(pa.nbx * Rn * L);
        pa.ob_P = grad_out; pa.ob_part = ob; pa.ob_C = Dout; pa.ob_nbx = (Dout + 63) / 64; pa.n_ob = (unsigned)(pa.ob_nbx * Rn);
        pa.edge = EdgeTermArgs{g->row_ptr, g->eid, g->tgt, g->loop_attr, edge_attr, F(w.draw), (long long)(w.nz / 4), N, m->edge_dim, ep};
        hipLaunchKernelGGL(backward_end_partials_kernel, dim3(pa.n_att + pa.n_ob + (any_edge ? (unsigned)(EDGE_BWD_WGS * L) : 0u)), dim3(256), 0, st, pa);
        EndFinals fa = {};
        fa.fin = fin; fa.nbx = (std::max(H, Dout) + 63) / 64;
        fa.n_fin = (unsigned)(fa.nbx * fin.n); fa.n_edge = any_edge ? (unsigned)L : 0u;
        fa.ev = edge_jobs; fa.edge_part = ep; fa.H = H; fa.edge_dim = m->edge_dim; fa.accumulate = acc;
        fa.slabs = defer->b;                                        // (the weight gradients' slab sums: after the last product above)
        hipLaunchKernelGGL(backward_end_finals_kernel, dim3(fa.n_fin + fa.n_edge + defer->b.blocks), dim3(256), 0, st, fa);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_triplet_workspace_bytes(int32_t n_triplets) { return n_triplets > 0 ? align256((size_t)n_triplets * 4) : 0; }

int nsc_triplet_loss(const float *emb, const int64_t *anchors, const int64_t *positives, const int64_t *negatives,
                     int32_t T, int32_t N, int32_t D, float margin, float scale, float *loss, float *grad_emb,
                     void *ws, size_t ws_bytes, void *stream_)
{
    if (T < 0 || N < 0 || D < 1 || (T > 0 && N == 0)) return NSC_EINVAL;
    if (!loss) return NSC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (grad_emb) fill_zero(st, grad_emb, (long long)N * D);
    if (T == 0) {
        fill_zero(st, loss, 1);
        return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
    }
    if (!emb || !anchors || !positives || !negatives) return NSC_EINVAL;
    if (!ws || ws_bytes < nsc_triplet_workspace_bytes(T)) return NSC_EWORKSPACE;
    float *per = static_cast<float *>(ws);
    hipLaunchKernelGGL(triplet_kernel, dim3((T + 3) / 4), dim3(256), 0, st, emb, reinterpret_cast<const long long *>(anchors),
                       reinterpret_cast<const long long *>(positives), reinterpret_cast<const long long *>(negatives), T, N, D,
                       margin, scale, per, grad_emb);
    hipLaunchKernelGGL(triplet_reduce_kernel, dim3(1), dim3(256), 0, st, per, T, scale, loss);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

}  // extern "C"
