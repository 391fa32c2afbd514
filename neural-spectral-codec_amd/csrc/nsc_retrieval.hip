// nsc_retrieval.hip -- stage-1 Wasserstein retrieval on gfx950 ("next" row 1).
//
// Path (reference file:line):
//   wasserstein_distance_batch_torch      src/retrieval/wasserstein.py:134-172  (query vs database)
//   wasserstein_distance_matrix_torch     src/retrieval/wasserstein.py:232-273  (many queries)
//   WassersteinRetriever.query            src/retrieval/wasserstein.py:328-384  (top-k smallest)
//   TwoStageRetrieval._global_retrieval   src/retrieval/two_stage_retrieval.py:145-202 (spatial filter)
//
// 1-D W1 between histograms = L1 distance of their CDFs.  One wavefront per database row: the row is
// streamed once from HBM (3 200 B for 800 bins), normalised and prefix-summed in registers (16 bins per
// lane + a wave scan of the lane totals), and compared with every query CDF of the batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nsc.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sumf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// inclusive CDF of one row held as PER consecutive bins per lane; returns the row sum
template <int PER>
__device__ __forceinline__ float row_cdf(float (&v)[PER], int lane, float eps, int divide_plain)
{
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) tot += v[k];
    const float sum = wave_sumf(tot);
    // wasserstein.py:153-163: query / sum (if sum > eps); database row / (sum + eps) (if sum > eps)
    float scale = 1.0f;
    if (sum > eps) scale = divide_plain ? sum : sum + eps;
    float run = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = (sum > eps) ? v[k] / scale : v[k]; run += v[k]; v[k] = run; }
    // exclusive scan of the lane totals (Hillis-Steele over 64 lanes)
    float inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const float off = inc - run;
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] += off;
    return sum;
}

template <int PER>
__device__ __forceinline__ void load_row(const float *__restrict__ row, int D, int lane, float (&v)[PER])
{
#pragma unroll
    for (int k = 0; k < PER; k += 4) {
        const int c = lane * PER + k;
        if (c + 3 < D) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(row + c);
            v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[k + j] = (c + j < D) ? row[c + j] : 0.0f;
        }
    }
}

// CDFs of the queries (wave per query), written to global for the distance kernel
template <int PER>
__global__ __launch_bounds__(256) void w1_cdf_kernel(const float *__restrict__ h, int n, int D, float eps,
                                                     int divide_plain, float *__restrict__ cdf)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float v[PER];
    load_row<PER>(h + (long long)i * D, D, lane, v);
    row_cdf<PER>(v, lane, eps, divide_plain);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane * PER + k;
        if (c < D) cdf[(long long)i * D + c] = v[k];
    }
}

// dist[q][i] = sum_k |cdf_db[i][k] - cdf_q[q][k]|, +inf where the spatial filter excludes the pair
template <int PER>
__global__ __launch_bounds__(256) void w1_dist_kernel(const float *__restrict__ db, int N, int D, float eps,
                                                      const float *__restrict__ qcdf, int Q,
                                                      const float *__restrict__ db_pos,
                                                      const float *__restrict__ q_pos, float min_dist,
                                                      float *__restrict__ dist)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    float v[PER];
    load_row<PER>(db + (long long)i * D, D, lane, v);
    row_cdf<PER>(v, lane, eps, 0);
    float px = 0.f, py = 0.f, pz = 0.f;
    if (db_pos) { px = db_pos[i * 3]; py = db_pos[i * 3 + 1]; pz = db_pos[i * 3 + 2]; }
    for (int q = 0; q < Q; ++q) {
        float w[PER];
        load_row<PER>(qcdf + (long long)q * D, D, lane, w);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane * PER + k;
            if (c < D) s += fabsf(v[k] - w[k]);
        }
        s = wave_sumf(s);
        if (db_pos && q_pos) {       // two_stage_retrieval.py:160-170: skip database frames that are too close
            const float dx = px - q_pos[q * 3], dy = py - q_pos[q * 3 + 1], dz = pz - q_pos[q * 3 + 2];
            if (sqrtf(dx * dx + dy * dy + dz * dz) < min_dist) s = INFINITY;
        }
        if (lane == 0) dist[(long long)q * N + i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Distances against a database of PRE-NORMALISED CDF rows (WassersteinRetriever keeps them next to the raw
// histograms: the database is persistent, its normalise + prefix-sum does not have to be redone per query).
//
// w1_stream_kernel<QT>: 1..4 queries.  HBM-bound: the query CDFs sit in registers, every wave walks its
//   share of the rows with the next row's loads in flight (3 200 B per row, read once).
// w1_tile_kernel: query batches.  VALU-bound (2 ops per |a - b| element): a 64-row x 64-query tile per
//   workgroup, 4 x 4 results per thread, k in chunks of 32 staged k-major through LDS -- the structure of a
//   register-tiled SGEMM with |a - b| in place of a * b.  Each (row, query) sum is accumulated by ONE thread in
//   ascending k, so it does not depend on the tiling.
// ---------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(256) void w1_stream_kernel(const float *__restrict__ dbc, int N, int D,
                                                        const float *__restrict__ qc, int Q,
                                                        const float *__restrict__ db_pos,
                                                        const float *__restrict__ q_pos, float min_dist,
                                                        float *__restrict__ dist)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int nvec = D >> 2;                                   // D % 4 == 0 (checked by the host)
    constexpr int NV = 4;                                      // float4 chunks per lane: D <= 1024
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 qv[QT][NV];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            qv[t][j] = (t < Q && c < nvec) ? reinterpret_cast<const f32x4 *>(qc + (long long)t * D)[c] : zero;
        }
    auto load = [&](int i, f32x4 (&v)[NV]) {
        const f32x4 *row = reinterpret_cast<const f32x4 *>(dbc + (long long)i * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j;
            v[j] = (c < nvec) ? __builtin_nontemporal_load(&row[c]) : zero;
        }
    };
    f32x4 cur[NV], nxt[NV];
    int i = wave;
    if (i < N) load(i, cur);
    for (; i < N; i += nwaves) {
        const int in = i + nwaves;
        if (in < N) load(in, nxt);
        float s[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float a = 0.0f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                a += fabsf(cur[j].x - qv[t][j].x);
                a += fabsf(cur[j].y - qv[t][j].y);
                a += fabsf(cur[j].z - qv[t][j].z);
                a += fabsf(cur[j].w - qv[t][j].w);
            }
            s[t] = wave_sumf(a);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                if (t >= Q) break;
                float r = s[t];
                if (db_pos && q_pos) {                          // two_stage_retrieval.py:160-170
                    const float dx = db_pos[i * 3] - q_pos[t * 3], dy = db_pos[i * 3 + 1] - q_pos[t * 3 + 1],
                                dz = db_pos[i * 3 + 2] - q_pos[t * 3 + 2];
                    if (sqrtf(dx * dx + dy * dy + dz * dz) < min_dist) r = INFINITY;
                }
                dist[(long long)t * N + i] = r;
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) cur[j] = nxt[j];
    }
}

constexpr int TL_I = 64, TL_K = 32, TL_LD = 68;               // LDS rows padded to 68 floats

template <int NQ>   // queries per thread: the tile is 64 rows x 16 NQ queries
__global__ __launch_bounds__(256) void w1_tile_kernel(const float *__restrict__ dbc, int N, int D,
                                                      const float *__restrict__ qc, int Q,
                                                      const float *__restrict__ db_pos,
                                                      const float *__restrict__ q_pos, float min_dist,
                                                      float *__restrict__ dist)
{
    __shared__ __attribute__((aligned(16))) float As[2][TL_K * TL_LD];   // [k][row]
    __shared__ __attribute__((aligned(16))) float Bs[2][TL_K * TL_LD];   // [k][query]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    constexpr int TL_Q = 16 * NQ;
    constexpr int HB = (TL_Q + 31) / 32;                            // staging passes for the query tile
    const int i0 = blockIdx.x * TL_I, q0 = blockIdx.y * TL_Q;
    // staging: thread -> (tile row sr / sr + 32, k offset 4 sk): 128 contiguous bytes per 8 threads
    const int sr = tid >> 3, sk = (tid & 7) * 4;
    const float *ga[2], *gb[HB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ri = i0 + sr + 32 * h;
        ga[h] = dbc + (long long)(ri < N ? ri : N - 1) * D + sk;
    }
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        const int rq = q0 + sr + 32 * h;
        gb[h] = qc + (long long)(rq < Q ? rq : Q - 1) * D + sk;
    }
    const int nchunks = (D + TL_K - 1) / TL_K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int ch, f32x4 (&ra)[2], f32x4 (&rb)[HB]) {
        const int k = ch * TL_K + sk;                                // past D: zeros on both sides add |0 - 0|
#pragma unroll
        for (int h = 0; h < 2; ++h)
            ra[h] = (k + 4 <= D) ? *reinterpret_cast<const f32x4 *>(ga[h] + ch * TL_K) : zero;
#pragma unroll
        for (int h = 0; h < HB; ++h)
            rb[h] = (k + 4 <= D && sr + 32 * h < TL_Q) ? *reinterpret_cast<const f32x4 *>(gb[h] + ch * TL_K) : zero;
    };
    auto stage = [&](int buf, const f32x4 (&ra)[2], const f32x4 (&rb)[HB]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][(sk + j) * TL_LD + sr + 32 * h] = ra[h][j];
#pragma unroll
        for (int h = 0; h < HB; ++h)
            if (sr + 32 * h < TL_Q)
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[buf][(sk + j) * TL_LD + sr + 32 * h] = rb[h][j];
    };
    float acc[4][NQ];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NQ; ++b) acc[a][b] = 0.0f;

    f32x4 ra[2], rb[HB];
    gload(0, ra, rb);
    stage(0, ra, rb);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) gload(ch + 1, ra, rb);
        const float *as = As[cur], *bs = Bs[cur];
#pragma unroll 8
        for (int k = 0; k < TL_K; ++k) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(&as[k * TL_LD + 4 * tx]);
            float bv[NQ];
#pragma unroll
            for (int b = 0; b < NQ; ++b) bv[b] = bs[k * TL_LD + NQ * ty + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NQ; ++b) acc[a][b] += fabsf(av[a] - bv[b]);
        }
        if (ch + 1 < nchunks) stage(cur ^ 1, ra, rb);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const int q = q0 + NQ * ty + b;
        if (q >= Q) continue;
        float r[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = i0 + 4 * tx + a;
            r[a] = acc[a][b];
            if (db_pos && q_pos && i < N) {
                const float dx = db_pos[i * 3] - q_pos[q * 3], dy = db_pos[i * 3 + 1] - q_pos[q * 3 + 1],
                            dz = db_pos[i * 3 + 2] - q_pos[q * 3 + 2];
                if (sqrtf(dx * dx + dy * dy + dz * dz) < min_dist) r[a] = INFINITY;
            }
        }
        float *o = dist + (long long)q * N + i0 + 4 * tx;
        if (i0 + 4 * tx + 3 < N && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0)) {
            *reinterpret_cast<f32x4 *>(o) = f32x4{r[0], r[1], r[2], r[3]};
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (i0 + 4 * tx + a < N) o[a] = r[a];
        }
    }
}

// k smallest of each row of dist, ascending, ties to the smaller index.  Two stages:
//   stage 1: grid (chunks, Q): a workgroup keeps its chunk of 2 048 distances in registers (8 per
//            thread) and extracts the chunk's k smallest by k rounds of (thread-local min, LDS reduce);
//   stage 2: one workgroup per query merges the chunks' candidates the same way.
// Picks are ordered by (value, index), so the result does not depend on the chunking.
constexpr int TK_CHUNK = 2048, TK_PER = 8;

__device__ __forceinline__ bool lex_less(float v, int i, float w, int j) { return v < w || (v == w && i < j); }

template <int PERT>
__device__ __forceinline__ void select_k(float (&v)[PERT], int (&id)[PERT], int k, float *sv, int *si,
                                         float *out_v, int *out_i)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < k; ++r) {
        float bv = INFINITY; int bi = 0x7fffffff, bs = -1;
#pragma unroll
        for (int u = 0; u < PERT; ++u)
            if (lex_less(v[u], id[u], bv, bi)) { bv = v[u]; bi = id[u]; bs = u; }
        float wv = bv; int wi = bi;                          // wave argmin in registers, then 4 candidates in LDS
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(wv, o); const int oi = __shfl_xor(wi, o);
            if (lex_less(ov, oi, wv, wi)) { wv = ov; wi = oi; }
        }
        if (lane == 0) { sv[wave] = wv; si[wave] = wi; }
        __syncthreads();
        wv = sv[0]; wi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (lex_less(sv[w], si[w], wv, wi)) { wv = sv[w]; wi = si[w]; }
        if (tid == 0) { out_v[r] = wv; out_i[r] = wi; }
        if (bs >= 0 && bi == wi && bv == wv) {              // the winning thread retires its element
#pragma unroll
            for (int u = 0; u < PERT; ++u) if (u == bs) { v[u] = INFINITY; id[u] = 0x7fffffff; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void topk_stage1_kernel(const float *__restrict__ dist, int N, int k,
                                                          float *__restrict__ cand_v, int *__restrict__ cand_i)
{
    __shared__ float sv[256];
    __shared__ int si[256];
    const int q = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const float *d = dist + (long long)q * N;
    float v[TK_PER]; int id[TK_PER];
#pragma unroll
    for (int u = 0; u < TK_PER; ++u) {
        const int i = chunk * TK_CHUNK + u * 256 + tid;
        v[u] = (i < N) ? d[i] : INFINITY;
        id[u] = (i < N) ? i : 0x7fffffff;
    }
    const long long base = ((long long)q * gridDim.x + chunk) * k;
    select_k<TK_PER>(v, id, k, sv, si, cand_v + base, cand_i + base);
}

__global__ __launch_bounds__(256) void topk_stage2_kernel(const float *__restrict__ cand_v, const int *__restrict__ cand_i,
                                                          int ncand, int k, long long *__restrict__ idx_out,
                                                          float *__restrict__ val_out)
{
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ float ov[256];
    __shared__ int oi[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    constexpr int PERT = 16;                                  // up to 4 096 candidates per query
    float v[PERT]; int id[PERT];
#pragma unroll
    for (int u = 0; u < PERT; ++u) {
        const int c = u * 256 + tid;
        v[u] = (c < ncand) ? cand_v[(long long)q * ncand + c] : INFINITY;
        id[u] = (c < ncand) ? cand_i[(long long)q * ncand + c] : 0x7fffffff;
    }
    select_k<PERT>(v, id, k, sv, si, ov, oi);
    if (tid < k) {
        idx_out[(long long)q * k + tid] = (oi[tid] == 0x7fffffff) ? -1 : oi[tid];
        val_out[(long long)q * k + tid] = ov[tid];
    }
}

// ---------------------------------------------------------------------------------------------
// hard-negative triplet mining inside one sequence
//   TripletMiner._mine_sequence_triplets  src/gnn/triplet_miner.py:141-229
//   TripletMiner._select_hard_negative    src/gnn/triplet_miner.py:314-359
// One wavefront per anchor.  Candidates are found by brute force over the sequence with float64
// distances (the reference asks a cKDTree the same inclusive-radius questions); the hard negative is
// the candidate with the smallest W1(anchor, candidate) = L1 of the pre-normalised CDF rows.
// ---------------------------------------------------------------------------------------------
struct MineParams {
    double pos_dmax, neg_dmin, neg_dmax;
    int pos_tmin, neg_tmin;
    int strategy;              // 0 hard (argmin W1), 1 random, 2 semi-hard (median W1)
    int per_anchor;            // triplets per anchor
    unsigned long long seed;
};

__device__ __forceinline__ unsigned mine_hash(unsigned long long seed, unsigned a, unsigned k)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)a * 1315423911ull + k + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}

// index of the r-th set candidate (0-based) among `ok(lo)` over lo = 0..n-1, by one wave
template <class F>
__device__ __forceinline__ int nth_candidate(int n, int lane, int r, F ok)
{
    int seen = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int lo = c0 + lane;
        const unsigned long long m = __ballot(lo < n && ok(lo));
        const int cnt = __popcll(m);
        if (r < seen + cnt) {
            int need = r - seen;                 // need-th set bit of m
            unsigned long long mm = m;
            for (int t = 0; t < need; ++t) mm &= mm - 1;
            return c0 + __ffsll((long long)mm) - 1;
        }
        seen += cnt;
    }
    return -1;
}

__global__ __launch_bounds__(256) void mine_kernel(const double *__restrict__ pos, const float *__restrict__ cdf,
                                                   int n, int D, MineParams p, int *__restrict__ out_pos,
                                                   int *__restrict__ out_neg, int *__restrict__ counts,
                                                   float *__restrict__ ws)
{
    const int lane = threadIdx.x & 63, la = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (la >= n) return;
    const double ax = pos[la * 3], ay = pos[la * 3 + 1], az = pos[la * 3 + 2];
    auto dist = [&](int lo) {
        const double dx = pos[lo * 3] - ax, dy = pos[lo * 3 + 1] - ay, dz = pos[lo * 3 + 2] - az;
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    auto is_pos = [&](int lo) {                        // :177-184
        return lo != la && abs(lo - la) >= p.pos_tmin && dist(lo) <= p.pos_dmax;
    };
    auto is_neg = [&](int lo) {                        // :186-201 (inside the outer ball, not inside the inner)
        if (lo == la || abs(lo - la) < p.neg_tmin) return false;
        const double d = dist(lo);
        return d <= p.neg_dmax && !(d <= p.neg_dmin);
    };
    int npos = 0, nneg = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int lo = c0 + lane;
        npos += __popcll(__ballot(lo < n && is_pos(lo)));
        nneg += __popcll(__ballot(lo < n && is_neg(lo)));
    }
    if (lane == 0) { counts[la * 2] = npos; counts[la * 2 + 1] = nneg; }
    if (npos == 0 || nneg == 0) {                      // :208-209
        for (int k = lane; k < p.per_anchor; k += 64) { out_pos[la * p.per_anchor + k] = -1; out_neg[la * p.per_anchor + k] = -1; }
        return;
    }
    // hard negative: argmin W1 over the candidates, (distance, index) order
    int hard = -1;
    if (p.strategy == 0) {
        float best = INFINITY;
        int first_nan = -1;
        const float *ca = cdf + (long long)la * D;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int lo = c0 + lane;
            unsigned long long m = __ballot(lo < n && is_neg(lo));
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int cand = c0 + b;
                const float *cb = cdf + (long long)cand * D;
                float s = 0.0f;
                for (int c = lane; c < D; c += 64) s += fabsf(ca[c] - cb[c]);
                s = wave_sumf(s);
                if (s < best) { best = s; hard = cand; }
                if (s != s && first_nan < 0) first_nan = cand;
            }
        }
        // np.argmin (:346) treats NaN as the minimum and returns its first position: a NaN / garbage descriptor row among
        // the candidates is what the reference would pick; and +inf distances everywhere leave its first candidate
        if (first_nan >= 0) hard = first_nan;
        else if (hard < 0) hard = nth_candidate(n, lane, 0, is_neg);
    }
    if (p.strategy == 2) {
        // semi-hard (:352-357): the candidate at position len // 2 of the candidates sorted by W1.  The distances of
        // this anchor's candidates go to its row of the workspace as BIT PATTERNS compared as unsigned integers:
        // distances are >= 0, so unsigned order == float order; a NaN distance (NaN / garbage descriptor row) becomes the
        // canonical quiet NaN 0x7fc00000, above +inf -- np.argsort sorts NaN last too; rows that are no candidates hold
        // 0xffffffff, above every candidate, so neither the counts nor the final pick can ever land on one.  The value
        // of rank k is found by bisection on the bits, ties resolve to the smaller index (np.argsort's quicksort leaves
        // the order of exact ties undefined).
        unsigned *row = reinterpret_cast<unsigned *>(ws + (long long)la * n);
        const float *ca = cdf + (long long)la * D;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int lo = c0 + lane;
            unsigned long long m = __ballot(lo < n && is_neg(lo));
            if (lo < n) row[lo] = 0xffffffffu;
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float *cb = cdf + (long long)(c0 + b) * D;
                float sd = 0.0f;
                for (int c = lane; c < D; c += 64) sd += fabsf(ca[c] - cb[c]);
                sd = wave_sumf(sd);
                if (lane == b) row[c0 + b] = (sd != sd) ? 0x7fc00000u : __float_as_uint(sd);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the wave re-reads its own row across lanes
        const int k = nneg / 2;
        unsigned lo_b = 0u, hi_b = 0x7fc00000u;                      // smallest v with #(d <= v) >= k + 1
        while (lo_b < hi_b) {
            const unsigned mid = lo_b + (hi_b - lo_b) / 2u;
            int cnt = 0;
            for (int c0 = 0; c0 < n; c0 += 64) {
                const int lo = c0 + lane;
                cnt += __popcll(__ballot(lo < n && row[lo] <= mid));
            }
            if (cnt >= k + 1) hi_b = mid; else lo_b = mid + 1u;
        }
        int less = 0;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int lo = c0 + lane;
            less += __popcll(__ballot(lo < n && row[lo] < lo_b));
        }
        hard = nth_candidate(n, lane, k - less, [&](int lo) { return row[lo] == lo_b; });
    }
    for (int k = 0; k < p.per_anchor; ++k) {           // :211-216
        const int rp = (int)(mine_hash(p.seed, (unsigned)la, 2u * k) % (unsigned)npos);
        const int pc = nth_candidate(n, lane, rp, is_pos);
        int nc = hard;
        if (p.strategy == 1) {
            const int rn = (int)(mine_hash(p.seed, (unsigned)la, 2u * k + 1u) % (unsigned)nneg);
            nc = nth_candidate(n, lane, rn, is_neg);
        }
        if (lane == 0) { out_pos[la * p.per_anchor + k] = pc; out_neg[la * p.per_anchor + k] = nc; }
    }
}

// ---------------------------------------------------------------------------------------------
// validation recall for loop closure
//   GNNTrainer._compute_recall_loop_closure   src/gnn/trainer.py:306-387
// ---------------------------------------------------------------------------------------------
// first j >= i + skip with |p_i - p_j| < thr  -> query j revisits i (trainer.py:342-348); -1 if none
__global__ __launch_bounds__(256) void revisit_kernel(const double *__restrict__ pos, int n, int skip, double thr,
                                                      int *__restrict__ first)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = pos[i * 3], y = pos[i * 3 + 1], z = pos[i * 3 + 2];
    int r = -1;
    for (int j = i + skip; j < n; ++j) {
        const double dx = pos[j * 3] - x, dy = pos[j * 3 + 1] - y, dz = pos[j * 3 + 2] - z;
        if (sqrt(dx * dx + dy * dy + dz * dz) < thr) { r = j; break; }
    }
    first[i] = r;
}

// dist[q][c] = |e_q - e_c|_2 for candidates with |c - q| > skip (trainer.py:363-370), +inf otherwise.
// grid (ceil(n/64), Q): one wave per (query, 16 candidates).
__global__ __launch_bounds__(256) void pair_l2_kernel(const float *__restrict__ emb, const int *__restrict__ qidx,
                                                      int n, int D, int skip, float *__restrict__ dist)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.y, q = qidx[qi];
    const int c0 = (blockIdx.x * 4 + wave) * 16;
    const float *eq = emb + (long long)q * D;
    for (int t = 0; t < 16; ++t) {
        const int c = c0 + t;
        if (c >= n) break;
        float out = INFINITY;
        if (abs(c - q) > skip) {
            const float *ec = emb + (long long)c * D;
            double s = 0.0;
            for (int k = lane; k < D; k += 64) { const float d = eq[k] - ec[k]; s += (double)d * (double)d; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            out = (float)sqrt(s);
        }
        if (lane == 0) dist[(long long)qi * n + c] = out;
    }
}

// rank[qi] = 1-based position of the first of the k nearest candidates lying within thr of the query, 0 if none
__global__ __launch_bounds__(256) void recall_rank_kernel(const double *__restrict__ pos, const int *__restrict__ qidx,
                                                          const long long *__restrict__ topk_idx, int Q, int k, double thr,
                                                          int *__restrict__ rank)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= Q) return;
    const int q = qidx[qi];
    int r = 0;
    for (int t = 0; t < k; ++t) {
        const long long c = topk_idx[(long long)qi * k + t];
        if (c < 0) break;
        const double dx = pos[c * 3] - pos[q * 3], dy = pos[c * 3 + 1] - pos[q * 3 + 1], dz = pos[c * 3 + 2] - pos[q * 3 + 2];
        if (sqrt(dx * dx + dy * dy + dz * dz) < thr) { r = t + 1; break; }
    }
    rank[qi] = r;
}

int per_of(int D) { int p = ((D + 63) / 64 + 3) / 4 * 4; return p < 4 ? 4 : p; }

}  // namespace

extern "C" {

int nsc_w1_cdf(const float *hists, int32_t n, int32_t D, float eps, int32_t divide_plain, float *cdf, void *stream_)
{
    if (n < 0 || D < 1 || D > 1024) return NSC_EUNSUPPORTED;
    if (n == 0) return NSC_OK;
    if (!hists || !cdf) return NSC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const dim3 grid((n + 3) / 4), block(256);
    switch (per_of(D)) {
    case 4: hipLaunchKernelGGL(w1_cdf_kernel<4>, grid, block, 0, st, hists, n, D, eps, divide_plain, cdf); break;
    case 8: hipLaunchKernelGGL(w1_cdf_kernel<8>, grid, block, 0, st, hists, n, D, eps, divide_plain, cdf); break;
    case 12: hipLaunchKernelGGL(w1_cdf_kernel<12>, grid, block, 0, st, hists, n, D, eps, divide_plain, cdf); break;
    default: hipLaunchKernelGGL(w1_cdf_kernel<16>, grid, block, 0, st, hists, n, D, eps, divide_plain, cdf); break;
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_w1_distances(const float *db, int32_t N, int32_t D, float eps, const float *q_cdf, int32_t Q,
                     const float *db_pos, const float *q_pos, float min_dist, float *dist, void *stream_)
{
    if (N < 0 || Q < 0 || D < 1 || D > 1024) return NSC_EUNSUPPORTED;
    if (N == 0 || Q == 0) return NSC_OK;
    if (!db || !q_cdf || !dist) return NSC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const dim3 grid((N + 3) / 4), block(256);
    switch (per_of(D)) {
    case 4: hipLaunchKernelGGL(w1_dist_kernel<4>, grid, block, 0, st, db, N, D, eps, q_cdf, Q, db_pos, q_pos, min_dist, dist); break;
    case 8: hipLaunchKernelGGL(w1_dist_kernel<8>, grid, block, 0, st, db, N, D, eps, q_cdf, Q, db_pos, q_pos, min_dist, dist); break;
    case 12: hipLaunchKernelGGL(w1_dist_kernel<12>, grid, block, 0, st, db, N, D, eps, q_cdf, Q, db_pos, q_pos, min_dist, dist); break;
    default: hipLaunchKernelGGL(w1_dist_kernel<16>, grid, block, 0, st, db, N, D, eps, q_cdf, Q, db_pos, q_pos, min_dist, dist); break;
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_w1_distances_cdf(const float *db_cdf, int32_t N, int32_t D, const float *q_cdf, int32_t Q,
                         const float *db_pos, const float *q_pos, float min_dist, float *dist, void *stream_)
{
    if (N < 0 || Q < 0 || D < 4 || D > 1024 || (D & 3)) return NSC_EUNSUPPORTED;
    if (N == 0 || Q == 0) return NSC_OK;
    if (!db_cdf || !q_cdf || !dist) return NSC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(db_cdf) | reinterpret_cast<uintptr_t>(q_cdf)) & 15u) return NSC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (Q <= 4) {
        int wgs = (N + 3) / 4;
        if (wgs > 256 * 8) wgs = 256 * 8;                      // 8 resident workgroups per CU walk the rows
        const dim3 grid(wgs), block(256);
        if (Q == 1) hipLaunchKernelGGL(w1_stream_kernel<1>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
        else if (Q == 2) hipLaunchKernelGGL(w1_stream_kernel<2>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
        else hipLaunchKernelGGL(w1_stream_kernel<4>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
    } else {
        // tile of 64 rows x 16 / 32 / 64 queries: the smallest one that covers Q in as few column tiles as 64 does
        const int nq = Q <= 16 ? 1 : (Q <= 32 || (Q > 64 && Q <= 96) ? 2 : 4);
        const dim3 grid((N + TL_I - 1) / TL_I, (Q + 16 * nq - 1) / (16 * nq)), block(256);
        if (nq == 1) hipLaunchKernelGGL(w1_tile_kernel<1>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
        else if (nq == 2) hipLaunchKernelGGL(w1_tile_kernel<2>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
        else hipLaunchKernelGGL(w1_tile_kernel<4>, grid, block, 0, st, db_cdf, N, D, q_cdf, Q, db_pos, q_pos, min_dist, dist);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_revisit_queries(const double *positions, int32_t n, int32_t skip_frames, double distance_threshold,
                        int32_t *first_revisit, void *stream_)
{
    if (n < 0 || skip_frames < 0) return NSC_EINVAL;
    if (n == 0) return NSC_OK;
    if (!positions || !first_revisit) return NSC_EINVAL;
    hipLaunchKernelGGL(revisit_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), positions, n,
                       skip_frames, distance_threshold, first_revisit);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_pairwise_l2(const float *emb, const int32_t *query_idx, int32_t Q, int32_t n, int32_t D, int32_t skip_frames,
                    float *dist, void *stream_)
{
    if (Q < 0 || n < 0 || D < 1) return NSC_EINVAL;
    if (Q == 0 || n == 0) return NSC_OK;
    if (!emb || !query_idx || !dist) return NSC_EINVAL;
    if (Q > 65535) return NSC_EUNSUPPORTED;              // grid.y; callers chunk the queries
    hipLaunchKernelGGL(pair_l2_kernel, dim3((n + 63) / 64, Q), dim3(256), 0, static_cast<hipStream_t>(stream_), emb, query_idx,
                       n, D, skip_frames, dist);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_recall_rank(const double *positions, const int32_t *query_idx, const int64_t *topk_idx, int32_t Q, int32_t k,
                    double distance_threshold, int32_t *rank, void *stream_)
{
    if (Q < 0 || k < 1) return NSC_EINVAL;
    if (Q == 0) return NSC_OK;
    if (!positions || !query_idx || !topk_idx || !rank) return NSC_EINVAL;
    hipLaunchKernelGGL(recall_rank_kernel, dim3((Q + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), positions,
                       query_idx, reinterpret_cast<const long long *>(topk_idx), Q, k, distance_threshold, rank);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_mine_workspace_bytes(int32_t n, int32_t strategy)
{
    return (strategy == 2 && n > 0) ? (size_t)n * (size_t)n * sizeof(float) : 0;
}

int nsc_mine_triplets(const double *positions, const float *cdf, int32_t n, int32_t D, const NscMineParams *mp,
                      int32_t *out_pos, int32_t *out_neg, int32_t *counts, void *stream_)
{
    if (mp && mp->strategy == 2) return NSC_EWORKSPACE;          // semi-hard needs nsc_mine_triplets_ws
    return nsc_mine_triplets_ws(positions, cdf, n, D, mp, out_pos, out_neg, counts, nullptr, 0, stream_);
}

int nsc_mine_triplets_ws(const double *positions, const float *cdf, int32_t n, int32_t D, const NscMineParams *mp,
                         int32_t *out_pos, int32_t *out_neg, int32_t *counts, void *ws, size_t ws_bytes, void *stream_)
{
    if (!mp || n < 0 || D < 1) return NSC_EINVAL;
    if (mp->strategy < 0 || mp->strategy > 2) return NSC_EUNSUPPORTED;
    if (mp->strategy == 2 && n > 0 && (!ws || ws_bytes < nsc_mine_workspace_bytes(n, 2))) return NSC_EWORKSPACE;
    if (mp->triplets_per_anchor < 1) return NSC_EINVAL;
    if (n == 0) return NSC_OK;
    if (!positions || !cdf || !out_pos || !out_neg || !counts) return NSC_EINVAL;
    MineParams p;
    p.pos_dmax = mp->positive_distance_max; p.neg_dmin = mp->negative_distance_min; p.neg_dmax = mp->negative_distance_max;
    p.pos_tmin = mp->positive_temporal_min; p.neg_tmin = mp->negative_temporal_min;
    p.strategy = mp->strategy; p.per_anchor = mp->triplets_per_anchor; p.seed = mp->seed;
    hipLaunchKernelGGL(mine_kernel, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), positions, cdf, n, D,
                       p, out_pos, out_neg, counts, static_cast<float *>(ws));
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_topk_workspace_bytes(int32_t Q, int32_t N, int32_t k)
{
    if (Q <= 0 || N <= 0 || k <= 0) return 0;
    const size_t chunks = ((size_t)N + TK_CHUNK - 1) / TK_CHUNK;
    return (size_t)Q * chunks * k * 8;
}

int nsc_topk_smallest(const float *dist, int32_t Q, int32_t N, int32_t k, int64_t *idx, float *val, void *ws,
                      size_t ws_bytes, void *stream_)
{
    if (Q < 0 || N < 0 || k < 0 || k > N) return NSC_EINVAL;
    if (Q == 0 || k == 0) return NSC_OK;
    if (!dist || !idx || !val) return NSC_EINVAL;
    const int chunks = (N + TK_CHUNK - 1) / TK_CHUNK;
    if (k > 256 || (long long)chunks * k > 4096) return NSC_EUNSUPPORTED;   // N*k <= 8.4 M entries
    if (!ws || ws_bytes < nsc_topk_workspace_bytes(Q, N, k)) return NSC_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float *cv = static_cast<float *>(ws);
    int *ci = reinterpret_cast<int *>(cv + (size_t)Q * chunks * k);
    hipLaunchKernelGGL(topk_stage1_kernel, dim3(chunks, Q), dim3(256), 0, st, dist, N, k, cv, ci);
    hipLaunchKernelGGL(topk_stage2_kernel, dim3(Q), dim3(256), 0, st, cv, ci, chunks * k, k,
                       reinterpret_cast<long long *>(idx), val);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

}  // extern "C"
