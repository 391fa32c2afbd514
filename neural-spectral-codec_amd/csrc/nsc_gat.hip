// nsc_gat.hip -- gfx950 kernels + C ABI for the GNN enhancer forward (include/nsc.h).
//
// Path (reference file:line):
//   SpectralGNN.forward                      src/gnn/model.py:96-153
//   torch_geometric 2.4.0 GATConv (heads=1)  src/gnn/model.py:16,75-84,127 (third party)
//
// Kernels
//   csr_*                edge list -> CSR by target with PyG's self-loop convention (deterministic order)
//   gat_fold_kernel      weights-only folding: u_src = W^T att_src, u_dst = W^T att_dst (so the two
//                        attention dot products ride along the lin GEMM as 2 extra output columns),
//                        v = W_edge^T att_edge (edge term becomes an edge_dim-long dot product)
//   gemm_nt_kernel       C = A * B^T on v_mfma_f32_16x16x4_f32 (exact f32), LDS-staged 16*ACC x 64 x 64
//                        tiles fed from a register ring of prefetched chunks (the GEMMs are 0.2-0.8 GFLOP:
//                        latency-, not FLOP-bound), fused bias / BatchNorm(eval) / ReLU / residual epilogue
//   gat_aggregate_kernel one wavefront per target node: leaky-relu logits, wave-shuffle softmax over the
//                        node's in-edges, alpha-weighted sum of neighbour rows (float4 per lane), fused
//                        bias + BatchNorm(eval) + ReLU + residual epilogue
//   gemm_nt_direct_kernel / gat_aggregate_kernel<1,4,false>
//                        the NSC_GAT_CORESIDENT set: no LDS, < 64 VGPRs, bit-identical output -- fits beside a
//                        resident encoder grid so that the GNN of batch k runs under the encoder of batch k+1
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#include "../../include/nsc.h"
#include "../../include/nsc_debug.h"

namespace {

#include "nsc_gemm_glds.h"
#include "nsc_gat_banded.h"
#include "nsc_fill.h"

// ---------------------------------------------------------------------------------------------
// CSR build
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_count_kernel(const long long *__restrict__ ei, long long E, int N,
                                                        int *__restrict__ deg)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long s = ei[e], t = ei[E + e];
        if (s == t || s < 0 || s >= N || t < 0 || t >= N) continue;     // remove_self_loops
        atomicAdd(&deg[t], 1);
    }
}

// exclusive scan of (deg[i] + 1) -> row_ptr, single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int *__restrict__ deg, int N,
                                                        int *__restrict__ row_ptr)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (N + 1023) / 1024;
    const int b = tid * chunk, e = min(b + chunk, N);
    int s = 0;
    for (int i = b; i < e; ++i) s += deg[i] + 1;
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = b; i < e; ++i) { row_ptr[i] = run; run += deg[i] + 1; }
    if (tid == 1023) row_ptr[N] = part[1023];
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const long long *__restrict__ ei, long long E, int N,
                                                       const int *__restrict__ row_ptr,
                                                       int *__restrict__ cursor, int *__restrict__ eid)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long s = ei[e], t = ei[E + e];
        if (s == t || s < 0 || s >= N || t < 0 || t >= N) continue;
        const int pos = row_ptr[t] + atomicAdd(&cursor[t], 1);
        eid[pos] = (int)e;
    }
}

// per node: order the entries by edge id (= original edge order, what PyG's scatter sees), emit the
// sources, append the self loop, and average the incoming edge attributes for it (fill_value='mean')
__global__ __launch_bounds__(256) void csr_finalize_kernel(const long long *__restrict__ ei, int N,
                                                           const float *__restrict__ edge_attr, int edge_dim,
                                                           const int *__restrict__ row_ptr,
                                                           int *__restrict__ src, int *__restrict__ eid,
                                                           float *__restrict__ loop_attr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int b = row_ptr[i], e = row_ptr[i + 1] - 1;     // [b,e) real edges, slot e = self loop
    for (int a = b + 1; a < e; ++a) {                     // insertion sort (in-degrees are small)
        const int v = eid[a];
        int c = a - 1;
        while (c >= b && eid[c] > v) { eid[c + 1] = eid[c]; --c; }
        eid[c + 1] = v;
    }
    float acc[NSC_GAT_MAX_EDGE_DIM];
#pragma unroll
    for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d) acc[d] = 0.0f;
    for (int a = b; a < e; ++a) {
        const int id = eid[a];
        src[a] = (int)ei[id];
        if (edge_attr)
#pragma unroll
            for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d)
                if (d < edge_dim) acc[d] += edge_attr[(long long)id * edge_dim + d];
    }
    src[e] = i;
    eid[e] = -1;
    if (loop_attr) {
        const float cnt = (float)max(e - b, 1);
#pragma unroll
        for (int d = 0; d < NSC_GAT_MAX_EDGE_DIM; ++d)
            if (d < edge_dim) loop_attr[(long long)i * edge_dim + d] = acc[d] / cnt;
    }
}

// Banded form of the CSR (nsc_graph_band_entries): per target NSC_BAND_SLOTS 16-byte slots {source, edge_attr[0],
// edge_attr[1], CSR entry index or -1} in CSR order -- what gat_layer_banded_kernel fetches in one round trip -- and the two
// facts that decide whether a graph may take that kernel: info[0] = max |source - target|, info[1] = max entries per target.
__global__ __launch_bounds__(256) void band_entries_kernel(const int *__restrict__ row_ptr, const int *__restrict__ src,
                                                           const int *__restrict__ eid, const float *__restrict__ edge_attr,
                                                           int edge_dim, const float *__restrict__ loop_attr, int N,
                                                           f32x4 *__restrict__ ent, int *__restrict__ info)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    int maxoff = 0;
    for (int e = beg; e < end; ++e) {
        const int d = src[e] - i;
        maxoff = max(maxoff, d < 0 ? -d : d);
    }
#pragma unroll
    for (int s = 0; s < NSC_BAND_SLOTS; ++s) {
        const int e = beg + s;
        f32x4 o = {__int_as_float(i), 0.f, 0.f, __int_as_float(-1)};
        if (e < end) {
            const int id = eid[e];
            float e0 = 0.f, e1 = 0.f;
            if (edge_attr && edge_dim > 0) {
                const float *ea = id >= 0 ? edge_attr + (long long)id * edge_dim : loop_attr + (long long)i * edge_dim;
                e0 = ea[0];
                if (edge_dim > 1) e1 = ea[1];
            }
            o = f32x4{__int_as_float(src[e]), e0, e1, __int_as_float(e)};
        }
        ent[(long long)i * NSC_BAND_SLOTS + s] = o;
    }
    atomicMax(&info[0], maxoff);
    atomicMax(&info[1], end - beg);
}

// ---------------------------------------------------------------------------------------------
// weights-only folding
// ---------------------------------------------------------------------------------------------
struct PrepLayer {
    const float *w, *att_src, *att_dst, *w_edge, *att_edge;
};
struct PrepArgs {
    PrepLayer l[NSC_GAT_MAX_LAYERS];
    int H, edge_dim;
};

// grid (n_layers, 3): y = 0 -> u_src, 1 -> u_dst, 2 -> v.  folded layout per layer (fold_stride(H) floats): [u_src H][u_dst H][v 8]
__host__ __device__ inline int fold_stride(int H) { return 2 * H + NSC_GAT_MAX_EDGE_DIM; }

__global__ __launch_bounds__(256) void gat_fold_kernel(PrepArgs a, float *__restrict__ folded)
{
    const int l = blockIdx.x, which = blockIdx.y, H = a.H;
    float *dst = folded + (long long)l * fold_stride(H);
    if (which < 2) {
        const float *att = which == 0 ? a.l[l].att_src : a.l[l].att_dst;
        const float *w = a.l[l].w;
        for (int k = threadIdx.x; k < H; k += 256) {          // coalesced over k, 8 rows in flight
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int c = 0;
            for (; c + 8 <= H; c += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] = __builtin_fmaf(w[(long long)(c + u) * H + k], att[c + u], s[u]);
            }
            for (; c < H; ++c) s[0] = __builtin_fmaf(w[(long long)c * H + k], att[c], s[0]);
            dst[which * H + k] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
    } else if (threadIdx.x < NSC_GAT_MAX_EDGE_DIM) {
        const int d = threadIdx.x;
        float s = 0.0f;
        if (a.l[l].w_edge && d < a.edge_dim)
            for (int c = 0; c < H; ++c)
                s = __builtin_fmaf(a.l[l].w_edge[(long long)c * a.edge_dim + d], a.l[l].att_edge[c], s);
        dst[2 * H + d] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T, f32 MFMA 16x16x4.  Workgroup 256 threads = 4 waves, tile 32 x 64:
// wave w owns columns [16w, 16w+16) x 32 rows (two accumulators share the B operand).
// Columns >= n_main come from the extra rows Bx (the folded attention vectors) and are written to
// aux0/aux1 instead of C.
// ---------------------------------------------------------------------------------------------
// EPI: 0 = plain store + aux columns (lin), 1 = bias + BatchNorm + ReLU (input_proj),
//      2 = bias + residual (output_proj / residual_proj)
//
// Tiles of A (16*ACC x 64) and B (64 x 64) are staged through LDS: global loads are 256-byte row segments
// (16 lanes x 16 B), each element is fetched once per workgroup, rows are padded to 68 floats; the MFMA operands are
// ds_read_b128.  Two LDS stages fed from a register ring of prefetched chunks (the operand fetch -- L2 / Infinity
// Cache round trips, the A rows were written by the previous kernel on other XCDs -- is what bounds these GEMMs).
// The epilogue goes back through LDS so that every output row leaves as 256 contiguous bytes (float4 per lane);
// the residual is read the same way.  (Measured and dropped in round 2: two MFMA chains per tile, operand fetches
// one block ahead of the MFMAs, a two-slot ring -- 20 % slower on the 13-chunk input projection.)
// ACN = 2 (round 2): the four waves form a 2 x 2 grid and every wave owns ACC x 2 accumulator blocks (a 32 x 32 wave tile on a
// 64 x 64 workgroup tile).  These GEMMs are LDS-bandwidth-bound (0.27 B of LDS traffic per FLOP with
// 32 x 16 wave tiles, every wave reading the whole A tile); the square wave tile needs 0.18 B/FLOP.  The k order of every
// output element is unchanged, so all variants (and the LDS-free set) stay bit-identical.
template <int ACC, int EPI, int ACN = 1>   // wave tile (16*ACC) x (16*ACN); workgroup tile (16*ACC*WR) x 64
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float *__restrict__ A, int lda,
                                                      const float *__restrict__ B, int ldb,
                                                      const float *__restrict__ Bx, int M, int N,
                                                      int n_main, int K, float *__restrict__ C, int ldc,
                                                      GemmEpi ep)
{
    constexpr int WC = 4 / ACN, WR = 4 / WC;   // waves across the columns / the rows of the workgroup tile
    constexpr int BM = 16 * ACC * WR, BN = 64, BK = 64, LD = BK + 4;
    constexpr int NA = BM * (BK / 4) / 256;    // float4 loads per thread for the A tile (1, 2 or 4)
    constexpr int NB = BN * (BK / 4) / 256;    // 4
    extern __shared__ __attribute__((aligned(16))) float gemm_lds[];   // As[2][BM*LD] | Bs[2][BN*LD] (69.6 KB at BM = 64)
    float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(gemm_lds);
    float (*Bs)[BN * LD] = reinterpret_cast<float (*)[BN * LD]>(gemm_lds + 2 * BM * LD);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;  // this wave's block of the workgroup tile
    const int r = lane & 15, q = lane >> 4;
    const unsigned tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int m0 = (int)(tile / gridDim.x) * BM, n0 = (int)(tile % gridDim.x) * BN;

    // staging map: float4 f = tid + 256 i -> tile row f / 16, k offset 4 (f % 16).  Rows / columns
    // past the matrix re-read a valid row: their products only reach outputs that are never stored.
    const float *ga[NA];
    int sa[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
        const int gr = m0 + row;
        ga[i] = A + (long long)(gr < M ? gr : M - 1) * lda + 4 * c4;
        sa[i] = row * LD + 4 * c4;
    }
    const float *gb[NB];
    int sb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
        int gc = n0 + row;
        gc = gc < N ? gc : N - 1;
        gb[i] = ((gc < n_main) ? B + (long long)gc * ldb : Bx + (long long)(gc - n_main) * ldb) + 4 * c4;
        sb[i] = row * LD + 4 * c4;
    }
    const int c4t = tid & 15;                  // every float4 of this thread sits at k offset 4 * c4t of a chunk

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[ACC][ACN];
#pragma unroll
    for (int h = 0; h < ACC; ++h)
#pragma unroll
        for (int g = 0; g < ACN; ++g) acc[h][g] = zero;
    // A wave whose columns all lie beyond N only helps with the staging (the fifth column block of the lin GEMMs
    // carries just the two attention columns): it skips the operand reads and the MFMAs.
    const bool active = n0 + wc * 16 * ACN < N;

    // Register ring of PD chunks: every global load of the next PD chunks is in flight while the current
    // chunk's MFMAs run, so one L2 round trip is exposed per kernel instead of one per chunk (these GEMMs
    // are latency-bound: K = 256 is only 4 chunks).  A short last chunk (K % 64 != 0) re-reads valid
    // columns; those k-blocks are skipped below.
    constexpr int PD = (ACC * WR == 1) ? 4 : 3;
    const int nchunks = K / BK + ((K % BK) ? 1 : 0);
    f32x4 ra[PD][NA], rb[PD][NB];
    auto load_chunk = [&](int ch, f32x4 (&xa)[NA], f32x4 (&xb)[NB]) {
        const int kn = ch * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c4 = (tid + 256 * i) & 15, k = kn + 4 * c4;
            xa[i] = *reinterpret_cast<const f32x4 *>(ga[i] + (k + 4 <= K ? kn : K - 4 - 4 * c4));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c4 = (tid + 256 * i) & 15, k = kn + 4 * c4;
            xb[i] = *reinterpret_cast<const f32x4 *>(gb[i] + (k + 4 <= K ? kn : K - 4 - 4 * c4));
        }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) load_chunk(s < nchunks ? s : nchunks - 1, ra[s], rb[s]);
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(&As[0][sa[i]]) = ra[0][i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4 *>(&Bs[0][sb[i]]) = rb[0][i];
    if (PD < nchunks) load_chunk(PD, ra[0], rb[0]);          // slot 0 is free again
    __syncthreads();

    for (int ch0 = 0; ch0 < nchunks; ch0 += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            const int ch = ch0 + s;
            if (ch < nchunks) {                                // workgroup-uniform
                const int cur = ch & 1;
                const float *as = As[cur], *bs = Bs[cur];
                const int kleft = K - ch * BK;
#pragma unroll
                for (int d = 0; d < BK / 16; ++d) {
                    if (16 * d < kleft && active) {
                        f32x4 bv[ACN], av[ACC];
#pragma unroll
                        for (int g = 0; g < ACN; ++g)
                            bv[g] = *reinterpret_cast<const f32x4 *>(&bs[(wc * 16 * ACN + 16 * g + r) * LD + 16 * d + 4 * q]);
#pragma unroll
                        for (int h = 0; h < ACC; ++h)
                            av[h] = *reinterpret_cast<const f32x4 *>(&as[(wr * 16 * ACC + 16 * h + r) * LD + 16 * d + 4 * q]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int h = 0; h < ACC; ++h)
#pragma unroll
                                for (int g = 0; g < ACN; ++g)
                                    acc[h][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][t], bv[g][t], acc[h][g], 0, 0, 0);
                    }
                }
                if (ch + 1 < nchunks) {
                    const int ns = (s + 1) % PD;                // ring slot holding chunk ch + 1 (static)
#pragma unroll
                    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(&As[cur ^ 1][sa[i]]) = ra[ns][i];
#pragma unroll
                    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4 *>(&Bs[cur ^ 1][sb[i]]) = rb[ns][i];
                    if (ch + 1 + PD < nchunks) load_chunk(ch + 1 + PD, ra[ns], rb[ns]);
                }
                __syncthreads();
            }
        }
    }

    // epilogue: the tile goes through LDS (stage buffers are free after the last barrier), rows leave as float4
    float *Cs = Bs[0];                                         // BM x LD floats fit (BM <= 64)
#pragma unroll
    for (int h = 0; h < ACC; ++h)
#pragma unroll
        for (int g = 0; g < ACN; ++g)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)                 // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
                Cs[(wr * 16 * ACC + 16 * h + 4 * q + reg) * LD + wc * 16 * ACN + 16 * g + r] = acc[h][g][reg];
    __syncthreads();
    const int cg = n0 + 4 * c4t;                               // first of this thread's 4 columns
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, bn_scale[4] = {1.f, 1.f, 1.f, 1.f}, bn_shift[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = (cg + j < N) ? cg + j : N - 1;
        if (EPI != 0) bias[j] = ep.bias[col];
        if (EPI == 1) {
            // torch batch_norm eval: alpha = invstd * weight, beta = bias - mean * alpha
            const float invstd = 1.0f / sqrtf(ep.bn_var[col] + ep.bn_eps);
            bn_scale[j] = invstd * ep.bn_w[col];
            bn_shift[j] = ep.bn_b[col] - ep.bn_mean[col] * bn_scale[j];
        }
    }
    const bool vec = (cg + 3 < n_main) && !(ldc & 3) && !(reinterpret_cast<unsigned long long>(C) & 15) &&
                     (EPI != 2 || !ep.resid || (!(ep.ldr & 3) && !(reinterpret_cast<unsigned long long>(ep.resid) & 15)));
#pragma unroll
    for (int pass = 0; pass < BM / 16; ++pass) {
        const int lr = pass * 16 + (tid >> 4), row = m0 + lr;
        if (row >= M) continue;
        const f32x4 t = *reinterpret_cast<const f32x4 *>(&Cs[lr * LD + 4 * c4t]);
        float v[4] = {t.x, t.y, t.z, t.w};
        f32x4 rs = zero;
        if (EPI == 2 && ep.resid && vec) rs = *reinterpret_cast<const f32x4 *>(ep.resid + (long long)row * ep.ldr + cg);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (EPI != 0) v[j] = v[j] + bias[j];
            if (EPI == 1) v[j] = fmaxf(v[j] * bn_scale[j] + bn_shift[j], 0.0f);
        }
        if (vec) {
            if (EPI == 2 && ep.resid) { v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w; }
            *reinterpret_cast<f32x4 *>(C + (long long)row * ldc + cg) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cg + j;
                if (col >= N) continue;
                if (col < n_main) {
                    float o = v[j];
                    if (EPI == 2 && ep.resid) o = o + ep.resid[(long long)row * ep.ldr + col];
                    C[(long long)row * ldc + col] = o;
                } else {
                    float *aux = (col == n_main) ? ep.aux0 : ep.aux1;
                    aux[row] = v[j];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-free variant of the same GEMM (NSC_GAT_CORESIDENT): operands are loaded from global memory into a
// register ring of P k-blocks and moved to the MFMA layout by ds_bpermute (the LDS crossbar, no LDS
// allocation); the four waves of a workgroup re-read the A rows through L1.  0 bytes of LDS and < 64 VGPRs,
// so one workgroup fits on a CU next to the four resident workgroups of encode_fused_kernel -- the GNN
// forward of batch k can then run on a second stream under the encoder of batch k+1.  Same k order and
// operand assignment as gemm_nt_kernel, so the results are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int ACC, int EPI, int P = 4>     // P: k-blocks in the register ring
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56))) void gemm_nt_direct_kernel(const float *__restrict__ A, int lda,
                                                             const float *__restrict__ B, int ldb,
                                                             const float *__restrict__ Bx, int M, int N,
                                                             int n_main, int K, float *__restrict__ C, int ldc,
                                                             GemmEpi ep)
{
    constexpr int BM = 16 * ACC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const unsigned tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int m0 = (int)(tile / gridDim.x) * BM, n0 = (int)(tile % gridDim.x) * 64;
    const int cb = n0 + wave * 16 + r;
    // A wave whose 16 columns all lie beyond N has nothing to compute (no barriers in this kernel): in the lin GEMMs the
    // fifth column block carries only the two attention columns, three of its four waves leave here.
    if (n0 + wave * 16 >= N) return;

    // Load mapping: lane L fetches 16 bytes of row L >> 2 at k offset 4 (L & 3), so 4 neighbouring lanes read
    // 64 contiguous bytes and a 16-lane quad touches 4 cache lines.  (Loading in the MFMA operand layout
    // instead -- row L & 15, offset 4 (L >> 4) -- makes every quad touch 16 lines, 64 L1 requests per load
    // instruction: those requests, not the MFMAs, were what slowed the co-running encoder.)  One ds_bpermute per
    // dword then moves the data to the operand layout: lane (r, q) pulls from lane 4 r + q.
    const int lr = lane >> 2, lq = lane & 3;
    const int perm = 4 * (4 * r + q);
    const float *pa[ACC];
#pragma unroll
    for (int h = 0; h < ACC; ++h) {
        const int gr = m0 + 16 * h + lr;
        pa[h] = A + (long long)(gr < M ? gr : M - 1) * lda + 4 * lq;
    }
    const int lcol = (n0 + wave * 16 + lr) < N ? (n0 + wave * 16 + lr) : N - 1;
    const float *pb = ((lcol < n_main) ? B + (long long)lcol * ldb : Bx + (long long)(lcol - n_main) * ldb) + 4 * lq;
    const int col = cb < N ? cb : N - 1;
    auto to_operand = [&](const f32x4 &v) {
        f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float e = v[t];                  // (bit_cast straight from a vector element reads element 0)
            o[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(e)));
        }
        return o;
    };


    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[ACC];
#pragma unroll
    for (int h = 0; h < ACC; ++h) acc[h] = zero;

    const int nblk = K >> 4;                       // K is a multiple of 16 (check_model)
    f32x4 ra[P][ACC], rb[P];
    auto load_blk = [&](int blk, f32x4 (&xa)[ACC], f32x4 &xb) {
        const int k = blk << 4;
#pragma unroll
        for (int h = 0; h < ACC; ++h) xa[h] = *reinterpret_cast<const f32x4 *>(pa[h] + k);
        xb = *reinterpret_cast<const f32x4 *>(pb + k);
    };
#pragma unroll
    for (int s = 0; s < P; ++s) load_blk(s < nblk ? s : nblk - 1, ra[s], rb[s]);
    for (int b0 = 0; b0 < nblk; b0 += P) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int blk = b0 + s;
            if (blk < nblk) {                      // workgroup-uniform
                f32x4 xa[ACC];
#pragma unroll
                for (int h = 0; h < ACC; ++h) xa[h] = to_operand(ra[s][h]);
                const f32x4 xb = to_operand(rb[s]);
                if (blk + P < nblk) load_blk(blk + P, ra[s], rb[s]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int h = 0; h < ACC; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[h][t], xb[t], acc[h], 0, 0, 0);
            }
        }
    }

    if (cb >= N) return;
    // (epilogue operands fetched here, not ahead of the loop: three registers less through it -- the 32-row form has to fit
    // 56 VGPRs for two of its waves to share a SIMD with five encoder workgroups; their round trip hides under other waves)
    float bias = 0.f, bn_scale = 1.f, bn_shift = 0.f;
    if (EPI != 0) bias = ep.bias[col];
    if (EPI == 1) {
        const float invstd = 1.0f / sqrtf(ep.bn_var[col] + ep.bn_eps);
        bn_scale = invstd * ep.bn_w[col];
        bn_shift = ep.bn_b[col] - ep.bn_mean[col] * bn_scale;
    }
    const bool main_col = cb < n_main;
#pragma unroll
    for (int h = 0; h < ACC; ++h) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = m0 + 16 * h + 4 * q + reg;
            if (row >= M) continue;
            float v = acc[h][reg];
            if (main_col) {
                if (EPI != 0) v = v + bias;
                if (EPI == 1) v = fmaxf(v * bn_scale + bn_shift, 0.0f);
                if (EPI == 2 && ep.resid) v = v + ep.resid[(long long)row * ep.ldr + cb];
                C[(long long)row * ldc + cb] = v;
            } else {
                float *aux = (cb == n_main) ? ep.aux0 : ep.aux1;
                aux[row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small-LDS co-resident GEMM (NSC_GAT_CORESIDENT | NSC_GAT_SHARED_B, round 3).  Beside the encoder the GNN costs its
// THROUGHPUT share, not its latency (experiment 17c), and gemm_nt_direct_kernel spends per MFMA two
// ds_bpermute (on the LDS pipe the encoder's ds_min atomics use), half a 16-byte load and the addressing around them:
// its 16 x 64 workgroup tile re-reads the whole B (weight) tile for 16 rows, and its four waves each fetch and transpose
// the same A rows.  Here a workgroup owns 64 rows x 64 columns: every wave its own 16 rows (A: one coalesced load + 4
// bpermutes per 16-k block, as before) and all 64 columns (4 accumulators); the B block (64 columns x 16 k = 4 KB) is
// loaded ONCE per workgroup -- one 16-byte load per thread -- and shared through a double-buffered 2 x 5 KB LDS tile that
// the waves read as MFMA operands (ds_read_b128).  Per 16 MFMAs: 2 loads, 4 bpermutes, 1 ds_write, 4 ds_read (the direct
// kernel: 8 loads, 32 bpermutes), and B is fetched from L2 a quarter as often.  10 KB of LDS: two workgroups fit in the 20 KB
// a CU has left beside five resident encoder workgroups.  Same k order and operand assignment as the other GEMMs: the
// results are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_share_kernel(const float *__restrict__ A, int lda,
                                                            const float *__restrict__ B, int ldb,
                                                            const float *__restrict__ Bx, int M, int N,
                                                            int n_main, int K, float *__restrict__ C, int ldc,
                                                            GemmEpi ep)
{
    constexpr int LDB = 20;                        // floats per staged B column: 16 k + 4 of padding (bank spread)
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const unsigned tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int m0 = (int)(tile / gridDim.x) * 64 + 16 * wave, n0 = (int)(tile % gridDim.x) * 64;

    // A: this wave's 16 rows, coalesced map (lane L: row L >> 2, k offset 4 (L & 3)) + ds_bpermute to the operand layout
    const int lr = lane >> 2, lq = lane & 3;
    const int perm = 4 * (4 * r + q);
    const int gr = m0 + lr;
    const float *pa = A + (long long)(gr < M ? gr : M - 1) * lda + 4 * lq;
    // B: thread t stages column t >> 2 at k offset 4 (t & 3) -- 4 lanes read 64 contiguous bytes
    const int sc = tid >> 2, sq = tid & 3;
    int gc = n0 + sc;
    gc = gc < N ? gc : N - 1;
    const float *pb = ((gc < n_main) ? B + (long long)gc * ldb : Bx + (long long)(gc - n_main) * ldb) + 4 * sq;
    const int sidx = sc * LDB + 4 * sq;
    auto to_operand = [&](const f32x4 &v) {
        f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float e = v[t];
            o[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(e)));
        }
        return o;
    };
    const int ncb = min(4, (N - n0 + 15) >> 4);    // 16-column blocks of this tile that hold real columns (uniform)

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero, zero, zero, zero};
    const int nblk = K >> 4;                       // K is a multiple of 16 (check_model)
    f32x4 ra[2], rb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k = (s < nblk ? s : nblk - 1) << 4;
        ra[s] = *reinterpret_cast<const f32x4 *>(pa + k);
        rb[s] = *reinterpret_cast<const f32x4 *>(pb + k);
    }
    *reinterpret_cast<f32x4 *>(&Bs[0][sidx]) = rb[0];
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int blk = b0 + s;
            if (blk < nblk) {                      // workgroup-uniform
                const float *bs = Bs[s];
                // B of block blk + 1 goes to the other buffer (last read in iteration blk - 1, a barrier ago)
                if (blk + 1 < nblk) *reinterpret_cast<f32x4 *>(&Bs[s ^ 1][sidx]) = rb[s ^ 1];
                const f32x4 xa = to_operand(ra[s]);
                if (blk + 2 < nblk) {              // both register sets of this parity are free: refill two blocks ahead
                    const int k = (blk + 2) << 4;
                    ra[s] = *reinterpret_cast<const f32x4 *>(pa + k);
                    rb[s] = *reinterpret_cast<const f32x4 *>(pb + k);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < ncb) {
                        const f32x4 xb = *reinterpret_cast<const f32x4 *>(&bs[(16 * g + r) * LDB + 4 * q]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t], xb[t], acc[g], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    }

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cb = n0 + 16 * g + r;
        if (cb >= N) continue;
        const bool main_col = cb < n_main;
        float bias = 0.f, bn_scale = 1.f, bn_shift = 0.f;
        if (EPI != 0 && main_col) bias = ep.bias[cb];
        if (EPI == 1 && main_col) {
            const float invstd = 1.0f / sqrtf(ep.bn_var[cb] + ep.bn_eps);
            bn_scale = invstd * ep.bn_w[cb];
            bn_shift = ep.bn_b[cb] - ep.bn_mean[cb] * bn_scale;
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = m0 + 4 * q + reg;
            if (row >= M) continue;
            float v = acc[g][reg];
            if (main_col) {
                if (EPI != 0) v = v + bias;
                if (EPI == 1) v = fmaxf(v * bn_scale + bn_shift, 0.0f);
                if (EPI == 2 && ep.resid) v = v + ep.resid[(long long)row * ep.ldr + cb];
                C[(long long)row * ldc + cb] = v;
            } else {
                float *aux = (cb == n_main) ? ep.aux0 : ep.aux1;
                aux[row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention + aggregation, one wavefront per target node
// ---------------------------------------------------------------------------------------------
struct AggArgs {
    const int *row_ptr, *src, *eid;
    const float *loop_attr;      // (N, edge_dim) or null
    const float *edge_attr;      // (E, edge_dim) or null
    const float *v;              // (edge_dim) folded edge vector or null
    const float *a_src, *a_dst;  // (N)
    const float *G;              // (N, H) transformed features
    const float *bias, *bn_w, *bn_b, *bn_mean, *bn_var;
    const float *resid;          // (N, H) or null
    float *out;                  // (N, H)
    float *alpha_out;            // (nnz capacity) or null
    float bn_eps, slope;
    int N, H, edge_dim, relu;
};

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sumf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// CH = ceil(H / 256): float4 chunks per lane; UR neighbour rows in flight; PRE: epilogue operands prefetched.
// <CH, 8, true> is the standalone configuration; <1, 4, false> stays under 64 VGPRs for NSC_GAT_CORESIDENT.
template <int CH, int UR, bool PRE>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(AggArgs a)
{
    const int lane = threadIdx.x & 63;
    const int i = (int)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);   // an XCD takes a contiguous node range
    if (i >= a.N) return;
    const int beg = a.row_ptr[i], end = a.row_ptr[i + 1];
    const float ad = a.a_dst[i];
    const bool use_edge = a.edge_attr && a.v && a.edge_dim > 0;

    // folded edge vector in registers (edge_dim = 2 is the reference's shape)
    const bool ed2 = use_edge && a.edge_dim == 2;
    const float v0 = use_edge ? a.v[0] : 0.0f, v1 = ed2 ? a.v[1] : 0.0f;

    // the loads of one entry go out in two dependent rounds only: {src, eid}, then {a_src[j], edge_attr}
    auto logit = [&](int e, int &j) -> float {
        j = a.src[e];
        const int id = use_edge ? a.eid[e] : 0;
        float t = 0.0f;
        if (use_edge) {
            const float *ea = id >= 0 ? a.edge_attr + (long long)id * a.edge_dim
                                      : a.loop_attr + (long long)i * a.edge_dim;
            if (ed2) {
                const float e0 = ea[0], e1 = ea[1];
                t = __builtin_fmaf(e1, v1, e0 * v0);
            } else {
                t = ea[0] * v0;
                for (int d = 1; d < a.edge_dim; ++d) t = __builtin_fmaf(ea[d], a.v[d], t);
            }
        }
        const float l = (a.a_src[j] + ad) + t;
        return l > 0.0f ? l : a.slope * l;                        // leaky_relu
    };

    // epilogue operands do not depend on the gather: with PRE their loads go out first and land under it
    f32x4 ep_bias[CH], ep_w[CH], ep_b[CH], ep_mean[CH], ep_var[CH], ep_res[CH];
    auto load_epilogue = [&]() {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = 4 * lane + 256 * c;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const bool in = col < a.H;
            ep_bias[c] = in ? *reinterpret_cast<const f32x4 *>(a.bias + col) : z;
            ep_w[c] = (in && a.bn_w) ? *reinterpret_cast<const f32x4 *>(a.bn_w + col) : z;
            ep_b[c] = (in && a.bn_w) ? *reinterpret_cast<const f32x4 *>(a.bn_b + col) : z;
            ep_mean[c] = (in && a.bn_w) ? *reinterpret_cast<const f32x4 *>(a.bn_mean + col) : z;
            ep_var[c] = (in && a.bn_w) ? *reinterpret_cast<const f32x4 *>(a.bn_var + col) : z;
            ep_res[c] = (in && a.resid) ? *reinterpret_cast<const f32x4 *>(a.resid + (long long)i * a.H + col) : z;
        }
    };
    if (PRE) load_epilogue();

    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int deg = end - beg;

    if (deg <= 64) {
        // common case (temporal chain: deg <= 5): one logit per lane, everything stays in registers
        const int e = beg + lane;
        int j = 0;
        float l = -INFINITY;
        if (e < end) l = logit(e, j);
        const float m = wave_max(l);
        const float p = (e < end) ? expf(l - m) : 0.0f;
        const float den = wave_sumf(p) + 1e-16f;                  // PyG softmax
        const float al = p / den;
        if (a.alpha_out && e < end) a.alpha_out[e] = al;
        for (int t0 = 0; t0 < deg; t0 += UR) {                    // UR neighbour rows in flight
            f32x4 gv[UR][CH];
            float at[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int t = t0 + u;
                at[u] = (t < deg) ? __shfl(al, t & 63) : 0.0f;
                const int jt = __shfl(j, t & 63);
                const float *g = a.G + (long long)((t < deg) ? jt : i) * a.H;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int col = 4 * lane + 256 * c;
                    gv[u][c] = (col < a.H) ? *reinterpret_cast<const f32x4 *>(g + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u)                          // entries in edge order, loop last
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    acc[c].x = __builtin_fmaf(at[u], gv[u][c].x, acc[c].x);
                    acc[c].y = __builtin_fmaf(at[u], gv[u][c].y, acc[c].y);
                    acc[c].z = __builtin_fmaf(at[u], gv[u][c].z, acc[c].z);
                    acc[c].w = __builtin_fmaf(at[u], gv[u][c].w, acc[c].w);
                }
        }
    } else {
        float m = -INFINITY;
        for (int e = beg + lane; e < end; e += 64) { int j; m = fmaxf(m, logit(e, j)); }
        m = wave_max(m);
        float s = 0.0f;
        for (int e = beg + lane; e < end; e += 64) { int j; s += expf(logit(e, j) - m); }
        s = wave_sumf(s);
        const float den = s + 1e-16f;
        for (int c0 = beg; c0 < end; c0 += 64) {
            const int e = c0 + lane;
            int j = 0;
            float al = 0.0f;
            if (e < end) {
                al = expf(logit(e, j) - m) / den;
                if (a.alpha_out) a.alpha_out[e] = al;
            }
            const int cnt = min(64, end - c0);
            for (int t = 0; t < cnt; ++t) {
                const float at = __shfl(al, t);
                const int jt = __shfl(j, t);
                const float *g = a.G + (long long)jt * a.H;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int col = 4 * lane + 256 * c;
                    if (col < a.H) {
                        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + col);
                        acc[c].x = __builtin_fmaf(at, gv.x, acc[c].x);
                        acc[c].y = __builtin_fmaf(at, gv.y, acc[c].y);
                        acc[c].z = __builtin_fmaf(at, gv.z, acc[c].z);
                        acc[c].w = __builtin_fmaf(at, gv.w, acc[c].w);
                    }
                }
            }
        }
    }

    if (!PRE) load_epilogue();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = 4 * lane + 256 * c;
        if (col >= a.H) continue;
        f32x4 o = acc[c];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = o[t] + ep_bias[c][t];
            if (a.bn_w) {
                const float invstd = 1.0f / sqrtf(ep_var[c][t] + a.bn_eps);
                const float sc = invstd * ep_w[c][t];
                v = v * sc + (ep_b[c][t] - ep_mean[c][t] * sc);
            }
            if (a.relu) v = fmaxf(v, 0.0f);
            if (a.resid) v += ep_res[c][t];
            o[t] = v;
        }
        *reinterpret_cast<f32x4 *>(a.out + (long long)i * a.H + col) = o;
    }
}

#ifdef NSC_DEV_TUNING
// Diagnostic co-runner (nsc_debug_burn, development builds only): one kind of operation per launch, in the co-resident
// kernels' footprint.
__global__ __launch_bounds__(256) void burn_kernel(int mode, int per_wave, float *__restrict__ scratch)
{
    const int lane = threadIdx.x & 63;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};
    float a = 1.0f + (float)lane * 1e-3f, b = 0.5f + (float)(gid & 7) * 1e-3f, v = a;
    asm volatile("" : "+v"(a), "+v"(b));
    if (mode == 0) {
        for (int i = 0; i < per_wave; i += 4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
    } else if (mode == 1) {
        float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
        for (int i = 0; i < per_wave; i += 4) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        }
        v = (x0 + x1) + (x2 + x3);
    } else if (mode == 2 || mode == 4 || mode == 5) {
        // 2: 1 MB = 65 536 quads, every wave walks all of it from its own start (L2 hits); 4: every wave of the chip walks the
        // same 16 KB (L1 hits); 5: the four waves of a workgroup read the same addresses of the 1 MB (one L2 fetch, three L1 hits)
        const f32x4 *src = reinterpret_cast<const f32x4 *>(scratch);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const unsigned mask = mode == 4 ? 1023u : 65535u;
        unsigned idx = mode == 4 ? 0u : mode == 5 ? (blockIdx.x * 256u * 64u) & 65535u : (gid * 64u) & 65535u;
        for (int i = 0; i < per_wave; i += 4) {
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = src[(idx + 64u * c + lane) & mask];
#pragma unroll
            for (int c = 0; c < 4; ++c) s += t[c];
            idx += 256u;
        }
        v = (s.x + s.y) + (s.z + s.w);
    } else {
        int x = __float_as_int(a);
        for (int i = 0; i < per_wave; ++i) x = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, x);
        v = __int_as_float(x);
    }
    if (mode == 0) v = (acc[0].x + acc[1].y) + (acc[2].z + acc[3].w);
    if (v == 12345.678f) scratch[(1u << 18) + (gid & 1023u)] = v;          // keeps the work alive; practically never true
}

#endif  // NSC_DEV_TUNING

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct GatWs {
    size_t h0, h1, g, a_src, a_dst, total;
};

GatWs gat_ws(const NscGatModel *m, int N)
{
    GatWs w;
    size_t o = 0;
    const size_t nh = align256((size_t)N * m->hidden * sizeof(float));
    w.h0 = o; o += nh;
    w.h1 = o; o += nh;
    w.g = o;  o += nh;
    w.a_src = o; o += align256((size_t)N * sizeof(float));
    w.a_dst = o; o += align256((size_t)N * sizeof(float));
    w.total = o;
    return w;
}

int check_model(const NscGatModel *m)
{
    if (!m) return NSC_EINVAL;
    if (m->n_layers < 1 || m->n_layers > NSC_GAT_MAX_LAYERS) return NSC_EUNSUPPORTED;
    if (m->hidden < 16 || m->hidden > 1024 || (m->hidden & 15)) return NSC_EUNSUPPORTED;   // GEMM k-blocks of 16
    if (m->in_dim < 16 || (m->in_dim & 15) || m->out_dim < 1) return NSC_EUNSUPPORTED;
    if (!m->in_bn_w || !m->in_bn_b || !m->in_bn_mean || !m->in_bn_var) return NSC_EINVAL;
    if (m->edge_dim < 0 || m->edge_dim > NSC_GAT_MAX_EDGE_DIM) return NSC_EUNSUPPORTED;
    if (!m->in_w || !m->in_b || !m->out_w || !m->out_b) return NSC_EINVAL;
    if (m->residual && m->in_dim != m->out_dim && (!m->res_w || !m->res_b)) return NSC_EINVAL;
    for (int l = 0; l < m->n_layers; ++l) {
        const NscGatLayer &L = m->layers[l];
        if (!L.lin_w || !L.att_src || !L.att_dst || !L.bias) return NSC_EINVAL;
        if (m->edge_dim > 0 && (!L.lin_edge_w || !L.att_edge)) return NSC_EINVAL;
    }
    return NSC_OK;
}


template <int EPI>
void launch_gemm(hipStream_t st, int cores, const float *A, int lda, const float *B, int ldb, const float *Bx,
                 int M, int N, int n_main, int K, float *C, int ldc, const GemmEpi &ep)
{
    // Tile choice (measured at M = 1 024 and 4 541, round 2): 32-row tiles (two accumulators share the B operand, 1.7x
    // less operand traffic per MFMA) as soon as they still give >= 1.5 workgroups per CU; below that 16-row tiles, so
    // that every SIMD of the chip gets a wave -- these GEMMs are operand-latency-, not MFMA-bound.
    // cores: 0 stand-alone (gemm_glds_kernel where it can run), 1 / 2 the co-resident sets, 3 the round-2 LDS-tiled kernels
    if (cores == 0 && launch_glds<EPI>(st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep)) return;
    const bool coresident = cores == 1 || cores == 2;
    if (cores == 2) {                              // small-LDS co-resident form: 64 x 64 tiles, B shared through 10 KB of LDS
        const dim3 gs((N + 63) / 64, (M + 63) / 64);
        hipLaunchKernelGGL((gemm_nt_share_kernel<EPI>), gs, dim3(256), 0, st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep);
        return;
    }
    const long long w2 = (long long)((N + 63) / 64) * ((M + 31) / 32);
    const long long w4 = (long long)((N + 63) / 64) * ((M + 63) / 64);
    bool two = w2 >= 384 && !coresident;
    // 64 x 64 tiles with 32 x 32 wave tiles (a third less LDS traffic per FLOP) where they still give two workgroups to
    // every CU (70 KB of LDS each): output_proj at M = 4 541 (923 tiles); input_proj and lin there have 284 / 355
    bool four = w4 >= 512 && !coresident;
    if (four) {
        // The 64 x 64 form takes 69.6 KB of dynamic LDS: above 64 KB a kernel has to be opted in, and the attribute is
        // per DEVICE.  State: one atomic per (instantiation, device) -- 0 not tried, 1 opted in, 2 refused (then the
        // 32-row tiles run: same results, bit for bit).  Racing threads at worst both set the attribute (idempotent).
        static std::atomic<int> opted[16];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
            four = false;
        } else {
            int s = opted[dev].load(std::memory_order_acquire);
            if (s == 0) {
                s = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_kernel<2, EPI, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess ? 1 : 2;
                opted[dev].store(s, std::memory_order_release);
            }
            if (s != 1) four = false;
        }
    }
    const int bm = four ? 64 : two ? 32 : 16;
    const dim3 grid((N + 63) / 64, (M + bm - 1) / bm);
    const unsigned lds = (unsigned)(2 * (bm + 64) * 68 * sizeof(float));
    if (coresident) {
        // Round 3: 32-row tiles (two accumulators share every B operand) with a 2-deep register ring -- 54 VGPRs, so two of
        // its waves still share a SIMD with five encoder workgroups.  Beside an HBM-saturating kernel a co-runner pays for
        // the bytes it fetches from beyond L1 (experiment 17e): the weight block is fetched once per 32
        // rows instead of once per 16.  Step 303.9 -> 299.9 us (3 of 3 interleaved pairs); round 2's 32-row form had a 4-deep
        // ring and 68+ VGPRs (one wave per SIMD) and measured slower.  A single 16-row tile keeps the 16-row form.
        if (M > 16) {
            const dim3 g2((N + 63) / 64, (M + 31) / 32);
            hipLaunchKernelGGL((gemm_nt_direct_kernel<2, EPI, 2>), g2, dim3(256), 0, st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep);
        } else {
            hipLaunchKernelGGL((gemm_nt_direct_kernel<1, EPI>), grid, dim3(256), 0, st, A, lda, B, ldb, Bx, M, N,
                               n_main, K, C, ldc, ep);
        }
    } else if (four) {
        hipLaunchKernelGGL((gemm_nt_kernel<2, EPI, 2>), grid, dim3(256), lds, st, A, lda, B, ldb, Bx, M, N, n_main,
                           K, C, ldc, ep);
    } else {
        if (two)
            hipLaunchKernelGGL((gemm_nt_kernel<2, EPI>), grid, dim3(256), lds, st, A, lda, B, ldb, Bx, M, N, n_main,
                               K, C, ldc, ep);
        else
            hipLaunchKernelGGL((gemm_nt_kernel<1, EPI>), grid, dim3(256), lds, st, A, lda, B, ldb, Bx, M, N, n_main,
                               K, C, ldc, ep);
    }
}

}  // namespace

extern "C" {

size_t nsc_graph_workspace_bytes(int32_t n_nodes, int64_t n_edges)
{
    (void)n_edges;
    if (n_nodes <= 0) return 0;
    return align256((size_t)n_nodes * sizeof(int)) * 2;    // in-degree counts + fill cursors
}

int nsc_graph_build_csr(const int64_t *edge_index, int64_t E, int32_t N, const float *edge_attr,
                        int32_t edge_dim, int32_t *row_ptr, int32_t *src, int32_t *eid, float *loop_attr,
                        void *ws, size_t ws_bytes, void *stream_)
{
    if (N < 0 || E < 0 || edge_dim < 0 || edge_dim > NSC_GAT_MAX_EDGE_DIM) return NSC_EINVAL;
    if (N == 0) return NSC_OK;
    if (!row_ptr || !src || !eid || (E > 0 && !edge_index)) return NSC_EINVAL;
    if (edge_attr && edge_dim > 0 && !loop_attr) return NSC_EINVAL;
    const size_t need = nsc_graph_workspace_bytes(N, E);
    if (!ws || ws_bytes < need) return NSC_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    int *deg = static_cast<int *>(ws);
    int *cursor = reinterpret_cast<int *>(static_cast<char *>(ws) + need / 2);
    nsc_fill_u32(st, ws, 0u, (long long)(need / 4));
    const long long *ei = reinterpret_cast<const long long *>(edge_index);
    long long blocks = (E + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(csr_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ei, (long long)E, N, deg);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, deg, N, row_ptr);
    hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ei, (long long)E, N, row_ptr,
                       cursor, eid);
    hipLaunchKernelGGL(csr_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, ei, N,
                       (edge_dim > 0) ? edge_attr : nullptr, edge_dim, row_ptr, src, eid,
                       (edge_attr && edge_dim > 0) ? loop_attr : nullptr);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

int nsc_graph_band_entries(const NscGraph *g, const float *edge_attr, int32_t edge_dim, float *entries, int32_t *info,
                           void *stream_)
{
    if (!g || g->n_nodes < 0 || edge_dim < 0 || edge_dim > NSC_GAT_MAX_EDGE_DIM) return NSC_EINVAL;
    if (g->n_nodes == 0) return NSC_OK;
    if (!g->row_ptr || !g->src || !g->eid || !entries || !info) return NSC_EINVAL;
    if (edge_attr && edge_dim > 0 && !g->loop_attr) return NSC_EINVAL;
    if (reinterpret_cast<unsigned long long>(entries) & 15) return NSC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    nsc_fill_u32(st, info, 0u, 2);
    hipLaunchKernelGGL(band_entries_kernel, dim3((g->n_nodes + 255) / 256), dim3(256), 0, st, g->row_ptr, g->src, g->eid,
                       (edge_dim > 0) ? edge_attr : nullptr, edge_dim, g->loop_attr, g->n_nodes,
                       reinterpret_cast<f32x4 *>(entries), info);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_gat_folded_floats(const NscGatModel *m)
{
    if (check_model(m) != NSC_OK) return 0;
    return (size_t)m->n_layers * fold_stride(m->hidden);
}

int nsc_gat_fold_weights(const NscGatModel *m, float *folded, void *stream_)
{
    int stt = check_model(m);
    if (stt != NSC_OK) return stt;
    if (!folded) return NSC_EINVAL;
    PrepArgs pa;
    pa.H = m->hidden;
    pa.edge_dim = m->edge_dim;
    for (int l = 0; l < m->n_layers; ++l) {
        pa.l[l].w = m->layers[l].lin_w;
        pa.l[l].att_src = m->layers[l].att_src;
        pa.l[l].att_dst = m->layers[l].att_dst;
        pa.l[l].w_edge = m->layers[l].lin_edge_w;
        pa.l[l].att_edge = m->layers[l].att_edge;
    }
    hipLaunchKernelGGL(gat_fold_kernel, dim3(m->n_layers, 3), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       pa, folded);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

size_t nsc_gat_workspace_bytes(const NscGatModel *m, int32_t n_nodes)
{
    if (check_model(m) != NSC_OK || n_nodes <= 0) return 0;
    return gat_ws(m, n_nodes).total;
}

int nsc_gat_forward(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                    float *out, float *alpha_out, void *ws, size_t ws_bytes, void *stream_)
{
    return nsc_gat_forward_ex(m, g, x, edge_attr, out, alpha_out, ws, ws_bytes, 0u, stream_);
}

#ifdef NSC_DEV_TUNING
int nsc_debug_burn(int32_t mode, int32_t workgroups, int32_t per_wave, float *scratch, size_t scratch_bytes, void *stream_)
{
    if (mode < 0 || mode > 5 || workgroups < 0 || per_wave < 0 || !scratch || scratch_bytes < (1u << 20)) return NSC_EINVAL;
    if (workgroups == 0) return NSC_OK;
    hipLaunchKernelGGL(burn_kernel, dim3((unsigned)workgroups), dim3(256), 0, static_cast<hipStream_t>(stream_), mode, per_wave, scratch);
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}
#endif

int nsc_gat_gemm_tile(int32_t M, int32_t N, int32_t K, int32_t *tile_rows, int32_t *tile_cols, int32_t *lds_bytes,
                      int64_t *workgroups)
{
    if (M <= 0 || N <= 0 || K <= 0 || (K & 15)) return NSC_EINVAL;
    const int t = glds_pick_tile(M, N, K);
    const int acc = t & 15, bc = 1 + (t >> 4), nst = bc == 1 ? 3 : 2;
    if (tile_rows) *tile_rows = 16 * acc;
    if (tile_cols) *tile_cols = 64 * bc;
    if (lds_bytes) *lds_bytes = nst * (16 * acc + 64 * bc) * 256;
    if (workgroups) *workgroups = (int64_t)((N + 64 * bc - 1) / (64 * bc)) * ((M + 16 * acc - 1) / (16 * acc));
    return NSC_OK;
}

int nsc_gat_forward_ex(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                       float *out, float *alpha_out, void *ws, size_t ws_bytes, uint32_t flags, void *stream_)
{
    if (flags & ~(uint32_t)(NSC_GAT_CORESIDENT | NSC_GAT_SHARED_B | NSC_GAT_LDS_TILED | NSC_GAT_GENERIC)) return NSC_EINVAL;
    if ((flags & NSC_GAT_SHARED_B) && !(flags & NSC_GAT_CORESIDENT)) return NSC_EINVAL;
    if ((flags & NSC_GAT_LDS_TILED) && (flags & NSC_GAT_CORESIDENT)) return NSC_EINVAL;
    const int cores = (flags & NSC_GAT_CORESIDENT) ? ((flags & NSC_GAT_SHARED_B) ? 2 : 1) : (flags & NSC_GAT_LDS_TILED) ? 3 : 0;
    int stt = check_model(m);
    if (stt != NSC_OK) return stt;
    if (!g || g->n_nodes < 0) return NSC_EINVAL;
    const int N = g->n_nodes;
    if (N == 0) return NSC_OK;
    if (!x || !out || !g->row_ptr || !g->src || !g->eid || !m->folded) return NSC_EINVAL;
    const GatWs w = gat_ws(m, N);
    if (!ws || ws_bytes < w.total) return NSC_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    char *base = static_cast<char *>(ws);
    float *h0 = reinterpret_cast<float *>(base + w.h0), *h1 = reinterpret_cast<float *>(base + w.h1);
    float *G = reinterpret_cast<float *>(base + w.g);
    float *a_src = reinterpret_cast<float *>(base + w.a_src), *a_dst = reinterpret_cast<float *>(base + w.a_dst);
    const float *aux = m->folded;
    const int H = m->hidden, L = m->n_layers;
    const bool use_edge = m->edge_dim > 0 && edge_attr && g->loop_attr;      // model.py:126

    // input_proj + input_norm + relu                                       model.py:116-118
    GemmEpi ep = {};
    ep.bias = m->in_b;
    ep.bn_w = m->in_bn_w; ep.bn_b = m->in_bn_b; ep.bn_mean = m->in_bn_mean; ep.bn_var = m->in_bn_var;
    ep.bn_eps = m->bn_eps;
    ep.relu = 1;
    launch_gemm<1>(st, cores, x, m->in_dim, m->in_w, m->in_dim, nullptr, N, H, H, m->in_dim, h0, H, ep);

    // A banded graph (nsc_graph_band_entries vouches: sources within NSC_BAND_HALO rows of their target, at most
    // NSC_BAND_SLOTS entries per target) takes ONE launch per layer; its entries carry the edge attributes, edge_dim 2 only.
    const bool banded = !(flags & NSC_GAT_GENERIC) && cores == 0 && g->band_entries && g->band > 0 && g->band <= NSC_BAND_HALO &&
                        (!use_edge || m->edge_dim == 2);

    float *cur = h0, *nxt = h1;
    for (int l = 0; l < L; ++l) {
        const NscGatLayer &Ly = m->layers[l];
        const float *auxl = aux + (size_t)l * fold_stride(H);
        if (banded) {
            BandArgs b = {};
            b.A = cur; b.B = Ly.lin_w; b.Bx = auxl; b.M = N; b.H = H;
            b.ent = reinterpret_cast<const f32x4 *>(g->band_entries);
            b.v = use_edge ? auxl + 2 * H : nullptr;
            b.bias = Ly.bias;
            b.bn_w = Ly.bn_w; b.bn_b = Ly.bn_b; b.bn_mean = Ly.bn_mean; b.bn_var = Ly.bn_var;
            b.bn_eps = m->bn_eps; b.slope = m->negative_slope;
            b.relu = (l < L - 1);
            b.resid = (m->residual && l > 0 && l < L - 1) ? cur : nullptr;
            b.out = nxt;
            b.alpha_out = alpha_out ? alpha_out + (size_t)l * g->nnz : nullptr;
            if (launch_banded_layer(st, b)) {
                float *t = cur; cur = nxt; nxt = t;
                continue;
            }
        }
        // G = cur * W^T, plus a_src = cur . u_src and a_dst = cur . u_dst as columns H, H+1
        GemmEpi e2 = {};
        e2.aux0 = a_src;
        e2.aux1 = a_dst;
        launch_gemm<0>(st, cores, cur, H, Ly.lin_w, H, auxl, N, H + 2, H, H, G, H, e2);

        AggArgs a = {};
        a.row_ptr = g->row_ptr; a.src = g->src; a.eid = g->eid;
        a.loop_attr = use_edge ? g->loop_attr : nullptr;
        a.edge_attr = use_edge ? edge_attr : nullptr;
        a.v = use_edge ? auxl + 2 * H : nullptr;
        a.a_src = a_src; a.a_dst = a_dst; a.G = G;
        a.bias = Ly.bias;
        a.bn_w = Ly.bn_w; a.bn_b = Ly.bn_b; a.bn_mean = Ly.bn_mean; a.bn_var = Ly.bn_var;
        a.bn_eps = m->bn_eps;
        a.slope = m->negative_slope;
        a.relu = (l < L - 1);                                               // model.py:135-137
        a.resid = (m->residual && l > 0 && l < L - 1) ? cur : nullptr;      // model.py:140-141
        a.out = nxt;
        a.alpha_out = alpha_out ? alpha_out + (size_t)l * g->nnz : nullptr;
        a.N = N; a.H = H; a.edge_dim = m->edge_dim;
        const dim3 grid((N + 3) / 4), block(256);
        switch ((H + 255) / 256) {
        case 1:
            if (cores) hipLaunchKernelGGL((gat_aggregate_kernel<1, 4, false>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((gat_aggregate_kernel<1, 8, true>), grid, block, 0, st, a);
            break;
        case 2: hipLaunchKernelGGL((gat_aggregate_kernel<2, 8, true>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((gat_aggregate_kernel<3, 8, true>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((gat_aggregate_kernel<4, 8, true>), grid, block, 0, st, a); break;
        }
        float *t = cur; cur = nxt; nxt = t;
    }

    // output_proj + input residual                                          model.py:144-151
    GemmEpi e3 = {};
    e3.bias = m->out_b;
    if (m->residual && m->in_dim == m->out_dim) { e3.resid = x; e3.ldr = m->in_dim; }
    launch_gemm<2>(st, cores, cur, H, m->out_w, H, nullptr, N, m->out_dim, m->out_dim, H, out, m->out_dim, e3);
    if (m->residual && m->in_dim != m->out_dim) {
        // out += residual_proj(x): second GEMM accumulating through the residual epilogue
        GemmEpi e4 = {};
        e4.bias = m->res_b;
        e4.resid = out; e4.ldr = m->out_dim;
        launch_gemm<2>(st, cores, x, m->in_dim, m->res_w, m->in_dim, nullptr, N, m->out_dim, m->out_dim, m->in_dim, out,
                       m->out_dim, e4);
    }
    return hipGetLastError() == hipSuccess ? NSC_OK : NSC_ELAUNCH;
}

}  // extern "C"
