// The stand-alone f32-MFMA GEMM of the GNN path, C[M,N] = A[M,K] * B[N,K]^T with fused epilogues: LDS-DMA staged, wave-
// specialised (gemm_glds_kernel), plus what it shares with the other GEMM kernels of nsc_gat.hip (epilogue descriptor,
// XCD-aware tile order).  Included by nsc_gat.hip (inference forward, reference src/gnn/model.py:116,127,144) and by
// nsc_gat_train.hip (the same three projections in the training forward, src/gnn/trainer.py:205); everything lives in
// the including file's anonymous namespace (the includer has <hip/hip_runtime.h> and <atomic> before it opens that namespace).
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, dispatch), each with
// its own L2: with the natural order the 4 (or 13) column blocks that share an A row block -- and neighbouring
// target nodes that share their neighbour rows -- land on different XCDs and every one of them fetches the rows
// again from beyond L2.  Renumbering (bijective for any grid size) gives the workgroups of ONE XCD consecutive
// tiles.  Speed only: the result does not depend on placement.
__device__ __forceinline__ unsigned xcd_tile(unsigned id, unsigned n)
{
#ifdef NSC_DEV_NOREMAP
    return id;
#endif
    const unsigned xcd = id & 7u, q = n >> 3, r = n & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (id >> 3);
}

// Columns >= n_main of a GEMM come from the extra rows Bx (the folded attention vectors of the inference forward) and are
// written to aux0 / aux1 instead of C.
struct GemmEpi {
    const float *bias;                              // per column, nullable
    const float *bn_w, *bn_b, *bn_mean, *bn_var;    // BatchNorm eval, nullable as a group
    float bn_eps;
    int relu;
    const float *resid;                             // (M, ldr) added last, nullable
    int ldr;
    float *aux0, *aux1;                             // columns n_main, n_main+1 (M each), nullable
};

// EPI: 0 = plain store + aux columns (lin), 1 = bias + BatchNorm + ReLU (input_proj),
//      2 = bias + residual (output_proj / residual_proj)

// ---------------------------------------------------------------------------------------------
// Round 3: the stand-alone GEMM (one launch owns the chip).  What the round-2 ablations showed about gemm_nt_kernel:
// its skeleton without a single MFMA (register-ring loads, ds_write_b128 staging at 79 B/clk/CU, a barrier per chunk)
// takes 23 of input_proj's 35 us, the matrix pipe needs 12, and the two ADD UP: 2.2 workgroups per CU in lock step, wave
// quantisation on top (568 tiles cost what 768 do).  This kernel changes the decomposition, the staging and who does what:
//   * ONE round of workgroups where the shape allows it: the tile is (16 ACC) x 64 with ACC picked on the host
//     (glds_pick_tile) so that the grid is at most one workgroup per CU (M = 4 541: ACC = 5 -> 228 tiles for input_proj,
//     ACC = 6 -> 240 for lin; output_proj takes two rounds of ACC = 8).  A computing wave owns ACC accumulators (16 ACC rows
//     x 16 columns): ACC independent MFMA chains, ACC + 1 operand reads per 4 ACC MFMAs, and 2 (1/(16 ACC) + 1/64) B of
//     operand traffic per FLOP from L2 (0.056 at ACC = 5 against 0.094 for the 32 x 64 tiles of round 2).
//   * staging by LDS-DMA (global_load_lds_dwordx4): no register ring, no ds_write pass.  A wave-instruction writes 1 KiB
//     = 4 tile rows of one 64-deep chunk (256 B per row), lane-linear; bank conflicts are avoided by swizzling on the
//     SOURCE side: the 16-byte slot s of tile row R holds the k-quad s ^ (R & 15), the operand read of lane (r, q) for
//     k-block d takes slot (4 d + q) ^ r -- every 16-lane group of a ds_read_b128 hits 16 different slots.
//   * three LDS stages, ONE raw s_barrier per chunk, counted vmcnt: chunk c + 2 is issued right after the barrier that
//     retires chunk c, so one chunk stays in flight across every barrier (__syncthreads() would drain it).
//   * wave specialisation (512 threads): waves 0-3 only compute (operand reads + MFMAs), waves 4-7 only stage (LDS-DMA issue
//     + counted waits).  An LDS-DMA instruction costs the issuing wave 60-185 cycles of its in-order instruction stream
//     (MI355X_MICROARCH.md, cycle constants): with the 9 of a chunk issued by the computing waves themselves the same kernel
//     measured 24.7 us on input_proj at 4 541 rows, 22.8 us with them on waves of their own (round 2: 34.6).  The cyclic
//     wave -> SIMD placement gives every SIMD one wave of each kind.
// Every output element is still the chain chunk-ascending, d = 0..3, t = 0..3 of v_mfma_f32_16x16x4_f32 with operand
// element t of lane (r, q) = k 64 c + 16 d + 4 q + t: bit-identical to the other GEMM kernels (tools/native/gemm_glds_probe.hip
// compares every configuration with gemm_nt_kernel bit for bit).
// ---------------------------------------------------------------------------------------------
template <int N_> __device__ __forceinline__ void glds_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N_) : "memory");
}

// BKM (round 4, the training backward's dX = dY W products, BC = 1 only): B is K-MAJOR, B(n, k) = B[k * ldb + n] -- the weight as it lies
// in memory when the product contracts over its rows.  Its tile goes to LDS untransposed, [64 k][64 n] (one LDS-DMA quad = four
// consecutive n of one k; slot of n-quad nq in row k: nq ^ 4 ((k / 4) mod 4), so the four k rows 4 q + t of an operand read land in
// four bank groups), and lane (r, q) reads its operand element t = B[k = 16 d + 4 q + t][n] as one ds_read_b32: the same k for the
// same (d, q, t) as the k-contiguous form, so the chain per output element is unchanged -- no transposed copy of the weight.
template <int ACC, int EPI, int NST = 3, int BC = 1, bool BKM = false>   // tile (16 ACC) x (64 BC); a computing wave owns ACC x BC accumulators
__global__ __launch_bounds__(512) void gemm_glds_kernel(const float *__restrict__ A, int lda,
                                                           const float *__restrict__ B, int ldb,
                                                           const float *__restrict__ Bx, int M, int N,
                                                           int n_main, int K, float *__restrict__ C, int ldc,
                                                           GemmEpi ep)
{
    constexpr int BM = 16 * ACC, BN = 64 * BC, ROWS = BM + BN, NPW = ROWS / 16;   // NPW: LDS-DMA pieces (4 rows) per staging wave and chunk
    constexpr int STAGE = ROWS * 64;                                // floats per stage
    constexpr int LD = BN + 4;                                      // epilogue staging stride
    static_assert(BM * LD + 3 * BN <= NST * STAGE, "the epilogue tile (+ the folded BatchNorm of its columns) reuses the stages");
    static_assert(!BKM || BC == 1, "the k-major B tile is laid out for 64 columns");
    extern __shared__ __attribute__((aligned(1024))) float gemm_lds[];   // the ONLY LDS object of the kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const int r = lane & 15, q = lane >> 4;
    const unsigned tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int m0 = (int)(tile / gridDim.x) * BM, n0 = (int)(tile % gridDim.x) * BN;
    const int nchunks = (K + 63) >> 6;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[ACC][BC];
#pragma unroll
    for (int h = 0; h < ACC; ++h)
#pragma unroll
        for (int g = 0; g < BC; ++g) acc[h][g] = zero;
    const bool active = n0 + wave * 16 * BC < N;      // a computing wave whose columns all lie beyond N has nothing to do

    // epilogue operands: the bias loads of the computing waves go out at once and stay in flight across the main loop (their
    // barriers do not wait for vector memory); the staging waves fetch theirs after the last chunk.  The BatchNorm operands
    // of EPI 1 are folded by the staging waves after their last chunk and handed over through LDS (in every wave ahead of the
    // loop, or ahead of the barriers, their 20 loads per thread measured +0.8 us on input_proj)
    const int c4t = tid & (BN / 4 - 1);
    const int cg = n0 + 4 * c4t;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    float bn_pre[3] = {1.f, 0.f, 0.f};                             // EPI 1: scale, shift, bias of column n0 + tid - 256 (staging waves)
    const bool vec = (cg + 3 < n_main) && !(ldc & 3) && !(reinterpret_cast<unsigned long long>(C) & 15) &&
                     (EPI != 2 || !ep.resid || (!(ep.ldr & 3) && !(reinterpret_cast<unsigned long long>(ep.resid) & 15)));
    constexpr int RPP = 2048 / BN;                                 // rows per epilogue pass of the 512 threads (a float4 each)
    constexpr int NPASS = (BM + RPP - 1) / RPP;
    f32x4 rs[NPASS];                                               // this thread's residual quads (EPI 2): 14.5 MB of reads for
#pragma unroll                                                     // output_proj at 4 541 rows, hidden behind the main loop
    for (int pass = 0; pass < NPASS; ++pass) rs[pass] = zero;
    auto load_epilogue = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = (cg + j < N) ? cg + j : N - 1;
            if (EPI != 0) bias[j] = ep.bias ? ep.bias[col] : 0.0f;     // (EPI 2 without a bias: the training path's plain / accumulating GEMMs)
        }
    };
    // The residual quads are fetched right after a wave's last chunk: the staging waves are done a chunk before the
    // computing waves, so half of these reads land under the last MFMAs.  (Measured and dropped: with the bias at the start
    // of the kernel +1.3 us on output_proj -- they queue ahead of the first chunks; in the computing waves' second-to-last
    // chunk +7 us -- the pinned read / MFMA order of the loop does not survive a block of ordinary loads.)
    auto load_residual = [&]() {
        if (EPI == 2 && ep.resid && vec) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int lr = pass * RPP + tid / (BN / 4), row = m0 + lr;
                if (lr < BM && row < M) rs[pass] = *reinterpret_cast<const f32x4 *>(ep.resid + (long long)row * ep.ldr + cg);
            }
        }
    };

    if (wave8 >= 4) {
        // ---- staging waves: piece j of a chunk = tile rows 4 (wave + 4 j) .. + 3, lane -> row + lane / 16, k-quad (lane % 16) ^ (row % 16)
        const float *src[NPW];
        const int kq_base = lane & 15;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int R = 4 * (wave + 4 * j) + (lane >> 4);
            const int kq = kq_base ^ (R & 15);
            const float *p;
            if (4 * (wave + 4 * j) < BM) {                          // wave-uniform: a piece is all A or all B
                const int gr = m0 + R;
                p = A + (long long)(gr < M ? gr : M - 1) * lda;
            } else if (!BKM) {
                int gc = n0 + R - BM;
                gc = gc < N ? gc : N - 1;
                p = (gc < n_main) ? B + (long long)gc * ldb : Bx + (long long)(gc - n_main) * ldb;
            } else {
                // k-major B: this piece holds k rows R - BM .. of the chunk, the lane its row (lane / 16) and physical slot lane % 16
                const int kr = R - BM;                              // k row within the chunk: 4 (piece) + lane / 16
                int gn = n0 + 4 * ((lane & 15) ^ (((kr >> 2) & 3) << 2));
                gn = gn + 3 < N ? gn : N - 4;                       // quads past the matrix re-read valid columns (never stored)
                src[j] = B + (long long)kr * ldb + gn;              // (+ chunk row offset at issue time)
                continue;
            }
            src[j] = p + 4 * kq;
        }
        auto issue = [&](int ch, int stage) {
            const int kn = ch << 6;
            float *dst0 = gemm_lds + stage * STAGE + wave * 256;
            if (kn + 64 <= K) {                                     // workgroup-uniform
#pragma unroll
                for (int j = 0; j < NPW; ++j) {
                    const bool brow = BKM && 4 * (wave + 4 * j) >= BM;     // a k-major B piece: the chunk is kn ROWS further down
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (brow ? (long long)kn * ldb : (long long)kn)),
                                                     (__attribute__((address_space(3))) void *)(dst0 + j * 1024), 16, 0, 0);
                }
            } else {
                // a short last chunk (K % 64 != 0): quads past K re-read quad 0 of the chunk (those k-blocks are skipped)
#pragma unroll
                for (int j = 0; j < NPW; ++j) {
                    const int R = 4 * (wave + 4 * j) + (lane >> 4);
                    const int kq = kq_base ^ (R & 15);
                    const float *g = src[j] + kn - ((kn + 4 * kq + 4 <= K) ? 0 : 4 * kq);
                    if (BKM && 4 * (wave + 4 * j) >= BM) {          // k-major B piece: rows past K re-read the last row (skipped k-blocks)
                        const int kr = R - BM;
                        g = src[j] + (long long)(kn + kr < K ? kn : K - 1 - kr) * ldb;
                    }
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)(dst0 + j * 1024), 16, 0, 0);
                }
            }
        };
        issue(0, 0);
        if (NST == 3 && 1 < nchunks) issue(1, 1);
        // (not ahead of the loop: with ordinary loads pending beside LDS-DMA hipcc drains the queue inside the loop -- measured
        // +1 us on input_proj; the staging waves have nothing else to do after their last chunk anyway)
        for (int c = 0; c < nchunks; ++c) {
            // barrier c: this wave's pieces of chunk c have landed (with three stages at most the younger chunk's are
            // outstanding), and every computing wave is done reading chunk c - 1, whose stage the next issue refills
            if (NST == 3 && c + 1 < nchunks) glds_wait_barrier<NPW>();
            else glds_wait_barrier<0>();
#if defined(NSC_GLDS_ABL) && (NSC_GLDS_ABL & 2)                               // ablation build: no refills after the prologue
            continue;
#endif
            if (c + NST - 1 < nchunks) issue(c + NST - 1, (c + NST - 1) % NST);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (EPI != 1) load_epilogue();
        load_residual();
        if (EPI == 1 && tid < 256 + BN) {
            // BatchNorm (eval) of this tile's columns, folded to a scale and a shift by the staging waves, idle from their last
            // chunk on; handed over through LDS behind the tile (the 20 loads per thread that every wave used to issue after
            // the second epilogue barrier were an exposed round trip).  torch batch_norm eval: alpha = invstd * weight,
            // beta = bias - mean * alpha.
            const int cl = tid - 256, col = (n0 + cl < N) ? n0 + cl : N - 1;
            const float invstd = 1.0f / sqrtf(ep.bn_var[col] + ep.bn_eps);
            bn_pre[0] = invstd * ep.bn_w[col];
            bn_pre[1] = ep.bn_b[col] - ep.bn_mean[col] * bn_pre[0];
            bn_pre[2] = ep.bias[col];
        }
    } else {
        // ---- computing waves
        const int boff = (BM + 16 * BC * wave + r) * 64, aoff = r * 64;
        // (k-major B: element t of the operand = row 16 d + 4 q + t of the [64 k][64 n] tile, slot (n / 4) ^ 4 q, element n % 4)
        const int bkoff = BM * 64 + 4 * (((16 * wave + r) >> 2) ^ (q << 2)) + (r & 3) + 4 * q * 64;
        auto frags = [&](const float *st, int d, f32x4 (&bv)[BC], f32x4 (&av)[ACC]) {
            const int slot = 4 * ((4 * d + q) ^ r);
            if (BKM) {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[0][t] = st[bkoff + (16 * d + t) * 64];
            } else
#pragma unroll
            for (int g = 0; g < BC; ++g) bv[g] = *reinterpret_cast<const f32x4 *>(&st[boff + g * 1024 + slot]);
#pragma unroll
            for (int h = 0; h < ACC; ++h) av[h] = *reinterpret_cast<const f32x4 *>(&st[aoff + h * 1024 + slot]);
        };
        auto mfmas = [&](const f32x4 (&bv)[BC], const f32x4 (&av)[ACC], int t0, int t1) {
#pragma unroll
            for (int t = t0; t < t1; ++t)
#pragma unroll
                for (int h = 0; h < ACC; ++h)
#pragma unroll
                for (int g = 0; g < BC; ++g) {
#if defined(NSC_GLDS_ABL) && (NSC_GLDS_ABL & 1)                                       // ablation build: no MFMAs, operands kept alive
                    acc[h][g][t] += av[h][t] * bv[g][t];
                    continue;
#endif
                    acc[h][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][t], bv[g][t], acc[h][g], 0, 0, 0);
                }
        };
        if (EPI != 1) load_epilogue();
        asm volatile("s_barrier" ::: "memory");                    // barrier 0 (no LDS-DMA of this wave's to wait for)
        f32x4 bv0[BC], bv1[BC], av0[ACC], av1[ACC];
        if (active) frags(gemm_lds, 0, bv0, av0);
        // One iteration = one 64-deep chunk = 16 ACC MFMAs.  The operand reads of k-block d + 1 go out between the two halves
        // of the MFMAs of block d (the scheduler would sink every read to just before its first use -- seen in the ISA --
        // hence the pinned order); the hand-over to the next chunk sits between the halves of the LAST block: by then this
        // wave has read everything it needs from the stage (lgkmcnt(0)), after the barrier the next chunk's first operands
        // are fetched and the second half of the MFMAs covers their latency.
#ifdef NSC_GLDS_CLOCK                                     // diagnostic build of the probe: in-kernel clock of the main loop
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int ch = 0; ch < nchunks; ++ch) {
            const float *st = gemm_lds + (ch % NST) * STAGE;
            const float *nx = gemm_lds + ((ch + 1) % NST) * STAGE;
            const int kleft = K - (ch << 6);
            const bool last = ch + 1 == nchunks;
            if (kleft >= 64) {
                if (active) {
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(bv0, av0, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    frags(st, 1, bv1, av1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(bv0, av0, 2, 4);
                    mfmas(bv1, av1, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    frags(st, 2, bv0, av0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(bv1, av1, 2, 4);
                    mfmas(bv0, av0, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    frags(st, 3, bv1, av1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(bv0, av0, 2, 4);
                    mfmas(bv1, av1, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!last) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier ch + 1
                    if (active) frags(nx, 0, bv0, av0);
                }
                if (active) {
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(bv1, av1, 2, 4);
                }
            } else {                                               // the short last chunk (K % 64 != 0)
                if (active) {
                    mfmas(bv0, av0, 0, 4);
                    for (int d = 1; 16 * d < kleft; ++d) {
                        frags(st, d, bv0, av0);
                        mfmas(bv0, av0, 0, 4);
                    }
                }
            }
        }
        load_residual();
#ifdef NSC_GLDS_CLOCK
        if (wave8 == 0 && lane == 0 && ep.aux0 && ep.aux1) {       // shader cycles and 100 MHz ticks of this workgroup's main loop
            reinterpret_cast<int *>(ep.aux0)[tile] = (int)(__builtin_amdgcn_s_memtime() - clk0);
            reinterpret_cast<int *>(ep.aux1)[tile] = (int)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
    }
    __syncthreads();                                               // nothing in flight, all reads done

    // epilogue as in gemm_nt_kernel: the tile goes through LDS, rows leave as 256 contiguous bytes; all 8 waves store
    float *Cs = gemm_lds;
    if (EPI == 1 && tid >= 256 && tid < 256 + BN) {
        Cs[BM * LD + tid - 256] = bn_pre[0];
        Cs[BM * LD + BN + tid - 256] = bn_pre[1];
        Cs[BM * LD + 2 * BN + tid - 256] = bn_pre[2];
    }
    if (wave8 < 4 && active) {
#pragma unroll
        for (int h = 0; h < ACC; ++h)
#pragma unroll
            for (int g = 0; g < BC; ++g)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)                  // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
                    Cs[(16 * h + 4 * q + reg) * LD + (wave * BC + g) * 16 + r] = acc[h][g][reg];
    }
    __syncthreads();
    float bn_scale[4] = {1.f, 1.f, 1.f, 1.f}, bn_shift[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == 1) {
        const float *pre = Cs + BM * LD;                           // 3 x BN floats behind the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) { bn_scale[j] = pre[4 * c4t + j]; bn_shift[j] = pre[BN + 4 * c4t + j]; bias[j] = pre[2 * BN + 4 * c4t + j]; }
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int lr = pass * RPP + tid / (BN / 4), row = m0 + lr;
        if (lr >= BM || row >= M) continue;
        const f32x4 t = *reinterpret_cast<const f32x4 *>(&Cs[lr * LD + 4 * c4t]);
        float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (EPI != 0) v[j] = v[j] + bias[j];
            if (EPI == 1) v[j] = fmaxf(v[j] * bn_scale[j] + bn_shift[j], 0.0f);
        }
        if (vec) {
            if (EPI == 2 && ep.resid) { v[0] += rs[pass].x; v[1] += rs[pass].y; v[2] += rs[pass].z; v[3] += rs[pass].w; }
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

// One configuration of gemm_glds_kernel.  Above 64 KB of dynamic LDS a kernel has to be opted in, and the attribute is
// per DEVICE: one atomic per (instantiation, device) -- 0 not tried, 1 opted in, 2 refused.  Returns false when the
// configuration cannot run here (the caller then takes gemm_nt_kernel: same results, bit for bit).
template <int ACC, int EPI, int NST = 3, int BC = 1, bool BKM = false>
bool launch_glds_cfg(hipStream_t st, const float *A, int lda, const float *B, int ldb, const float *Bx, int M, int N,
                     int n_main, int K, float *C, int ldc, const GemmEpi &ep)
{
    constexpr unsigned lds = NST * (16 * ACC + 64 * BC) * 256;
    static_assert(lds <= 160 * 1024, "LDS of a CU");
    if (lds > 64 * 1024) {
        static std::atomic<int> opted[16];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
        int s = opted[dev].load(std::memory_order_acquire);
        if (s == 0) {
            s = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_glds_kernel<ACC, EPI, NST, BC, BKM>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : 2;
            opted[dev].store(s, std::memory_order_release);
        }
        if (s != 1) return false;
    }
    const dim3 grid((N + 64 * BC - 1) / (64 * BC), (M + 16 * ACC - 1) / (16 * ACC));
    hipLaunchKernelGGL((gemm_glds_kernel<ACC, EPI, NST, BC, BKM>), grid, dim3(512), lds, st, A, lda, B, ldb, Bx, M, N, n_main, K, C,
                       ldc, ep);
    return true;
}

// Tile of the stand-alone GEMM, (16 ACC) rows x (64 BC) columns: the grid should be ONE round of at most a workgroup per CU
// (256), and among such grids the one with the least work per workgroup; when no tile gives one round, the cost is
// rounds x (MFMA time of a tile + what a round costs besides: first operand round trip, epilogue, ramp).  BC = 2 (two
// stages: 2 x (16 ACC + 128) x 256 B of LDS) gives mid-sized wide outputs one round (output_proj at 2 500 rows: 224 tiles
// of 80 x 128, 17.0 us against 21.9 for round 2's kernel).  At 4 541 rows output_proj would need 128 x 128 tiles for that
// (252 of them): measured 26.1 us against 26.5 for two rounds of 128 x 64 -- and 64 + 80 accumulator and operand registers
// plus the compiler's copies do not fit 256: the instantiation spilled inside the loop, so it is not offered
// (tests/test_abi_cpu.py::test_glds_gemm_code_objects holds every instantiation to 0 B of scratch).  Returns ACC + 16 (BC - 1).
inline int glds_pick_tile(int M, int N, int K, int max_bc = 2)
{
    const long long nch = (K + 63) / 64;
    int best = 1;
    long long best_cost = -1;
    for (int bc = 1; bc <= max_bc; ++bc) {
        const long long ncb = (N + 64 * bc - 1) / (64 * bc);
        for (int a = 1; a <= (bc == 1 ? 8 : 7); ++a) {             // 128 x 128 tiles (a = 8, bc = 2) do not fit 256 registers: they spill
            const long long tiles = ncb * ((M + 16 * a - 1) / (16 * a));
            const long long rounds = (tiles + 255) / 256;
            const long long cost = rounds * (a * bc * nch * 512 + 3000);    // cycles: 16 MFMAs of 32 per chunk and accumulator
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = a + 16 * (bc - 1); }
        }
    }
    return best;
}

template <int EPI>
bool launch_glds(hipStream_t st, const float *A, int lda, const float *B, int ldb, const float *Bx, int M, int N,
                 int n_main, int K, float *C, int ldc, const GemmEpi &ep)
{
    // LDS-DMA moves 16 bytes per lane: rows must be 16-byte aligned at every k-quad
    if ((lda & 3) || (ldb & 3) || (reinterpret_cast<unsigned long long>(A) & 15) ||
        (reinterpret_cast<unsigned long long>(B) & 15) || (Bx && (reinterpret_cast<unsigned long long>(Bx) & 15)) || (K & 15))
        return false;
#define NSC_GLDS_CASE(a)                                                                                              \
    case a: return launch_glds_cfg<a, EPI, 3, 1>(st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep);               \
    case 16 + a: return launch_glds_cfg<a, EPI, 2, 2>(st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep);
    switch (glds_pick_tile(M, N, K)) {
        NSC_GLDS_CASE(1) NSC_GLDS_CASE(2) NSC_GLDS_CASE(3) NSC_GLDS_CASE(4)
        NSC_GLDS_CASE(5) NSC_GLDS_CASE(6) NSC_GLDS_CASE(7)
        case 8: return launch_glds_cfg<8, EPI, 3, 1>(st, A, lda, B, ldb, Bx, M, N, n_main, K, C, ldc, ep);
    }
#undef NSC_GLDS_CASE
    return false;
}


// C = A * B for a K-MAJOR B (B[k * ldb + n]): the dX = dY W products of the training backward, no transposed copy of the weight
// (EPI 2: plain / accumulating store).  false: the operands do not fit (the caller transposes and takes launch_glds).
inline bool launch_glds_bkm(hipStream_t st, const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C,
                            int ldc, const GemmEpi &ep)
{
    if ((lda & 3) || (ldb & 3) || (N & 3) || N < 4 || (K & 15) || (reinterpret_cast<unsigned long long>(A) & 15) ||
        (reinterpret_cast<unsigned long long>(B) & 15))
        return false;
#define NSC_GLDS_BKM_CASE(a) case a: return launch_glds_cfg<a, 2, 3, 1, true>(st, A, lda, B, ldb, nullptr, M, N, N, K, C, ldc, ep);
    switch (glds_pick_tile(M, N, K, 1)) {
        NSC_GLDS_BKM_CASE(1) NSC_GLDS_BKM_CASE(2) NSC_GLDS_BKM_CASE(3) NSC_GLDS_BKM_CASE(4)
        NSC_GLDS_BKM_CASE(5) NSC_GLDS_BKM_CASE(6) NSC_GLDS_BKM_CASE(7) NSC_GLDS_BKM_CASE(8)
    }
#undef NSC_GLDS_BKM_CASE
    return false;
}
