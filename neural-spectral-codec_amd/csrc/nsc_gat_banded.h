// One GATConv layer of a BANDED graph as ONE launch (round 4): lin GEMM + attention logits + per-target softmax +
// alpha-weighted aggregation + bias + BatchNorm(eval) (+ ReLU, + middle-layer residual).
//   SpectralGNN.forward, layer loop        src/gnn/model.py:124-141
//   torch_geometric 2.4.0 GATConv          src/gnn/model.py:16,75-84,127
//   the temporal chain it runs on          src/keyframe/graph_manager.py:520-532 (abs(i - j) <= M // 2 = 2)
//
// The temporal graph of every reference caller is banded: a target's sources lie within HALO = 2 rows of it.  A row tile of
// h = x W^T plus a 2-row halo either side therefore holds every neighbour row the tile's targets aggregate -- the fact
// distributed.halo_window exploits across GPUs, used here inside one.  A workgroup owns (16 ACC - 4) target rows x 64 output
// columns: it computes h for 16 ACC rows (the owned rows + halo) x 64 columns on gemm_glds_kernel's main loop (LDS-DMA staging
// with source-side swizzle, three stages, one raw s_barrier per chunk, four computing + four staging waves + two more).  h stays in
// LDS: no gat_aggregate_kernel launch, no round trip of h (4.65 MB written + 5 x read at 4 541 keyframes) through L2.
//
// The two attention columns a_src = x . u_src, a_dst = x . u_dst (nsc_gat_fold_weights) of the same rows come off the VALU of
// two ATTENTION waves (waves 8, 9 of the 640-thread workgroup), one row per lane, from the A chunks the computing waves
// multiply: v_mfma_f32_16x16x4_f32 is exactly the sequential fused-multiply-add chain over its four k terms in lane-group
// (q) order -- measured: tools/native/mfma_fma_probe.hip, 409 600 outputs of 256-deep chains over four operand distributions
// incl. denormal products, 0 differ (the reversed order differs on 54 %) -- so c = fma(x[k], u[k], c) over
// k = 64 c + 16 d + 4 q + t in the order (c, d, t, q) gives the bits the generic kernels get from a fifth 16-column MFMA block
// of which 14 columns are waste (a quarter more matrix-pipe time per tile, unevenly dealt: 7 / 6 / 6 / 6 accumulators per
// wave at 80 rows).  What the chains cost was measured form by form (tools/native/banded_probe.hip, per layer at 4 541
// keyframes, in-kernel clock; without any chain the kernel takes 12.0 us and its main loop runs at the MFMA issue rate,
// 10.2 k cycles): the MFMA block 14.6 us; the chains on the STAGING waves 17.5 us (they delay the LDS-DMA issue); on two
// extra waves fed by scalar loads 15.4 us (five exposed round trips per chunk), by v_readlane from a VGPR 14.2 us, by scalar
// loads a block ahead 14.0 us, with all 16 LDS reads of a chunk issued up front 13.9 us, + s_setprio 13.6 us (this form);
// INTERLEAVED with the MFMAs of the computing waves themselves 15.3 us (main loop 19.3 k cycles: a VALU instruction between
// two MFMAs of a wave costs the matrix pipe its full duration).  In every two-wave form the main loop takes ~14.9 k cycles:
// a dependent VALU chain in a wave that shares its SIMD with a wave saturating the matrix pipe advances one step per ~58
// cycles whatever feeds it.
//
// Bit-identical to the generic kernel sets by construction: every h element is the same MFMA chain (chunk ascending,
// d = 0..3, t = 0..3, operand element t of lane (r, q) = k 64 c + 16 d + 4 q + t), a_src / a_dst the same chain as fmas; the
// softmax reduces with the same xor butterfly (a 16-lane group here, the 64-lane wave there: with at most 16 entries the
// upper levels of the 64-lane butterfly only add zeros); the aggregation is the same fma chain in CSR entry order; the
// epilogue is the same expression.  tests/test_gat_gpu.py compares the sets bit for bit.
//
// The graph comes as banded entries (nsc_graph_band_entries, built once per graph next to the CSR): 8 slots of 16 bytes per
// target {source node, edge_attr[0], edge_attr[1], CSR entry index (-1 = empty slot)} in CSR order (self loop last, its
// attributes = the mean of the incoming ones), so the kernel fetches a target's whole neighbourhood in ONE round trip issued
// ahead of the main loop instead of the three dependent ones of the CSR walk (row_ptr -> src / eid -> edge_attr).
#pragma once

#define NSC_BAND_HALO 2
#define NSC_BAND_SLOTS 8

struct BandArgs {
    const float *A;              // (M, H) layer input, row stride H
    const float *B;              // (H, H) lin_src.weight
    const float *Bx;             // folded [u_src H][u_dst H]
    int M, H;
    const f32x4 *ent;            // (M, 8) banded entries
    const float *v;              // folded edge vector (2 floats) or null: no edge term
    const float *bias, *bn_w, *bn_b, *bn_mean, *bn_var;
    const float *resid;          // (M, H) or null
    float *out;                  // (M, H)
    float *alpha_out;            // (nnz) or null
    float bn_eps, slope;
    int relu;
#ifdef NSC_BAND_CLOCK
    unsigned long long *clk;     // diagnostic build of tools/native/banded_probe.hip: (tiles, 8) s_memtime stamps
#endif
};

#ifdef NSC_BAND_CLOCK
#define NSC_BAND_STAMP(slot) a.clk[(size_t)tile * 8 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define NSC_BAND_STAMP(slot)
#endif
#ifndef NSC_BAND_ABL
#define NSC_BAND_ABL 0           // ablation builds of the probe (timings only): 1 no aggregation, 2 no attention chains, 4 no entries
#endif

// counted wait of the staging waves: `pieces` LDS-DMA pieces of younger chunks may stay in flight across the barrier
template <int NPW> __device__ __forceinline__ void band_wait_barrier(int younger_chunks)
{
    switch (younger_chunks) {
    case 0: glds_wait_barrier<0>(); break;
    case 1: glds_wait_barrier<NPW>(); break;
    case 2: glds_wait_barrier<2 * NPW>(); break;
    default: glds_wait_barrier<3 * NPW>(); break;
    }
}

// xor butterflies inside a 16-lane DPP row without the LDS pipe (__shfl_xor compiles to ds_bpermute_b32: a dependent chain of
// eight LDS round trips per target pass): lane ^ 8 = row_ror:8, lane ^ 4 = row_half_mirror then quad_perm [3,2,1,0]
// (7 - i = i ^ 7, then ^ 3), lane ^ 2 / ^ 1 = quad_perm [2,3,0,1] / [1,0,3,2].  Same partners, same order of additions as
// the 64-lane butterfly of gat_aggregate_kernel (whose upper levels add zeros when a target has at most 16 entries).
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float xor8(float v) { return dpp_f<0x128>(v); }
__device__ __forceinline__ float xor4(float v) { return dpp_f<0x1B>(dpp_f<0x141>(v)); }
__device__ __forceinline__ float xor2(float v) { return dpp_f<0x4E>(v); }
__device__ __forceinline__ float xor1(float v) { return dpp_f<0xB1>(v); }
__device__ __forceinline__ float group16_max(float v)
{
    v = fmaxf(v, xor8(v));
    v = fmaxf(v, xor4(v));
    v = fmaxf(v, xor2(v));
    return fmaxf(v, xor1(v));
}
__device__ __forceinline__ float group16_sum(float v)
{
    v += xor8(v);
    v += xor4(v);
    v += xor2(v);
    return v + xor1(v);
}

template <int ACC, int NST = 3>
__global__ __launch_bounds__(640) void gat_layer_banded_kernel(BandArgs a, const float *__restrict__ ux)
{
    constexpr int HALO = NSC_BAND_HALO, SLOTS = NSC_BAND_SLOTS;
    constexpr int BM = 16 * ACC, OWN = BM - 2 * HALO;              // rows computed / rows owned
    constexpr int BNR = 64;                                        // B rows of a stage: 64 columns of W
    constexpr int ROWS = BM + BNR, NPW = ROWS / 16, STAGE = ROWS * 64;
    constexpr int LD = 68;                                         // h tile row stride (floats)
    constexpr int NE = (OWN * SLOTS + 255) / 256;                  // banded entries per computing thread
    // epilogue map over the (then free) stages, in floats: h tile | a_src | a_dst | folded BatchNorm of the 64 columns | entries
    constexpr int O_AS = BM * LD, O_AD = O_AS + BM, O_BN = O_AD + BM, O_EN = O_BN + 192;
    static_assert(O_EN + OWN * SLOTS * 4 <= NST * STAGE, "the epilogue reuses the stages");
    extern __shared__ __attribute__((aligned(1024))) float gemm_lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const int r = lane & 15, q = lane >> 4;
    const unsigned tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int trow = (int)(tile / gridDim.x), n0 = (int)(tile % gridDim.x) * 64;
    const int own0 = trow * OWN, rowbase = own0 - HALO;            // tile row R <-> node rowbase + R
    const int M = a.M, K = a.H;
    const int nchunks = K >> 6;                                    // H is a multiple of 64 (host check)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    if (tid == 0) NSC_BAND_STAMP(0);
    f32x4 acc[ACC];
#pragma unroll
    for (int h = 0; h < ACC; ++h) acc[h] = zero;
    f32x4 pe[NE];                                                  // prefetched entries (computing waves)
    float bn_pre[3] = {1.f, 0.f, 0.f};
    // attention chains: lane l of attention wave 8 + a owns tile row 64 a + l (16 consecutive lanes = 16 consecutive rows:
    // their swizzled reads of one k-quad hit 16 different slots); dead lanes repeat row 0
    const int xrow = 64 * (wave8 - 8) + lane;
    const bool xlive = wave8 >= 8 && xrow < BM;
    float cs = 0.0f, cd = 0.0f;

    if (wave8 >= 8) {
        // ---- attention waves: after barrier c chunk c has landed for every wave; this lane's row of it joins the two chains
        // in the MFMA's k order (c, d, t, q), done before barrier c + 1, after which the stage may be refilled.  All 16
        // quads of the row go out right after the barrier (one exposed LDS round trip per chunk: the reads queue behind the
        // computing waves' operand reads); u costs no memory access inside a chunk -- lane l holds u[64 c + l] in a VGPR (one
        // coalesced load per vector, fetched a chunk ahead) and v_readlane broadcasts element k into the fma.
        __builtin_amdgcn_s_setprio(3);
        float usn = ux[lane], udn = ux[K + lane];
        for (int c = 0; c < nchunks; ++c) {
            const float usv = usn, udv = udn;
            if (c + 1 < nchunks) {
                usn = ux[((c + 1) << 6) + lane];
                udn = ux[K + ((c + 1) << 6) + lane];
            }
            asm volatile("s_barrier" ::: "memory");                // barrier c
            const float *st = gemm_lds + (c % NST) * STAGE + (xlive ? xrow : 0) * 64;
            f32x4 aq[16];
#pragma unroll
            for (int kq = 0; kq < 16; ++kq) aq[kq] = *reinterpret_cast<const f32x4 *>(&st[4 * (kq ^ (xrow & 15))]);
            if (!(NSC_BAND_ABL & 2)) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int k = 16 * d + 4 * qq + t;
                            const float su = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(usv), k));
                            const float sd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(udv), k));
                            cs = __builtin_fmaf(aq[4 * d + qq][t], su, cs);
                            cd = __builtin_fmaf(aq[4 * d + qq][t], sd, cd);
                        }
            }
        }
        if (tid == 512) NSC_BAND_STAMP(7);
    } else if (wave8 >= 4) {
        // ---- staging waves (as gemm_glds_kernel): piece j = tile rows 4 (wave + 4 j) .. + 3 of a 64-deep chunk
        const float *src[NPW];
        const int kq_base = lane & 15;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int R = 4 * (wave + 4 * j) + (lane >> 4);
            const int kq = kq_base ^ (R & 15);
            const float *p;
            if (4 * (wave + 4 * j) < BM) {                          // wave-uniform: a piece is all A or all B
                int gr = rowbase + R;                               // halo rows outside the graph re-read a valid row: no
                gr = gr < 0 ? 0 : (gr < M ? gr : M - 1);            // target inside the graph has a source there
                p = a.A + (long long)gr * K;
            } else {
                p = a.B + (long long)(n0 + R - BM) * K;
            }
            src[j] = p + 4 * kq;
        }
        auto issue = [&](int ch, int stage) {
            float *dst0 = gemm_lds + stage * STAGE + wave * 256;
#pragma unroll
            for (int j = 0; j < NPW; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (ch << 6)),
                                                 (__attribute__((address_space(3))) void *)(dst0 + j * 1024), 16, 0, 0);
        };
        // NST stages: chunks 0 .. NST - 2 go out at once, chunk c + NST - 1 right after the barrier that retires chunk c - 1's
        // stage; the wait for chunk c leaves the younger chunks' pieces in flight.  With NST = 4 and K = 256 (four chunks:
        // the reference's hidden size) three of the four chunks are in flight from the start -- a 4-chunk K has no steady
        // state, what bounds it is how soon the operand bytes arrive.
#pragma unroll
        for (int c0 = 0; c0 < NST - 1; ++c0)
            if (c0 < nchunks) issue(c0, c0);
        if (tid == 256) NSC_BAND_STAMP(5);
        for (int c = 0; c < nchunks; ++c) {
            const int issued = c + NST - 1 < nchunks ? c + NST - 1 : nchunks;       // chunks issued so far
            band_wait_barrier<NPW>(issued - 1 - c);
            if (c + NST - 1 < nchunks) issue(c + NST - 1, (c + NST - 1) % NST);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 256) NSC_BAND_STAMP(6);
        if (tid < 256 + 64) {
            // BatchNorm (eval) of this tile's 64 columns folded to scale / shift by the staging waves, idle from their last
            // chunk on (gat_aggregate_kernel's expression: sc = invstd * w, shift = b - mean * sc)
            const int col = n0 + tid - 256;
            bn_pre[2] = a.bias[col];
            if (a.bn_w) {
                const float invstd = 1.0f / sqrtf(a.bn_var[col] + a.bn_eps);
                bn_pre[0] = invstd * a.bn_w[col];
                bn_pre[1] = a.bn_b[col] - a.bn_mean[col] * bn_pre[0];
            }
        }
    } else {
        // ---- computing waves
        // the tile's banded entries: one 16-byte load per (target, slot), in flight across the main loop
#pragma unroll
        for (int c = 0; c < NE; ++c) {
            const int idx = tid + 256 * c, lo = idx >> 3;
            const int i = own0 + lo;
            f32x4 e = {0.f, 0.f, 0.f, __int_as_float(-1)};
            if (lo < OWN && i < M && !(NSC_BAND_ABL & 4)) e = a.ent[(long long)i * SLOTS + (idx & 7)];
            pe[c] = e;
        }
        const int boff = (BM + 16 * wave + r) * 64, aoff = r * 64;
        auto frags = [&](const float *st, int d, f32x4 &bv, f32x4 (&av)[ACC]) {
            const int slot = 4 * ((4 * d + q) ^ r);
            bv = *reinterpret_cast<const f32x4 *>(&st[boff + slot]);
#pragma unroll
            for (int h = 0; h < ACC; ++h) av[h] = *reinterpret_cast<const f32x4 *>(&st[aoff + h * 1024 + slot]);
        };
        auto mfmas = [&](const f32x4 &bv, const f32x4 (&av)[ACC], int t0, int t1) {
#pragma unroll
            for (int t = t0; t < t1; ++t)
#pragma unroll
                for (int h = 0; h < ACC; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][t], bv[t], acc[h], 0, 0, 0);
        };
        asm volatile("s_barrier" ::: "memory");                    // barrier 0
        if (tid == 0) NSC_BAND_STAMP(1);
        f32x4 bv0, bv1, av0[ACC], av1[ACC];
        frags(gemm_lds, 0, bv0, av0);
        for (int ch = 0; ch < nchunks; ++ch) {
            const float *st = gemm_lds + (ch % NST) * STAGE;
            const float *nx = gemm_lds + ((ch + 1) % NST) * STAGE;
            const bool last = ch + 1 == nchunks;
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
            if (!last) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // barrier ch + 1
                frags(nx, 0, bv0, av0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(bv1, av1, 2, 4);
        }
        if (tid == 0) NSC_BAND_STAMP(2);
    }
    // the residual quads of this thread's (target, column quad) pairs: requested here, a barrier pair and the softmax ahead
    // of their use
    const int grp = tid >> 4, s16 = tid & 15;
    constexpr int NG = 640 / 16, NPASS = (OWN + NG - 1) / NG;
    f32x4 rs[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int lo = pass * NG + grp, i = own0 + lo;
        rs[pass] = zero;
        if (a.resid && lo < OWN && i < M) rs[pass] = *reinterpret_cast<const f32x4 *>(a.resid + (long long)i * K + n0 + 4 * s16);
    }
    __syncthreads();                                               // nothing in flight, every operand read done

    // ---- h tile, attention columns, folded BatchNorm and the entries go to LDS
    float *Cs = gemm_lds;
    if (wave8 >= 8) {
        if (xlive) {
            Cs[O_AS + xrow] = cs;
            Cs[O_AD + xrow] = cd;
        }
    } else if (wave8 >= 4) {
        if (tid < 256 + 64) {
            Cs[O_BN + tid - 256] = bn_pre[0];
            Cs[O_BN + 64 + tid - 256] = bn_pre[1];
            Cs[O_BN + 128 + tid - 256] = bn_pre[2];
        }
    } else {
#pragma unroll
        for (int h = 0; h < ACC; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)                      // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
                Cs[(16 * h + 4 * q + reg) * LD + 16 * wave + r] = acc[h][reg];
        const bool use_edge = a.v != nullptr;
        const float v0 = use_edge ? a.v[0] : 0.0f, v1 = use_edge ? a.v[1] : 0.0f;
#pragma unroll
        for (int c = 0; c < NE; ++c) {
            const int idx = tid + 256 * c;
            if (idx < OWN * SLOTS) {
                const f32x4 e = pe[c];
                const int eidx = __float_as_int(e.w);
                int jl = __float_as_int(e.x) - rowbase;            // tile row of the source
                jl = (eidx >= 0 && jl >= 0 && jl < BM) ? jl : (idx >> 3) + HALO;      // empty slot: the target's own row, weight 0
                const float t = use_edge ? __builtin_fmaf(e.z, v1, e.y * v0) : 0.0f;  // gat_aggregate_kernel's edge term
                *reinterpret_cast<f32x4 *>(&Cs[O_EN + 4 * idx]) = f32x4{__int_as_float(jl), t, e.w, 0.f};
            }
        }
    }
    __syncthreads();

    if (tid == 0) NSC_BAND_STAMP(3);
    // ---- attention + aggregation: a 16-lane group per target (4 targets per wave), lane s = entry slot s and column quad s.
    // The softmaxes of all passes first, then the aggregations: independent latency chains (LDS gathers, cross-lane
    // reductions) that the scheduler can interleave.
    const f32x4 bnsc = *reinterpret_cast<const f32x4 *>(&Cs[O_BN + 4 * s16]);
    const f32x4 bnsh = *reinterpret_cast<const f32x4 *>(&Cs[O_BN + 64 + 4 * s16]);
    const f32x4 bnbi = *reinterpret_cast<const f32x4 *>(&Cs[O_BN + 128 + 4 * s16]);
    float al_[NPASS];
    int jl_[NPASS], dw_[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int lo = pass * NG + grp, i = own0 + lo;
        const bool live = lo < OWN && i < M;                       // uniform over the 16-lane group
        const int lc = live ? lo : 0;
        f32x4 en = {__int_as_float(lc + HALO), 0.f, __int_as_float(-1), 0.f};
        if (s16 < SLOTS) en = *reinterpret_cast<const f32x4 *>(&Cs[O_EN + 4 * (lc * SLOTS + s16)]);
        const int jl = __float_as_int(en.x), eidx = __float_as_int(en.z);
        const bool valid = eidx >= 0;
        float l = -INFINITY;
        if (valid) {
            l = (Cs[O_AS + jl] + Cs[O_AD + lc + HALO]) + en.y;
            l = l > 0.0f ? l : a.slope * l;                        // leaky_relu
        }
        const float m = group16_max(l);
        const float p = valid ? expf(l - m) : 0.0f;
        const float den = group16_sum(p) + 1e-16f;                 // PyG softmax
        const float al = p / den;
        if (a.alpha_out && valid && live && n0 == 0) a.alpha_out[eidx] = al;
        al_[pass] = al;
        jl_[pass] = jl;
        // entries of the fullest of this wave's four targets (valid slots are a prefix of the 8): the aggregation below stops
        // there -- on the temporal chain 5 of the 8 slots, and this phase is bound by the LDS pipe (ten waves gathering rows)
        const unsigned long long vm = __ballot(valid);
        int dmax = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dmax = max(dmax, __popc((unsigned)(vm >> (16 * g4)) & 0xffffu));
        dw_[pass] = __builtin_amdgcn_readfirstlane(dmax);
    }
    // the passes' gathers interleaved slot by slot (independent chains of a cross-lane broadcast, an LDS row read and four fmas)
    f32x4 o_[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) o_[pass] = zero;
#pragma unroll
    for (int t = 0; t < ((NSC_BAND_ABL & 1) ? 1 : SLOTS); ++t) {   // entries in CSR order, self loop last
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (t >= dw_[pass]) continue;                          // wave-uniform: the remaining slots carry weight 0 (x + 0 * h = x)
            const float at = __shfl(al_[pass], t, 16);
            const int jt = __shfl(jl_[pass], t, 16);
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(&Cs[jt * LD + 4 * s16]);
            o_[pass].x = __builtin_fmaf(at, gv.x, o_[pass].x);
            o_[pass].y = __builtin_fmaf(at, gv.y, o_[pass].y);
            o_[pass].z = __builtin_fmaf(at, gv.z, o_[pass].z);
            o_[pass].w = __builtin_fmaf(at, gv.w, o_[pass].w);
        }
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int lo = pass * NG + grp, i = own0 + lo;
        const bool live = lo < OWN && i < M;
        f32x4 o = o_[pass];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = o[t] + bnbi[t];
            if (a.bn_w) v = v * bnsc[t] + bnsh[t];
            if (a.relu) v = fmaxf(v, 0.0f);
            if (a.resid) v += rs[pass][t];
            o[t] = v;
        }
        if (live) *reinterpret_cast<f32x4 *>(a.out + (long long)i * K + n0 + 4 * s16) = o;
    }
    if (tid == 0) NSC_BAND_STAMP(4);
}

// Tile of the fused layer: 16 ACC rows computed, 16 ACC - 4 owned, 64 columns; the cost of a workgroup is the MFMA time of a
// computing wave (ACC accumulators) + what a round costs besides.
inline int band_pick_acc(int M, int H)
{
    const long long nch = H / 64, ncb = H / 64;
    int best = 1;
    long long best_cost = -1;
    for (int acc = 1; acc <= 6; ++acc) {
        const int own = 16 * acc - 2 * NSC_BAND_HALO;
        const long long tiles = ncb * ((M + own - 1) / own);
        const long long rounds = (tiles + 255) / 256;
        const long long cost = rounds * (acc * nch * 512 + 3000);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = acc; }
    }
    return best;
}

template <int ACC, int NST = 3>
bool launch_banded_cfg(hipStream_t st, const BandArgs &a)
{
    constexpr unsigned lds = NST * (16 * ACC + 64) * 256;
    static_assert(lds <= 160 * 1024, "LDS of a CU");
    if (lds > 64 * 1024) {                                         // per-device opt-in, as launch_glds_cfg
        static std::atomic<int> opted[16];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
        int s = opted[dev].load(std::memory_order_acquire);
        if (s == 0) {
            s = hipFuncSetAttribute(reinterpret_cast<const void *>(&gat_layer_banded_kernel<ACC, NST>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : 2;
            opted[dev].store(s, std::memory_order_release);
        }
        if (s != 1) return false;
    }
    constexpr int own = 16 * ACC - 2 * NSC_BAND_HALO;
    const dim3 grid(a.H / 64, (a.M + own - 1) / own);
    hipLaunchKernelGGL((gat_layer_banded_kernel<ACC, NST>), grid, dim3(640), lds, st, a, a.Bx);
    return true;
}

// false: this layer cannot take the fused kernel here (the caller runs lin GEMM + gat_aggregate_kernel: same bits)
inline bool launch_banded_layer(hipStream_t st, const BandArgs &a)
{
    if (a.H < 64 || (a.H & 63) || (reinterpret_cast<unsigned long long>(a.A) & 15) ||
        (reinterpret_cast<unsigned long long>(a.B) & 15) || (reinterpret_cast<unsigned long long>(a.Bx) & 15) ||
        (reinterpret_cast<unsigned long long>(a.out) & 15) || (a.resid && (reinterpret_cast<unsigned long long>(a.resid) & 15)))
        return false;
    switch (band_pick_acc(a.M, a.H)) {
    case 1: return launch_banded_cfg<1>(st, a);
    case 2: return launch_banded_cfg<2>(st, a);
    case 3: return launch_banded_cfg<3>(st, a);
    case 4: return launch_banded_cfg<4>(st, a);
    case 5: return launch_banded_cfg<5>(st, a);
    default: return launch_banded_cfg<6>(st, a);
    }
}
