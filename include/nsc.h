/*
 * nsc.h -- C ABI of libnsc_hip.so, the MI355X (gfx950) implementation of the Neural-Spectral-Codec
 * descriptor hot path.
 *
 * The reference is pure Python and has no FFI layer; its boundary for this path is two nn.Module
 * APIs.  Each entry point below replaces the device work behind one of those
 * calls; the Python classes in neural-spectral-codec_amd/{encoding,gnn}/ keep the reference's
 * names and signatures and bind these functions with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (e.g. torch tensor storage);
 *     parameter structs are HOST pointers, read before the call returns
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing,
 *     keeps no mutable global state and is safe to capture into a hipGraph
 *   - return value: 0 = NSC_OK, negative = NscStatus error; nothing is launched on error
 *   - no torch types, no C++ types, no exceptions across the boundary
 */
#ifndef NSC_H
#define NSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSC_ABI_VERSION 4

typedef enum NscStatus {
    NSC_OK            = 0,
    NSC_EINVAL        = -1,  /* null pointer / negative size / inconsistent arguments          */
    NSC_EUNSUPPORTED  = -2,  /* shape outside what the kernels are built for (see each call)   */
    NSC_EWORKSPACE    = -3,  /* workspace smaller than nsc_*_workspace_bytes()                  */
    NSC_ELAUNCH       = -4   /* hipLaunchKernel reported an error (hipGetLastError has details) */
} NscStatus;

int         nsc_abi_version(void);
const char *nsc_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Encoder: point cloud -> E x 360 min-range image -> circular gap interpolation -> row-wise
 * 360-point real FFT magnitude -> per-row 181->n_bins histogram -> global L1 normalisation.
 * Replaces SpectralEncoder.encode_points / encode_range_image / forward
 *   (reference src/encoding/spectral_encoder.py:206, :160, :231) and what they call:
 *   RangeImageProjector.project (src/encoding/range_image.py:129) and
 *   interpolate_range_image    (src/encoding/range_image.py:15).
 * ------------------------------------------------------------------------------------------ */
typedef struct NscEncParams {
    int32_t n_elevation;   /* projector rows E: 1..64          range_image.py:104              */
    int32_t n_azimuth;     /* projector cols A: must be 360    range_image.py:105              */
    int32_t n_bins;        /* histogram bins per row, 1..176   spectral_encoder.py:39          */
    int32_t target_rows;   /* target_elevation_bins R, 1..16   spectral_encoder.py:43          */
    double  elev_min_rad;  /* np.deg2rad(elevation_range[0])   range_image.py:126              */
    double  elev_max_rad;  /* np.deg2rad(elevation_range[1])   range_image.py:127              */
    float   min_range;     /* 1.0                              range_image.py:108              */
    float   max_range;     /* 80.0                             range_image.py:107              */
    float   epsilon;       /* 1e-8                             spectral_encoder.py:42          */
    int32_t interpolate;   /* interpolate_empty                spectral_encoder.py:44,220      */
    int32_t elev_f64;      /* 1 = row math in float64 (numpy >= 2 promotion of range_image.py:186),
                              0 = float32 (numpy 1.24 value-based casting)                      */
} NscEncParams;

void nsc_enc_default_params(NscEncParams *p);

/* Bytes of scratch nsc_encode_clouds needs for this batch (0 when the fused one-workgroup-per-
 * cloud kernel is used; non-zero when clouds are split across workgroups). */
size_t nsc_encode_clouds_workspace_bytes(int32_t n_clouds, int64_t total_points,
                                         const NscEncParams *p);

/* Which kernel set nsc_encode_clouds launches for a batch of this shape (the launcher calls the same decision
 * function): NSC_ENC_PATH_FAST = encode_fast_kernel, the streaming kernel of the configuration every reference caller
 * uses (16 x 360 image, no pooling, (N,4) points, default FOV / range window; any number of clouds, a single cloud of
 * >= 2^27 points takes a slower loop inside it); NSC_ENC_PATH_FUSED = encode_fused_kernel (any other shape);
 * NSC_ENC_PATH_SPLIT = scatter_split_kernel + finish_kernel (few large clouds).  Negative = NscStatus error. */
#define NSC_ENC_PATH_FAST 1
#define NSC_ENC_PATH_FUSED 2
#define NSC_ENC_PATH_SPLIT 3
int nsc_encode_clouds_path(int32_t n_clouds, int64_t total_points, int32_t stride_floats, const NscEncParams *p);

/* SpectralEncoder.encode_points for a packed batch of clouds.
 *   pts            (total_points, stride) float32, row-major AoS [x,y,z(,intensity)]; stride 3 or 4
 *   cloud_offsets  (n_clouds+1) int64 point offsets, cloud c = [off[c], off[c+1]); off[0] = 0
 *   lut            (181) int32  frequency -> bin table (spectral_encoder.py:136-145), monotone
 *   out_desc       (n_clouds, target_rows*n_bins) float32
 *   out_raw        nullable (n_clouds, E, 360) float32  image before interpolation
 *   out_interp     nullable (n_clouds, E, 360) float32  image after interpolation
 *   ws, ws_bytes   scratch of at least nsc_encode_clouds_workspace_bytes() (may be NULL if 0) */
int nsc_encode_clouds(const float *pts, const int64_t *cloud_offsets, int32_t n_clouds,
                      int64_t total_points, int32_t stride_floats, const NscEncParams *p,
                      const int32_t *lut, float *out_desc, float *out_raw, float *out_interp,
                      void *ws, size_t ws_bytes, void *stream);

/* The two halves of nsc_encode_clouds as separate launches, so a caller can overlap the (ALU-bound)
 * finish of batch k with the (HBM-bound) scatter of batch k+1 on two streams:
 *   nsc_scatter_clouds: points -> (n_clouds, E, 360) uint32 images of the MINIMUM SQUARED range
 *                       (float32 bit pattern, 0xffffffff = empty pixel)    range_image.py:129-208
 *   nsc_finish_images : those images -> sqrt, interpolation, FFT, histogram, normalisation */
int nsc_scatter_clouds(const float *pts, const int64_t *cloud_offsets, int32_t n_clouds,
                       int64_t total_points, int32_t stride_floats, const NscEncParams *p,
                       uint32_t *out_sqr, void *stream);
int nsc_finish_images(const uint32_t *sqr, int32_t n_images, const NscEncParams *p, const int32_t *lut,
                      float *out_desc, float *out_raw, float *out_interp, void *stream);

/* SpectralEncoder.forward / encode_range_image for a batch of range images (no projection, no
 * interpolation; rows != target_rows are average-pooled like adaptive_avg_pool2d).
 *   imgs      (n_images, rows, 360) float32, rows 1..64
 *   out_desc  (n_images, target_rows*n_bins) float32 */
int nsc_encode_range_images(const float *imgs, int32_t n_images, int32_t rows,
                            const NscEncParams *p, const int32_t *lut, float *out_desc,
                            void *stream);

/* Intensity image of RangeImageProjector.project(points, keep_intensity=True) (reference range_image.py:216-228):
 * per pixel the maximum intensity over the points whose float32 range equals the pixel's minimum range, 0 where no
 * point fell (and never below 0: the reference max-reduces into a zero image).  points are (N,4) rows; range_raw is
 * the raw range image batch of the same clouds (out_raw of nsc_encode_clouds).  A NaN intensity among a pixel's closest
 * points makes the pixel NaN (np.maximum.at propagates it; the result is the canonical quiet NaN). */
int nsc_project_intensity(const float *points, const int64_t *cloud_offsets, int32_t n_clouds, int64_t total_points,
                          const NscEncParams *p, const float *range_raw, float *out_intensity, void *stream);

/* interpolate_range_image(img, 'linear') (reference range_image.py:15-89) for a batch of float32 range
 * images (n_images, rows, 360), 0 = empty pixel; rows 1..64. */
int nsc_interpolate_range_images(const float *imgs, int32_t n_images, int32_t rows, const int32_t *lut,
                                 float *out, void *stream);
/* The same with the reference's `method` argument: NSC_INTERP_LINEAR (range_image.py:52-64) or NSC_INTERP_NEAREST
 * (:66-75: the circularly nearest valid pixel of the row, the smaller column on a tie).  NscEncParams.interpolate
 * takes the same values (0 = off). */
#define NSC_INTERP_LINEAR 1
#define NSC_INTERP_NEAREST 2
int    nsc_interpolate_range_images_ex(const float *imgs, int32_t n_images, int32_t rows, const int32_t *lut,
                                       int32_t method, float *out, void *stream);


/* ------------------------------------------------------------------------------------------
 * GNN enhancer: Linear 800->256 + BN + ReLU, 3 x GATConv(256->256, heads=1, edge_dim) + BN,
 * Linear 256->800 + input residual.  Replaces SpectralGNN.forward / forward_with_attention
 * (reference src/gnn/model.py:96-153, :155-201) and the torch_geometric 2.4.0 GATConv it calls
 * (model.py:16,75-84,127; algorithm restated in SURVEY.md Appendix B).
 * ------------------------------------------------------------------------------------------ */
#define NSC_GAT_MAX_LAYERS 8
#define NSC_GAT_MAX_EDGE_DIM 8

typedef struct NscGatLayer {      /* device pointers; names are the reference's state-dict keys */
    const float *lin_w;           /* convs.{l}.lin_src.weight (H,H)  (lin_dst is the same tensor)   */
    const float *att_src;         /* convs.{l}.att_src (H)                                          */
    const float *att_dst;         /* convs.{l}.att_dst (H)                                          */
    const float *lin_edge_w;      /* convs.{l}.lin_edge.weight (H,edge_dim), NULL without edge_dim  */
    const float *att_edge;        /* convs.{l}.att_edge (H), NULL without edge_dim                  */
    const float *bias;            /* convs.{l}.bias (H)                                             */
    const float *bn_w, *bn_b, *bn_mean, *bn_var;   /* batch_norms.{l}.{weight,bias,running_*} (H)  */
} NscGatLayer;

typedef struct NscGatModel {
    int32_t in_dim;               /* 800   model.py:33 (multiple of 16) */
    int32_t hidden;               /* 256   model.py:34 (multiple of 16, <= 1024) */
    int32_t out_dim;              /* 800   model.py:35 */
    int32_t n_layers;             /* 3     model.py:36 */
    int32_t edge_dim;             /* 0 = GATConv built without edge_dim (pipeline.py:158-166) */
    int32_t residual;             /* model.py:39 */
    float   bn_eps;               /* 1e-5 */
    float   negative_slope;       /* 0.2 (GATConv default) */
    const float *in_w, *in_b;     /* input_proj.{weight (H,in), bias (H)}      model.py:67 */
    const float *in_bn_w, *in_bn_b, *in_bn_mean, *in_bn_var;   /* input_norm.* model.py:68 */
    const float *out_w, *out_b;   /* output_proj.{weight (out,H), bias (out)}  model.py:88 */
    const float *res_w, *res_b;   /* residual_proj.* (out,in) or NULL when in_dim == out_dim (model.py:91-94) */
    const float *folded;          /* nsc_gat_fold_weights() output, nsc_gat_folded_floats() floats */
    NscGatLayer layers[NSC_GAT_MAX_LAYERS];
} NscGatModel;

/* Graph in CSR-by-target form with PyG's self-loop convention applied (existing self loops removed,
 * one loop per node appended last, its edge attribute = mean of the node's incoming edge attributes). */
typedef struct NscGraph {
    int32_t n_nodes;
    int32_t nnz;                  /* capacity of src/eid = E + n_nodes (stride of alpha_out layers) */
    const int32_t *row_ptr;       /* (n_nodes+1) */
    const int32_t *src;           /* (nnz) source node of each entry, entries of a target in edge order */
    const int32_t *eid;           /* (nnz) index into the caller's edge list, -1 for the self loop */
    const float   *loop_attr;     /* (n_nodes, edge_dim) or NULL */
    /* transposed view (entries grouped by SOURCE), needed by nsc_gat_backward only; NULL otherwise */
    const int32_t *t_ptr;         /* (n_nodes+1) */
    const int32_t *t_entry;       /* (nnz) entry indices into src/eid, ascending per source */
    const int32_t *tgt;           /* (nnz) target node of each entry */
    /* banded form (nsc_graph_band_entries), NULL / 0 otherwise: lets nsc_gat_forward run a GATConv layer as one launch */
    const float   *band_entries;  /* (n_nodes, 8, 4) float32, 16-byte aligned */
    int32_t        band;          /* half bandwidth the caller vouches for (max |source - target| <= band, and at most 8
                                     entries per target incl. the self loop), 0 = not banded.  The kernels support <= 2 */
} NscGraph;

size_t nsc_graph_workspace_bytes(int32_t n_nodes, int64_t n_edges);

/* edge_index (2,E) int64 [row 0 = source j, row 1 = target i] -> CSR arrays (caller-allocated:
 * row_ptr n_nodes+1, src/eid E+n_nodes, loop_attr n_nodes*edge_dim); row_ptr[n_nodes] is the entry
 * count.  Edges with an endpoint outside [0,n_nodes) are dropped. */
int nsc_graph_build_csr(const int64_t *edge_index, int64_t n_edges, int32_t n_nodes,
                        const float *edge_attr, int32_t edge_dim, int32_t *row_ptr, int32_t *src,
                        int32_t *eid, float *loop_attr, void *ws, size_t ws_bytes, void *stream);

/* Banded form of a graph whose CSR exists: per target 8 slots {source (int32 bits), edge_attr[0], edge_attr[1], CSR entry
 * index (int32 bits, -1 = empty slot)} in CSR order, self loop last with loop_attr -- the neighbourhood of a target in ONE
 * 128-byte fetch -- plus info[0] = max |source - target| and info[1] = max entries per target over the graph (int32,
 * device).  A caller that reads info[0] <= 2 and info[1] <= 8 sets NscGraph.band_entries / band = 2: nsc_gat_forward then
 * runs every GATConv layer as one launch (gat_layer_banded_kernel: lin GEMM + 2-row halo, softmax, aggregation, BatchNorm
 * in one kernel, h never leaves LDS) -- the temporal chain of the reference (src/keyframe/graph_manager.py:520-532,
 * abs(i - j) <= 2) always qualifies.  The entries embed the edge attributes (edge_dim 0 or 2; otherwise the banded path is
 * not taken): rebuild them when edge_attr changes.  entries: (n_nodes, 8, 4) float32, 16-byte aligned. */
int nsc_graph_band_entries(const NscGraph *g, const float *edge_attr, int32_t edge_dim, float *entries, int32_t *info,
                           void *stream);

/* Weights-only folding, redone only when the GATConv parameters change: per layer
 * u_src = W^T att_src, u_dst = W^T att_dst (the attention dot products then ride along the lin GEMM as
 * two extra output columns) and v = W_edge^T att_edge (the edge term becomes an edge_dim-long dot). */
size_t nsc_gat_folded_floats(const NscGatModel *m);
int    nsc_gat_fold_weights(const NscGatModel *m, float *folded, void *stream);

size_t nsc_gat_workspace_bytes(const NscGatModel *m, int32_t n_nodes);

/* Inference forward (BatchNorm running statistics, no dropout) == model.eval(); model(data).
 *   x          (n_nodes, in_dim) float32
 *   edge_attr  (E, edge_dim) float32 indexed by g->eid, or NULL (then the edge term is skipped,
 *              as model.py:126-129 does when data has no edge_attr)
 *   out        (n_nodes, out_dim) float32
 *   alpha_out  nullable (n_layers, nnz) attention coefficients (forward_with_attention) */
int nsc_gat_forward(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                    float *out, float *alpha_out, void *ws, size_t ws_bytes, void *stream);
/* Same forward with launch options.  NSC_GAT_CORESIDENT: every kernel uses 0 bytes of LDS and <= 56 VGPRs, so two waves
 * of it fit on a SIMD beside a resident nsc_encode_clouds grid (up to five encoder workgroups per CU: 5 x 27.9 KB of LDS,
 * 5 x 80 VGPRs): issue it on its own stream to run the GNN of batch k under the encoders of batches k+1, k+2.
 * NSC_GAT_SHARED_B (with NSC_GAT_CORESIDENT): the GEMMs use 64 x 64 tiles that share the weight block through 10 KB of LDS
 * (two such workgroups fit in the 20 KB a CU has left beside five encoder workgroups): fewer LDS-pipe and L1 operations
 * per FLOP.  NSC_GAT_LDS_TILED (alone): the stand-alone forward with the round-2 GEMMs (register-staged 16/32/64 x 64 tiles)
 * instead of the default LDS-DMA, wave-specialised GEMM of round 3 -- kept for A/B measurements and as the fall-back when
 * that kernel cannot run (unaligned operands, LDS opt-in refused).  Bit-identical output in every combination. */
#define NSC_GAT_CORESIDENT 1u
#define NSC_GAT_SHARED_B 2u
#define NSC_GAT_LDS_TILED 4u
/* NSC_GAT_GENERIC: ignore NscGraph.band_entries -- every layer as lin GEMM + gat_aggregate_kernel (what irregular graphs
 * always get).  Bit-identical to the banded path; for A/B measurements and the parity tests. */
#define NSC_GAT_GENERIC 8u
int nsc_gat_forward_ex(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                       float *out, float *alpha_out, void *ws, size_t ws_bytes, uint32_t flags, void *stream);
/* Which tile the default (LDS-DMA) GEMM takes for C[M,N] = A[M,K] B[N,K]^T -- host function, no device work: rows and
 * columns of a workgroup tile, its LDS bytes and the number of workgroups (a grid of at most 256 is one round on the
 * 256 CUs).  Returns NSC_OK, or NSC_EINVAL for non-positive sizes / K not a multiple of 16. */
int nsc_gat_gemm_tile(int32_t M, int32_t N, int32_t K, int32_t *tile_rows, int32_t *tile_cols, int32_t *lds_bytes,
                      int64_t *workgroups);

/* ------------------------------------------------------------------------------------------
 * Training step of the GNN (BASELINE configs[4]): model.train(); emb = model(graph);
 * loss = TripletLoss(emb[a], emb[p], emb[n]); loss.backward()
 * (reference src/gnn/trainer.py:205-213, TripletLoss :44-68; model.py:96-153 in train mode:
 * BatchNorm batch statistics, feature dropout :137, GATConv attention dropout).
 * ------------------------------------------------------------------------------------------ */
typedef struct NscGatTrainCfg {
    float    dropout_p;            /* model.py:38; 0 for parity runs (the reference never seeds its RNG) */
    float    bn_momentum;          /* 0.1 (nn.BatchNorm1d default) */
    uint64_t seed;                 /* counter-based dropout masks: same seed in forward and backward */
    int32_t  update_running_stats; /* 1: running_mean / running_var of the model are updated IN PLACE */
    int32_t  accumulate_grads;     /* nsc_gat_backward: 1 = every PARAMETER gradient is ADDED to what its NscGatGrads buffer
                                      holds (gradient accumulation over the batches of an optimizer step, trainer.py:207-221,
                                      without a pass of axpy kernels behind the backward); 0 = overwritten.  grads->x is
                                      always overwritten */
    const uint64_t *seed_dev;      /* nullable DEVICE pointer: when set, the kernels read the seed from this word at run
                                      time instead of `seed` -- a training step captured into a hipGraph then replays
                                      with a fresh mask per step (the caller rewrites the word between replays) */
} NscGatTrainCfg;

typedef struct NscGatGradLayer {   /* device buffers shaped like the NscGatLayer parameters; overwritten, or added to
                                      with NscGatTrainCfg.accumulate_grads */
    float *lin_w, *att_src, *att_dst, *lin_edge_w, *att_edge, *bias, *bn_w, *bn_b;
} NscGatGradLayer;

typedef struct NscGatGrads {
    float *in_w, *in_b, *in_bn_w, *in_bn_b, *out_w, *out_b;
    float *x;                      /* nullable: gradient w.r.t. the input features (n_nodes, in_dim) */
    float *res_w, *res_b;          /* residual_proj (model.py:91-94): required iff residual && in_dim != out_dim */
    NscGatGradLayer layers[NSC_GAT_MAX_LAYERS];
} NscGatGrads;

size_t nsc_graph_transpose_workspace_bytes(int32_t n_nodes);
int    nsc_graph_transpose(const NscGraph *g, int32_t *t_ptr, int32_t *t_entry, int32_t *tgt, void *ws,
                           size_t ws_bytes, void *stream);

/* The workspace holds the activations the forward saves for the backward: pass the SAME buffer (and
 * cfg) to nsc_gat_backward.  residual && in_dim != out_dim trains residual_proj (m->res_w / res_b). */
size_t nsc_gat_train_workspace_bytes(const NscGatModel *m, const NscGraph *g);
int    nsc_gat_forward_train(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                             const NscGatTrainCfg *cfg, float *out, void *ws, size_t ws_bytes, void *stream);
int    nsc_gat_backward(const NscGatModel *m, const NscGraph *g, const float *x, const float *edge_attr,
                        const NscGatTrainCfg *cfg, const float *grad_out, const NscGatGrads *grads, void *ws,
                        size_t ws_bytes, void *stream);

/* TripletLoss forward (+ backward when grad_emb != NULL): loss = scale * mean_t relu(|a-p|^2 - |a-n|^2 + margin);
 * grad_emb (n_nodes, dim) is zeroed and receives d loss / d emb.  Indices follow embeddings[idx] (trainer.py:207-209):
 * values in [-n_nodes, 0) wrap; a triplet with an index outside [-n_nodes, n_nodes) reads and writes nothing and turns
 * the loss into NaN (the reference raises IndexError there). */
size_t nsc_triplet_workspace_bytes(int32_t n_triplets);
int    nsc_triplet_loss(const float *emb, const int64_t *anchors, const int64_t *positives,
                        const int64_t *negatives, int32_t n_triplets, int32_t n_nodes, int32_t dim,
                        float margin, float scale, float *loss, float *grad_emb, void *ws, size_t ws_bytes,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Stage-1 retrieval: 1-D Wasserstein distance = L1 of CDFs, top-k.
 * Replaces wasserstein_distance_batch_torch / _matrix_torch (reference src/retrieval/wasserstein.py
 * :134-172, :232-273), WassersteinRetriever.query (:328-384) and the spatial filter of
 * TwoStageRetrieval._global_retrieval (src/retrieval/two_stage_retrieval.py:145-202).
 * ------------------------------------------------------------------------------------------ */
/* cdf[i] = cumsum(normalised hists[i]);  divide_plain = 1: h / sum (query form, wasserstein.py:153-155),
 * 0: h / (sum + eps) (database / matrix form, :158-163); rows with sum <= eps stay unnormalised. */
int nsc_w1_cdf(const float *hists, int32_t n, int32_t dim, float eps, int32_t divide_plain, float *cdf,
               void *stream);
/* dist (Q, N): W1 of every query CDF against every database row (raw histograms, normalised on the
 * fly).  db_pos (N,3) / q_pos (Q,3) nullable: pairs closer than min_dist get +inf (spatial filter). */
int nsc_w1_distances(const float *db, int32_t N, int32_t dim, float eps, const float *q_cdf, int32_t Q,
                     const float *db_pos, const float *q_pos, float min_dist, float *dist, void *stream);
/* Same distances against a database whose rows are already CDFs (= nsc_w1_cdf(db, divide_plain = 0), kept by
 * WassersteinRetriever next to the raw histograms so that the normalise + prefix-sum is not redone per query).
 * dim % 4 == 0, both matrices 16-byte aligned.  Q <= 4: HBM-streaming kernel (rows read once, queries in
 * registers); Q > 4: register-tiled |a-b| kernel (VALU-bound). */
int nsc_w1_distances_cdf(const float *db_cdf, int32_t N, int32_t dim, const float *q_cdf, int32_t Q,
                         const float *db_pos, const float *q_pos, float min_dist, float *dist, void *stream);
/* idx/val (Q, k): the k smallest entries of each row of dist, ascending, ties to the smaller index
 * (k <= 256 and ceil(N/2048)*k <= 4096). */
size_t nsc_topk_workspace_bytes(int32_t Q, int32_t N, int32_t k);
int nsc_topk_smallest(const float *dist, int32_t Q, int32_t N, int32_t k, int64_t *idx, float *val,
                      void *ws, size_t ws_bytes, void *stream);

/* Hard-negative triplet mining inside ONE sequence: replaces
 * TripletMiner._mine_sequence_triplets / _select_hard_negative (reference src/gnn/triplet_miner.py
 * :141-229, :314-359).  Members of the sequence are rows 0..n-1 in temporal order. */
typedef struct NscMineParams {
    double   positive_distance_max;   /* 5.0   triplet_miner.py:43 */
    double   negative_distance_min;   /* 10.0  :45 */
    double   negative_distance_max;   /* 50.0  :46 */
    int32_t  positive_temporal_min;   /* 30    :44 */
    int32_t  negative_temporal_min;   /* 30    :47 */
    int32_t  strategy;                /* 0 = "hard" (argmin W1, :347-350), 1 = "random" (:333-334),
                                       * 2 = "semi-hard" (candidate at position len // 2 of the W1 order, :352-357) */
    int32_t  triplets_per_anchor;
    uint64_t seed;                    /* counter-based choice of the positive (np.random.choice in the reference) */
} NscMineParams;
/* positions (n,3) float64 translations; cdf (n,dim) = nsc_w1_cdf(descriptors, divide_plain = 1);
 * out_pos / out_neg (n, triplets_per_anchor) local row indices or -1 when the anchor has no positive
 * or no negative candidate; counts (n,2) = [#positive, #negative candidates]. */
int nsc_mine_triplets(const double *positions, const float *cdf, int32_t n, int32_t dim,
                      const NscMineParams *mp, int32_t *out_pos, int32_t *out_neg, int32_t *counts,
                      void *stream);
/* The same with a workspace: strategy 2 keeps one row of candidate distances per anchor (n * n floats). */
size_t nsc_mine_workspace_bytes(int32_t n, int32_t strategy);
int nsc_mine_triplets_ws(const double *positions, const float *cdf, int32_t n, int32_t dim,
                         const NscMineParams *mp, int32_t *out_pos, int32_t *out_neg, int32_t *counts,
                         void *ws, size_t ws_bytes, void *stream);

/* Validation recall for loop closure: the pieces of
 * GNNTrainer._compute_recall_loop_closure (reference src/gnn/trainer.py:306-387).
 *   nsc_revisit_queries: first_revisit[i] = first j >= i + skip_frames with |p_i - p_j| < thr, else -1  (:342-348)
 *   nsc_pairwise_l2    : dist (Q,n) = euclidean embedding distance of query q to every c with |c - q| > skip_frames,
 *                        +inf for the excluded temporal neighbours (:355-370); feed to nsc_topk_smallest
 *   nsc_recall_rank    : rank[q] = 1-based position of the first of the k nearest candidates that lies within
 *                        thr of the query pose, 0 if none (:376-383)  ->  recall@K = mean(0 < rank <= K) */
int nsc_revisit_queries(const double *positions, int32_t n, int32_t skip_frames, double distance_threshold,
                        int32_t *first_revisit, void *stream);
int nsc_pairwise_l2(const float *emb, const int32_t *query_idx, int32_t Q, int32_t n, int32_t dim,
                    int32_t skip_frames, float *dist, void *stream);
int nsc_recall_rank(const double *positions, const int32_t *query_idx, const int64_t *topk_idx, int32_t Q,
                    int32_t k, double distance_threshold, int32_t *rank, void *stream);

/* ------------------------------------------------------------------------------------------
 * Keyframe-side helpers either side of the path.
 * ------------------------------------------------------------------------------------------ */
/* Offline temporal graph: replaces the O(N*M) Python loop of build_graph_from_keyframes_batch (reference
 * src/keyframe/graph_manager.py:515-596).  Edges [i, i+off], off = -M/2..M/2 except 0, node-major in ascending
 * offset, then two edges [q,m],[m,q] per loop closure (loops (n_loops,2), already range-checked by the host
 * as :555).  poses (n,4,4) float64 row-major, nullable -> no edge_attr.  edge_index (2,E) int64,
 * edge_attr (E,2) float32 = [log1p(|t_i - t_j|)/5, arccos(clip((clip(tr(R_j R_i^T),-1,3)-1)/2,-1,1))/pi]. */
int64_t nsc_chain_graph_num_edges(int32_t n_nodes, int32_t temporal_neighbors, int32_t n_loops);
int nsc_build_chain_graph(const double *poses, int32_t n_nodes, int32_t temporal_neighbors, const int64_t *loops,
                          int32_t n_loops, int64_t *edge_index, float *edge_attr, void *stream);

/* 16-bit descriptor wire format: HistogramQuantizer.quantize / dequantize (reference
 * src/encoding/quantization.py:131-191) for n histograms of dim bins (dim <= 4096; the float32 sums follow
 * numpy's pairwise order bit for bit), and CompressedDescriptor.to_bytes / from_bytes (:41-110) generalised
 * from 50 bins to dim: records of nsc_record_bytes(dim) = 2*dim + 120 bytes (220 for 50 bins):
 * [dim x u16][pose 7 x f32][timestamp f64][keyframe id u32][20-byte hash][60 zero bytes]. */
int nsc_quantize_descriptors(const float *hist, int32_t n, int32_t dim, float eps, uint16_t *quantized, void *stream);
int nsc_dequantize_descriptors(const uint16_t *quantized, int32_t n, int32_t dim, float eps, float *hist, void *stream);
size_t nsc_record_bytes(int32_t dim);
int nsc_pack_records(const uint16_t *quantized, const float *pose7, const double *timestamps,
                     const uint32_t *keyframe_ids, const uint8_t *hashes, int32_t n, int32_t dim,
                     uint8_t *records, void *stream);
int nsc_unpack_records(const uint8_t *records, int32_t n, int32_t dim, uint16_t *quantized, float *pose7,
                       double *timestamps, uint32_t *keyframe_ids, uint8_t *hashes, void *stream);

/* Geometric-novelty test of the keyframe selector: compute_overlap (reference src/data/pose_utils.py:323-389,
 * called from keyframe/criteria.py:123) after its random down-sampling (:340-347, done by the caller) for
 * n_pairs cloud pairs at once.  points1/points2: packed float32 rows of stride_floats (3 or 4) columns,
 * offsets (n_pairs+1) int64; transforms (n_pairs,4,4) float64 maps cloud 1 into the frame of cloud 2.
 * max_pair_points = host-known max over pairs of n1+n2 (<= 12288; the reference caps each cloud at 5000).
 * counts (n_pairs,3) = [|voxels 1|, |voxels 2|, |intersection|], iou (n_pairs) float64. */
size_t nsc_voxel_overlap_workspace_bytes(int64_t total_points1, int64_t total_points2);
int nsc_voxel_overlap(const float *points1, const int64_t *offsets1, const float *points2, const int64_t *offsets2,
                      int32_t n_pairs, int64_t total_points1, int64_t total_points2, int64_t max_pair_points,
                      int32_t stride_floats, const double *transforms, double voxel_size, int32_t *counts,
                      double *iou, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSC_H */
