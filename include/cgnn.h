/*
 * cgnn.h -- C ABI of the MI355X (gfx950) Interaction-Network message-passing engine.
 *
 * The reference (mattpan-peregrinus/Cosmology_GNN_Simulation) has no FFI: its hot
 * path is the Python API of graph_network.py / data_utils.py, whose native work
 * happens inside third-party ops.  Every
 * entry point below replaces one of those call sites; the reference file:line it
 * stands in for is given with each declaration.  The Python side
 * (cosmology_gnn_simulation_amd/graph_network.py, data_utils.py) binds these
 * symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *    parameter name ends in _host;
 *  - tensors are dense row-major float32; index arrays are int32;
 *  - no allocation, no ownership transfer: scratch is a caller-provided workspace
 *    whose size comes from the matching *_workspace_bytes() query;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, a negative cgnn_status otherwise; cgnn_last_error()
 *    returns a thread-local message for the last failure.
 */
#ifndef CGNN_H_
#define CGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGNN_VERSION 100          /* 0.1.0 */
#define CGNN_MAX_HIDDEN_LAYERS 6  /* mlp_num_hidden_layers upper bound */

typedef enum {
    CGNN_OK = 0,
    CGNN_ERR_INVALID_ARG = -1,  /* null pointer, negative size, bad enum        */
    CGNN_ERR_UNSUPPORTED = -2,  /* shape outside the compiled specialisations    */
    CGNN_ERR_WORKSPACE = -3,    /* workspace too small                           */
    CGNN_ERR_HIP = -4           /* a HIP runtime call or kernel launch failed    */
} cgnn_status;

typedef enum {
    CGNN_F32 = 0,      /* f32 operands, v_mfma_f32_32x32x2_f32, exact f32 (parity mode)     */
    CGNN_BF16 = 1,     /* bf16 operands, v_mfma_f32_32x32x16_bf16, f32 accumulate           */
    CGNN_BF16_N16 = 2, /* same arithmetic as CGNN_BF16, weights packed for the 16-edge-per-
                          wave kernels (v_mfma_f32_16x16x32_bf16): cgnn_edge_block, and
                          cgnn_mlp_rows as the edge encoder (input <= 32 features, LayerNorm,
                          CGNN_TILED32 output)                                             */
    CGNN_F32X3 = 3,    /* f32 emulated on the bf16 matrix cores: operands split into three bf16
                          terms (8+8+8 significand bits), the six products whose weight is
                          >= 2^-16 accumulated in f32 (a1b1,a1b2,a2b1,a2b2,a1b3,a3b1); dropped
                          terms are <= 2^-24 relative.  2.7x the f32-MFMA rate at f32-level
                          error; cgnn_mlp_rows and cgnn_node_block                          */
    CGNN_F32X3_N16 = 4, /* CGNN_F32X3 arithmetic, weights packed for the 16-row-per-wave node kernel
                          (v_mfma_f32_16x16x32_bf16, two waves per SIMD); cgnn_node_block only,
                          square layers, its projection epilogue takes CGNN_BF16_N16 weights    */
    CGNN_F16X2_N16 = 5, /* f32 emulated on the fp16 matrix cores (v_mfma_f32_16x16x32_f16): operands split
                          into two fp16 terms, x = hi + lo/2048 with lo = fp16((x - hi) * 2048) (11 + 11
                          significand bits, the residual scaled so it stays a normal fp16 number); three
                          products per element (hi.hi, hi.lo, lo.hi; lo.lo <= 2^-22 relative dropped), the
                          scaled ones summed in a second f32 accumulator.  Half the matrix work of
                          CGNN_F32X3 at the same error.  Range: |activation| < 65520 (fp16), beyond it the
                          row turns into inf/NaN (never a silently wrong number); use CGNN_F32X3 for
                          unnormalised inputs.  16-row packing; cgnn_node_block (square layers <= 128,
                          projection epilogue CGNN_BF16_N16) and cgnn_edge_block                  */
    CGNN_F16X2 = 6      /* the same two-fp16-term arithmetic in the 32-row packing (v_mfma_f32_32x32x16_f16):
                          wherever CGNN_F32X3 is accepted -- cgnn_mlp_rows, cgnn_node_block (any supported
                          latent / hidden pair), cgnn_project_nodes (CGNN_P_F32 tables), cgnn_mlp_backward --
                          at half its matrix work; same range limit as CGNN_F16X2_N16              */
} cgnn_precision;

/* Element type / row order of the Ps, Pd gather tables (cgnn_project_nodes -> cgnn_edge_block).
 * Feature f of a row of H values:
 *   CGNN_P_F32       float32, position f                                  (mlp precision CGNN_F32, CGNN_F16X2, CGNN_F16X2_N16)
 *   CGNN_P_BF16_S32  bf16, f = 32t+8g+4h+c at h*(H/2) + (4t+g)*4 + c      (mlp precision CGNN_BF16)
 *   CGNN_P_BF16_S16  bf16, f = 16O+4q+i    at (4*(O/2) + q)*8 + 4*(O%2) + i   (mlp precision CGNN_BF16_N16: the
 *                    16-edge kernel's MFMA B-operand order, so the rows enter the accumulators through the matrix pipe)
 *   CGNN_P_F16_S32   IEEE fp16 (projection weights CGNN_BF16 / CGNN_BF16_N16, the f32 sums rounded to fp16: 11
 *                    significand bits instead of 8, |value| < 65520), H = 128 only; with u = (4t+g)*4 + c the position of
 *                    f within CGNN_P_BF16_S32's half h, f sits at (u/32)*64 + h*32 + u%32: the halves are interleaved in
 *                    64-byte segments, so that a 128-byte line holds what a 16-row writer produces per row at a time
 *                    (whole-line stores) and both halves of a gathered sub-row share one line.  The table format of
 *                    cgnn_edge_stream_run_w8, which adds Ps[src] + Pd[dst] on the vector pipe with one v_fma_mix_f32 per
 *                    value (fp16 widens for free there; bf16 rows need four selector MFMAs per row tile instead)
 * i.e. each lane of the consuming kernel reads one contiguous run. */
typedef enum { CGNN_P_F32 = 0, CGNN_P_BF16_S32 = 1, CGNN_P_BF16_S16 = 2, CGNN_P_F16_S32 = 3 } cgnn_ptable;

/*
 * Memory layout of an [n, width] float32 matrix.
 * CGNN_ROWS    dense row-major.
 * CGNN_TILED32 (width % 32 == 0) whole tiles of 32 rows; element (32*T + r, 32*t + 8*g + 4*h + c), g<4, h<2,
 *              c<4, lives at float offset  T*32*width + ((4*t + g)*64 + 32*h + r)*4 + c.  A wavefront then
 *              moves its 32-row tile with width/8 fully coalesced 1-KiB instructions in exactly the register
 *              order of the MFMA accumulators.  The buffer holds cgnn_tiled_rows(n) >= n rows (padding rows
 *              are scratch).  Used for the edge-latent stream, which never leaves the engine; convert with
 *              cgnn_relayout at the API boundary.
 */
typedef enum { CGNN_ROWS = 0, CGNN_TILED32 = 1 } cgnn_layout;

/* One Linear layer, weights already in MFMA-fragment order (cgnn_pack_linear). */
typedef struct {
    const void* w;      /* packed weights, cgnn_packed_linear_bytes() long */
    const float* b;     /* bias [out_dim] (natural order), may be NULL      */
    int32_t in_dim;     /* logical K                                        */
    int32_t out_dim;    /* logical number of outputs                        */
} cgnn_linear;

/*
 * build_mlp(hidden, nh, out) [+ LayerNorm(out)]  (reference graph_network.py:15-32,
 * :133-135): nh hidden layers Linear+ReLU, a final Linear, optional LayerNorm
 * (eps 1e-5, biased variance, affine).
 */
typedef struct {
    int32_t precision;                 /* cgnn_precision of the packed weights   */
    int32_t num_hidden_layers;         /* nh >= 1                                 */
    cgnn_linear layer[CGNN_MAX_HIDDEN_LAYERS + 1]; /* nh hidden + 1 output layer   */
    const float* ln_gamma;             /* [out_dim] or NULL = no LayerNorm        */
    const float* ln_beta;              /* [out_dim] or NULL                       */
} cgnn_mlp;

/* ---- library queries -------------------------------------------------------- */
int cgnn_version(void);
const char* cgnn_arch(void);           /* "gfx950" */
const char* cgnn_last_error(void);

/* ---- weight packing (host-side nn.Linear.weight [out,in] -> MFMA fragments) -- */
/* Packs columns [col0, col0+ncols) of the row-major weight w[out_dim, ld]:
 * the column slice is how the first edge/node layer is split into its sender /
 * receiver / edge (resp. node / aggregate) blocks, following the concatenation
 * order of reference graph_network.py:89 and :94. */
size_t cgnn_packed_linear_bytes(int32_t out_dim, int32_t ncols, int32_t precision);
int cgnn_pack_linear(const float* w, int32_t out_dim, int32_t ld, int32_t col0, int32_t ncols,
                     int32_t precision, void* packed, void* stream);

/* ---- K4/K10: row-wise MLP (+LayerNorm): encoders and decoders ----------------
 * y[n, out] = MLP(x[n, in]) ; reference graph_network.py:54,57 (encoder),
 * :158-159 (decoders).  ld_x / ld_y are row strides in floats. */
int cgnn_mlp_rows(const cgnn_mlp* mlp, const float* x, int64_t n, int32_t ld_x,
                  float* y, int32_t ld_y, int32_t y_layout, void* stream);

/* rows of a CGNN_TILED32 buffer holding n logical rows (n rounded up to 32) */
int64_t cgnn_tiled_rows(int64_t n);
/* dst (layout `to`) = src (layout `from`), logical shape [n, width]; padding rows of a tiled dst are zeroed. */
int cgnn_relayout(const float* src, int32_t from, float* dst, int32_t to, int64_t n, int32_t width, void* stream);

/* ---- first-layer split: per-node projections consumed by cgnn_edge_block ------
 * ps[n,H] = x[n,D] * Ws^T ; pd[n,H] = x[n,D] * Wd^T + b1, where [Ws|Wd|We] is the
 * column split of the edge model's first Linear (reference graph_network.py:89-90:
 * cat([x[src], x[dest], edge_attr])).  Either output may be NULL.
 * `precision` is the packing of ws/wd: CGNN_F32 or CGNN_F16X2 (both write CGNN_P_F32 tables) or CGNN_BF16 (either
 * bf16 table order); `p_format` (cgnn_ptable) the layout of the tables, which must be the one the consuming
 * cgnn_edge_block expects.  From 4096 rows on, bf16 / two-fp16-term matrices of up to 128 x 128 are copied into LDS once
 * per workgroup instead of being streamed from L2 by every wave (same results bit for bit). */
int cgnn_project_nodes(const cgnn_linear* ws, const cgnn_linear* wd, int32_t precision,
                       const float* x, int64_t n, void* ps, void* pd, int32_t p_format, void* stream);

/* ---- K5+K6+K9: fused edge update ---------------------------------------------
 * For every edge e = (src[e] -> dst[e]):
 *   u   = LayerNorm(MLP(cat[x[src], x[dst], e_in[e]]))      graph_network.py:89-90
 *   e_out[e] = e_in[e] + u  (residual != 0, graph_network.py:182)  or  u
 *   e_upd[e] = u            (if e_upd != NULL; feeds message_source="edge")
 * with the first layer evaluated as ps[src] + pd[dst] + e_in * We^T.
 * mlp->layer[0] holds We (in_dim = D); e_out may alias e_in.
 * e_in / e_out / e_upd are CGNN_TILED32 buffers of cgnn_tiled_rows(num_edges) rows;
 * ps / pd are cgnn_project_nodes tables in the cgnn_ptable format matching mlp->precision.
 * Kernels by mlp->precision: CGNN_F32 (exact, any compiled shape), CGNN_BF16 (32-edge tiles), CGNN_BF16_N16 (16-edge
 * tiles; weights resident in LDS up to 128 x 128, streamed through an LDS ring at latent = hidden = 256, where the
 * fused aggregation below is not available), CGNN_F16X2_N16 (f32 accuracy on the fp16 matrix cores, latent = hidden =
 * 128, 1..3 hidden layers, a bias on every Linear, CGNN_P_F32 tables).
 *
 * Optional fused aggregation (agg_out != NULL; CGNN_BF16_N16 kernels, receiver-sorted edges with
 * fixed in-degree seg_k in {8, 16}): the same launch also writes the receivers' aggregate
 *   agg_out[i] = sum_{e: dst[e]==i} x_gather[src[e]]     (x_gather != NULL: PyG's default message)
 *   agg_out[i] = sum_{e: dst[e]==i} u[e]                 (x_gather == NULL: message_source "edge")
 * i.e. cgnn_aggregate folded in (graph_network.py:92): a 16-edge wave tile is exactly one (k=16) or two
 * (k=8) receivers, so the sum is a cross-lane reduction and the e_upd round trip disappears. */
int cgnn_edge_block(const cgnn_mlp* mlp, const void* ps, const void* pd,
                    const int32_t* src, const int32_t* dst, int64_t num_edges,
                    const float* e_in, float* e_out, float* e_upd, int32_t residual,
                    int32_t latent, const float* x_gather, float* agg_out, int32_t seg_k, void* stream);

/* ---- K7: aggregation (PyG propagate, aggr='add') ------------------------------
 * out[i] = sum over edges e with dst[e]==i of table[gather ? gather[e] : e].
 *  - fixed_k > 0: edges are receiver-sorted with exactly fixed_k edges per
 *    receiver (the layout data_utils.preprocess produces): segmented
 *    gather-sum, no atomics, bit-reproducible; dst is ignored (may be NULL).
 *  - fixed_k == 0: general edge list: out is zeroed, then run-length-reduced
 *    float atomics keyed by dst (sum order not reproducible).
 * table_layout: CGNN_ROWS, or CGNN_TILED32 for per-edge messages (gather == NULL) straight from
 * cgnn_edge_block's e_upd.  out is CGNN_ROWS.
 * reference graph_network.py:92 -> torch_geometric MessagePassing.propagate. */
int cgnn_aggregate(const float* table, int32_t table_layout, const int32_t* gather, const int32_t* dst,
                   int64_t num_edges, int32_t fixed_k, int64_t num_nodes, int32_t width,
                   float* out, void* stream);

/* The same sum for a graph that is aggregated more than once (every round of a forward): a per-graph plan lists, for
 * each block of 64 consecutive receivers, the DISTINCT sender rows and, per edge, its sender's position in that list; the
 * kernel stages a block's distinct rows in LDS once and sums from there (about 3.4x fewer row reads at k = 16 on a
 * spatially ordered graph).  Same summation order as cgnn_aggregate(fixed_k): bit-identical results.
 *   gather   the receiver-sorted sender list, fixed_k per receiver (1 <= fixed_k <= 32); the plan is valid for exactly
 *            this list (rebuild after any change)
 *   plan     cgnn_aggregate_plan_bytes(num_nodes, fixed_k) bytes of device memory (0: not plannable)
 *   width    multiple of 32; table is CGNN_ROWS.  reference graph_network.py:92 as above. */
size_t cgnn_aggregate_plan_bytes(int64_t num_nodes, int32_t fixed_k);
int cgnn_aggregate_plan_build(const int32_t* gather, int64_t num_nodes, int32_t fixed_k, void* plan, void* stream);
int cgnn_aggregate_planned(const float* table, const int32_t* gather, const void* plan, int64_t num_nodes,
                           int32_t fixed_k, int32_t width, float* out, void* stream);
/* The same with the table's row count (>= 1 + the largest sender id; a spatial shard's table holds ghost rows behind the
 * receivers'): known, and with table and output below 4 GiB at width 128 / 256, the kernel addresses rows by 32-bit buffer
 * offsets without branches (faster, same bits). */
int cgnn_aggregate_planned_rows(const float* table, int64_t table_rows, const int32_t* gather, const void* plan,
                                int64_t num_nodes, int32_t fixed_k, int32_t width, float* out, void* stream);

/* ---- K8+K9: fused node update --------------------------------------------------
 *   u = LayerNorm(MLP(cat[x, agg]))                          graph_network.py:94-96
 *   x_out = x + u (residual != 0, graph_network.py:181) or u
 * w_x / w_agg are the column split of the node model's first Linear; the rest of
 * the MLP (hidden layers 1.., output layer, LayerNorm) is in `mlp` with
 * mlp->layer[0] ignored.  x_out may alias x.
 * Optional epilogue (ws_next != NULL): the next round's cgnn_project_nodes(ws_next,
 * wd_next, proj_precision, x_out, n, ps_next, pd_next, p_format) in the same call --
 * fused into the kernel where a fused specialisation exists, otherwise run after it. */
int cgnn_node_block(const cgnn_mlp* mlp, const cgnn_linear* w_x, const cgnn_linear* w_agg,
                    const float* x, const float* agg, int64_t n, float* x_out,
                    int32_t residual, int32_t latent,
                    const cgnn_linear* ws_next, const cgnn_linear* wd_next, int32_t proj_precision,
                    void* ps_next, void* pd_next, int32_t p_format, void* stream);

/* ---- all rounds of the edge stream in one launch (reference-faithful message only) -----------------------------
 * graph_network.py:89-90,182 for round = 0..L-1.  Under the reference's aggregation (PyG's default message: sender
 * NODE latents) the node stream never reads the edge stream, so Ps_r / Pd_r of every round can be computed
 * first (cgnn_node_block's epilogue) and each edge's latent tile then stays in registers through all L updates
 *     e <- e + LN_r(MLP_r(Ps_r[src] + Pd_r[dst] + We_r e)),
 * crossing HBM once instead of L times; the rounds' layers cycle through an LDS ring.  Same arithmetic as
 * cgnn_edge_block (CGNN_BF16_N16): results are bit-identical to L per-round calls.
 *   rounds[r]       edge model of round r, CGNN_BF16_N16, layer[0] = the We column block (as for cgnn_edge_block)
 *   ps_all, pd_all  CGNN_P_BF16_S16 tables of all rounds; round r starts at element r * round_stride
 *   e_in, e_out     CGNN_TILED32 edge latents (may be the same buffer)
 *   encoder         optional (NULL: start from e_in): the edge encoder of graph_network.py:57 (CGNN_BF16_N16, <= 32
 *                   inputs, LayerNorm), run on each tile's rows of edge_attr [E, ld_attr] before round 0, so that its
 *                   E x latent output is never written to memory; e_in is then ignored
 * Built for hidden == latent in {32, 64, 128}. */
int cgnn_edge_stream(const cgnn_mlp* rounds, int32_t num_rounds, const void* ps_all, const void* pd_all,
                     int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                     const float* e_in, float* e_out, int32_t latent, const cgnn_mlp* encoder, const float* edge_attr,
                     int32_t ld_attr, void* stream);

/* ---- the same, second generation: 32 edges per MFMA tile, one wave per SIMD, two tiles per wave ---------------------
 * graph_network.py:57 (optional edge encoder) and :89-90,182 for round = 0..L-1, as cgnn_edge_stream, for
 * hidden == latent in {32, 64, 128}.  The layers of all rounds travel as one contiguous IMAGE of self-contained chunks
 * (packed CGNN_BF16 weights followed by the layer's bias and, for output layers, LayerNorm gamma / beta) that the kernel
 * cycles through a four-slot LDS ring; build it once per model:
 *   cgnn_edge_stream_image_bytes  size of the image (0 when the shape is not supported)
 *   cgnn_edge_stream_image_build  rounds[r] = edge model of round r (CGNN_BF16, layer[0] = the We column block as for
 *                                 cgnn_edge_block, LayerNorm required); encoder = NULL or the edge encoder (CGNN_BF16,
 *                                 <= 16 input features, LayerNorm); device-to-device copies on `stream`
 *   cgnn_edge_stream_run          ps_all / pd_all: CGNN_P_BF16_S32 tables of all rounds (round r at element
 *                                 r * round_stride); e_in / e_out: CGNN_TILED32 (may alias); enc_in_dim > 0: the image
 *                                 starts with the encoder, the initial latents are computed from edge_attr [E, ld_attr]
 *                                 in the launch and e_in is ignored
 * Arithmetic: bf16 operands, f32 accumulation, f32 LayerNorm and residual (as cgnn_edge_block with CGNN_BF16). */
size_t cgnn_edge_stream_image_bytes(int32_t latent, int32_t num_hidden_layers, int32_t num_rounds, int32_t with_encoder);
int cgnn_edge_stream_image_build(const cgnn_mlp* rounds, int32_t num_rounds, const cgnn_mlp* encoder, int32_t latent,
                                 void* image, size_t image_bytes, void* stream);
int cgnn_edge_stream_run(const void* image, size_t image_bytes, int32_t latent, int32_t num_hidden_layers,
                         int32_t num_rounds, int32_t enc_in_dim, const void* ps_all, const void* pd_all,
                         int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                         const float* e_in, float* e_out, const float* edge_attr, int32_t ld_attr, void* stream);

/* ---- the same, third generation: 32 edges per MFMA tile, TWO waves per SIMD, one tile per wave ----------------------
 * Same image, tables, layouts and arithmetic as cgnn_edge_stream_run (reference graph_network.py:57,:89-90,:182); the
 * two co-resident waves of a SIMD overlap one's vector work (bf16 pack, LayerNorm) with the other's matrix work in
 * hardware, and the P rows travel through LDS (whole cache lines per load instruction).  For the graphs
 * data_utils.preprocess emits: fixed_k = the fixed in-degree, the caller guarantees dst[e] == e / fixed_k (receiver-sorted;
 * as for cgnn_aggregate), fixed_k in {8, 16, 32, 64, 96, ...}, latent == hidden == 128, 1..3 hidden layers
 * (cgnn_edge_stream_w8_supported); every other shape or edge list: cgnn_edge_stream_run.
 * lag = 1: the second wave of every SIMD runs one layer behind the first (their LayerNorms never coincide), 0: in step;
 * results do not depend on lag.
 * p_format: CGNN_P_F16_S32 (fp16 tables: Ps[src] + Pd[dst] is one v_fma_mix_f32 per value on the vector pipe, most of them
 * under the first layer's MFMAs; lag = 0 only) or CGNN_P_BF16_S32 (the tables of cgnn_edge_stream_run: the rows enter the
 * accumulators through four selector MFMAs per row tile, 14 % more matrix work per round).
 * Its image differs from cgnn_edge_stream_run's in one thing: every bias sits one chunk early (the kernel reads a layer's
 * bias while the previous layer still computes); cgnn_edge_stream_image_build_w8 builds it (same arguments and size).
 * flags: CGNN_STREAM_FOLDED (CGNN_P_F16_S32 tables only) = the caller promises an image whose LayerNorms were folded:
 *   (1) every pass's output Linear is CENTRED -- the mean over its output features was subtracted from each weight column
 *       and from the bias (W[:, i] -= mean(W[:, i]), b -= mean(b): LayerNorm(y) == LayerNorm(y - mean(y)), so the model
 *       is unchanged) -- its outputs then have zero mean up to the bf16 rounding of the centred weights (~2e-4 standard
 *       deviations), and the kernel normalises with var = E[y^2] and no mean;
 *   (2) the LayerNorm shift (beta) of every round that has a successor is zero in the image and was carried forward
 *       instead: with B_r = beta_0 + ... + beta_{r-1} the residual stream the kernel holds is e_r - B_r, round r's Pd
 *       table was projected with the bias b1_r + We_r B_r (We_r: the edge-latent block of round r's first Linear, as
 *       rounded to bf16 in the image), and the LAST round's beta in the image is B_L = the sum of all rounds' betas
 *       (the stored latents are e_L itself).  The encoder's LayerNorm keeps its beta.
 * 128 fewer vector instructions per tile and round (LayerNorm: 64 adds of the mean's sum, 64 adds of beta); results
 * differ from the unfolded kernel's by bf16 operand roundings of e_r - B_r instead of e_r (same error class).  Without
 * the flag any image runs as before (a folded image is also a valid plain image: mean ~ 0 is computed, beta = 0 added). */
#define CGNN_STREAM_FOLDED 1
int cgnn_edge_stream_w8_supported(int32_t latent, int32_t num_hidden_layers, int32_t fixed_k);
int cgnn_edge_stream_image_build_w8(const cgnn_mlp* rounds, int32_t num_rounds, const cgnn_mlp* encoder, int32_t latent,
                                    void* image, size_t image_bytes, void* stream);
int cgnn_edge_stream_run_w8(const void* image, size_t image_bytes, int32_t latent, int32_t num_hidden_layers,
                            int32_t num_rounds, int32_t enc_in_dim, const void* ps_all, const void* pd_all,
                            int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                            const float* e_in, float* e_out, const float* edge_attr, int32_t ld_attr, int32_t lag,
                            int32_t fixed_k, int32_t p_format, int32_t flags, void* stream);

/* ---- backward of a row-wise MLP (+LayerNorm): the node stream of train.py:263 ------------------------
 * In reference-faithful mode only the node path carries gradient (the edge models' parameters get
 * none), so training needs the backward of cgnn_mlp_rows / cgnn_node_block and the transpose of the
 * aggregation (= cgnn_aggregate with src and dst swapped, general path).
 *
 * cgnn_mlp_backward recomputes the forward of one 32-row tile from its inputs (no activations are kept from the
 * forward pass), then walks the chain backwards:
 *     y = [LayerNorm](W_nh relu(... relu(W_0a u1 + W_0b u2 + b_0) ...) + b_nh),      given dy = dL/dy
 * and writes, all row-major f32:
 *   buf->h[l]    [n, H]        post-ReLU activation of hidden layer l (recomputed)          l = 0..nh-1
 *   buf->g_a[l]  [n, H]        dL/d(pre-activation of hidden layer l)
 *   buf->g_o     [n, 32*ceil(out/32)]  dL/d(output layer's pre-LayerNorm output)
 *   buf->zhat    [n, out]      normalised output (LayerNorm only; else unused)
 *   du1 / du2                  dL/du1, dL/du2 (either may be NULL when not needed)
 * The parameter gradients then are plain reductions over rows (cgnn_weight_grad, cgnn_col_dot):
 *   dW_nh = g_o^T h[nh-1], dW_l = g_a[l]^T h[l-1], dW_0a = g_a[0]^T u1, dW_0b = g_a[0]^T u2, db = column sums,
 *   dgamma = colsum(dy * zhat), dbeta = colsum(dy).
 * `fwd` holds the forward weights (layer[0] = W_0a; fwd_part2 = W_0b or NULL), `bwd` the TRANSPOSED weights
 * packed the same way (bwd->layer[l] = W_l^T, in_dim = out_l, out_dim = in_l; bwd_part2 = W_0b^T or NULL).
 * Arithmetic by (fwd->precision, bwd->precision): (CGNN_F32, CGNN_F32) exact; (CGNN_F32X3, CGNN_F32X3) three bf16 terms;
 * (CGNN_F16X2, CGNN_F32X3) recomputes the forward on two fp16 terms (inputs must respect its range: latents, not raw
 * features) and runs the gradient chain, whose values can be far below fp16's range, on three bf16 terms.
 * fwd_part2 / bwd_part2 are packed like fwd / bwd. */
typedef struct {
    float* h[CGNN_MAX_HIDDEN_LAYERS];
    float* g_a[CGNN_MAX_HIDDEN_LAYERS];
    float* g_o;
    float* zhat;
} cgnn_mlp_bwd_buffers;

int cgnn_mlp_backward(const cgnn_mlp* fwd, const cgnn_linear* fwd_part2, const cgnn_mlp* bwd,
                      const cgnn_linear* bwd_part2, const float* u1, int32_t ld1, const float* u2, int32_t ld2,
                      const float* dy, int32_t ld_dy, int64_t n, const cgnn_mlp_bwd_buffers* buf,
                      float* du1, int32_t ld_du1, float* du2, int32_t ld_du2, void* stream);

/* dw[o, col0 + i] += sum_r g[r, o] * a[r, i]   (o < out_dim, i < in_dim; f32 MFMA, float atomics across row chunks:
 * the caller zeroes dw; summation order over row chunks is not reproducible).  db (optional): db[o] += sum_r g[r, o],
 * the bias gradient, from the operands already loaded. */
int cgnn_weight_grad(const float* g, int32_t ld_g, int32_t out_dim, const float* a, int32_t ld_a, int32_t in_dim,
                     int64_t n, float* dw, int32_t ld_dw, int32_t col0, float* db, void* stream);

/* cgnn_weight_grad with the same bits on every run, any shape: the row chunks' products go to `workspace`
 * (cgnn_weight_grad_workspace_bytes(n, out_dim, in_dim) bytes of device memory, contents irrelevant) and a second kernel adds
 * them in a fixed order: dw[o, col0 + i] += ..., db[o] += ... (db may be NULL).  Exact f32 MFMA as cgnn_weight_grad. */
size_t cgnn_weight_grad_workspace_bytes(int64_t n, int32_t out_dim, int32_t in_dim);
int cgnn_weight_grad_ordered(const float* g, int32_t ld_g, int32_t out_dim, const float* a, int32_t ld_a, int32_t in_dim,
                             int64_t n, float* dw, int32_t ld_dw, int32_t col0, float* db, void* workspace,
                             size_t workspace_bytes, void* stream);

/* The same reduction for a 128 x 128 Linear (out_dim == in_dim == 128) on the bf16 matrix cores, reproducible:
 * g and a are split into three bf16 terms in registers (six products, f32 accumulation: f32-level error, f32 exponent
 * range), every wave keeps the whole 128 x 128 product of its row range, writes it to `workspace`
 * (cgnn_weight_grad_x3_workspace_bytes() bytes, device memory, contents irrelevant) and a second kernel adds the
 * partial products in a fixed order: dw[o, col0 + i] += ..., db[o] += ... (db may be NULL) with the same bits on
 * every run.  g and a must be 16-byte aligned with ld_g % 4 == ld_a % 4 == 0 (else CGNN_ERR_UNSUPPORTED: use
 * cgnn_weight_grad). */
size_t cgnn_weight_grad_x3_workspace_bytes(void);
int cgnn_weight_grad_x3(const float* g, int32_t ld_g, const float* a, int32_t ld_a, int64_t n, float* dw, int32_t ld_dw,
                        int32_t col0, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* out[c] += sum_r a[r, c] * (b ? b[r, c] : 1)   for c < width (bias / LayerNorm-affine gradients). */
int cgnn_col_dot(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out,
                 void* stream);
/* Both LayerNorm-affine gradients from one pass over dy: out_ab[c] += sum_r a[r, c] * b[r, c] (dgamma, a = dy,
 * b = zhat) and out_a[c] += sum_r a[r, c] (dbeta). */
int cgnn_col_dot2(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out_ab,
                  float* out_a, void* stream);

/* The same sums reproducibly: per-workgroup partial sums in `workspace` (cgnn_col_dot_workspace_bytes(n, width) bytes of
 * device memory, contents irrelevant), added in a fixed order by a second kernel -- the same bits on every run (the two
 * entries above meet in float atomics).  out_a NULL: out_ab only (b may then be NULL: plain column sums). */
size_t cgnn_col_dot_workspace_bytes(int64_t n, int32_t width);
int cgnn_col_dot_ordered(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out_ab,
                         float* out_a, void* workspace, size_t workspace_bytes, void* stream);

/* ---- transpose of the aggregation (backward of graph_network.py:92 `propagate`) --------------------------------
 * cgnn_csr_build groups an edge list by `key`: row_ptr[r]..row_ptr[r+1] delimit, in col[], the `val` (or, when val
 * is NULL, the edge index) of every edge whose key is r, in ascending order (deterministic).  With key = senders and
 * val = receivers this is the sender-major adjacency; cgnn_aggregate_csr then sums, for every sender, the rows of its
 * receivers: out[r] = sum_{p in row} table[col[p]] -- an atomic-free gather like the forward, summed in a fixed order.
 * row_ptr: num_rows + 1 ints; col: num_edges ints; workspace: cgnn_csr_workspace_bytes(num_rows).  cgnn_csr_build
 * validates the keys and therefore synchronises the stream once (it runs once per graph). */
size_t cgnn_csr_workspace_bytes(int64_t num_rows);
int cgnn_csr_build(const int32_t* key, const int32_t* val, int64_t num_edges, int64_t num_rows, int32_t* row_ptr,
                   int32_t* col, void* workspace, size_t workspace_bytes, void* stream);
int cgnn_aggregate_csr(const float* table, const int32_t* row_ptr, const int32_t* col, int64_t num_rows, int32_t width,
                       float* out, void* stream);
/* The same sum plus up to two row-aligned addends (either may be NULL):
 *   out[r] = add1[r] + add2[r] + sum_{p in row r} table[col[p]]
 * -- the backward of a residual round, dx_i = dx_{i+1} + du1 + A^T du2, in one pass.  out may alias add1 or add2. */
int cgnn_aggregate_csr_add(const float* table, const int32_t* row_ptr, const int32_t* col, int64_t num_rows,
                           int32_t width, const float* add1, const float* add2, float* out, void* stream);

/* ---- K1+K2+K3: periodic k-NN graph + edge features -----------------------------
 * For each query particle q (all n, or query_ids[0..nq) when non-NULL) the k
 * nearest of the 27 periodic images of all particles, ordered by (float32 squared
 * distance, image index); the particle itself comes first (distance 0).
 *   senders[i*k + j]      = index in [0,n) of the j-th neighbour of query i
 *   edge_attr[(i*k+j)*4..] = (pos[sender] - pos[query], |.|)   NOT minimum-image
 * reference data_utils.py:9-33 (27 shifts), :148-152 (torch_cluster.knn + swap +
 * mapping), :162-164 (edge features). edge_attr may be NULL. */
size_t cgnn_knn_workspace_bytes(int64_t n, int32_t k);
int cgnn_knn_periodic(const float* pos, int64_t n, float box_size, int32_t k,
                      const int32_t* query_ids, int64_t nq,
                      int32_t* senders, float* edge_attr,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Optional by-product of the last cgnn_knn_periodic on this workspace: the
 * cell-sorted particle order (a locality-improving permutation), perm[i] = original
 * index of the i-th particle in sorted order. */
int cgnn_knn_sorted_order(const void* workspace, int64_t n, int32_t* perm, void* stream);

/* ---- window -> node features (reference data_utils.py:91-92, :100-107, :127-145) -------------------
 * pos_seq [W, N, 3] and temp_seq [W, N] (frame-major, as the drivers hold a window), optional additive
 * noise pos_noise [N, W, 3] / temp_noise [N, W] (NULL = none).  Writes
 *   x[n, 3t + c]      = ((wrap(p[t+1] - p[t]) / dt) - vel_mean) / vel_std,   t < W-1   (p = remainder(pos + noise, box))
 *   x[n, 3(W-1) + t]  = ((temp[t] + noise) - temp_mean) / temp_std,          t < W
 *   recent_pos[n, :]  = p[W-1]
 * with wrap(d) = d + box if d < -box/2, then d - box if d > box/2: float32, one rounding per operation, the
 * order of the reference's tensor expressions. */
int cgnn_window_features(const float* pos_seq, const float* temp_seq, const float* pos_noise, const float* temp_noise,
                         int32_t window, int64_t n, float box_size, float dt, float vel_mean, float vel_std,
                         float temp_mean, float temp_std, float* x, float* recent_pos, void* stream);

/* ---- K11: momentum-conservation term ------------------------------------------
 * sums[g, c] = sum_{i: batch[i]==g} acc[i, c] in float64 (batch sorted ascending,
 * NULL = one graph); reference train.py:107-118.  sums is [num_graphs, width] f64,
 * zeroed by the call. */
int cgnn_segment_colsum(const float* acc, const int32_t* batch, int64_t n, int32_t width,
                        int32_t num_graphs, double* sums, void* stream);

/* ---- halo pack / unpack (multi-GPU ghost rows) and row permutation -------------
 * out[i, :] = table[idx[i], :]   and   table[idx[i], :] = rows[i, :]  */
int cgnn_gather_rows(const float* table, const int32_t* idx, int64_t n_idx, int32_t width,
                     float* out, void* stream);
int cgnn_scatter_rows(const float* rows, const int32_t* idx, int64_t n_idx, int32_t width,
                      float* table, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGNN_H_ */
