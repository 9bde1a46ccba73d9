/* gat_mi355.h -- C ABI of libgat_mi355.so: the GAT layer hot path on MI355X (gfx950).
 *
 * The reference (gokcantali/gat) has no FFI layer of its own: its hot path is the
 * Python call  GATConv(...)(x, edge_index)  at /root/reference/gat/model.py:26,28,30
 * (construction) and :40,:43,:46 (forward), differentiated by autograd at :54.  This
 * header is the boundary a host language binds to replace that call; each entry point
 * cites the step of that call it stands for (steps as listed in SURVEY.md 8a).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless it says "host";
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*) except
 *    gatmi_graph_build, which synchronises the stream once to report its counts;
 *  - no entry point allocates device memory: scratch comes in through `ws`
 *    (size from the matching *_workspace_bytes query), 256-byte aligned;
 *  - return value 0 = success; anything else is an error code and
 *    gatmi_last_error() returns a thread-local message;
 *  - fp32 values, int32 indices inside the library, int64 edge_index at the door.
 *
 * Feature rows live in a head-padded layout: a node's projected features are
 * H blocks of Cp = round_up(C, 4) floats (pad = 0), Dp = H*Cp floats per node, so that
 * every head starts on a 16-byte boundary and one lane moves one float4.
 */
#ifndef GAT_MI355_H
#define GAT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GATMI_OK 0
#define GATMI_ERR_ARG 1     /* bad argument / shape */
#define GATMI_ERR_HIP 2     /* a HIP runtime call failed */
#define GATMI_ERR_RANGE 3   /* node id outside [0, N) in edge_index */
#define GATMI_ERR_WS 4      /* workspace too small */

#define GATMI_MASK_NONE 0   /* eval mode / p = 0 */
#define GATMI_MASK_BYTES 1  /* keep-mask given as bytes [E', H] in dst-CSR edge order */
#define GATMI_MASK_HASH 2   /* keep-mask derived on the fly from (seed, edge, head) */
#define GATMI_MASK_HASH_DEV 3 /* as HASH, the 64-bit seed read from device memory: `mask` points at it
                               (8-byte aligned).  For steps captured in a HIP graph: the caller bumps
                               the seed on the device between replays and every replay draws a new mask */

const char *gatmi_last_error(void);
int gatmi_version(void);

/* Kernel-selection switch for A/B runs and for the parity tests of the fallback kernels
 * (no counterpart in the reference).  name = one of GATMI_STREAM, GATMI_FWD3, GATMI_BWD3,
 * GATMI_BWD2, GATMI_EDGE8, GATMI_GROUPROWS, GATMI_GEMM_V2, GATMI_QGROUP, GATMI_OUTGROUP,
 * GATMI_KRES, GATMI_KREG, GATMI_KREG_TAIL, GATMI_KREG_WIDE, GATMI_DW_STREAM, GATMI_DW_TAIL, GATMI_SPLIT3,
 * GATMI_DW16_ROWS, GATMI_ADST_FUSED, GATMI_BWD_HOIST, GATMI_KREG_DBUF, GATMI_FUSED, GATMI_GCOMPACT,
 * GATMI_NARROW
 * (what each selects: DESIGN.md section 5, "A/B switches"); value >= 0 sets it, value < 0 hands
 * the decision back to the environment variable of the same name. */
int gatmi_set_option(const char *name, int value);

/* Per-kernel timing for benchmarks (no counterpart in the reference, which only prints
 * epoch wall time at model.py:89,105): while enabled, every kernel the library launches
 * is bracketed by HIP events on its launch stream.  gatmi_prof_read waits for the
 * recorded events, copies up to max_records (tag[32], milliseconds) pairs out (host
 * pointers), clears the list and returns the number copied. */
int gatmi_prof_enable(int on);
int gatmi_prof_read(char *tags /* host [max_records][32] */, float *ms /* host */,
                    int max_records);

/* One half of the graph layout: a CSR over `num_rows` rows plus its segment plan.
 * Rows longer than seg_len edges are cut into segments; every segment is a work item
 * (item_row, item_beg); item_slot >= 0 marks a segment of a cut row and names its
 * partial-result slot, split_row/split_slot list the cut rows for the merge pass. */
typedef struct gatmi_plan {
    int32_t n_items;
    int32_t n_split_rows;
    int32_t n_slots;
    int32_t _pad;
    const int32_t *item_row;   /* [n_items] */
    const int32_t *item_beg;   /* [n_items] */
    const int32_t *item_slot;  /* [n_items] */
    const int32_t *split_row;  /* [n_split_rows] */
    const int32_t *split_slot; /* [n_split_rows + 1] */
} gatmi_plan;

/* Destination-sorted CSR with self-loops, and its transpose.
 * Stands for: remove_self_loops + add_self_loops inside GATConv.forward (SURVEY 8a row
 * a4; call sites model.py:40,43,46) applied to the edge_index built at
 * /root/reference/gat/converter.py:12-15, followed by a stable sort by destination.
 *   rowptr[i]..rowptr[i+1] : in-edges of node i in normalised-list order
 *   col[e]  : source of edge e          perm[e] : index of e in the normalised list
 *   rowptr_t/col_t : same edges grouped by source (col_t = destination), ordered by e
 *   pos_t[t] : dst-CSR position of transposed edge t;   d2s[e] : its inverse        */
typedef struct gatmi_graph {
    int32_t num_src;    /* rows of the feature table the edges gather from (all nodes)      */
    int32_t num_dst;    /* destination rows; == num_src except for a destination partition  */
    int32_t num_edges;  /* E' */
    int32_t seg_len;
    int32_t dst_offset; /* partition only: source id of destination row 0 (set by caller)   */
    int32_t _pad;
    const int32_t *rowptr, *col, *perm;
    const int32_t *rowptr_t, *col_t, *pos_t, *d2s;
    gatmi_plan plan;   /* over destination rows */
    gatmi_plan plan_t; /* over source rows */
} gatmi_graph;

/* Upper bounds a caller needs to allocate the arrays gatmi_graph_build fills. */
typedef struct gatmi_graph_sizes {
    int64_t max_edges;      /* E + N (E when add_self_loops == 0) */
    int64_t max_items;      /* per plan */
    int64_t max_split_rows; /* per plan */
    int64_t workspace_bytes;
} gatmi_graph_sizes;

int gatmi_graph_sizes_query(int64_t num_edges, int64_t num_src, int64_t num_dst,
                            int add_self_loops, int seg_len, gatmi_graph_sizes *out /* host */);

/* Fills the arrays `g` points to (caller-allocated to the bounds above; the const is
 * cast away) and sets the counts in *g (host struct).  edge_index: int64 [2, E].
 * Synchronises `stream` before returning.
 * A destination partition (one GPU's share of a large graph) is the
 * rectangular case: sources 0..num_src-1 are all nodes, destinations 0..num_dst-1 are
 * the partition's own nodes renumbered from 0; add_self_loops must then be 0 (the
 * partitioner appends each own node's loop itself). */
int gatmi_graph_build(const int64_t *edge_index, int64_t num_edges, int64_t num_src,
                      int64_t num_dst, int add_self_loops, int seg_len, gatmi_graph *g /* host */,
                      void *ws, size_t ws_bytes, void *stream);

/* Shape of one layer.  C = out_channels per head, Cp = round_up(C, 4). */
typedef struct gatmi_shape {
    int32_t num_nodes;
    int32_t in_features;  /* F */
    int32_t heads;        /* H */
    int32_t channels;     /* C */
} gatmi_shape;

/* Projection h = lin(x) into the head-padded layout, plus the
 * destination logit a_dst[n,h] = <h[n,h,:], att_dst[h,:]> (row a6).  fp32 in, fp32 out, at the
 * fp32 parity bar: fp32 MFMA, or -- tall layers with F % 8 == 0, Cp % 8 == 0 and at most 256 columns
 * either way -- six bf16 MFMAs per product on operands cut exactly into three bf16 pieces each
 * (gemm_split3.hip; gatmi_set_option("GATMI_SPLIT3", 0) keeps them on the fp32 MFMA kernels).  The
 * workspace query covers both.
 *   x [N,F]   weight [H*C,F]   att_dst [H*C]   ->   hp [N,Dp]   a_dst [N,H]           */
size_t gatmi_project_workspace_bytes(const gatmi_shape *s);
int gatmi_project_fwd(const gatmi_shape *s, const float *x, const float *weight,
                      const float *att_dst, float *hp, float *a_dst,
                      void *ws, size_t ws_bytes, void *stream);

/* Backward of the projection (row a11, last two formulae): d_weight = d_hp^T x,
 * d_x = d_hp W (skipped when d_x == NULL).   d_hp [N,Dp] -> d_weight [H*C,F], d_x [N,F]
 * d_a_dst != NULL ([N,H], with att_dst [H*C]): d_hp is what gatmi_attn_bwd leaves when it
 * hands out d_a_dst, and the missing term d_a_dst[n,h] * att_dst[h,:] is added while the
 * d_hp tiles are staged for the two GEMMs (it never goes through HBM). */
int gatmi_project_bwd(const gatmi_shape *s, const float *x, const float *weight,
                      const float *d_hp, const float *d_a_dst, const float *att_dst,
                      float *d_weight, float *d_x, void *ws, size_t ws_bytes, void *stream);

/* Arguments of the fused attention kernels. */
typedef struct gatmi_attn {
    gatmi_shape shape;
    float negative_slope;  /* LeakyReLU slope, GATConv default 0.2 */
    float dropout_p;       /* attention dropout probability (model.py:26,28,30)      */
    int32_t mask_mode;     /* GATMI_MASK_*                                            */
    int32_t feat_bf16;     /* 0: hp is float32; 1: hp is bfloat16 (see "bf16 feature tables") */
    uint64_t seed;         /* GATMI_MASK_HASH                                         */
    const uint8_t *mask;   /* GATMI_MASK_BYTES: [E', H] keep flags, dst-CSR order     */
    const void *hp;        /* [N, Dp] projected features, float32 or bfloat16         */
    const float *a_dst;    /* [N, H]                                                  */
    const float *att_src;  /* [H*C]                                                   */
    const float *att_dst;  /* [H*C]   (backward only)                                 */
    const float *bias;     /* [H*C] or NULL                                           */
} gatmi_attn;

/* Fused logits + LeakyReLU + segment softmax + attention dropout + aggregation + bias
 * (rows a6-a10): one pass over the dst-CSR, each source row gathered once.
 *   out [N, H*C];  lse [N, H] = log-sum-exp of the row's logits (NULL when no backward) */
size_t gatmi_attn_fwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s);
int gatmi_attn_fwd(const gatmi_graph *g, const gatmi_attn *a, float *out, float *lse,
                   void *ws, size_t ws_bytes, void *stream);

/* gatmi_attn_fwd without a workspace argument.  NO COUNTERPART IN THE REFERENCE: it is for
 * callers that cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The only scratch the forward uses, the partial
 * (max, sum, accumulator) triples of destination rows cut into segments, lives in a device
 * array of the library itself: GATMI_STATIC_SCRATCH_BYTES per GPU the library runs on, held
 * until the process ends.  Same kernels and bit-identical results as gatmi_attn_fwd; returns
 * GATMI_ERR_WS when gatmi_attn_fwd_workspace_bytes(g, &a->shape) exceeds the array (about
 * 7,900 segments of a 4x64 layer).  Calls no hipMalloc / hipFree.  All calls on one GPU share
 * the array: calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
#define GATMI_STATIC_SCRATCH_BYTES (8u << 20)
int gatmi_attn_fwd_static(const gatmi_graph *g, const gatmi_attn *a, float *out, float *lse,
                          void *stream);

/* The keep-mask GATMI_MASK_HASH derives from (seed, edge, head), written out as bytes
 * [E', H] in dst-CSR order: what F.dropout's Bernoulli draw is to the reference (row a8).
 * Running GATMI_MASK_BYTES with this mask reproduces the GATMI_MASK_HASH run exactly. */
int gatmi_dropout_mask(const gatmi_graph *g, int heads, float dropout_p, uint64_t seed,
                       uint8_t *mask, void *stream);

/* Backward of rows a6-a10 (row a11): a per-node preparation, ONE pass over the
 * transposed CSR that gathers each grad_out row once and does all per-edge work
 * (softmax/LeakyReLU/dropout gradients, d h accumulation, d a_src), a light pass over the
 * dst-CSR for d a_dst, then the column reductions for d att_src, d att_dst, d bias.
 *   out, lse : what gatmi_attn_fwd produced;  grad_out [num_dst, H*C]
 *   -> d_hp [num_src, Dp], d_att_src [H*C], d_att_dst [H*C], d_bias [H*C] (NULL to skip)
 * d_a_dst_out == NULL (square graphs only): d_hp is complete on return.
 * d_a_dst_out != NULL: [num_dst, H] receives d a_dst and d_hp lacks the term
 * d_a_dst[i] * att_dst of its destination rows: gatmi_project_bwd adds it on the fly
 * (saves a read-modify-write of d_hp); a destination partition adds it after the
 * reduce-scatter of d_hp, there or with gatmi_add_dst_term. */
size_t gatmi_attn_bwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s);
int gatmi_attn_bwd(const gatmi_graph *g, const gatmi_attn *a, const float *out,
                   const float *lse, const float *grad_out, float *d_hp,
                   float *d_att_src, float *d_att_dst, float *d_bias, float *d_a_dst_out,
                   void *ws, size_t ws_bytes, void *stream);

/* ---- Source chunks: the same layer call (model.py:40,43,46; backward model.py:54) when the
 * feature table arrives in pieces -- the pipelined all-gather of a destination partition
 * ("chunk the all-gather by source range and start aggregating rows whose sources have
 * arrived"; no counterpart in the single-process reference).  Chunk k owns a slice of the table and
 * a rectangular layout of its own: the edges whose sources lie in the slice, source ids counted
 * from the slice's first row (num_src = rows of the slice), all num_dst destination rows
 * (gatmi_graph_build with add_self_loops = 0; a destination's loop goes into the chunk that holds
 * its own row).  a->hp points at the slice.  Nothing allocates; the kernels are those of
 * gatmi_attn_fwd / gatmi_attn_bwd.
 *
 * Forward: gatmi_attn_fwd_partial per chunk (as soon as its slice has arrived) leaves the
 * un-normalised online-softmax triple of every destination row over the chunk's edges,
 *     m [num_dst, H] running maximum (-inf: no edge),  s [num_dst, H] = sum exp(z - m),
 *     acc [num_dst, Dp] = sum keep/(1-p) exp(z - m) h[j];
 * gatmi_attn_fwd_combine joins the triples of all chunks, in the order given (results are bitwise
 * reproducible for a fixed chunking), into out and lse exactly as gatmi_attn_fwd joins the
 * segments of a cut row.  Workspace of the partial call: gatmi_attn_fwd_workspace_bytes(g, s). */
#define GATMI_MAX_PARTS 16
typedef struct gatmi_attn_partial {
    float *m, *s, *acc;
} gatmi_attn_partial;
int gatmi_attn_fwd_partial(const gatmi_graph *g, const gatmi_attn *a, const gatmi_attn_partial *p,
                           void *ws, size_t ws_bytes, void *stream);
int gatmi_attn_fwd_combine(const gatmi_shape *s /* num_nodes = destination rows */, int n_parts,
                           const gatmi_attn_partial *parts /* host array [n_parts] */,
                           const float *bias, float *out, float *lse, void *stream);

/* Backward over source chunks.  gatmi_attn_bwd_begin once (per-node preparation into `node_ws`,
 * which the later calls read: gatmi_attn_bwd_node_workspace_bytes; a->a_dst, a->bias, shape and
 * feat_bf16 are used, a->hp is not).  gatmi_attn_bwd_chunk per chunk, in any order, on the same
 * stream: d_hp [g->num_src, Dp] receives the COMPLETE gradient of the chunk's source rows from
 * this rank's edges (ready for that chunk's reduce-scatter while the next chunk computes; the
 * d_a_dst * att_dst term of destination rows is not in it, as with gatmi_attn_bwd's
 * d_a_dst_out); d_a_dst [num_dst, H] is written when accumulate == 0 and added to otherwise;
 * `chunk` < max_chunks names the chunk's slot for the d_att_src partial sums.
 * gatmi_attn_bwd_end: a->hp = the table of the destination rows themselves [num_dst, Dp];
 * -> d_att_src, d_att_dst, d_bias [H*C] (this rank's partial sums; NULL to skip). */
size_t gatmi_attn_bwd_node_workspace_bytes(const gatmi_shape *s, int max_chunks, int feat_bf16);
size_t gatmi_attn_bwd_chunk_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s);
int gatmi_attn_bwd_begin(const gatmi_attn *a, int max_chunks, const float *out, const float *lse,
                         const float *grad_out, void *node_ws, size_t node_ws_bytes, void *stream);
int gatmi_attn_bwd_chunk(const gatmi_graph *g, const gatmi_attn *a, int chunk, int max_chunks,
                         int accumulate, const float *grad_out, void *node_ws, size_t node_ws_bytes,
                         float *d_hp, float *d_a_dst, void *ws, size_t ws_bytes, void *stream);
int gatmi_attn_bwd_end(const gatmi_attn *a, int max_chunks, const float *d_a_dst, void *node_ws,
                       size_t node_ws_bytes, float *d_att_src, float *d_att_dst, float *d_bias,
                       void *stream);

/* ---- bf16 feature tables (BASELINE.json north_star: "bf16 stated separately").  The edge
 * kernels are bound by the bytes of the rows they gather, so the projected features can be kept
 * as bfloat16 rows (same head-padded layout, 2 bytes per channel): gatmi_attn.feat_bf16 = 1 and
 * hp = the bf16 table gatmi_project_fwd_bf16 writes.  Every value is widened to f32 in registers;
 * logits, softmax, accumulation, out, lse and all gradients stay f32.  The backward gathers
 * grad_out rows from a bf16 copy it makes itself (S_i is taken from the same rounded rows, d_bias
 * from the unrounded ones), so the gradients are those of the function "h and the gradient
 * arriving at the aggregate rounded to bf16" -- tests state that tolerance separately from the
 * f32 path's.  Needs rows of more than 32 float4 slots (gatmi_attn_bf16_supported); the backward
 * workspace is gatmi_attn_bwd_workspace_bytes_bf16. */
int gatmi_attn_bf16_supported(const gatmi_shape *s);
size_t gatmi_attn_bwd_workspace_bytes_bf16(const gatmi_graph *g, const gatmi_shape *s);

/* The projection of such a layer on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, f32
 * accumulation): x, weight, d_hp stay float32 in memory and are rounded to bf16 on their way into
 * the MFMA; the forward writes the bf16 table hp16 [N, Dp] and a_dst [N, H] taken from the ROUNDED
 * rows.  Same arguments as gatmi_project_fwd / gatmi_project_bwd otherwise.  Covered shapes
 * (gatmi_project_bf16_supported): F % 8 == 0, round_up(C, 4) % 8 == 0, H * round_up(C, 4) a multiple
 * of 32, both <= 256, and gatmi_attn_bf16_supported. */
int gatmi_project_bf16_supported(const gatmi_shape *s);
size_t gatmi_project_bf16_workspace_bytes(const gatmi_shape *s);
int gatmi_project_fwd_bf16(const gatmi_shape *s, const float *x, const float *weight,
                           const float *att_dst, void *hp16, float *a_dst, void *ws, size_t ws_bytes,
                           void *stream);
int gatmi_project_bwd_bf16(const gatmi_shape *s, const float *x, const float *weight,
                           const float *d_hp, const float *d_a_dst, const float *att_dst,
                           float *d_weight, float *d_x, void *ws, size_t ws_bytes, void *stream);

/* d_hp[i, h, :] += d_a_dst[i, h] * att_dst[h, :] for i < s->num_nodes (rows of one shard). */
int gatmi_add_dst_term(const gatmi_shape *s, float *d_hp, const float *d_a_dst,
                       const float *att_dst, void *stream);

/* ---- Input-side aggregation: the same layer call (model.py:40, rows a5-a11) for a layer
 * whose input needs no gradient -- the reference's conv1, whose x is `data.x` (model.py:38-40)
 * -- and whose rows are narrower than its output (F < H*Cp).  The projection moves behind
 * the aggregation,
 *     out[i,h,:] = W_h (sum_e alpha~_e x_j) + bias_h,    a_src[j,h] = <x_j, W_h^T att_src_h>,
 * so every edge gathers an x row (F floats) instead of a projected row (H*C floats); the
 * backward runs on the destination CSR alone and yields the parameter gradients (there is
 * no d_x).  Results equal gatmi_project_fwd + gatmi_attn_fwd up to fp32 re-association.
 *
 * Step 1, gatmi_inagg_pack: rows of x -> rows of the packed source table
 *     xs[n] = x[n, 0..F) zero-padded to Fp = round_up(F,4) | a_src[n, 0..H) | zero pad
 * (Fs = Fp + round_up(H,4) floats per row: the source logit rides at the end of the row an
 * edge gathers anyway) and a_dst [n, H].  A destination partition packs its own rows and
 * all-gathers xs; a_dst stays local.
 * Step 2, gatmi_inagg_fwd / gatmi_inagg_bwd over the graph. */
int gatmi_inagg_supported(const gatmi_shape *s); /* 1: H <= 32 and Fp/4 <= 4*64/next_pow2(H) */
size_t gatmi_inagg_xs_floats(const gatmi_shape *s); /* Fs */
size_t gatmi_inagg_z_floats(const gatmi_shape *s);  /* H * Fp */

size_t gatmi_inagg_pack_workspace_bytes(const gatmi_shape *s);
int gatmi_inagg_pack(const gatmi_shape *s /* num_nodes = rows to pack */, const float *x,
                     const float *weight, const float *att_src, const float *att_dst,
                     float *xs /* [rows, Fs] */, float *a_dst /* [rows, H] */,
                     void *ws, size_t ws_bytes, void *stream);

/* bf16 source rows for the gathers of this path (no counterpart in the reference; the "bf16"
 * of BASELINE.json's north_star applied to the gathered table of model.py:40's layer; tolerance
 * stated in tests/test_gpu_bf16.py).  gatmi_inagg_xs16_bytes: bytes per row (256), 0 when the
 * shape is not covered (F <= 112; H <= 4, or 5..8 heads with F > 32).  gatmi_inagg_table16: xs [num_nodes, Fs] -> xs16. */
size_t gatmi_inagg_xs16_bytes(const gatmi_shape *s);
/* gatmi_inagg_pack that also writes the bf16 table of the rows it packs (the same bytes
 * gatmi_inagg_table16 makes of xs, without the pass that reads xs back). */
int gatmi_inagg_pack16(const gatmi_shape *s /* num_nodes = rows to pack */, const float *x,
                       const float *weight, const float *att_src, const float *att_dst,
                       float *xs /* [rows, Fs] */, float *a_dst /* [rows, H] */,
                       void *xs16 /* [rows, 256 bytes] */, void *ws, size_t ws_bytes, void *stream);
int gatmi_inagg_table16(const gatmi_shape *s /* num_nodes = rows */, const float *xs, void *xs16,
                        void *stream);

typedef struct gatmi_inagg {
    gatmi_shape shape;     /* num_nodes = destination rows                              */
    float negative_slope;
    float dropout_p;
    int32_t mask_mode;     /* GATMI_MASK_*                                              */
    int32_t stream;        /* forward kernel: 0 = pick by table size (LDS-DMA streaming when
                              the packed table exceeds the 256 MiB Infinity Cache), 1 = streaming,
                              2 = register pipeline; same results up to fp32 re-association */
    uint64_t seed;
    const uint8_t *mask;   /* GATMI_MASK_BYTES: [E', H] keep flags, dst-CSR order       */
    const float *xs;       /* [num_src, Fs] packed source rows (all nodes of the graph) */
    const float *a_dst;    /* [num_dst, H]                                              */
    const float *weight;   /* [H*C, F]                                                  */
    const float *att_src;  /* [H*C]                                                     */
    const float *att_dst;  /* [H*C]                                                     */
    const float *bias;     /* [H*C] or NULL                                             */
    const void *xs16;      /* NULL, or the bf16 copy of xs made by gatmi_inagg_table16: the edge
                              kernels then gather its 256-byte rows (two lines instead of four
                              for F = 100) and keep all arithmetic and every output in f32     */
} gatmi_inagg;

/*   -> out [num_dst, H*C],  z [num_dst, H*Fp] (the per-head aggregated inputs, kept for the
 *      backward),  lse [num_dst, H] (NULL when no backward follows)                     */
size_t gatmi_inagg_fwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s);
int gatmi_inagg_fwd(const gatmi_graph *g, const gatmi_inagg *a, float *out, float *z, float *lse,
                    void *ws, size_t ws_bytes, void *stream);

/*   z, lse: what gatmi_inagg_fwd produced;  grad_out [num_dst, H*C]
 *   -> d_weight [H*C, F], d_att_src [H*C], d_att_dst [H*C], d_bias [H*C] (NULL to skip).
 * On a destination partition these are the rank's partial sums (all-reduce them).      */
size_t gatmi_inagg_bwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s);
int gatmi_inagg_bwd(const gatmi_graph *g, const gatmi_inagg *a, const float *z, const float *lse,
                    const float *grad_out, float *d_weight, float *d_att_src, float *d_att_dst,
                    float *d_bias, void *ws, size_t ws_bytes, void *stream);

/* ---- Inter-layer glue of the reference's network, fused: F.elu -> BatchNorm1d -> Dropout
 * between two GAT layers (/root/reference/gat/model.py:40-42 and :43-45; SURVEY 8a row a12,
 * 8f rank 1).  y = dropout(batch_norm(elu(x))).  torch.nn.BatchNorm1d semantics: training
 * normalises with the batch's biased variance and updates running_mean / running_var (the
 * latter with the unbiased variance) by `momentum`; eval uses the running statistics.  The
 * dropout mask is the counter hash (seed, row, column), active only in training. */
typedef struct gatmi_glue {
    int64_t rows;          /* N < 2^32 */
    int32_t cols;          /* D <= 1024 */
    int32_t training;
    float eps;             /* BatchNorm1d eps (1e-5)                                   */
    float momentum;        /* BatchNorm1d momentum (0.1)                               */
    float dropout_p;       /* nn.Dropout p (model.py:31), [0, 1)                       */
    int32_t seed_on_device; /* 1: `seed` holds the ADDRESS of a 64-bit seed in device memory (HIP graphs) */
    uint64_t seed;
    const float *gamma;    /* BatchNorm weight [D]                                     */
    const float *beta;     /* BatchNorm bias [D]                                       */
    float *running_mean;   /* [D], updated in place in training (NULL: not tracked)    */
    float *running_var;    /* [D]                                                      */
} gatmi_glue;

size_t gatmi_glue_workspace_bytes(int cols);
/*   x [N, D] -> y [N, D];  save_mean, save_rstd [D]: the statistics used (for the backward) */
int gatmi_glue_fwd(const gatmi_glue *g, const float *x, float *y, float *save_mean,
                   float *save_rstd, void *ws, size_t ws_bytes, void *stream);
/*   grad_y [N, D] -> grad_x [N, D], grad_gamma [D], grad_beta [D] (NULL to skip) */
int gatmi_glue_bwd(const gatmi_glue *g, const float *x, const float *grad_y,
                   const float *save_mean, const float *save_rstd, float *grad_x,
                   float *grad_gamma, float *grad_beta, void *ws, size_t ws_bytes, void *stream);
/* The same glue on a ROW PARTITION (one rank's rows of a destination-partitioned network; SURVEY 8e:
 * "BatchNorm batch statistics need a cross-GPU all-reduce of per-column sums for exact parity" --
 * model.py:27,29,41,44 on all N rows).  The two reductions and the two passes are calls of their own
 * and the caller combines between them; g->rows = this rank's rows (0 is allowed), dropout as above:
 *   gatmi_glue_stats      -> mean [D], m2 [D] = sum (elu(x) - mean)^2 over THESE rows (caller: Chan's
 *                            combination of (rows, mean, m2) over the ranks -> mean, rstd of all rows)
 *   gatmi_glue_apply      y = dropout(gamma (elu(x) - mean) rstd + beta) with the combined statistics
 *   gatmi_glue_bwd_sums   -> sums [2][D] = sum g', sum g' xhat over these rows (this rank's share of
 *                            grad_beta, grad_gamma; caller: all-reduce)
 *   gatmi_glue_bwd_apply  grad_x with the all-reduced sums and total_rows = the rows of all ranks. */
int gatmi_glue_stats(const gatmi_glue *g, const float *x, float *mean, float *m2, void *ws,
                     size_t ws_bytes, void *stream);
int gatmi_glue_apply(const gatmi_glue *g, const float *x, const float *mean, const float *rstd, float *y,
                     void *stream);
int gatmi_glue_bwd_sums(const gatmi_glue *g, const float *x, const float *grad_y, const float *mean,
                        const float *rstd, float *sums, void *ws, size_t ws_bytes, void *stream);
int gatmi_glue_bwd_apply(const gatmi_glue *g, int64_t total_rows, const float *x, const float *grad_y,
                         const float *mean, const float *rstd, const float *sums, float *grad_x,
                         void *stream);
/* The keep-mask of (seed, row, column), bytes [N, D] -- what nn.Dropout's Bernoulli draw is to
 * the reference; for inspection and tests. */
int gatmi_glue_mask(int64_t rows, int32_t cols, float dropout_p, uint64_t seed, uint8_t *mask,
                    void *stream);

/* ---- log_softmax + nll_loss tail ------------------------------------------------------------
 * Replaces  F.log_softmax(x, dim=1)  (/root/reference/gat/model.py:47) and  F.nll_loss(out, y)
 * (model.py:53, mean reduction) and their autograd backward.  logits, logp: float32
 * [rows, classes], classes <= 256; target: int64 [rows] (values outside [0, classes) are not
 * counted, as torch's ignore_index).  loss_stats: float32 [2] = {mean loss, 1 / counted rows}. */
size_t gatmi_tail_workspace_bytes(void);
int gatmi_tail_fwd(int64_t rows, int32_t classes, const float *logits, const int64_t *target,
                   float *logp, float *loss_stats, void *ws, size_t ws_bytes, void *stream);
/*   grad_loss: device scalar (NULL = 0); grad_logp [rows, classes] or NULL: a gradient arriving
 *   at the log-probabilities from elsewhere -> grad_logits [rows, classes] */
int gatmi_tail_bwd(int64_t rows, int32_t classes, const float *logp, const int64_t *target,
                   const float *loss_stats, const float *grad_loss, const float *grad_logp,
                   float *grad_logits, void *stream);

/* ---- kNN graph construction -----------------------------------------------------------------
 * Replaces  create_knn_graph(X, k)  (/root/reference/gat/converter.py:7-15): scikit-learn's
 * kneighbors_graph(X, k, mode="connectivity", include_self=True) -> COO -> edge_index.
 * x: float64 [n, f] row-major (X.values), f <= 64, k <= 16, k <= n.  edge_index: int64 [2, k*n],
 * row 0 = query i (k times, ascending), row 1 = its k nearest by ascending distance, itself
 * included; exact fp64 brute force, equal distances resolved towards the lower index. */
size_t gatmi_knn_workspace_bytes(int64_t n);
int gatmi_knn_graph(const double *x, int64_t n, int32_t f, int32_t k, int64_t *edge_index,
                    void *ws, size_t ws_bytes, void *stream);
/* The same, and tie_flags [n] (bytes, NULL to skip) = 1 for every query whose k + 1 nearest hold
 * two neighbouring distances that agree to 2^-40 of their magnitude: exact duplicates and near-ties,
 * whose order in scikit-learn's result is implementation-defined (its heap, thread count and BLAS
 * rounding).  Queries with flag 0 have a strictly ordered list -- the one kneighbors_graph returns;
 * a caller that needs converter.py:7-15's exact output on tied data recomputes with scikit-learn
 * when any flag is set (gat_amd.data.create_knn_graph, backend "auto", does). */
int gatmi_knn_graph_ties(const double *x, int64_t n, int32_t f, int32_t k, int64_t *edge_index,
                         uint8_t *tie_flags, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GAT_MI355_H */
