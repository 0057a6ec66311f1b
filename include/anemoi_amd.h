/*
 * anemoi_amd.h -- C ABI of the MI355X (gfx950) forward-path kernels.
 *
 * Drop-in boundary for the encoder -> processor -> decoder forward path of
 * ecmwf/anemoi-models.  The reference has no FFI of its own: the boundary there
 * is the set of nn.Module.forward methods in anemoi/models/layers (SURVEY.md
 * section 8b).  Each entry point below replaces the ATen / torch_geometric op
 * sequence of one such method; the citation names the reference lines it
 * replaces (paths relative to /root/reference/src/anemoi/models).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     stated otherwise; `stream` is a hipStream_t passed as void*.
 *   - node / edge feature matrices are row-major with an explicit leading
 *     dimension (in elements), so slices of wider buffers can be passed.
 *   - `dtype` selects the storage type of activations and GEMM weights:
 *     ANEMOI_F32 (exact f32 MFMA / f32 math) or ANEMOI_BF16 (bf16 storage,
 *     f32 accumulate).  Biases, LayerNorm affine parameters, edge attributes and
 *     the lin_edge weights are always f32.
 *   - every function returns ANEMOI_OK (0) or an error code; the message of the
 *     last error on the calling thread is available from anemoi_last_error().
 *     Nothing is allocated, no global state is kept, all launches go to `stream`.
 */
#ifndef ANEMOI_AMD_H
#define ANEMOI_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANEMOI_OK 0
#define ANEMOI_ERR_INVALID 1     /* bad argument; Python shim raises ValueError        */
#define ANEMOI_ERR_UNSUPPORTED 2 /* unsupported shape/dtype; shim raises NotImplementedError */
#define ANEMOI_ERR_LAUNCH 3      /* HIP launch failure; shim raises RuntimeError          */

#define ANEMOI_F32 0
#define ANEMOI_BF16 1

#define ANEMOI_ACT_NONE 0
#define ANEMOI_ACT_GELU 1 /* exact erf form, nn.GELU() default */
#define ANEMOI_ACT_SILU 2
#define ANEMOI_ACT_RELU 3

typedef void* anemoi_stream_t; /* hipStream_t */

/* ABI version (bumped on any signature change) and last error text of this thread. */
int anemoi_abi_version(void);
/* Target architecture and the hipcc / clang the library was built with (static string).  The inline-asm kernels carry
 * hand-counted wait states hipcc does not check; a report of wrong results should quote this line. */
const char* anemoi_build_info(void);
const char* anemoi_last_error(void);

/*
 * LayerNorm over the last dimension: y[r,:] = (x[r,:] - mean_r) * rstd_r * gamma + beta.
 * Replaces nn.LayerNorm calls layers/block.py:614 (layer_norm1), :491-494 (layer_norm1/2 of the
 * mapper block), node_dst_mlp[0] :349-351, node_data_extractor[0] layers/mapper.py:408-410.
 * Statistics in f32 (two-pass, biased variance, eps inside the sqrt) as ATen does.
 */
int anemoi_layer_norm(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                      int64_t ldy, int64_t rows, int C, float eps, anemoi_stream_t stream);

/* y = LayerNorm(x) + residual in one pass (each term rounded to the activation dtype first, as the two separate
 * operations of the reference round: the trailing LayerNorm of layers/mlp.py:74-84 followed by "+ x" / "+ edge_attr" in
 * layers/block.py:222, layers/conv.py:70). */
int anemoi_layer_norm_residual(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta,
                               const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t rows, int C, float eps,
                               anemoi_stream_t stream);

/* The same, and stats[r] = { rstd_r, -mean_r * rstd_r } (f32 pairs, as anemoi_row_stats leaves them) out of the same pass:
 * the training forward keeps them for anemoi_layer_norm_backward instead of reading x a second time. */
int anemoi_layer_norm_stats(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                            int64_t ldy, float* stats, int64_t rows, int C, float eps, anemoi_stream_t stream);

/*
 * Fused Linear: y = act(x @ W^T + bias) + residual, on MFMA.
 *   x [M, K] (ldx), W [N, K] row-major contiguous in `dtype` (nn.Linear layout), bias [N] f32 or NULL,
 *   residual [M, N] (ldr) in `dtype` or NULL, y [M, N] (ldy) in `out_dtype`.
 * K must be a multiple of 32 (the host pads the K dimension of x and W with zeros).
 * Replaces the nn.Linear calls layers/block.py:615-618 (lin_self/query/key/value as ONE GEMM over the
 * concatenated weight), :630 (projection, + x_skip :632 as residual), node_dst_mlp[1..3] :633
 * (GELU and the "+ out" residual fused), layers/mapper.py:112-113,416 (emb_nodes_*), :100
 * (node_data_extractor[1]), layers/attention.py:68,110 and the MLPs of layers/mlp.py:74-84.
 */
int anemoi_linear(int dtype, int out_dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                  const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act,
                  anemoi_stream_t stream);

/*
 * LayerNorm statistics of the rows of x [rows, C] (ldx): stats[r] = { rstd_r, -mean_r * rstd_r } (f32 pairs, same
 * two-pass arithmetic as anemoi_layer_norm).  First half of a LayerNorm -> Linear pair (next entry point).
 */
int anemoi_row_stats(int dtype, const void* x, int64_t ldx, float* stats, int64_t rows, int C, float eps,
                     anemoi_stream_t stream);

/*
 * Linear with the LayerNorm of its input folded in:  y = act(LN(x) @ W^T + b) + residual  computed as
 *   y[m,n] = act( rstd_m * (x @ W'^T)[m,n] + (-mean_m rstd_m) * colsum[n] + bias[n] ) + residual[m,n]
 * where the caller passes  W' = W * gamma (input columns scaled by the LayerNorm weight, in `dtype`),
 * colsum[n] = sum_k W'[n,k] (f32, from the rounded W'), bias = b + W beta (f32) and stats from anemoi_row_stats(x).
 * x is the UN-normalised input: the normalised activation is never written to memory.  Replaces the pairs
 * layer_norm1 -> lin_query/key/value/self (layers/block.py:614-618, :491-497), node_dst_mlp[0] -> node_dst_mlp[1]
 * (:349-351) and layer_norm -> lin_qkv / mlp[0] of the transformer block (:99-105).  Same shapes / alignment as
 * anemoi_linear; colsum 16-byte aligned for the fast path.
 */
int anemoi_linear_ln(int dtype, int out_dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                     const float* colsum, const float* stats, const void* residual, int64_t ldr, void* y, int64_t ldy,
                     int64_t M, int N, int K, int act, anemoi_stream_t stream);

/*
 * anemoi_linear_ln / anemoi_linear without activation, plus the LayerNorm statistics of the RESULT's rows
 * (stats_out [M, 2] f32 = { rstd, -mean * rstd } of y[m, 0:N], the format of anemoi_row_stats) for the LayerNorm that
 * consumes y next (reference layers/block.py:631-633 node_dst_mlp[0], :614 layer_norm1 of the next block,
 * layers/mapper.py:416 node_data_extractor[0]): on the bf16 fast path the GEMM's epilogue leaves per-row partial sums of
 * the bf16 values it stores in `workspace` (>= M * (N / 128) * 8 bytes, 8-byte aligned; N % 256 == 0) and a tiny kernel
 * folds them -- y is not read again.  Other shapes / dtypes (or workspace == NULL): same result through
 * anemoi_row_stats on y.  colsum / stats_in: both NULL = plain Linear, both set = LayerNorm-folded input as in
 * anemoi_linear_ln.  x, y, residual in `dtype`.
 */
int anemoi_linear_stats(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, const float* colsum,
                        const float* stats_in, const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int N,
                        int K, void* workspace, int64_t workspace_bytes, float eps, float* stats_out,
                        anemoi_stream_t stream);

/*
 * Edge attributes in CSR (destination-sorted) order:
 *   out[e, :] = [ a0[perm[e] % rows0, 0:d0] | a1[perm[e] % rows0, 0:d1] | 0 ... ]   (row stride ld_out)
 * and, when one_col >= 0, out[e, one_col] = 1 (the constant attribute that carries the lin_edge bias through the
 * folded kernel below).  `perm[e]` is the original (batched) edge id of CSR slot e.  Replaces TrainableTensor.forward
 * (layers/graph.py:37-44: repeat over the batch + concat of the trainable tensor) composed with the edge
 * gather that torch_geometric's propagate performs per block.  a1 may be NULL (d1 = 0).
 */
int anemoi_edge_attr_csr(const float* a0, int d0, const float* a1, int d1, int64_t rows0, const int32_t* perm,
                         float* out, int ld_out, int one_col, int64_t n_edges, anemoi_stream_t stream);

/*
 * Fused GraphTransformer edge phase for all destinations (K1 + K2 of SURVEY.md section 2a):
 *   e_ij   = W_e a_ij + b_e                                        (lin_edge, layers/block.py:499,620)
 *   s_ij,h = q_i,h . (k_j,h + e_ij,h) / sqrt(D)                    (layers/conv.py:134-137)
 *   alpha  = exp(s - max_i) / (sum_i exp(s - max_i) + 1e-16)       (PyG softmax, layers/conv.py:139)
 *   out_i  = sum_j alpha_ij,h (v_j,h + e_ij,h)  [+ x_r_i]          (layers/conv.py:142 + scatter-sum;
 *                                                                   "+ x_r" of layers/block.py:531,630)
 * Graph given as CSR by destination: rowptr [n_dst+1], col [E] (source index), edge_attr [E, ea_ld] f32 in
 * CSR order.  Edges of one destination are accumulated in CSR order (= original edge order when the CSR is a
 * stable sort), destinations without edges give out_i = 0 (+ x_r_i).  One pass, no atomics, no [E, C]
 * temporaries.  q/k/v/x_r/out are [*, C] slices with leading dimensions; H heads of D = C/H channels.
 */
int anemoi_gt_edge_attention(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                             const void* x_r, int64_t ldr, const float* edge_attr, int ea_ld, int edge_dim,
                             const float* w_edge, const float* b_edge, const int32_t* rowptr, const int32_t* col,
                             void* out, int64_t ldo, int64_t n_dst, int C, int H, anemoi_stream_t stream);

/*
 * The same edge phase with lin_edge folded into the neighbouring GEMMs (the fast path used by the block mirrors).
 * lin_edge is linear, so with W_e' = [W_e | b_e], a'_ij = [a_ij | 1] and per head h:
 *   q_i,h . e_ij,h       = u_i,h . a'_ij      with u_i,h = W_h'^T q_i,h  -> H*up extra OUTPUT columns of the q/k/v GEMM
 *   sum_j alpha e_ij,h   = W_h' t_i,h         with t_i,h = sum_j alpha a'_ij -> H*up extra INPUT columns of `projection`
 * The kernel therefore never touches W_e:  u [n_dst, H, up] (ldu) is read next to q, edge_attr is [E, up] f32 in CSR
 * order with a constant 1 in column edge_dim, and out (ldo >= C + H*up) receives
 *   out[:, 0:C] = sum_j alpha_ij v_j (+ x_r)      out[:, C:C+H*up] = t_i,h      (alpha as defined above, incl. 1e-16).
 * up is 4, 8, 12 or 16; D = C/H must be a multiple of the 16-byte vector width with D/vec a power of two <= 16.
 * lse (optional, f32 [n_dst, H]): the softmax normaliser max + log(sum exp + 1e-16) per destination and head, i.e.
 * alpha_ij = exp(s_ij - lse_i); what the backward needs to rebuild alpha in one sweep.  NULL: not written.
 */
int anemoi_gt_edge_attention_folded(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                    const void* x_r, int64_t ldr, const void* u, int64_t ldu, const float* edge_attr,
                                    int up, const int32_t* rowptr, const int32_t* col, void* out, int64_t ldo, float* lse,
                                    int64_t n_dst, int C, int H, anemoi_stream_t stream);

/*
 * anemoi_gt_edge_attention_folded on a graph whose destinations all have exactly three in-edges, stored as CSR with
 * rowptr[d] = 3 d (the mesh -> grid decoder of the reference, layers/mapper.py:348-418, on anemoi-graphs' 3-nearest-neighbour
 * edges): consecutive destinations fed by the SAME three sources form a run and share one gather of the three k / v rows.
 *   run_ptr  int32 [n_runs + 1]  first destination of every run (ascending, run_ptr[n_runs] = n_dst)
 *   run_perm int32 [n_runs]      6 bits per destination d = 0, 1 of the run (runs hold at most TWO); bits 6 d + 2 s .. + 1 =
 *                                position (0 .. 2) inside that destination's CSR segment of its edge to the s-th source in
 *                                ascending source order; all destinations of a run have the same source set
 * Same result as the plain entry point up to the f32 rounding of another summation order (deterministic).  bf16, head sizes
 * 64 / 32; other shapes, or run_ptr == NULL, run the plain kernel.  anemoi_models_amd/runtime.py::EdgePlan.runs3 builds the lists.
 */
int anemoi_gt_edge_attention_folded_runs(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                         const void* x_r, int64_t ldr, const void* u, int64_t ldu, const float* edge_attr,
                                         int up, const int32_t* rowptr, const int32_t* col, const int32_t* run_ptr,
                                         const int32_t* run_perm, int64_t n_runs, void* out, int64_t ldo, float* lse,
                                         int64_t n_dst, int C, int H, anemoi_stream_t stream);

/*
 * The same launch on GROUPS: every destination fed by the same three sources -- the grid points of one mesh triangle, wherever
 * they lie in the grid's own order (5.4 per triangle at N320 -> ico-6, against runs of 1.5 consecutive ones) -- is walked
 * by one wave behind one gather of the three k / v row slices:
 *   grp_ptr  int32 [n_groups + 1]  group g = entries grp_ptr[g] .. grp_ptr[g + 1] - 1 (1 .. 8 of them) of
 *   grp_dst  int32 [n_dst]         the destinations, each exactly once (sorted by source triple)
 *   grp_perm int32 [n_dst]         per entry, bits 2 s .. 2 s + 1 = CSR position of its edge to the s-th source, ascending
 * n_src = rows of k / v.  Results equal anemoi_gt_edge_attention_folded_runs bit for bit (same per-destination arithmetic).
 * bf16, head size 64 / 32, every matrix below 4 GiB; anything else, or grp_ptr == NULL, runs the plain kernel.
 * anemoi_models_amd/runtime.py::EdgePlan.runs3 builds the lists (reference layers/mapper.py:348-418 on 3-NN decoder edges).
 */
int anemoi_gt_edge_attention_folded_groups(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                           const void* x_r, int64_t ldr, const void* u, int64_t ldu, const float* edge_attr,
                                           int up, const int32_t* rowptr, const int32_t* col, const int32_t* grp_ptr,
                                           const int32_t* grp_dst, const int32_t* grp_perm, int64_t n_groups, int64_t n_src,
                                           void* out, int64_t ldo, float* lse, int64_t n_dst, int C, int H,
                                           anemoi_stream_t stream);

/*
 * anemoi_gt_edge_attention_folded with a DESTINATION SCHEDULE (round 5): the same result bit for bit -- same arithmetic,
 * same per-destination summation order -- from a launch in which
 *   - a host-built static schedule names the destinations every wave slot walks.  At any step the slots of an XCD work on
 *     one contiguous group of destinations (same L2 window as the plain kernel), but inside the group the destinations
 *     with many in-edges go to the slots with the least work so far: on a multi-scale icosahedral mesh (in-degree 6 ... 36)
 *     the plain round-robin leaves the busiest wave with 1.6 x the mean work, and the launch lasts as long as that wave;
 *   - schedule entry -> row pointers -> source ids are resolved by scalar loads one destination AHEAD each, and all of a
 *     destination's first loads (2 U row gathers, attribute rows, q / u / x_r) are in flight under one wait.
 * sched: int32 [8][slots][steps], XCD x's lists hold every destination of [n_dst x / 8, n_dst (x + 1) / 8) exactly once,
 * each list ends with at least three -1; slots / steps as anemoi_edge_schedule_shape returns them for (dtype, n_dst, C).
 * bf16 with 32- or 64-channel heads and every operand matrix -- the attribute matrix [n_edges, up] f32 included, n_edges =
 * rowptr[n_dst] stated by the caller -- below 4 GiB; other cases (or sched == NULL, or n_edges <= 0) run the plain kernel.
 * anemoi_models_amd/runtime.py::EdgePlan.schedule builds the lists.
 */
int anemoi_edge_schedule_shape(int dtype, int64_t n_dst, int C, int* slots, int* steps);
int anemoi_gt_edge_attention_folded_sched(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                          const void* x_r, int64_t ldr, const void* u, int64_t ldu, const float* edge_attr,
                                          int up, const int32_t* rowptr, const int32_t* col, const int32_t* sched, int slots,
                                          int steps, int64_t n_src, int64_t n_edges, void* out, int64_t ldo, float* lse,
                                          int64_t n_dst, int C, int H, anemoi_stream_t stream);

/*
 * anemoi_gt_edge_attention_folded on LDS TILES (round 6): the same result bit for bit -- same arithmetic, same
 * per-destination summation order -- from a launch that stages every source row ONCE per tile instead of gathering it once
 * per edge.  A tile is a run of <= 32 consecutive destinations (an internal order that keeps graph neighbours together is
 * what makes it pay: the mesh's Morton order) whose in-edges name <= src_cap distinct sources and <= edge_cap edges; a
 * workgroup takes one tile x one 128-channel slice, copies the k | v slices of the tile's sources, the tile's attribute rows
 * and the per-edge LDS slot bytes into LDS, and a wave then walks four destinations at a time (one per 16-lane row) reading
 * its sources from LDS.  Lists (host-built, anemoi_models_amd/runtime.py::EdgeTiles; all on the device):
 *   tile_hdr  int32 [n_tiles][8]      first CSR slot e0, edge count, offset into tile_src, source count, offset into
 *                                     tile_slot (a multiple of 16), destination count, 0, 0
 *   tile_dst  int32 [n_tiles][32][2]  per (pass of four, row): destination id (-1: none), (first edge - e0) << 8 | in-degree
 *   tile_src  int32 [...]             the tiles' distinct sources
 *   tile_slot uint8 [...]             per tile, per edge in CSR order: index of its source in the tile's list
 *   tile_xcd  int32 [9]               tile index range of each of the 8 XCDs (destinations [n_dst x / 8, n_dst (x + 1) / 8))
 * bf16, 32- or 64-channel heads, C a multiple of 128, up in {4, 8, 12, 16}, every operand matrix below 2 GiB; other cases
 * (or tile_hdr == NULL) run the plain kernel.  Reference: layers/conv.py:98-142 (+ PyG propagate / softmax / scatter).
 */
int anemoi_gt_edge_attention_folded_tiles(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                          const void* x_r, int64_t ldr, const void* u, int64_t ldu, const float* edge_attr,
                                          int up, const int32_t* rowptr, const int32_t* col, const int32_t* tile_hdr,
                                          const int32_t* tile_dst, const int32_t* tile_src, const uint8_t* tile_slot,
                                          const int32_t* tile_xcd, int max_tiles_per_xcd, int src_cap, int edge_cap,
                                          int64_t n_src, int64_t n_edges, void* out, int64_t ldo, float* lse, int64_t n_dst,
                                          int C, int H, anemoi_stream_t stream);

/*
 * GraphTransformerConv with explicit per-edge features (the callable the reference exposes, layers/conv.py:98-142):
 *   s_ij = q_i . (k_j + e_ij) / sqrt(D),  alpha = softmax over the in-edges of i (+1e-16),  out_i = sum_j alpha (v_j + e_ij)
 * q [n_dst, C], k / v [n_src, C], edges [E, C] in the CSR order of (rowptr, col) (= lin_edge(edge_attr)[perm]), all in
 * the activation dtype; x_r (optional) is added to the result, lse (optional f32 [n_dst, H]) receives the softmax
 * normaliser for the backward.  The block mirrors fold lin_edge away (see above) and come here only for edge_dim values
 * the folded / fused kernels do not cover (any edge_dim works on this route).
 * Backward (same structure as the folded backward below, with k + e, v + e in place of k, v):
 *   _dst: alpha, w [E, H], dsum [n_dst, H] (f32), dq;   _src: dk, dv and d edges [E, C] (CSR order) = alpha dout_i + scale ds q_i.
 * dropout_p / dropout_seed / dropout_seed_dev (ABI v41): the conv's `dropout` argument in training mode (reference
 * layers/conv.py:89,140 -- `dropout(alpha, p, training)` on alpha [E, H]): out_i = sum_j alpha keep / (1 - p) (v_j + e_ij) with
 * one counter-based keep decision per (CSR edge position, head) and seed (no mask tensor; the two backward entry points
 * take the same three arguments and rebuild it).  dropout_seed_dev: optional device word whose low 32 bits the kernels add
 * to the seed when they run (what a captured training step advances; NULL otherwise).  dropout_p = 0: the plain kernels.
 */
int anemoi_gt_conv(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* edges,
                   int64_t lde, const void* x_r, int64_t ldr, const int32_t* rowptr, const int32_t* col, void* out,
                   int64_t ldo, float* lse, int64_t n_dst, int C, int H, float dropout_p, uint32_t dropout_seed,
                   const void* dropout_seed_dev, anemoi_stream_t stream);
int anemoi_gt_conv_backward_dst(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                const void* edges, int64_t lde, const void* dout, int64_t ldd, const float* lse,
                                const int32_t* rowptr, const int32_t* col, float* alpha, float* w, float* dsum, void* dq,
                                int64_t lddq, int64_t n_dst, int C, int H, float dropout_p, uint32_t dropout_seed,
                                const void* dropout_seed_dev, anemoi_stream_t stream);
int anemoi_gt_conv_backward_src(int dtype, const void* q, int64_t ldq, const void* dout, int64_t ldd, const float* alpha,
                                const float* w, const float* dsum, const int32_t* rowptr_t, const int32_t* eid_t,
                                const int32_t* dst_t, void* dk, void* dv, int64_t ldg, void* dedges, int64_t ldde,
                                int64_t n_src, int C, int H, float dropout_p, uint32_t dropout_seed,
                                const void* dropout_seed_dev, anemoi_stream_t stream);

/*
 * Input assembly (I/O glue K9): rows (b, ens, g) of
 *   out = [ x[b, 0..T-1, ens, g, 0..V-1] (time-major) | latlons[g, 0:n_ll] | trainable[g, 0:n_tr] | 0-pad ]
 * x is f32 [B, T, Ens, G, V] contiguous; out is `dtype` with leading dimension ldo >= T*V + n_ll + n_tr.
 * Replaces einops.rearrange + NamedNodesAttributes + torch.cat, models/encoder_processor_decoder.py:173-181.
 * With x == NULL (T = 0) it produces the hidden-node attribute matrix of line :181.
 * in_mul / in_add (both NULL or both [V] f32): x is the RAW state and `x * in_mul[v] + in_add[v]` -- the
 * InputNormalizer's transform (preprocessing/normalizer.py:134-164) -- is applied while it is read.
 */
int anemoi_assemble_nodes(int dtype, const float* x, int B, int T, int Ens, int64_t G, int V, const float* latlons,
                          int n_ll, const float* trainable, int n_tr, void* out, int64_t ldo, const float* in_mul,
                          const float* in_add, anemoi_stream_t stream);

/*
 * The same for a LIST of grid nodes (batch 1, ensemble 1): out row i is the row of node rows[i] (int64 [n_rows], any order,
 * repeats allowed).  A rank of a node-partitioned run (models/encoder_processor_decoder.py:168-181 under a model group)
 * assembles only the grid rows its encoder reads and its decoder writes instead of the whole grid.
 */
int anemoi_assemble_node_rows(int dtype, const float* x, int T, int64_t G, int V, const float* latlons, int n_ll,
                              const float* trainable, int n_tr, const int64_t* rows, int64_t n_rows, void* out,
                              int64_t ldo, const float* in_mul, const float* in_add, anemoi_stream_t stream);

/*
 * One pass over the f32 output [B, Ens, G, V_out] that ends AnemoiModelInterface.predict_step on a raw input state:
 *   y[.., c] += normalised x[b, T-1, ens, g, src[c]]      where src[c] >= 0 (prognostic residual,
 *                                                          models/encoder_processor_decoder.py:227; src int32 [V_out])
 *   y[.., c]  = (y[.., c] - out_add[c]) / out_mul[c]       (InputNormalizer.inverse_transform,
 *                                                          preprocessing/normalizer.py:166-205; NULL/NULL = skipped)
 * with normalised x = x * in_mul[v] + in_add[v] (NULL/NULL: x is already normalised).
 */
int anemoi_finalize_output(float* y, int V_out, const float* x, int B, int T, int Ens, int64_t G, int V_in,
                           const int32_t* src, const float* in_mul, const float* in_add, const float* out_mul,
                           const float* out_add, anemoi_stream_t stream);

/*
 * The same on the rows a rank decodes (batch 1, ensemble 1): y is [n_rows, V_out], row i belongs to grid node rows[i]
 * (int64 [n_rows]) -- the residual and the de-normalisation are row-local, so a node-partitioned run finishes its own rows
 * BEFORE the final all-gather (models/encoder_processor_decoder.py:223-233, layers/mapper.py:99-102).
 */
int anemoi_finalize_output_rows(float* y, int V_out, const float* x, int T, int64_t G, int V_in, const int32_t* src,
                                const int64_t* rows, int64_t n_rows, const float* in_mul, const float* in_add,
                                const float* out_mul, const float* out_add, anemoi_stream_t stream);

/*
 * Output boundings, in place on the f32 output rows [rows, V_out] (rows = B * Ens * G) after the prognostic residual:
 * replaces the chained ReluBounding / HardtanhBounding / FractionBounding modules of
 * models/encoder_processor_decoder.py:229-231 (layers/bounding.py:60-124).  Every row applies, IN ORDER i = 0..n_ops-1,
 *   y[col[i]] = clamp(y[col[i]], lo[i], hi[i]) * (mul[i] >= 0 ? y[mul[i]] : 1)
 * (ReLU: lo = 0, hi = +inf; Hardtanh: lo = min_val, hi = max_val; the fraction step: lo = -inf, hi = +inf, mul = column
 * of total_var; NaN stays NaN), then de-normalises the n_fin columns fin_col[j]: y = (y - fin_add[j]) / fin_mul[j]
 * (InputNormalizer.inverse_transform of the columns the boundings had to see normalised; n_fin = 0: none).
 * All lists are device arrays (int32 / f32).
 */
int anemoi_bound_output(float* y, int V_out, int64_t rows, int n_ops, const int32_t* op_col, const float* op_lo,
                        const float* op_hi, const int32_t* op_mul, int n_fin, const int32_t* fin_col,
                        const float* fin_mul, const float* fin_add, anemoi_stream_t stream);

/*
 * Prognostic residual (models/encoder_processor_decoder.py:227), in place on the f32 output:
 *   y[b, ens, g, out_idx[p]] += x[b, T-1, ens, g, in_idx[p]]   for p < n_prog.
 */
int anemoi_prognostic_residual(float* y, int V_out, const float* x, int B, int T, int Ens, int64_t G, int V_in,
                               const int32_t* out_idx, const int32_t* in_idx, int n_prog, anemoi_stream_t stream);

/*
 * Autoregressive rollout: next model input from the current one and the prediction, in place on x (f32
 * [B, T, Ens, G, V_in]):  x[:, t] <- x[:, t+1] for t < T-1, then for the last time slice
 *   colmap[v] >= 0  : x[b, T-1, ens, g, v] = y[b, ens, g, colmap[v]]            (prognostic variables)
 *   colmap[v] <= -2 : x[b, T-1, ens, g, v] = forcing[b, ens, g, -2 - colmap[v]]  (forcings of the new time; skipped
 *                     when forcing == NULL: the previous value persists)
 *   colmap[v] == -1 : the previous value persists.
 * The reference repository stops at one step (interface/__init__.py:97-123); this is the caller's loop
 * (anemoi-training `advance_input`: roll the time axis, write the prognostic outputs, write the new forcings), i.e.
 * BASELINE config 4.  y is f32 [B, Ens, G, V_out], forcing f32 [B, Ens, G, F] or NULL.
 */
int anemoi_advance_input(float* x, int B, int T, int Ens, int64_t G, int V_in, const float* y, int V_out,
                         const float* forcing, int F, const int32_t* colmap, anemoi_stream_t stream);

/* dtype conversion / K-padding copy: dst[r, 0:cols] = src[r, 0:cols], dst[r, cols:ld_dst] = 0. */
int anemoi_convert_pad(int src_dtype, const void* src, int64_t ld_src, int dst_dtype, void* dst, int64_t ld_dst,
                       int64_t rows, int cols, anemoi_stream_t stream);

/* y = a + b elementwise over [rows, cols] slices (processor skip, models/encoder_processor_decoder.py:204). */
int anemoi_add(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows,
               int cols, anemoi_stream_t stream);

/*
 * GNN edge phase, part 1 (K5): out[e, :] = act(t[e, :] + p_dst[dst[e], :] + p_src[src[e], :]).
 * With t = e W1c^T + b1, p_dst = x W1a^T, p_src = x W1b^T this is the first Linear + activation of the edge MLP applied
 * to cat[x_i, x_j, e] (layers/conv.py:68-69, layers/mlp.py:74) without materialising the [E, 3C] concatenation.
 * dst / src: int32 [E] node index per edge (CSR order: dst ascending).
 */
int anemoi_gather_add_act(int dtype, const void* t, int64_t ldt, const void* p_dst, int64_t ldpd, const void* p_src,
                          int64_t ldps, const int32_t* dst, const int32_t* src, void* out, int64_t ldo,
                          int64_t n_edges, int C, int act, anemoi_stream_t stream);

/*
 * GNN edge phase, part 2: out[i, :] = sum of v[e, :] over the CSR row i (edges sorted by destination), f32 accumulation
 * in CSR order.  Replaces torch_geometric scatter(reduce="sum") at layers/conv.py:73-76.
 */
int anemoi_segment_sum(int dtype, const void* v, int64_t ldv, const int32_t* rowptr, void* out, int64_t ldo,
                       int64_t n_dst, int C, anemoi_stream_t stream);

/*
 * The same pass producing the node MLP's input of a GNN block, out[i, :] = [ x[i, 0:C] | sum of v[e, :] over CSR row i ]
 * (out is [n_dst, ldo >= 2C]): torch.cat([x, aggregated], dim=1) at layers/block.py:217, 276 without a separate copy of x.
 */
int anemoi_segment_sum_cat(int dtype, const void* v, int64_t ldv, const int32_t* rowptr, const void* x, int64_t ldx,
                           void* out, int64_t ldo, int64_t n_dst, int C, anemoi_stream_t stream);

/*
 * Mesh-node multi-head self attention (K7): out[b*S + i, h*D:(h+1)*D] = softmax_j(q_i . k_j / sqrt(D)) v_j per head,
 * flash style, on the fused lin_qkv output qkv [B*S, 3C] = q | k | v (leading dimension ld), C = H*D.
 * Replaces the rearranges + flash_attn_func / scaled_dot_product_attention of layers/attention.py:76-108.
 * window < 0: global attention (the reference's SDPA fallback); window >= 0: flash-attn sliding window |i - j| <= window.
 * bf16 with D = 64 or 32 runs on MFMA and needs `workspace` of anemoi_mhsa_workspace_bytes() bytes (V transposed);
 * other cases use a VALU kernel (workspace may be NULL).
 * dropout_p in [0, 1] (attention dropout of the reference in training mode, layers/attention.py:90): probabilities are
 * dropped AFTER normalisation by a counter-based hash of (batch, head, query, key) and `dropout_seed`, kept ones scaled
 * by 1 / (1 - p); the backward rebuilds the same mask from the same seed.  One 32-bit hash decides the key pair
 * (2 j, 2 j + 1) of a query, 16 bits each (p is resolved to 2^-16).  The MFMA kernels apply the mask to their packed
 * probabilities (0 < p < 1, B x heads x S < 2^32); p = 1 takes the VALU kernel.  `dropout_h0` / `dropout_h_total`: the
 * GLOBAL index of head 0 of this call and the head count of the whole attention (0 = H) -- a head-sharded call
 * (sequence-parallel attention, distributed/transformer.py:85-130) then draws exactly the mask of the unsharded one.
 * `dropout_seed_dev` (may be NULL): a 4-byte aligned DEVICE word whose value every kernel of the call adds to
 * `dropout_seed` when it starts -- the part of the seed a captured HIP graph can advance between replays (kernel arguments
 * are frozen at capture; anemoi_models_amd/runtime.py::DeviceDropout).  The backward must be given the same word, holding
 * the value the forward saw.
 */
int64_t anemoi_mhsa_workspace_bytes(int dtype, int B, int S, int H, int D);
int anemoi_mhsa(int dtype, const void* qkv, int64_t ld, void* out, int64_t ldo, void* workspace, float* lse, int B, int S,
                int H, int D, int window, float dropout_p, uint32_t dropout_seed, const void* dropout_seed_dev,
                int dropout_h0, int dropout_h_total, anemoi_stream_t stream);

/*
 * Backward of anemoi_mhsa (what torch autograd derives for the reference's scaled_dot_product_attention call,
 * layers/attention.py:99-105, when anemoi-training calls .backward()).  `lse` f32 [B, H, S] is the forward's optional
 * output (natural-log sum-exp of the scaled scores per query and head; pass NULL to the forward when not training), `out`
 * the forward's result, `dout` its gradient; writes dqkv [B*S, 3C] = dq | dk | dv in `dtype`.  `delta` f32 [B, H, S] is
 * scratch.  Probabilities are recomputed from lse (nothing of size S x S is stored); no atomics.  bf16 with D = 64 / 32
 * and dropout_p < 1: MFMA kernels (one pass with the keys stationary for dK / dV, one with the queries stationary for dQ)
 * on a `workspace` of anemoi_mhsa_backward_workspace_bytes() bytes (Q^T, K^T, dO^T and the padded lse / delta rows the
 * dK / dV kernel streams); otherwise VALU kernels, O(S^2 D) on the vector pipe (workspace may be NULL).
 */
int64_t anemoi_mhsa_backward_workspace_bytes(int dtype, int B, int S, int H, int D);
int anemoi_mhsa_backward(int dtype, const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                         int64_t lddo, const float* lse, float* delta, void* dqkv, int64_t lddq, void* workspace, int B,
                         int S, int H, int D, int window, float dropout_p, uint32_t dropout_seed,
                         const void* dropout_seed_dev, int dropout_h0, int dropout_h_total, anemoi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward pass, dense half (first step): the pieces the autograd of the fused Linear and of
 * LayerNorm needs besides the forward GEMM entry points (anemoi_models_amd/autograd.py):
 *   dX = dpre W (anemoi_linear on dpre and W^T), dW = dpre^T X (anemoi_linear on the two transposes, f32 result),
 *   db = column sums of dpre, dpre = dy * act'(pre).  What torch.autograd derives from nn.Linear / nn.GELU / nn.SiLU /
 *   nn.LayerNorm in layers/block.py:504-508,631-633 and layers/mlp.py:74-84 when anemoi-training calls .backward().
 * No atomics: gradients are reproducible bit for bit.
 * ------------------------------------------------------------------------------------------------------------------ */

/* dst[c, r] = src[r, c]; dst has leading dimension ld_dst >= rows, its columns rows..ld_dst-1 are zero filled. */
int anemoi_transpose(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                     anemoi_stream_t stream);

/* Chunked form: rows [s * chunk_rows, (s + 1) * chunk_rows) of src become slab s of dst = [chunks, cols, ld_dst]
 * (ld_dst >= chunk_rows, zero filled behind the chunk's rows).  colsum_partial (optional, bf16 sources, cols % 4 == 0):
 * f32 [anemoi_transpose_colsum_rows(rows, chunk_rows), cols] -- every 64-row tile leaves the column sums of its source
 * rows there; their sum over the rows (anemoi_col_sum) is the column sum of src (the bias gradient, without a second pass
 * over dpre). */
int64_t anemoi_transpose_colsum_rows(int64_t rows, int64_t chunk_rows);
int anemoi_transpose_chunked(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                             int64_t chunk_rows, float* colsum_partial, anemoi_stream_t stream);

/* Weight gradient without transposed copies (bf16): partial[c] [N, K] f32 (contiguous, c = 0 .. ceil(M / chunk_rows) - 1)
 * <- sum over the rows m of chunk c of dy[m, :N]^T x[m, :K]; bias_partial (optional) [chunks, N] f32 <- the column sums of
 * dy over the chunk (the bias gradient, out of the same pass).  partial_stride / bias_stride: floats between the chunks of
 * either (>= N * K / >= N) -- both may live in one [chunks, N * K + N] buffer that one anemoi_col_sum reduces.  dy [M, ldy], x [M, ldx] row-major, 16-byte aligned, pitches
 * and N, K multiples of 8; chunk_rows a multiple of 64, >= 128, chunk_rows * pitch * 2 < 2 GiB.  The caller adds the
 * chunks (anemoi_col_sum).  Replaces autograd's grad_output.t() @ input of every nn.Linear under
 * models/encoder_processor_decoder.py:167-233 (training). */
int anemoi_weight_grad_tn(const void* dy, int64_t ldy, const void* x, int64_t ldx, void* partial, int64_t partial_stride,
                          void* bias_partial, int64_t bias_stride, int64_t M, int N, int K, int chunk_rows,
                          anemoi_stream_t stream);

/* `batch` independent products y[b] = x[b] w[b]^T (strides in elements; no bias / activation): the weight-gradient GEMMs
 * split their long reduction over the rows into `batch` chunks this way and add the partial [N, K] results with
 * anemoi_col_sum -- deterministic, and a small result still fills the chip. */
int anemoi_linear_batched(int dtype, int out_dtype, const void* x, int64_t ldx, int64_t stride_x, const void* w,
                          int64_t stride_w, void* y, int64_t ldy, int64_t stride_y, int batch, int64_t M, int N, int K,
                          anemoi_stream_t stream);

/* out[c] = sum_r x[r, c] (f32 result, two deterministic stages; workspace: anemoi_col_sum_workspace_floats floats). */
int64_t anemoi_col_sum_workspace_floats(int64_t rows, int cols);
int anemoi_col_sum(int dtype, const void* x, int64_t ldx, int64_t rows, int cols, float* out, float* workspace,
                   int64_t workspace_floats, anemoi_stream_t stream);

/* out[r] = sum_c a[r, c] * (b[r, c] - shift[c]) in f32 (shift optional: f32 [cols]).  The training route's folded
 * "embedding -> LayerNorm -> Linear" product y = rstd * (x F^T) + b' needs d rstd[r] = sum_c dy[r, c] (y[r, c] - b'[c]) / rstd[r]
 * (anemoi_models_amd/autograd.py::_ScaledLinear; reference layers/mapper.py:322-331 + layers/block.py:516-528 under autograd). */
int anemoi_row_dot(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, const float* shift, float* out,
                   int64_t rows, int cols, anemoi_stream_t stream);

/* out[r, c] = alpha * s[r] * x[r, c] (s: f32 per row; out in x's dtype, may alias x): the row scalings of the same folded
 * product's backward (dx = rstd (dy F), dF = dy^T (rstd x)) and of its variance term x^T (A^T A / C) x. */
int anemoi_row_scale(int dtype, const void* x, int64_t ldx, const float* s, float alpha, void* out, int64_t ldo,
                     int64_t rows, int cols, anemoi_stream_t stream);

/* out = act(pre) (+ residual): the differentiable forward keeps `pre` for act' and applies the activation in one pass. */
int anemoi_act_forward(int dtype, int act, const void* pre, int64_t ldp, const void* residual, int64_t ldr, void* out,
                       int64_t ldo, int64_t rows, int cols, anemoi_stream_t stream);

/* out = dy * act'(pre), pre = the Linear's result before its activation (ANEMOI_ACT_*; exact erf GELU derivative). */
int anemoi_act_backward(int dtype, int act, const void* pre, int64_t ldp, const void* dy, int64_t ldd, void* out,
                        int64_t ldo, int64_t rows, int cols, anemoi_stream_t stream);

/*
 * Training forward of Linear + activation: y = act(x W^T + b) and pre = x W^T + b (rounded to the activation dtype) from
 * ONE launch (the backward multiplies by act'(pre); nn.Linear + nn.GELU of layers/mlp.py:74-84, layers/block.py:504-508
 * under autograd).  bf16 only, M = a multiple of 256 plus at most 8 rows (the mesh sizes 10 * 4^k + 2: the last rows are
 * computed by the same launch), N >= 256, K >= 128 (multiple of 64), 16-byte aligned operands; ANEMOI_ERR_UNSUPPORTED
 * otherwise -- the caller then runs anemoi_linear followed by anemoi_act_forward.
 */
int anemoi_linear_dual(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* pre, int64_t ldp,
                       void* y, int64_t ldy, int64_t M, int N, int K, int act, anemoi_stream_t stream);

/*
 * Backward counterpart: y = (x W^T) * act'(pre) in one launch -- with x = dy and W = W2^T this is the dX GEMM of the
 * Linear BEHIND an activation delivering d pre of the Linear in front of it (what autograd derives for
 * Linear -> GELU -> Linear, layers/mlp.py:74-84), without a separate act' pass.  GELU': degree-8 polynomial in x^2 of the
 * exact derivative, |error| < 5.5e-4.  Same shape rules / fallback as anemoi_linear_dual.
 */
int anemoi_linear_actgrad(int dtype, const void* x, int64_t ldx, const void* w, const void* pre, int64_t ldp, void* y,
                          int64_t ldy, int64_t M, int N, int K, int act, anemoi_stream_t stream);

/*
 * LayerNorm backward from the forward's row statistics (stats [rows, 2] = { rstd, -mean * rstd }, anemoi_row_stats):
 *   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * gamma, xhat = x * rstd - mean * rstd;
 *   dgamma[c] = sum_r dy * xhat, dbeta[c] = sum_r dy (f32).  workspace: anemoi_layer_norm_backward_workspace_floats.
 *   dres (optional, [rows, ldr] in x's dtype): added to dx before the store -- the gradient that reaches x through the skip
 *   connection around the LayerNorm (layers/block.py:504-508, 614-635: x + mlp(norm(x))), instead of a separate add pass.
 */
int64_t anemoi_layer_norm_backward_workspace_floats(int64_t rows, int C);
int anemoi_layer_norm_backward(int dtype, const void* x, int64_t ldx, const float* stats, const float* gamma,
                               const void* dy, int64_t ldd, const void* dres, int64_t ldr, void* dx, int64_t ldo,
                               int64_t rows, int C, float* dgamma, float* dbeta, float* workspace,
                               int64_t workspace_floats, anemoi_stream_t stream);

/*
 * Backward of anemoi_gt_edge_attention_folded (the edge half of SURVEY.md section 8f-1; what torch.autograd derives from
 * GraphTransformerConv.message / softmax / aggregate, layers/conv.py:98-142, plus lin_edge through the fold).
 * With s_e = scale (q_i.k_j + u_i.a_e), alpha_e = exp(s_e - lse_i) (lse: the forward's optional output), out_i = sum alpha
 * v_j (+ x_r), t_i = sum alpha a_e, and for the incoming gradients dout [n_dst, C] and dt [n_dst, H*up]:
 *   w_e = alpha_e (dout_i,h.v_j,h + dt_i,h.a_e),   dsum[i, h] = sum_e w_e,   ds_e = w_e - alpha_e dsum[i, h]
 *   _dst  (forward CSR, ONE sweep over the in-edges):  alpha[E, H], w[E, H], dsum[n_dst, H] (f32);
 *                            dq_i = scale sum_e ds k_j,  du_i = scale sum_e ds a_e   (activation dtype, any leading dim)
 *   _src  (transposed CSR):  dk_j = scale sum_{e from j} ds q_i,  dv_j = sum_{e from j} alpha dout_i
 *   anemoi_gt_edge_attr_grad: dattr[e, a] = sum_h (scale ds[e, h] u[dst(e), h, a] + alpha[e, h] dt[dst(e), h, a]), the
 *                            gradient of the edge attributes [E, up] (CSR order; trainable edge tensor columns included)
 * (rowptr_t [n_src+1], eid_t [E] = position of the edge in the forward CSR, dst_t [E] = its destination; u / dt / du are
 * [n_dst, H*up] column ranges in the activation dtype, dst_of_edge [E] the destination of every CSR slot).  dsum is
 * accumulated in f32 over the edges, never rebuilt from the forward's rounded output.  No atomics.
 * dxr (optional, [n_dst, lddxr]): _dst also leaves a copy of dout there -- the gradient of the self term x_r of
 * out = sum alpha v + x_r -- so that a caller who keeps x_r | q | k | v | u in one matrix fills d x_r without a pass of its own.
 */
int anemoi_gt_edge_attention_folded_backward_dst(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                 int64_t ldkv, const void* dout, int64_t ldd, const void* u, int64_t ldu,
                                                 const void* dt, int64_t lddt, const float* lse, const float* edge_attr,
                                                 int up, const int32_t* rowptr, const int32_t* col, float* alpha,
                                                 float* w, float* dsum, void* dq, int64_t lddq, void* du, int64_t lddu,
                                                 void* dxr, int64_t lddxr, int64_t n_dst, int C, int H,
                                                 anemoi_stream_t stream);
int anemoi_gt_edge_attention_folded_backward_src(int dtype, const void* q, int64_t ldq, const void* dout, int64_t ldd,
                                                 const float* alpha, const float* w, const float* dsum,
                                                 const int32_t* rowptr_t, const int32_t* eid_t, const int32_t* dst_t,
                                                 void* dk, void* dv, int64_t ldg, int64_t n_src, int C, int H,
                                                 anemoi_stream_t stream);
int anemoi_gt_edge_attr_grad(int dtype, const float* alpha, const float* w, const float* dsum, const void* u, int64_t ldu,
                             const void* dt, int64_t lddt, const int32_t* dst_of_edge, float* dattr, int64_t n_edges,
                             int H, int up, int D, anemoi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Block-level entry points ("gt_block_fwd(x, ln / lin weights, edge attributes, csr, ..., out,
 * stream)"): a whole GraphTransformer block's launch sequence from ONE call on the caller's stream -- what
 * GraphTransformerProcessorBlock.forward (reference layers/block.py:602-635) and the back half of
 * GraphTransformerMapperBlock.forward (layers/block.py:516-550) do, on the bf16 LayerNorm-folded route: the host mirror
 * pays one FFI call per block instead of five or six (host enqueue of a forward: 2.6 -> 0.6 ms).  Weights arrive packed
 * as the op-level entry points above take them (LayerNorm folded in by the caller: W' = W * gamma, column sums of W',
 * b' = b + W beta; lin_edge folded into W_in / W_proj); every buffer -- workspaces included -- is the caller's.
 *
 *   anemoi_gt_block_tail              out = mlp(LN(y)) + y,  y = projection(edge_phase(q, k, v, u) + x_r | t) + res
 *                                     (edge phase = anemoi_gt_edge_attention_folded; projection = anemoi_linear_stats
 *                                     with the node MLP's LayerNorm statistics in its epilogue; mlp = LayerNorm-folded
 *                                     Linear + activation (anemoi_linear_ln), Linear + y (anemoi_linear_stats, or
 *                                     anemoi_linear when out_stats is NULL))
 *   anemoi_gt_processor_block_forward the x_r | q | k | v | u product of LayerNorm(x) first (anemoi_linear_ln into
 *                                     `sq`, whose column ranges then ARE q, k, v, x_r, u), then the tail with res = x.
 * `struct_bytes` must be sizeof(anemoi_gt_block_args) (layout check across the FFI).  Status codes as everywhere.
 */
typedef struct anemoi_gt_block_args {
  int64_t struct_bytes;
  int64_t n_dst;           /* destination rows (processor block: all rows)                                       */
  int32_t dtype;           /* ANEMOI_BF16                                                                        */
  int32_t C, H, up;        /* channels, heads, folded edge width per head                                        */
  int32_t hidden, act;     /* node MLP: hidden width, activation code                                            */
  int32_t k_proj, n_in;    /* projection K (= ld of att: C + H * up, slab padded); columns of the input product  */
  float eps_mlp, eps_out;  /* epsilon of the node MLP's LayerNorm; of the LayerNorm that consumes `out` next     */
  /* LayerNorm-folded input product (anemoi_gt_processor_block_forward only) */
  const void* x; int64_t ldx; const float* x_stats;                 /* [n_dst, C], its row statistics            */
  const void* w_in; const float* b_in; const float* cs_in;          /* [n_in, C] = x_r | q | k | v | u rows      */
  void* sq; int64_t ld_sq;                                          /* [n_dst, n_in] workspace                   */
  /* edge phase (the processor entry point overwrites these with column ranges of sq) */
  const void* q; const void* k; const void* v; const void* x_r; const void* u;
  int64_t ldq, ldkv, ldr, ldu;
  const float* edge_attr; const int32_t* rowptr; const int32_t* col; /* [E, up] f32 CSR order, int32 CSR           */
  void* att; int64_t ld_att;                                        /* [n_dst, k_proj] workspace; the caller has */
                                                                    /* ZEROED its columns >= C + H * up (K pad)  */
  /* projection (+ res) -> y and the statistics of the node MLP's LayerNorm */
  const void* w_proj; const float* b_proj; const void* res; int64_t ld_res; void* y; float* y_stats;
  /* node MLP */
  const void* w_fc1; const float* b_fc1; const float* cs_fc1; void* h;  /* [hidden, C] LayerNorm-folded; [n_dst, hidden] */
  const void* w_fc2; const float* b_fc2; void* out; float* out_stats;   /* [C, hidden]; out [n_dst, C]; NULL: no stats   */
  void* stats_ws; int64_t stats_ws_bytes;                           /* >= n_dst * max(C / 128, 1) * 8 bytes      */
  /* optional (NULL / 0: plain edge kernel): the runs of a uniform-degree-3 graph, anemoi_gt_edge_attention_folded_runs */
  const int32_t* run_ptr; const int32_t* run_perm; int64_t n_runs;
  /* optional (NULL: none; ignored when runs are given): the destination schedule of anemoi_gt_edge_attention_folded_sched */
  const int32_t* sched; int32_t sched_slots, sched_steps; int64_t n_src;
  /* optional (NULL: run_ptr / run_perm are the consecutive runs above): run_ptr / run_perm / run_dst are the GROUP lists of
   * anemoi_gt_edge_attention_folded_groups (groups of 1 .. 8 destinations out of run_dst, run_perm per destination; n_src set) */
  const int32_t* run_dst;
  /* rowptr[n_dst], stated by the caller: the scheduled kernel addresses attribute rows by 32-bit byte offsets and is taken
   * only when n_edges * up * 4 < 2^32 (<= 0: not stated, plain kernel) */
  int64_t n_edges;
  /* optional (NULL: none; taken before the schedule, ignored when runs are given): the tile lists of
   * anemoi_gt_edge_attention_folded_tiles (n_src, n_edges set) */
  const int32_t* tile_hdr; const int32_t* tile_dst; const int32_t* tile_src; const uint8_t* tile_slot; const int32_t* tile_xcd;
  int32_t tile_max_per_xcd, tile_src_cap, tile_edge_cap, tile_pad;
} anemoi_gt_block_args;
int anemoi_gt_block_tail(const anemoi_gt_block_args* args, anemoi_stream_t stream);
int anemoi_gt_processor_block_forward(const anemoi_gt_block_args* args, anemoi_stream_t stream);

/*
 * anemoi_transformer_block_forward: TransformerProcessorBlock.forward (reference layers/block.py:99-105 over
 * layers/attention.py:67-112) from one call, f32 or bf16 --
 *     y   = x + projection(attention(lin_qkv(layer_norm1(x))))
 *     out = y + mlp(layer_norm2(y)),    mlp = Linear -> activation -> Linear
 * as the launch sequence anemoi_layer_norm, anemoi_linear, anemoi_mhsa, anemoi_linear (+ x), anemoi_layer_norm,
 * anemoi_linear (+ activation), anemoi_linear (+ y) on the caller's stream.  Weights [out, K] in `dtype`, K zero padded to
 * the slab multiple as for anemoi_linear (C itself must be a multiple); LayerNorm parameters and biases f32 (biases may be
 * NULL); `qkv` / `att` / `y` / `h_ln` / `h` are the caller's workspaces, `mhsa_ws` of anemoi_mhsa_workspace_bytes() bytes;
 * window / dropout arguments as anemoi_mhsa.  `out` may not alias `x`.
 */
typedef struct anemoi_tfm_block_args {
  int64_t struct_bytes;
  int64_t rows;              /* B * S node rows                                                                    */
  int32_t dtype;             /* ANEMOI_F32 or ANEMOI_BF16                                                          */
  int32_t B, S, C, H;        /* batch, sequence length (rows = B * S), channels, heads                             */
  int32_t hidden, act;       /* MLP hidden width, activation code                                                  */
  int32_t window;            /* < 0: global attention                                                              */
  float eps1, eps2;          /* epsilon of layer_norm1 / layer_norm2                                               */
  float dropout_p; uint32_t dropout_seed; const void* dropout_seed_dev; int32_t dropout_h0, dropout_h_total;
  const void* x; int64_t ldx;                                       /* [rows, C]                                   */
  const float* ln1_w; const float* ln1_b; const float* ln2_w; const float* ln2_b;
  const void* w_qkv; const float* b_qkv;                            /* [3 C, C]                                    */
  const void* w_proj; const float* b_proj;                          /* [C, C]                                      */
  const void* w_fc1; const float* b_fc1;                            /* [hidden, C]                                 */
  const void* w_fc2; const float* b_fc2;                            /* [C, hidden]                                 */
  void* h_ln;                                                       /* [rows, C]      layer_norm1 / layer_norm2 out */
  void* qkv;                                                        /* [rows, 3 C]                                 */
  void* att;                                                        /* [rows, C]                                   */
  void* y;                                                          /* [rows, C]                                   */
  void* h;                                                          /* [rows, hidden]                              */
  void* mhsa_ws;                                                    /* anemoi_mhsa_workspace_bytes (may be NULL)   */
  void* out;                                                        /* [rows, C]                                   */
} anemoi_tfm_block_args;
int anemoi_transformer_block_forward(const anemoi_tfm_block_args* args, anemoi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ANEMOI_AMD_H */
