/*
 * gnnome_hip.h - C ABI of libgnnome_hip.so, the MI355X (gfx950) implementation of GNNome's
 * SymGatedGCN edge-scoring path.
 *
 * The reference (lbcb-sci/GNNome) has no FFI seam for this path: the seam is the Python module
 * `models.SymGatedGCNModel.forward(graph, x, e)` (models/full_graph.py:22-30), whose native work
 * is done by two third-party wheels (DGL 0.8.1 graph kernels, PyTorch 1.9 dense/elementwise ops).
 * Each entry point below names the reference call sites it replaces.  The Python host module
 * `gnnome_amd.models.SymGatedGCNModel` binds them through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is stream-ordered,
 *     nothing synchronises the host
 *   - matrices are row-major fp32; `ld*` arguments are row strides in ELEMENTS
 *   - graph indices are int32 (the reference casts graphs with g.int(), utils/data_utils.py:32)
 *   - return value: 0 on success, a negative GNNOME_E* code otherwise; gnnome_last_error() returns a
 *     thread-local message for the last failure
 *   - "sorted position" p in [0,E) is an edge's slot in destination-sorted (CSR-by-dst) order; all
 *     [E,H] edge tensors handed to the kernels live in that order
 */
#ifndef GNNOME_HIP_H
#define GNNOME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNOME_ABI_VERSION 17

#define GNNOME_OK 0
#define GNNOME_EINVAL (-1)    /* bad argument (null pointer, unsupported width, bad stride)      */
#define GNNOME_EHIP (-2)      /* a HIP runtime call or a kernel launch failed                     */
#define GNNOME_EWORKSPACE (-3) /* workspace too small                                             */

/* normalisation applied inside the fused kernels (gated_gcn_full.py:37-42) */
#define GNNOME_NORM_AFFINE 0  /* y = x*scale[c] + shift[c]: eval-mode BatchNorm1d folded to an affine,
                                 or train-mode BatchNorm1d once the batch statistics are known     */
#define GNNOME_NORM_LAYER 1   /* y = (x-mean_row)/sqrt(var_row+1e-5)*scale[c] + shift[c]: LayerNorm */
/* LayerNorm whose row statistics run over the first w channels only: a model narrower than the built width (64 / 128 / 256) runs
 * zero-padded - its padded channels hold exact zeros and zero scale / shift - and normalises over its OWN width
 * (configs/hyperparameters.py:22 takes any hidden_features).  Accepted wherever norm_kind is; w = hidden is GNNOME_NORM_LAYER. */
#define GNNOME_NORM_LAYER_OVER(w) (GNNOME_NORM_LAYER | ((w) << 8))

int gnnome_abi_version(void);
const char* gnnome_last_error(void);

/* Measurement knobs, not part of the reference-facing contract: select kernel variants for A/B runs
 * (tools/kernel_ab.py).  0 is always the shipped default.
 *   key 0 edge gate variant : 0 shipped default (H = 128: plane-form bf16x6 edge-tile kernel k_edge_gate_pl, layer 0
 *                             k_edge_gate_enc16; H = 64: k_edge_gate_bf; H = 256: the wave-specialised / streaming kernel),
 *                             1 tile-per-workgroup kernel, 5 exact-fp32 MFMA wave-specialised kernel (LDS-counter
 *                             hand-over), 6 the same with workgroup barriers, 7 = 0 spelled out (plane form),
 *                             8 second-generation bf16x6 kernel k_edge_gate_bf at H = 128 (split in the compute waves)
 *   key 1 edge gate ablation: bit mask, timing only (results are wrong): 1 no node gathers, 2 no stores,
 *                             4 no e loads, 8 no MFMA, 16 nontemporal e loads / e' stores
 *   key 2 linear variant    : 0 shipped default (bf16x6 streaming kernel, A-stationary from 400k rows), 1 tile kernel,
 *                             2 exact-fp32 weight-stationary, 3 bf16x6 with A staged through LDS, 4 8-wave workgroups,
 *                             5 2-wave workgroups, 6 A-stationary at every size, 7 streaming kernel with row-major 16-byte
 *                             stores (the default at K = 64)
 *   key 3 gate tile order   : 1 contiguous run per workgroup (default: interleaved, XCD-contiguous)
 *   key 4 gate experiment   : kernel-specific measurement switch (H = 256 streaming gate: tiles per workgroup piece)
 *   key 5 aggregation LDS   : KiB of unused dynamic LDS per workgroup (caps the resident workgroups per CU)
 *   key 6 aggregation hubs  : 1 = hub split path off
 *   key 7 aggregation variant: 1-5 items in flight / occupancy A/B, 6 unsplit item loop
 *   key 8 reference-order kernels: 0 the fp32 matrix cores (v_mfma_f32_32x32x2_f32 as a k-ascending fma chain), 1 the
 *                             scalar-fed VALU chains of round 2 (same bits)
 *   key 9 overlap edit distance: 0 banded (Ukkonen) pass first, the full matrix for what exceeds the band; 1 full matrix only
 *   key 10 forward arithmetic : 0 fp16x3 on the f16 matrix cores for the forward's dense products at H = 128 / 256 (round 4; see
 *                             gnnome_linear_f32), 1 bf16x6 everywhere (round 3's arithmetic; the plane-form kernels at H = 256)
 *   key 7 also: 9 / 10 two nodes per wave (U = 4 / 2), 11-13 persistent contiguous chunks (measured negatives) */
int gnnome_set_tuning(int key, int value);

/* Measurement only: when set to a device buffer of 256 x 8 int64, every launch of the edge-tile kernel leaves, per
 * workgroup, the shader-clock cycles its first compute wave spent [0] waiting for a slot, [1] in the tile prologue,
 * [2] in the MFMA loop, [3] writing x back, and [4] its tile count.  NULL (the default) turns it off. */
int gnnome_debug_gate_profile(void* counters);

/* ---- graph views -------------------------------------------------------------------------------
 * Replaces what DGL builds lazily inside g.update_all / dgl.reverse (gated_gcn_full.py:99,112-113,
 * 125-126): the in-edge (by dst) and out-edge (by src) orderings of one edge list.
 *   src,dst      int32[E]   edge endpoints in edge-id order (graph.edges())
 *   in_ptr       int32[N+1] CSR offsets by destination over sorted positions
 *   srt_src/dst  int32[E]   endpoints of the edge at sorted position p
 *   srt_eid      int32[E]   original edge id of sorted position p (stable: ties keep edge-id order)
 *   out_ptr      int32[N+1] CSR offsets by source
 *   out_pos      int32[E]   for each node, the sorted positions of its out-edges (ascending)
 *   out_dst      int32[E]   out_dst[q] = srt_dst[out_pos[q]], the far endpoint of that out-edge
 */
/* Building views also resets the aggregation's per-device hub cache: the list of nodes with more than 4096 incident edges is
 * keyed by the addresses of in_ptr / out_ptr; a caller that rewrites such arrays in place must call this entry (or use new arrays). */
int gnnome_graph_views_workspace_bytes(int64_t num_nodes, int64_t num_edges, size_t* bytes_host);
int gnnome_build_graph_views(const int32_t* src, const int32_t* dst, int64_t num_nodes, int64_t num_edges,
                             int32_t* in_ptr, int32_t* srt_src, int32_t* srt_dst, int32_t* srt_eid,
                             int32_t* out_ptr, int32_t* out_pos, int32_t* out_dst, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- encoders ----------------------------------------------------------------------------------
 * out[r,:] = W2 * relu(W1 * in[row(r),:] + b1) + b2, row(r) = gather ? gather[r] : r.
 * Replaces linear2_*(relu(linear1_*(.))) at models/full_graph.py:26-27.  `gather` = srt_eid turns
 * the edge-id-ordered e[E,F] into the sorted-order e0[E,H] without a separate permute pass.
 *   in [rows_in,F]  W1 [M,F]  b1 [M]  W2 [H,M]  b2 [H]  out [rows,H];  F <= 8, M <= 64
 */
int gnnome_encode_f32(const float* in, int64_t rows, int in_features, const int32_t* gather,
                      const float* W1, const float* b1, int hidden_ne, const float* W2, const float* b2,
                      int hidden, float* out, void* stream);

/* ---- dense linear on the matrix cores -----------------------------------------------------------
 * C[M,Nout] = A[M,K] * W[Nout,K]^T + bias (torch nn.Linear layout).
 * Replaces the five node projections A_1,A_2,A_3,B_1,B_2 (gated_gcn_full.py:91-96, one call with the
 * weights concatenated) and the node halves of predictor.W1 (score_predictor.py:13-14).
 * Arithmetic: the fp32-faithful "bf16x6" product on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): every fp32
 * operand is split EXACTLY into three bf16 and six of the nine partial products are accumulated in fp32; the dropped
 * terms are <= 3 * 2^-24 |a b| per product, i.e. one fp32 rounding, so |C - exact| <= ~(K + 3) * 2^-24 * sum|a b| like
 * any fp32 dot product - but the fp32 number produced is not the one a k-ordered fma chain produces (see
 * gnnome_linear_ref_f32 for that).  The exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32) stay selectable with
 * gnnome_set_tuning(2, 2).
 * Round 4: where the output is whole 128-column blocks (K = 128 with Nout >= 256, K = 256) - the node projections - and in the forward modes of
 * the edge gate at H = 128 / 256 the product runs as "fp16x3" on the f16 matrix cores (v_mfma_f32_32x32x16_f16): every fp32 operand as TWO fp16
 * planes x1 = RN16(x), x2 = RN16((x - x1) * 2048) and three of the four plane products, the two small ones in a second fp32 accumulator that is
 * folded in with 2^-11; dropped terms <= 3 * 2^-22 |a b| per product, measured no further from an fp64 product than an fp32 GEMM
 * (tests/test_f16x3_model.py: 8e-8 of sum |a b| against 3.3e-7 at K = 256).  Operands must lie inside fp16's range, |x| < 65504: an element beyond it
 * makes its output row NaN (never a wrong finite value).  gnnome_set_tuning(10, 1) keeps bf16x6 there too.
 *   K % 32 == 0; bias may be NULL; A and W 16-byte aligned with lda, ldw % 4 == 0
 */
int gnnome_linear_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                      int Nout, float* C, int ldc, void* stream);
/* C += A*W^T + bias: the residual form the backward uses (d e_in = d e' + dxe*W3; dh = dh_in + dP*Wcat). */
int gnnome_linear_acc_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                          int Nout, float* C, int ldc, void* stream);

/* Round 6 - the node projection with PREPARED weights (gated_gcn_full.py:91-96 as one GEMM over the row-concatenated A_1 | A_2 | A_3 | B_1 | B_2;
 * score_predictor.py:13-14's node halves): gnnome_weight_planes_f16 splits W[Nout, K] ONCE into the two fp16 planes of the fp16x3 arithmetic
 * (see gnnome_linear_f32), laid out in MFMA fragment order as 16 KB chunks [Nout / 32][K / 128][8 k steps][2 planes][64 lanes][8 halves] =
 * Nout * K * 4 bytes at `planes` (256-byte aligned); gnnome_linear_planes_f32 computes C[M, Nout] = A W^T + bias from them
 * (csrc/node_project.hip): a wave keeps its 32 rows of A as planes in registers for every output column, the chunks of W arrive from L2 in an
 * LDS ring by LDS-DMA, 16-byte row pieces leave straight from the accumulators.  K in {64, 128, 256}, Nout % 32 == 0 (K <= 128: % 64), Nout <= 1536;
 * lda, ldc % 4 == 0, A, C 16-byte aligned, C must not alias A; bias may be NULL.  A row's bits depend on that row and W alone (not on M).  Same
 * operand range as every fp16x3 product: an |element| >= 65504 makes its output row NaN.
 * gnnome_linear_planes_route: 1 when a product of this shape is to run on gnnome_linear_planes_f32 under the current tuning, 0 when it keeps
 * gnnome_linear_f32's routes - ONE rule for every host above the ABI (gnnome_amd.ops.linear, gnnome_model_forward_f32, the dispatcher operator
 * gnnome_hip::linear): the kernel's shapes; without planes from the caller (`given_planes` = 0) only the node projections' shapes at K >= 128, whole 128-column
 * blocks with K = 256 or Nout >= 256; not under gnnome_set_tuning(10, 1) (bf16x6) or a gnnome_set_tuning(2, .) variant of the older kernels; not
 * when the environment holds GNNOME_PLANES_LINEAR=0 (A/B against round 5's routes).  Alignment and aliasing stay the caller's to check. */
int gnnome_weight_planes_f16(const float* W, int ldw, int Nout, int K, void* planes, void* stream);
int gnnome_linear_planes_f32(const float* A, int64_t M, int K, int lda, const void* planes, const float* bias, int Nout, float* C,
                             int ldc, void* stream);
int gnnome_linear_planes_route(int64_t M, int K, int Nout, int given_planes);

/* ---- fused edge gate ----------------------------------------------------------------------------
 * For every sorted position p:
 *   e_out[p,:] = relu(norm_e(B1h[srt_src[p],:] + B2h[srt_dst[p],:] + e_in[p,:] * W3^T)) + e_in[p,:]
 * i.e. B_3(e), u_add_v, bn_e, relu and the residual of gated_gcn_full.py:97,104-110 (and the
 * bit-identical second evaluation at :117-122) in one pass.  B_3's bias must already be folded into
 * B2h.  e_out may alias e_in.  H in {64,128,256}.
 */
int gnnome_edge_gate_f32(const float* e_in, float* e_out, int64_t num_edges, int hidden,
                         const float* B1h, const float* B2h, int ld_node, const int32_t* srt_src,
                         const int32_t* srt_dst, const float* W3, int ldw, int norm_kind,
                         const float* norm_scale, const float* norm_shift, void* stream);

/* Layer-0 form of the gate with the edge encoder folded in (models/full_graph.py:27 + gated_gcn_full.py:97-110):
 *   e0[p,:]    = encW2 * relu(encW1 * e_raw[srt_eid[p],:] + encb1) + encb2          (in_features 2, hidden_ne 16)
 *   e_out[p,:] = relu(norm_e(B1h[srt_src[p],:] + B2h[srt_dst[p],:] + e0[p,:]*W3^T)) + e0[p,:]
 * e0 is produced tile by tile in LDS and never touches HBM.  hidden in {64,128}, affine norm only; 256 since round 4 (the encoder folded
 * algebraically into a K = 16 product on the fp16x3 edge-tile kernel; default kernels only).
 * At hidden = 128 the encoder is folded algebraically: with t = relu(encW1 * e_raw + encb1) (16 wide), e0*W3^T = t*(W3*encW2)^T +
 * W3*encb2, so the gate's product and the residual e0 are two K = 16 products of one tile; W3*encW2 and W3*encb2 are recomputed at
 * every call (fp32, fixed summation tree) into a small per-device buffer - calls on DIFFERENT streams of one device must not
 * overlap.  The result differs from the unfolded form by fp32 reassociation only (a few 1e-7 relative). */
int gnnome_edge_gate_encode_f32(const float* e_raw, const int32_t* srt_eid, const float* encW1, const float* encb1,
                                const float* encW2, const float* encb2, float* e_out, int64_t num_edges, int hidden,
                                const float* B1h, const float* B2h, int ld_node, const int32_t* srt_src,
                                const int32_t* srt_dst, const float* W3, int ldw, const float* norm_scale,
                                const float* norm_shift, void* stream);

/* ---- the same two operations in the REFERENCE'S ORDER of evaluation --------------------------------
 * torch's CPU nn.Linear evaluates every output element as one k-ascending chain of fused multiply-adds that starts
 * from zero and adds the bias afterwards; its eval-mode BatchNorm1d is fma(x, alpha, beta).  These two entry points
 * reproduce that sequence operation for operation, so their results equal the reference's CPU results BIT FOR BIT on
 * equal inputs (tests/test_reference_order.py) - which matters for layers whose normalisation magnifies reorder noise
 * (weights/weights.pt layer 0: bn_e gain up to 135).  One row per lane, weights through scalar
 * loads, fp32 VALU; by default the same chain on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 evaluates its two k in order).
 * K / hidden in {64,128}; 256 on the matrix-core form only (round 4; Nout % 32 == 0 there).
 *
 * gnnome_linear_ref_f32: C[M,Nout] = chain(A W^T) + bias;  Nout % 8 == 0  (gated_gcn_full.py:91-96)
 * gnnome_edge_gate_ref_f32 (gated_gcn_full.py:97,104-110; B_3's bias is NOT folded into B2h here):
 *   B3e  = chain(e W3^T) + b3
 *   x    = (B1h[srt_src[p]] + B2h[srt_dst[p]]) + B3e
 *   e'   = max(fma(x, norm_scale, norm_shift), 0) + e
 *   with e = e_in[p,:], or - when e_raw != NULL - the edge encoder's output for edge srt_eid[p], computed in registers
 *   in the same order (models/full_graph.py:27; in_features 2, hidden_ne 16).  e_out may alias e_in.
 */
int gnnome_linear_ref_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout,
                          float* C, int ldc, void* stream);
int gnnome_edge_gate_ref_f32(const float* e_in, float* e_out, int64_t num_edges, int hidden, const float* B1h,
                             const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3,
                             int ldw, const float* b3, const float* norm_scale, const float* norm_shift, const float* e_raw,
                             const int32_t* srt_eid, const float* encW1, const float* encb1, const float* encW2,
                             const float* encb2, void* stream);

/* ---- fused gated aggregation + node update -------------------------------------------------------
 * For every node i < num_nodes_out, with s_p = sigmoid(e[p,:]):
 *   fwd = sum_{p in in(i)}  s_p * A2h[srt_src[p],:] / (sum_{p in in(i)}  s_p + 1e-6)
 *   bwd = sum_{q in out(i)} s_p * A3h[out_dst[q],:] / (sum_{q in out(i)} s_p + 1e-6),  p = out_pos[q]
 *   h_out[i,:] = relu(norm_h(A1h[i,:] + fwd + bwd)) + h_in[i,:]
 * Replaces sigmoid + both update_all pairs + the node epilogue, gated_gcn_full.py:111-114,124-137.
 * The summation order depends on the graph only (never on scheduling): bit-reproducible.
 * h_out is dense [num_nodes_out, H]; it must not alias h_in.
 */
int gnnome_node_aggregate_f32(const float* e, int hidden, int64_t num_nodes_out, const float* A1h,
                              const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                              const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                              const int32_t* out_dst, const float* h_in, int ld_h, float* h_out, int norm_kind,
                              const float* norm_scale, const float* norm_shift, void* stream);

/* The aggregation's record form (round 6; same bits as the default kernel): per-node 256-byte records [ib, in-degree, ob, out-degree, 20 x srt_src,
 * 20 x out_pos, 20 x out_dst] (records[num_nodes][64], 256-byte aligned) and the switch that makes the calling thread's following gnnome_node_aggregate_f32
 * calls (BatchNorm, whole node range) take a node's pointers AND neighbour ids from ONE load; NULL: off.  -11 % per launch at hidden = 64, level from 128 on. */
int gnnome_build_node_records(const int32_t* in_ptr, const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                              const int32_t* out_dst, int64_t num_nodes, int32_t* records, void* stream);
int gnnome_debug_node_records(const int32_t* records);

/* The same update for the nodes [node_begin, node_end) only - all pointers and num_nodes_out describe the WHOLE graph
 * exactly as for gnnome_node_aggregate_f32, rows outside the range are not touched.  Lets a caller cut one aggregation
 * (gated_gcn_full.py:111-137) into consecutive launches and start the NEXT layer's node projection (:91-96) on the rows
 * that are complete, on a second stream, while the remaining ranges are still being reduced (engine.aggregate_then_project).
 * Results are bit-identical to the single launch.  The ranges of one aggregation must be issued in ascending order
 * starting at node 0: the range that begins at 0 also runs the long-list (hub) path for the whole graph. */
int gnnome_node_aggregate_range_f32(const float* e, int hidden, int64_t num_nodes_out, int64_t node_begin, int64_t node_end,
                                    const float* A1h, const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                                    const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                    const int32_t* out_dst, const float* h_in, int ld_h, float* h_out, int norm_kind,
                                    const float* norm_scale, const float* norm_shift, void* stream);

/* ---- streaming aggregation (round 5) -----------------------------------------------------------------
 * The same node update as gnnome_node_aggregate_f32 with norm_kind = GNNOME_NORM_AFFINE (gated_gcn_full.py:111-114, :124-127,
 * :129-137), with every e row read ONCE: the rows are streamed in destination order, the in-edge sums are reduced per run, and a
 * row's contribution to its SOURCE node (the out-edge sums of dgl.reverse(g), :125-126) is accumulated in a window of slots in LDS.
 * What a row does to which slot is a SCHEDULE, a function of the graph alone, built once per graph:
 *   - the nodes are cut into num_chunks ranges of consecutive destination nodes with about equal numbers of in-edge rows
 *     (chunk_node[num_chunks + 1]); one workgroup walks one chunk, a wave of it one 32-channel slice of the rows;
 *   - a chunk's rows are walked in STEPS of <= rows_per_step (16) rows of ONE destination node: steps[][4] =
 *     {first row, node, count | flags | slot << 16, pending index}; chunk c's steps start at in_ptr[chunk_node[c]] / 16 +
 *     chunk_node[c], chunk_steps[c] of them (capacity from gnnome_stream_schedule_sizes);
 *   - edge_meta[p]: 0xFF = FAR (source in another chunk, a repeated (src, dst) pair, or no free slot: the stream skips the row's
 *     out-edge half), else state << 6 | slot, state 0 / 1 = accumulate, 2 / 3 = the LAST event of the row's source - that row
 *     also runs the source's node update (3: the source is pending);
 *   - a node with far rows is PENDING (node_pend[node] = its index, pend_nodes[index] = node, counters[0] = their number): its
 *     last event parks (A1h + fwd, num_b, den_b) in pend_rows[index][3][H], and a second launch adds the far rows in out-list
 *     order and finishes the update.
 * counters[8] (device): [0] pending nodes, [1] far rows, [2] nodes that found no free slot, [3] most slots in use, [4] most
 * steps in a chunk.  A caller reads [0] back once to size pend_rows and decides from [1] / E whether the graph's numbering has
 * the locality this form needs (a uniform random graph is all far rows: use gnnome_node_aggregate_f32).  num_slots <= 62.
 * Results are bit-reproducible and equal gnnome_node_aggregate_f32's up to fp32 reassociation (and 1-ulp reciprocals).
 * h_out must not alias h_in; hidden in {64,128,256}. */
int gnnome_stream_schedule_sizes(int64_t num_nodes, int64_t num_edges, int rows_per_step, int64_t* steps_capacity_host,
                                 size_t* scratch_bytes_host);
int gnnome_build_stream_schedule(int64_t num_nodes, int64_t num_edges, const int32_t* in_ptr, const int32_t* srt_src,
                                 const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, int num_chunks,
                                 int rows_per_step, int num_slots, int32_t* chunk_node, int32_t* chunk_steps, int32_t* steps,
                                 uint8_t* edge_meta, int32_t* node_pend, int32_t* pend_nodes, int32_t* counters, void* scratch,
                                 size_t scratch_bytes, void* stream);
int gnnome_node_aggregate_stream_f32(const float* e, int hidden, int64_t num_nodes, int64_t num_edges, const float* A1h, const float* A2h,
                                     const float* A3h, int ld_node, const int32_t* in_ptr, const int32_t* srt_src,
                                     const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, const float* h_in,
                                     int ld_h, float* h_out, const float* norm_scale, const float* norm_shift, int num_chunks,
                                     int rows_per_step, int num_slots, const int32_t* chunk_node, const int32_t* chunk_steps,
                                     const int32_t* steps, const uint8_t* edge_meta, const int32_t* node_pend,
                                     const int32_t* pend_nodes, const int32_t* counters, int64_t num_pending, float* pend_rows,
                                     void* stream);

/* ---- fused edge scorer --------------------------------------------------------------------------
 * For every sorted position p < num_edges:
 *   z1 = relu(Ps[srt_src[p],:] + Qd[srt_dst[p],:] + e[p,:] * W1e^T)        (b1 folded into Qd)
 *   logits[eid(p)] = W3 . relu(W2 * z1 + b2) + b3,   eid(p) = srt_eid ? srt_eid[p] : p
 * Replaces ScorePredictor.apply_edges (score_predictor.py:12-17) without materialising the
 * [E,3H] concatenation; W1e is the third column block of predictor.W1 (row stride ldw1 = 3H).
 *   hidden in {64,128,256}; hidden_edge_scores in {32,64,128}; W2 is [32,hs], W3 is [32]
 */
int gnnome_edge_score_f32(const float* e, int64_t num_edges, int hidden, int hidden_edge_scores,
                          const float* Ps, const float* Qd, int ld_node, const int32_t* srt_src,
                          const int32_t* srt_dst, const int32_t* srt_eid, const float* W1e, int ldw1,
                          const float* W2, const float* b2, const float* W3, const float* b3, float* logits,
                          float* z1_out /* NULL, or [E,hs]: relu(z1) kept for the backward */, void* stream);

/* ---- the whole eval forward as ONE call (round 6) ---------------------------------------------------
 * models/full_graph.py:22-30 - `x = linear2_node(relu(linear1_node(x))); e = linear2_edge(relu(linear1_edge(e)));
 * x, e = gnn(graph, x, e); scores = predictor(graph, x, e)` - i.e. what inference.py:440 calls once per graph, enqueued from C on one
 * stream: encoders -> num_layers x (node projection, edge gate, gated aggregation + node update: layers/processor.py:16-19 over
 * gated_gcn_full.py:84-142) -> score_predictor.py:12-24.  It calls the per-kernel entries above in exactly the order
 * gnnome_amd/engine.py::run_stack calls them (same kernels, same bits; tests/test_model_forward_entry.py), so a caller that scores each graph
 * once (where a hipGraph replay does not help) pays one host call instead of ~30.  No allocation, no host synchronisation, capturable.
 *   gnnome_layer_params: one SymGatedGCN layer with eval semantics - Wcat = rows A_1 | A_2 | A_3 | B_1 | B_2 (gated_gcn_full.py:29-33,91-96),
 *     bcat their biases (not reference_order: B_3's bias added to the B_2 block, it rides on the B2h rows), Wcat_planes =
 *     gnnome_weight_planes_f16(Wcat) or NULL, W3 / b3 = B_3 (:34,97), norm_kind + scale / shift = bn_e and bn_h folded as the fused
 *     kernels take them (:37-42,106,132); reference_order != 0: the layer's dense products in the reference's order of evaluation
 *     (gnnome_linear_ref_f32 / gnnome_edge_gate_ref_f32)
 *   gnnome_model_params: encoders (models/full_graph.py:13-16), layers (a HOST array), predictor with W1 split as the scorer takes it:
 *     W_nodes [2 hs, H] = W1's x[src] block over its x[dst] block, b_nodes = [0 | b1], W1e [hs, H] = W1's e block (score_predictor.py:13)
 *   gnnome_views: the arrays of gnnome_build_graph_views; node_gather: NULL, or the row of x each node of the views' numbering reads
 *     (views over renumbered nodes); transposed != 0: the views describe dgl.reverse(g) (train.py:165) - src / dst roles swapped
 *   workspace: gnnome_model_forward_workspace_bytes(...) bytes, 256-byte aligned; logits [E] in EDGE-ID order (what the model returns,
 *     squeezed).  hidden in {64, 128, 256}, score_hidden in {32, 64, 128}: narrower models are zero-padded by the caller (engine.Prepared). */
typedef struct gnnome_layer_params {
    const float* Wcat;
    const void* Wcat_planes;
    const float* bcat;
    const float* W3;
    const float* b3;
    const float* scale_e;
    const float* shift_e;
    const float* scale_h;
    const float* shift_h;
    int32_t norm_kind;
    int32_t reference_order;
} gnnome_layer_params;

typedef struct gnnome_model_params {
    int32_t hidden, hidden_ne, num_layers, score_hidden, node_features, edge_features;
    const float *node_W1, *node_b1, *node_W2, *node_b2;   /* linear1_node [hidden_ne, node_features], linear2_node [hidden, hidden_ne] */
    const float *edge_W1, *edge_b1, *edge_W2, *edge_b2;   /* linear1_edge, linear2_edge */
    const gnnome_layer_params* layers_host;                /* [num_layers], host memory */
    const float* W_nodes;
    const void* W_nodes_planes;                            /* gnnome_weight_planes_f16(W_nodes) or NULL */
    const float* b_nodes;
    const float* W1e;
    int32_t ld_w1e, reserved;
    const float *W2, *b2, *W3, *b3;                        /* predictor.W2 [32, hs], W3 [32], biases */
} gnnome_model_params;

typedef struct gnnome_views {
    int64_t num_nodes, num_edges;
    const int32_t *in_ptr, *srt_src, *srt_dst, *srt_eid, *out_ptr, *out_pos, *out_dst;
    const int32_t* node_gather;
    int32_t transposed, reserved;
} gnnome_views;

/* Measurement only (bench.py's live `roofline`): four hipEvent_t (as void*) that every following gnnome_model_forward_f32 of the calling thread
 * records on its stream around layer `layer`'s edge-gate launch (start, stop) and its aggregation launch (start, stop); NULLs / layer < 0: off. */
int gnnome_debug_forward_events(void* gate_start, void* gate_stop, void* aggregate_start, void* aggregate_stop, int layer);
int gnnome_model_forward_workspace_bytes(int64_t num_nodes, int64_t num_edges, int hidden, int score_hidden, size_t* bytes_host);
int gnnome_model_forward_f32(const gnnome_model_params* params_host, const gnnome_views* views_host, const float* x, const float* e_raw,
                             float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* The same forward on buffers the caller allocated one by one instead of one workspace block (round 6): where the driver places a buffer in HBM is
 * worth ~3 % of the forward at configs[1], and separate allocations come out on the fast side more often than one block holding everything
 * (NOTES.md round 6).  h[0], h[1] [N, hidden]; P [N, 5 hidden]; e[0] [E, hidden], and e[1] (another [E, hidden], used at hidden = 256 only, may be NULL
 * otherwise); PQ [N, 2 score_hidden]; all 16-byte aligned, distinct, contents undefined on entry and on return. */
typedef struct gnnome_forward_buffers {
    float* h[2];
    float* P;
    float* e[2];
    float* PQ;
} gnnome_forward_buffers;
int gnnome_model_forward_buffers_f32(const gnnome_model_params* params_host, const gnnome_views* views_host, const float* x, const float* e_raw,
                                     float* logits, const gnnome_forward_buffers* buffers_host, void* stream);

/* ================================================================================================
 * Training step (train.py:138-145 + :328-330: forward in train mode, BCE-with-logits, loss.backward()).
 * The reference gets its backward from torch autograd over DGL's gspmm/gsddmm; here the layer backward is
 * written out in gnnome_amd/train.py and built from the entries below.  All are stream-ordered.
 * ================================================================================================ */

/* x_out[p,:] = B1h[srt_src[p],:] + B2h[srt_dst[p],:] + e_in[p,:] * W3^T  - the input of bn_e
 * (gated_gcn_full.py:104-106) materialised so that train-mode BatchNorm can take batch statistics. */
int gnnome_edge_gate_raw_f32(const float* e_in, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                             const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                             const float* W3, int ldw, void* stream);

/* The same, with the BatchNorm batch statistics gathered on the way (hidden in {64,128,256}): every workgroup writes
 * its shifted column sums  stats_partial[w][c] = sum (x - center[c]),  stats_partial[w][hidden + c] = sum (x - center[c])^2
 * over the rows it produced; w < R = gnnome_edge_gate_raw_stats_rows(hidden).  (hidden = 256: two stacked matrices instead,
 * stats_partial[0][w][c] the sums and stats_partial[1][w][c] the sums of squares, 2 * R * 256 floats in all, and x_out
 * must not alias e_in.)  Summing the rows of stats_partial (in any
 * FIXED order, e.g. gnnome_colsum2_f32) gives the sums the statistics need without a second pass over x_out.
 * center: any per-column value near the column mean (row 0 of x_out is one) - the shift keeps the second moment
 * free of cancellation. */
/* x_out == NULL (hidden = 128): the statistics alone - the first pass of the two-pass training forward (round 4): the product is cheap on the
 * matrix cores (fp16x3), a written and re-read [E,H] tensor is not.  The second pass is gnnome_edge_gate_bn_f32 below. */
int gnnome_edge_gate_raw_stats_rows(int hidden, int* rows_host);
int gnnome_edge_gate_raw_stats_f32(const float* e_in, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                                   const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                   const float* W3, int ldw, const float* center, float* stats_partial, void* stream);
/* e_out = relu(xe * scale + shift) + e_in  AND  x_out = xe = e_in W3^T + B1h[src] + B2h[dst]: the train-mode gate once the batch statistics are
 * known (scale = gamma * rstd, shift = beta - mean * scale): what gnnome_edge_gate_raw_f32 + gnnome_bn_relu_res_f32 did in two passes over
 * [E,H] (gated_gcn_full.py:97,104-110 under train()).  hidden = 128; e_out, x_out and e_in distinct. */
int gnnome_edge_gate_bn_f32(const float* e_in, float* e_out, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                            const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw,
                            const float* scale, const float* shift, void* stream);

/* Gated aggregation without the node epilogue.
 *   mode 1: v_out = A1h + fwd + bwd (input of bn_h, gated_gcn_full.py:129); aux0 = fwd, aux1 = 1/(den_f+1e-6),
 *           aux2 = bwd, aux3 = 1/(den_b+1e-6)   (kept for the backward)
 *   mode 2: aux0[i] = sum_{in(i)} s_p * A2h[srt_src[p]], aux2[i] = sum_{out(i)} s_p * A3h[out_dst[q]] (raw sums:
 *           with A2h := dv/(den_b+eps), A3h := dv/(den_f+eps) these are dA3h and dA2h) */
int gnnome_node_aggregate_raw_f32(const float* e, int hidden, int64_t num_nodes, int mode, const float* A1h,
                                  const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                                  const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                  const int32_t* out_dst, float* v_out, float* aux0, float* aux1, float* aux2,
                                  float* aux3, void* stream);

/* s1[c] = sum_r x'[r,c];  s2[c] = sum_r x'[r,c]*y'[r,c]  with x' = x - center[c] (center NULL: 0) and y NULL -> x'
 * (sum of squares).  s1/s2 are overwritten (rows = 0: left untouched).  hidden in {16,32,64,128,256}.  Two calls give train-mode
 * BatchNorm its batch statistics (gated_gcn_full.py:106,119,132): the mean, then the centred second moment - the
 * one-pass E[x^2]-E[x]^2 form cancels badly on this path (|e| ~ 500, spread of a few units).  Also bias gradients.
 * The sum over workgroups is deterministic (partials parked in `workspace`, added up in workgroup order by a second
 * small kernel), so BatchNorm statistics and bias gradients are bit-identical from run to run.  workspace:
 * gnnome_colsum_workspace_bytes() bytes, 16-byte aligned, not shared by launches that may run concurrently. */
int gnnome_colsum_workspace_bytes(size_t* bytes_host);
int gnnome_colsum2_f32(const float* x, const float* y, int64_t rows, int hidden, const float* center, float* s1,
                       float* s2, void* workspace, size_t workspace_bytes, void* stream);

/* out = relu(x*scale[c] + shift[c]) + res : train-mode bn + relu + residual (gated_gcn_full.py:106-110, 132-137) */
int gnnome_bn_relu_res_f32(const float* x, const float* scale, const float* shift, const float* res, int64_t rows,
                           int hidden, float* out, void* stream);

/* BatchNorm backward through the relu of out = relu(x*scale + shift) + res, m = (x*scale + shift > 0):
 *   stats: s1[c] = sum_r dy*m,  s2[c] = sum_r dy*m*(x - mean[c])   (overwritten)
 *   apply: dx = a[c] * (dy*m - c1[c] - (x - mean[c])*rstd[c]*c2[c]) */
int gnnome_bn_bwd_stats_f32(const float* dy, const float* x, const float* scale, const float* shift,
                            const float* mean, int64_t rows, int hidden, float* s1, float* s2, void* workspace,
                            size_t workspace_bytes, void* stream);
int gnnome_bn_bwd_apply_f32(const float* dy, const float* x, const float* scale, const float* shift, int64_t rows,
                            int hidden, const float* a, const float* c1, const float* c2, const float* mean,
                            const float* rstd, float* dx, void* stream);

/* gnnome_bn_bwd_apply_f32 for bn_h of a layer WITH the four node tables of the aggregation's backward in the same pass (round 5; what two
 * gnnome_mul23_f32 launches made of dx, each reading it again):  Tf = dx rdf,  Uf = Tf hf,  Tb = dx rdb,  Ub = Tb hb, all [rows, hidden]
 * contiguous; rdf / hf / rdb / hb are the forward's gnnome_node_aggregate_raw_f32 mode-1 outputs aux1 / aux0 / aux3 / aux2. */
int gnnome_bn_bwd_apply_tables_f32(const float* dy, const float* x, const float* scale, const float* shift, int64_t rows, int hidden,
                                   const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                                   const float* rdf, const float* hf, const float* rdb, const float* hb, float* dx, float* Tf, float* Uf,
                                   float* Tb, float* Ub, unsigned* amax_bits, void* stream);   /* amax_bits (NULL: skip): raised to max |dx| */

/* LayerNorm variant (normalization='layer', gated_gcn_full.py:40-42,106,119,132) of the two groups above:
 *   out = relu(LN(x) * gamma + beta) + res, LN over the `hidden` entries of each row (biased variance, eps 1e-5);
 *   backward: dx = rstd_row (g - mean_row(g) - xhat mean_row(g xhat)) with g = dy m gamma, m the relu mask;
 *   dbeta[c] = sum_r dy m, dgamma[c] = sum_r dy m xhat (deterministic; overwritten; workspace as colsum2). */
/* norm_width (round 5): the statistics run over the first norm_width <= hidden channels (0 = hidden) - see GNNOME_NORM_LAYER_OVER;
 * dx of a channel beyond norm_width is 0. */
int gnnome_ln_relu_res_f32(const float* x, const float* gamma, const float* beta, const float* res, int64_t rows, int hidden,
                           int norm_width, float* out, void* stream);
int gnnome_ln_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, int64_t rows, int hidden, int norm_width,
                      float* dx, float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes, void* stream);

/* o1 = a*b, o2 = a*b*c (count floats, count % 4 == 0);  out = a + b;  dx = dy*(y > 0) */
int gnnome_mul23_f32(const float* a, const float* b, const float* c, int64_t count, float* o1, float* o2, void* stream);
int gnnome_add_f32(const float* a, const float* b, int64_t count, float* out, void* stream);
int gnnome_relu_bwd_f32(const float* dy, const float* y, int64_t count, float* dx, void* stream);

/* out[i,:] = sum_{q in [ptr[i], ptr[i+1])} X[pos ? pos[q] : q, :]  width in {32,64,128,256}: the transposes of the
 * per-edge gathers (dB2h, dQd with ptr = in_ptr; dB1h, dPs with ptr = out_ptr, pos = out_pos). */
int gnnome_segment_sum_f32(const float* X, int width, const int32_t* ptr, const int32_t* pos, int64_t num_nodes,
                           float* out, int ld_out, void* stream);

/* Both segment sums of one tensor in one launch: out_in[i,:] = sum over node i's in-edge rows (ptr = in_ptr, contiguous),
 * out_out[i,:] = sum over its out-edge rows (out_ptr / out_pos).  Same values as two gnnome_segment_sum_f32 calls. */
int gnnome_segment_sum2_f32(const float* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                            int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, void* stream);

/* C[Ka,Kb] = A[rows,Ka]^T * B[rows,Kb]: nn.Linear weight gradients on the bf16 matrix cores as the fp32-faithful
 * bf16x6 product (see gnnome_linear_f32; error of the size of an fp32 dot product's), chunked over rows with a
 * deterministic second-stage sum.  Ka, Kb % 4 == 0; A, B 16-byte aligned. */
int gnnome_wgrad_workspace_bytes(int64_t rows, int Ka, int Kb, size_t* bytes_host);
int gnnome_wgrad_f32(const float* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C, int ldc,
                     void* workspace, size_t workspace_bytes, void* stream);
/* gnnome_wgrad_f32 for a GRADIENT operand A whose largest magnitude is known on the device (round 5; VERDICT r4: "fp16x3 for the backward's
 * products: gradients need a per-tensor scale"): amax_bits[0] = the bits of max |A| as a non-negative float.  Where the 128 x 128 tile
 * kernel runs, the product is formed as fp16x3 - A multiplied by 2^(13 - floor(log2 max|A|)) while it is split into two fp16 planes (exact),
 * the result by the inverse - three MFMAs per k step instead of bf16x6's six; error per term <= max(2^-22 |a|, 2^-49 max|A|) |b|
 * (measured 2.6e-9 of max sum |a||b| at 1M rows against bf16x6's 8.1e-9).  B must lie in fp16's range like every operand of the forward
 * (NaN otherwise, never a wrong finite value).  Operands that take the 256 x 256 kernel ignore amax_bits (bf16x6). */
int gnnome_wgrad_scaled_f32(const float* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, const unsigned* amax_bits,
                            float* C, int ldc, void* workspace, size_t workspace_bytes, void* stream);
/* The same for the column-block form (gnnome_wgrad_blocks_f32) with ONE maximum over all blocks - the slot the blocks' producers raised
 * (gnnome_bn_bwd_apply_tables_f32, gnnome_agg_bwd_fused_f32, gnnome_segment_sum2_amax_f32; the caller zeroes it once before them): fp16x3 with
 * that common scale (block_width a multiple of 128); a block much smaller than the largest keeps an absolute error of 2^-49 max|A| |b| per term. */
int gnnome_wgrad_blocks_scaled_f32(const float* const* A_blocks, int num_blocks, int block_width, int lda, const float* B, int ldb, int Kb,
                                   int64_t rows, const unsigned* amax_bits, float* C, int ldc, float* colsum, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* gnnome_linear_blocks_f32 as ONE fp16x3 launch with the blocks scaled by their common maximum (the same slot): a workgroup keeps a 128 x 128
 * tile of C in its accumulators over all of K = num_blocks x block_width (block_width a multiple of 32), so C is read and written once - the
 * data gradient dh += [dv | dA2 | dA3 | dB1 | dB2] Wcat of the node projection (gated_gcn_full.py:91-96 under autograd). */
int gnnome_linear_blocks_scaled_f32(const float* const* A_blocks, int num_blocks, int block_width, int64_t M, int lda, const float* W, int ldw,
                                    int Nout, const unsigned* amax_bits, float* C, int ldc, int accumulate, void* stream);
/* gnnome_segment_sum2_f32 that also raises amax_bits[0] to max |out_in|, |out_out| (not zeroed here) */
int gnnome_segment_sum2_amax_f32(const float* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                                 int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, unsigned* amax_bits, void* stream);

/* The same with A given as num_blocks (1..8) column blocks of equal width in separate buffers - A_blocks is a HOST array of
 * device pointers, each [rows, block_width] with row stride lda - so that the five gradients of a layer's node projections
 * (dA1h | dA2h | dA3h | dB1h | dB2h, gated_gcn_full.py:89-95) need not be concatenated: C[num_blocks*block_width, Kb].
 * colsum (may be NULL): [num_blocks*block_width] column sums of A over the rows = the bias gradients, from the same pass.
 * Workspace: gnnome_wgrad_workspace_bytes(rows, num_blocks*block_width, Kb). */
int gnnome_wgrad_blocks_f32(const float* const* A_blocks, int num_blocks, int block_width, int lda, const float* B, int ldb,
                            int Kb, int64_t rows, float* C, int ldc, float* colsum, void* workspace, size_t workspace_bytes,
                            void* stream);

/* C[M,Nout] (accumulate != 0: +=) [A_0 | A_1 | ...] * W[Nout, num_blocks*block_width]^T with the column blocks of A in separate
 * buffers as above (block_width % 32 == 0): the backward's dh += dP * Wcat (gnnome_linear_acc_f32 on the concatenation). */
int gnnome_linear_blocks_f32(const float* const* A_blocks, int num_blocks, int block_width, int64_t M, int lda, const float* W,
                             int ldw, int Nout, float* C, int ldc, int accumulate, void* stream);

/* Backward of the scorer tail (score_predictor.py:15-16) from the saved relu(z1) and d(score) (edge-id order,
 * read through srt_eid): dz1[E,hs], dz2[E,32], u[E,32] = dscore*z2.  hs in {32,64}. */
int gnnome_score_tail_bwd_f32(const float* z1, const float* dscore, const int32_t* srt_eid, int64_t num_edges,
                              int hidden_edge_scores, const float* W2, const float* b2, const float* W3, float* dz1,
                              float* dz2, float* u, void* stream);

/* de[p,:] += s(1-s) * (Tf[dst]*A2h[src] - Uf[dst] + Tb[src]*A3h[dst] - Ub[src]),  s = sigmoid(e[p,:]):
 * the gradient the two gated aggregations send back to the edge state (gated_gcn_full.py:111-114,124-127). */
int gnnome_agg_edge_bwd_f32(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                            const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                            const int32_t* srt_src, const int32_t* srt_dst, float* de, void* stream);

/* t[r,:] = relu(W1*in[row(r),:] + b1): an encoder's hidden layer, recomputed for its backward */
int gnnome_encode_hidden_f32(const float* in, int64_t rows, int in_features, const int32_t* gather, const float* W1,
                             const float* b1, int hidden_ne, float* t, void* stream);

/* ---- row gather (halo packing for the destination-range partition) ------------------------------
 * out[r,:] = in[idx[r],:]   rows of `width` floats, width % 4 == 0
 */
int gnnome_gather_rows_f32(const float* in, int ld_in, const int32_t* idx, int64_t rows, int width, float* out,
                           int ld_out, void* stream);

/* out[idx[r],:] += in[r,:]  - the transpose of the halo gather: in the partitioned training step the gradient of
 * a halo row travels back to the row's owner and is added there (no reference counterpart).  The
 * rows in idx must be DISTINCT within one call (one peer's block of the exchange plan), which makes the sum
 * order, and therefore the result, deterministic. */
int gnnome_scatter_add_rows_f32(const float* in, int ld_in, const int32_t* idx, int64_t rows, int width, float* out,
                                int ld_out, void* stream);

/* ================================================================================================
 * The callers' arithmetic either side of the model call: feature preparation
 * before it, loss and confusion counts after it.  Deterministic reductions; `workspace` for all three is
 * gnnome_closure_workspace_bytes() bytes, 8-byte aligned.
 * ================================================================================================ */
int gnnome_closure_workspace_bytes(size_t* bytes_host);

/* x[N,2] = [zscore(in_degree) | zscore(out_degree)] (columns swapped when `reverse`), degrees read off the CSR
 * pointers of the graph views, mean and UNBIASED std as torch.mean / torch.std.
 * Replaces inference.py:416-420 and train.py:112-122 (get_full_ne_features). */
int gnnome_degree_features_f32(const int32_t* in_ptr, const int32_t* out_ptr, int64_t num_nodes, int reverse, float* x,
                               void* workspace, size_t workspace_bytes, void* stream);

/* e[E,2] = [zscore(overlap_length) | overlap_similarity], unbiased std.
 * Replaces utils/data_utils.py:31-41 (preprocess_graph with use_similarities, configs/hyperparameters.py:17). */
int gnnome_edge_features_f32(const float* overlap_length, const float* overlap_similarity, int64_t num_edges, float* e,
                             void* workspace, size_t workspace_bytes, void* stream);

/* loss[0] = mean_i ( bce(logits_i) [+ bce(logits_rev_i) + alpha |logits_i - logits_rev_i|] ),
 * bce = F.binary_cross_entropy_with_logits(., labels, pos_weight) - train.py:144 (logits_rev NULL) and
 * train.py:103-109 symmetry_loss (logits_rev given).  pos_weight: DEVICE scalar.  dlogits / dlogits_rev (NULL: skip) =
 * grad_scale * d(sum of per-edge terms)/d logits, so grad_scale = 1/E gives the gradient of the mean.
 * tfpn (NULL: skip) = int64[4] TP, TN, FP, FN of round(sigmoid(logits)) against labels - utils/metrics.py:6-12. */
int gnnome_edge_loss_f32(const float* logits, const float* logits_rev, const float* labels, int64_t num_edges,
                         const float* pos_weight, float alpha, float grad_scale, float* loss, float* dlogits,
                         float* dlogits_rev, int64_t* tfpn, void* workspace, size_t workspace_bytes, void* stream);

/* gnnome_bn_bwd_apply_f32 followed by gnnome_linear_acc_f32 in one pass over the [rows,hidden] tensors (hidden in {64,128}):
 *   dxe[r,:] = a (C[r,:] m - c1 - (X[r,:] - mean) rstd c2),  m = (X scale + shift > 0)        (written out)
 *   C[r,:]  += dxe[r,:] * W^T                                                                  (in place)
 * i.e. train.py's  dxe = bn_e backward(de, xe);  d e_in = d e' + dxe W3  with C = de, X = xe, W = W3^T.
 * rows_once (0 <= rows_once <= rows): the mean-subtraction terms c1, c2 enter the first rows_once rows only - on a
 * destination-range partition the rows past the owned in-edges are replicas of edges another rank owns, and those terms
 * must enter once per edge of the whole graph; rows_once = rows on a single rank. */
int gnnome_bn_bwd_dgrad_f32(float* C, const float* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift,
                            const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                            const float* W, int ldw, float* dxe, void* stream);
/* gnnome_bn_bwd_dgrad_f32 (hidden = 128) that also leaves max |dxe| at amax_bits[0] - the bits of a non-negative float, zeroed by the call
 * and raised with atomicMax on the unsigned value (a maximum does not depend on the order it is formed in) - for gnnome_wgrad_scaled_f32. */
int gnnome_bn_bwd_dgrad_amax_f32(float* C, const float* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift,
                                 const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                                 const float* W, int ldw, float* dxe, unsigned* amax_bits, void* stream);
/* the same at hidden = 256, OUT OF PLACE (C_out != C_in): there two workgroups - one per column half - read whole rows of C,
 * so the updated rows cannot overwrite them: C_out = C_in + dxe W^T. */
int gnnome_bn_bwd_dgrad_out_f32(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                const float* mean, const float* rstd, const float* W, int ldw, float* dxe, void* stream);
/* ... and max |dxe| left at amax_bits[0] as the bits of a non-negative float (zeroed first), as gnnome_bn_bwd_dgrad_amax_f32 does at hidden = 128:
 * gnnome_wgrad_scaled_f32(dxe, e, amax_bits) then runs B_3's weight gradient as fp16x3 (round 6). */
int gnnome_bn_bwd_dgrad_out_amax_f32(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                     const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                     const float* mean, const float* rstd, const float* W, int ldw, float* dxe, unsigned* amax_bits, void* stream);

/* gnnome_agg_edge_bwd_f32 with the BatchNorm-backward statistics of its result gathered in the same pass:
 *   de[p,:] += s(1-s)(...)   as above, then   s1[c] = sum_p de[p,c] m,  s2[c] = sum_p de[p,c] m (xe[p,c] - mean[c]),
 *   m = (xe*scale + shift > 0): what gnnome_bn_bwd_stats_f32(de, xe, ...) would return, without its pass over de and xe. */
int gnnome_agg_edge_bwd_stats_f32(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                  const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                  const int32_t* srt_src, const int32_t* srt_dst, float* de, const float* xe, const float* scale,
                                  const float* shift, const float* mean, float* s1, float* s2, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* The aggregation's whole backward over the edges in ONE launch (round 5; the backward of gated_gcn_full.py:118-126's two gated
 * sums as autograd derives it): gnnome_node_aggregate_raw_f32 mode 2 with the tables (Tb at src, Tf at dst)
 *   sum_in[i,:]  = sum_{p: dst_p = i} sigmoid(e[p,:]) Tb[src_p,:]      sum_out[i,:] = sum_{p: src_p = i} sigmoid(e[p,:]) Tf[dst_p,:]
 * AND gnnome_agg_edge_bwd_stats_f32 (de updated in place, s1 / s2 of the result) - as two launches they stream e from HBM twice.
 * One wave per node over the destination-sorted views; num_nodes rows of Tf/Uf/Tb/Ub [., hidden] and A2h/A3h (row stride ld_node),
 * sum_in / sum_out [num_nodes, hidden] overwritten.  Equal to the two launches up to fp32 reassociation of the sums; every sum is
 * formed in an order that depends on the graph and the launch geometry only (same bits on every run).  workspace as gnnome_colsum2_f32.
 * amax_bits (NULL: skip): a slot holding the bits of a non-negative float, RAISED (atomicMax on the unsigned value, never zeroed here) to
 * max |sum_in|, |sum_out| - see gnnome_wgrad_blocks_scaled_f32. */
int gnnome_agg_bwd_fused_f32(const float* e, int64_t num_nodes, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                             const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                             const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, float* de,
                             const float* xe, const float* scale, const float* shift, const float* mean, float* sum_in, float* sum_out,
                             float* s1, float* s2, unsigned* amax_bits, void* workspace, size_t workspace_bytes, void* stream);

/* The per-channel arithmetic of a train-mode BatchNorm1d call in one launch (gated_gcn_full.py:106,119,132 with
 * nn.BatchNorm1d's buffer semantics): from shifted column sums d1 = sum(x - center), d2 = sum((x - center)^2) over `rows`
 * rows -> mean, rstd = 1/sqrt(biased var + eps), scale = gamma*rstd, shift = beta - mean*scale, and `updates` momentum
 * updates of running_mean / running_var (unbiased variance) + num_batches_tracked += updates.  center, running_* and
 * num_batches_tracked may be NULL. */
int gnnome_bn_train_finish_f32(const float* d1, const float* d2, const float* center, int64_t rows, int hidden, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float momentum, float eps, int updates, float* mean, float* rstd, float* scale, float* shift,
                               void* stream);
/* BatchNorm backward, one rank: the per-channel vectors between the statistics pass and the apply pass in one launch -
 * s2h = rstd * s2, c1 = s1 / rows, c2 = s2h / rows (what torch autograd derives for nn.BatchNorm1d in train mode,
 * gated_gcn_full.py:106,119,132; the reference spends three elementwise operators on them). */
int gnnome_bn_bwd_terms_f32(const float* s1, const float* s2, const float* rstd, int64_t rows, int hidden, float* s2h, float* c1,
                            float* c2, void* stream);
/* One layer's weights as the kernels read them, in one launch (gated_gcn_full.py:91-97: the five node projections run as ONE
 * product with the stacked weight): Wcat[5H,H] = A_1|A_2|A_3|B_1|B_2 weights stacked, bcat[5H] = their biases with B_3's added
 * to B_2's, WcatT[H,5H] and W3T[H,H] = the transposes the backward's data-gradient products read.  weights5 / biases5 are HOST
 * arrays of five device pointers; hidden a multiple of 32. */
int gnnome_pack_layer_f32(const float* const* weights5, const float* const* biases5, const float* B3_weight, const float* B3_bias,
                          int hidden, float* Wcat, float* bcat, float* WcatT, float* W3T, void* stream);
/* center[:] = B1h[srt_src[0],:] + B2h[srt_dst[0],:] + e[0,:] W3^T: row 0 of the raw gate, the shift of its batch statistics. */
int gnnome_gate_center_f32(const float* e, int64_t num_edges, int hidden, const float* B1h, const float* B2h, int ld_node,
                           const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw, float* center, void* stream);

/* ---- greedy decode of the edge scores into contig walks -----------------------------
 * What inference.py:70-165 (greedy_forwards, greedy_backwards_rc, run_greedy_both_ways) and :29-36 (get_contig_length)
 * do with Python dicts and sets, for all sampled start edges of one iteration of get_contigs_greedy (:193-300) in one
 * launch, one wavefront per candidate.
 *   succ_ptr/succ_nbr/succ_eid  successor lists in EDGE-ID order (graph_parser.py:31-37) and the edge id of every slot
 *   logp        float[E]  log(sigmoid(score)) by edge id (inference.py:184); prefix_len int32[E]; read_len int32[N]
 *   visited     uint8[N]  nodes consumed by earlier contigs; cand_* the sampled edges (endpoints and edge id)
 *   walks_f[c,:len_f[c]]  forward walk from dst; walks_b[c,:len_b[c]] the walk from src^1 on the reverse-complement
 *   strand in walk order (the contig is reversed(walks_b ^ 1) followed by walks_f, inference.py:157, :254)
 *   sum_f/sum_b  fp32 sums of the chosen log-probabilities in walk order; contig_len[c] int64 as get_contig_length
 *   status[c]    0, or bit 0: a walk reached `capacity` and was cut, bit 1: an edge of the backward half has no mate
 *   exact ties for the best successor are resolved as torch.topk(k=1) resolves them on the CPU (decode.hip)
 * Node 2r+1 is the reverse complement of node 2r (x ^ 1, as in the reference); N even.
 */
int gnnome_greedy_walks_workspace_bytes(int64_t num_nodes, int num_candidates, size_t* bytes_host);
int gnnome_greedy_walks(const int32_t* succ_ptr, const int32_t* succ_nbr, const int32_t* succ_eid, const float* logp,
                        const int32_t* prefix_len, const int32_t* read_len, const uint8_t* visited, int64_t num_nodes,
                        const int32_t* cand_src, const int32_t* cand_dst, const int32_t* cand_eid, int num_candidates,
                        int32_t* walks_f, int32_t* walks_b, int64_t capacity, int32_t* len_f, int32_t* len_b, float* sum_f,
                        float* sum_b, int64_t* contig_len, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
/* visited[n] = visited[n^1] = 1 for every node of `walk` and for every node it jumped over: for consecutive (ss, dd),
 * succs[ss] & preds[dd] and their mates (inference.py:313-318, :334). */
int gnnome_mark_walk_visited(const int32_t* succ_ptr, const int32_t* succ_nbr, const int32_t* walk, int64_t walk_len,
                             uint8_t* visited, void* stream);

/* ---- overlap similarity -------------------------------------------------------------------------------
 * Replaces graph_parser.py:101-117 (calculate_similarities): for every edge
 *     edit_distance = edlib.align(read_seqs[src][-ol:], read_seqs[dst][:ol])['editDistance']     (edlib defaults: global / NW)
 *     overlap_similarity = 1 - edit_distance / ol        (0.5 where ol == 0)
 * with read_seqs[2r] = read r, read_seqs[2r+1] = its reverse complement (:365).  The distance is the exact Levenshtein
 * distance (Myers' bit-vector programme, one wavefront per overlap: csrc/overlap_similarity.hip).
 *   reads       uint8[total]   the reads as the S lines give them, concatenated (forward strand only)
 *   read_off    int64[R+1]     offsets of read r in `reads`
 *   symtab      uint8[512]     [b] = index of byte b in the caller's alphabet, [256+b] = index of complement(b)
 *                              (the table Bio.Seq.reverse_complement applies); num_symbols <= 32
 *   src, dst    int32[E]       node ids (graph.edges()); overlap_length int32[E].  An edge with an endpoint outside [0, 2 num_reads)
 *                              is reported as -1 and not aligned; symtab entries >= num_symbols are read as num_symbols - 1
 *   dist_out    int32[E]       edit distances; entries the kernels cannot serve keep the caller's fill value (fill with -1):
 *                              a query longer than 65 536 bases, or an alphabet whose match masks exceed LDS at that length
 *   similarity_out float32[E]  (NULL: skip) 1 - dist / ol evaluated in double precision like the reference's Python floats */
int gnnome_overlap_workspace_bytes(size_t* bytes_host);
int gnnome_overlap_edit_distance(const uint8_t* reads, const int64_t* read_off, int64_t num_reads, const uint8_t* symtab,
                                 int num_symbols, const int32_t* src, const int32_t* dst, const int32_t* overlap_length,
                                 int64_t num_edges, int32_t* dist_out, float* similarity_out, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- Node order (round 4): locality for inputs whose node ids do not follow the layout ---------------------------------------
 * The reference numbers nodes in S-line order of the GFA (graph_parser.py:174-181: read r -> nodes 2r, 2r+1), which need not be
 * the order of the reads along the genome; gnnome_amd/node_order.py renumbers the READS once per graph so that the
 * destination-range partition (gnnome_amd/dist.py) cuts few edges and gathers hit L2.  Logits stay in edge-id order: nothing
 * the reference's callers see changes.  Two kernels; sorting, compaction and the final numbering are torch operators.
 *
 * gnnome_adjacency_support: for every entry p of a CSR adjacency with SORTED rows (row_of[p] = its row), supported[p] = 1 iff
 *   the two endpoints have a common neighbour (an overlap graph is locally transitive; repeat-induced edges are not).
 * gnnome_bfs_levels: breadth-first levels of every component, one workgroup for the whole graph (the frontier of an overlap graph
 *   is dozens of reads wide and tens of thousands of levels deep).  level_key[v] int32[num_nodes] = a counter that grows by one
 *   per level and does not restart between components (-1: never reached - no neighbours); seeds / num_seeds (device, NULL: the
 *   smallest unvisited id that has a neighbour starts the next component); far_node[c] = the smallest id in the last level of
 *   component c; *num_components (device).  frontier_workspace: int32[2 * num_nodes].  Levels and far nodes are functions of the
 *   graph alone. */
int gnnome_adjacency_support(const int32_t* ptr, const int32_t* adj, const int32_t* row_of, int64_t num_rows, int64_t nnz,
                             uint8_t* supported, void* stream);
int gnnome_bfs_levels(const int32_t* ptr, const int32_t* adj, int64_t num_nodes, const int32_t* seeds, const int32_t* num_seeds,
                      int32_t* level_key, int32_t* frontier_workspace, int32_t* far_node, int32_t* num_components, void* stream);

/* ---- bf16 STORAGE of two training activations (activation_storage = "bf16", BASELINE configs[2]) ------------------------------
 * The reference trains in fp32 throughout; as an OPTION the pre-normalisation gate output xe[E,H] (written once, read three
 * times per layer) and its gradient dxe[E,H] (written once, read twice) can live in HBM as bfloat16 - rounded to nearest even
 * when written, widened exactly when read; every sum, product and statistic is still fp32, and the residual streams e, e', de,
 * h stay fp32 (|e| reaches several hundred with per-layer updates of order one: bf16 there would drop the updates).  Each entry
 * is its _f32 namesake with the marked tensor as uint16_t (bf16 bits), contiguous, 8-byte aligned:
 *   raw_stats: x_out (the statistics are those of the ROUNDED values);  bn_relu_res: x;  agg_edge_bwd_stats, agg_bwd_fused: xe;
 *   bn_bwd_dgrad: X and dxe (C += dxe W^T uses the unrounded dxe);  segment_sum2: X;  wgrad: A (lda in elements). */
int gnnome_edge_gate_raw_stats_x16(const float* e_in, uint16_t* x_out, int64_t num_edges, int hidden, const float* B1h, const float* B2h,
                                   int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw,
                                   const float* center, float* stats_partial, void* stream);
/* gnnome_edge_gate_bn_f32 with x_out stored as bf16; e_out is computed from the ROUNDED rows (what gnnome_bn_relu_res_x16 reads back). */
int gnnome_edge_gate_bn_x16(const float* e_in, float* e_out, uint16_t* x_out, int64_t num_edges, int hidden, const float* B1h,
                            const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw,
                            const float* scale, const float* shift, void* stream);
int gnnome_bn_relu_res_x16(const uint16_t* x, const float* scale, const float* shift, const float* res, int64_t rows, int hidden,
                           float* out, void* stream);
int gnnome_agg_edge_bwd_stats_x16(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf, const float* Tb,
                                  const float* Ub, const float* A2h, const float* A3h, int ld_node, const int32_t* srt_src,
                                  const int32_t* srt_dst, float* de, const uint16_t* xe, const float* scale, const float* shift,
                                  const float* mean, float* s1, float* s2, void* workspace, size_t workspace_bytes, void* stream);
int gnnome_agg_bwd_fused_x16(const float* e, int64_t num_nodes, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                             const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                             const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, float* de,
                             const uint16_t* xe, const float* scale, const float* shift, const float* mean, float* sum_in, float* sum_out,
                             float* s1, float* s2, unsigned* amax_bits, void* workspace, size_t workspace_bytes, void* stream);
int gnnome_bn_bwd_dgrad_x16(float* C, const uint16_t* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift, const float* a,
                            const float* c1, const float* c2, const float* mean, const float* rstd, const float* W, int ldw,
                            uint16_t* dxe, void* stream);
/* hidden = 256 (round 4): the out-of-place form, gnnome_bn_bwd_dgrad_out_f32 with X and dxe as bf16; gnnome_edge_gate_raw_stats_x16 and
 * gnnome_wgrad_x16 take hidden / Ka = 256 as well (the raw gate on the fp16x3 kernel: not under gnnome_set_tuning(10, 1)). */
int gnnome_bn_bwd_dgrad_out_x16(const float* C_in, float* C_out, const uint16_t* X, int64_t rows, int64_t rows_once, int hidden,
                                const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                const float* mean, const float* rstd, const float* W, int ldw, uint16_t* dxe, void* stream);
int gnnome_segment_sum2_x16(const uint16_t* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                            int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, void* stream);
int gnnome_wgrad_x16(const uint16_t* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C, int ldc,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- clusters for mini-batch training (round 5) ----------------------------------------------
 * train.py:333-346 calls dgl.metis_partition(g.long(), num_clusters, extra_cached_hops=1): METIS 5.1.0's multilevel k-way partitioner
 * behind DGL 0.8.1 (requirements.txt:23; neither is in the reference checkout).  gnnome_amd/partition.py runs the published scheme
 * (Karypis & Kumar 1998: heavy-edge matching, a greedy-growing initial partition of the coarsest graph, greedy k-way boundary refinement
 * while uncoarsening) on the device; these are its two inner loops over an undirected weighted CSR (ptr int32[n+1], adj / wgt int32[nnz],
 * both directions stored, no self loops needed), deterministic and atomic-free:
 *   gnnome_hem_propose: proposal[v] = the heaviest unmatched neighbour u of an unmatched v (match[.] < 0) with vwgt[v] + vwgt[u] <=
 *     max_vwgt, ties to the lighter vertex, then the smaller id; -1 if none.  The caller pairs mutual proposals (handshake).
 *   gnnome_kway_gains: best_part[v] = the part, other than label[v], that v's neighbours connect it to most heavily (ties: smaller id;
 *     -1 for an interior vertex) and gain[v] = that connectivity - the connectivity to its own part. */
int gnnome_hem_propose(const int32_t* ptr, const int32_t* adj, const int32_t* wgt, const int32_t* vwgt, const int32_t* match,
                       int64_t num_vertices, int max_vwgt, int32_t* proposal, void* stream);
int gnnome_kway_gains(const int32_t* ptr, const int32_t* adj, const int32_t* wgt, const int32_t* label, int64_t num_vertices,
                      int32_t* best_part, int32_t* gain, void* stream);
/* The coarsest level's initial partition (METIS's greedy graph growing), on the HOST: all pointers are host memory, nothing is launched.
 * k - 1 regions grown one after the other from the free vertex with the smallest id, always taking the free vertex most heavily connected to the
 * region (ties: the one that became its neighbour first), until the region holds (remaining weight) / (remaining parts); the rest is the last
 * part.  label_host[v] in [0, num_parts).  (train.py:333: dgl.metis_partition -> METIS 5.1.0 InitKWayPartitioning; see csrc/partition.hip.) */
int gnnome_greedy_growing_host(const int32_t* ptr_host, const int32_t* adj_host, const int32_t* wgt_host, const int32_t* vwgt_host,
                               int64_t num_vertices, int num_parts, int32_t* label_host);

#ifdef __cplusplus
}
#endif
#endif /* GNNOME_HIP_H */
