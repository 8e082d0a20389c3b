/*
 * relpose_gnn_hip.h -- C ABI of the MI355X (gfx950) hot path of relpose-gnn.
 *
 * The reference has no FFI of its own: its "plugin API" for this path is the PyTorch
 * nn.Module contract of PoseNetX_R2 (/root/reference/python/niantic/modules/posenet.py:920-1091)
 * and all native arithmetic lives in third-party wheels (torchvision / aten conv+BN+ReLU,
 * torch_geometric MessagePassing.propagate, torch_scatter scatter-mean).  Each entry point below
 * names the reference interface it replaces.  The host-side mirror of the nn.Module contract
 * (relpose-gnn_amd/posenet.py) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned, 16-byte aligned memory (except where
 *     marked HOST); no call synchronises the device (except rpg_timing_read* and
 *     rpg_release_scratch);
 *   - the composite forwards (rpg_resnet_forward_*, rpg_gnn_forward_*) allocate nothing: all their
 *     memory, including the partial-tile scratch of the split-K / stream-K launches, is the
 *     caller's workspace (rpg_*_workspace_bytes), so they can be captured into a HIP graph as is.
 *     The fine-grained convolution / Linear entry points have no workspace argument: when a shape
 *     calls for a split-K pass they use a library-owned scratch block per (device, stream) that
 *     only grows (old blocks stay valid; never grown while the stream is being captured -- the
 *     launch then simply does not split); rpg_release_scratch() frees the pool;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - tensors are dense fp32, indices are int64 (the dtype of PyG's edge_index);
 *   - return value: 0 = RPG_OK, negative = error, never throws across the ABI.
 *   - activations inside the encoder are NHWC ("channels last"): x[n][h][w][c].
 */
#ifndef RELPOSE_GNN_HIP_H
#define RELPOSE_GNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPG_OK 0
#define RPG_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, unsupported shape/alignment */
#define RPG_ERR_LAUNCH (-2)       /* hipLaunch / runtime error (hipGetLastError text via rpg_last_error) */
#define RPG_ERR_WORKSPACE (-3)    /* workspace_bytes smaller than rpg_*_workspace_bytes() */

#define RPG_ABI_VERSION 1

int rpg_abi_version(void);
/* HOST: static string describing the last RPG_ERR_LAUNCH on this thread ("" if none). */
const char* rpg_last_error(void);
/* Synchronises the device and frees the library-owned split-K scratch pool (see conventions). */
int rpg_release_scratch(void);

/* ------------------------------------------------------------------------------------------- */
/* Encoder primitives (replace aten conv2d + batch_norm + relu + max_pool2d + adaptive_avg_pool */
/* as reached from torchvision resnet34, call site posenet.py:1037)                              */
/* ------------------------------------------------------------------------------------------- */

/* data.x.view(N,3,H,-1) (posenet.py:1035, NCHW) -> NHWC with the channel dim padded 3 -> 4 (zero). */
int rpg_nchw3_to_nhwc4_f32(const float* x_nchw, float* y_nhwc4, int n, int h, int w, void* stream);

/* y = act( conv(x, w) * scale[c] + shift[c] (+ residual) ),  implicit GEMM on f32 MFMA.
 *   x        [n][h][w][cin]           NHWC, cin % 4 == 0
 *   w_ohwi   [cout][kh][kw][cin]      (PyTorch OIHW weight permuted once at load time)
 *   scale/shift [cout]                folded eval-mode BatchNorm: scale = gamma/sqrt(var+eps),
 *                                     shift = beta - mean*scale; scale may be NULL (=1)
 *   residual [n][ho][wo][cout] or NULL; relu != 0 applies max(.,0) last
 *   y        [n][ho][wo][cout],  ho = (h + 2*pad - kh)/stride + 1 (floor), same for wo        */
int rpg_conv2d_bn_act_nhwc_f32(const float* x, const float* w_ohwi, const float* scale, const float* shift,
                               const float* residual, float* y, int n, int h, int w, int cin, int cout,
                               int kh, int kw, int stride, int pad, int relu, void* stream);

/* The same op for the 3x3 / stride 1 / pad 1 case as a 1-D Winograd F(4,3) along the width: half the f32 MFMA work.
 *   u  [6][cout][3][cin]   transformed weights from rpg_wino43_transform_weights_f32 (once per weight load);
 *      rpg_wino43_weights_floats(cout, cin) floats = 18 * cout * cin (42 * cout * cin in a probe build that carries the nested
 *      F(4x2, 3x3) experiment of round 4, tools/probes/winograd2d.hip: its image [24][cout][cin] follows the 1-D one)
 *   cin % 4 == 0, cout % 4 == 0; x [n][h][w][cin] -> y [n][h][w][cout]; other arguments as above.            */
size_t rpg_wino43_weights_floats(int cout, int cin);
int rpg_wino43_transform_weights_f32(const float* w_ohwi /* [cout][3][3][cin] */, float* u, int cout, int cin,
                                     void* stream);
int rpg_conv3x3_wino43_bn_act_nhwc_f32(const float* x, const float* u, const float* scale, const float* shift,
                                       const float* residual, float* y, int n, int h, int w, int cin, int cout,
                                       int relu, void* stream);

/* The whole ResNet stem in one kernel: conv1 (7x7, stride 2, pad 3, 3 -> 64 channels, no bias) + bn1 (eval) + ReLU +
 * MaxPool2d(3, stride 2, padding 1) of torchvision's resnet34 (call site posenet.py:1037), straight from the NCHW input
 * of posenet.py:1035 to the pooled NHWC tensor:  x_nchw [n][3][h][w] -> y [n][hp][wp][64],
 * hc = (h - 1) / 2 + 1, hp = (hc - 1) / 2 + 1 (same for w).
 *   wpack [74][2][64]  the weight operands of the kernel's 74 tap pairs with the BatchNorm scale folded in:
 *                      wpack[kp][nf][l] = scale[ch] * W[ch][tap], ch = 32 nf + (l & 31), tap = tap_a[kp] for l < 32 and
 *                      tap_b[kp] (0 if tap_b[kp].c == -1) for l >= 32, with the table of rpg_stem_pair_table
 *                      (relpose-gnn_amd/params.py pack_stem_pairs builds it)
 *   shift [64]         folded BatchNorm shift
 * rpg_stem_pair_table: HOST arrays tap_a[74][3], tap_b[74][3] = (c, kh, kw) of the two taps of every pair.            */
int rpg_stem_conv7x7s2_bn_relu_maxpool_f32(const float* x_nchw, const float* wpack, const float* shift, float* y_nhwc,
                                           int n, int h, int w, void* stream);
int rpg_stem_pair_table(int* tap_a, int* tap_b);

/* nn.MaxPool2d(3, stride 2, padding 1), NHWC, c % 4 == 0. */
int rpg_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n, int h, int w, int c, void* stream);

/* nn.AdaptiveAvgPool2d(1) + flatten: [n][hw][c] -> [n][c]. */
int rpg_global_avgpool_nhwc_f32(const float* x, float* y, int n, int hw, int c, void* stream);

/* Whole encoder: torchvision-0.9.1 ResNet (BasicBlock) forward incl. the replaced fc
 * (posenet.py:942-945, :1037).  `tensors` is a HOST array of device pointers laid out as
 * documented in relpose-gnn_amd/params.py (stem, then per block conv1/conv2/(downsample), each
 * as {w_ohwi, scale, shift, u_wino43}, then fc weight [feat][512] and bias).  u_wino43 may be NULL
 * (always for strided and 1x1 convolutions): that convolution then takes the direct implicit-GEMM
 * kernel.  The stem's fourth slot holds the wpack of rpg_stem_conv7x7s2_bn_relu_maxpool_f32 (planes[0]
 * == 64) or NULL (= re-layout + generic convolution + max-pool kernels).  `blocks[4]`/`planes[4]` HOST.
 * x_nchw [n][3][h][w] -> feat [n][feat_dim].                                                     */
size_t rpg_resnet_workspace_bytes(int n, int h, int w, const int* planes);
int rpg_resnet_forward_f32(const float* const* tensors, int n_tensors, const int* blocks, const int* planes,
                           int feat_dim, const float* x_nchw, int n, int h, int w, float* feat,
                           void* workspace, size_t workspace_bytes, void* stream);

/* bf16 encoder (BASELINE.json configs[2]: bf16 activations + bf16 MFMA convolutions, fp32 accumulate / BN epilogue).
 *   rpg_conv2d_bn_act_nhwc_bf16: x, w_ohwi, residual, y are bf16 (y is fp32 when out_f32 != 0); scale/shift fp32;
 *                                cin % 8 == 0, cout % 4 == 0; other arguments as rpg_conv2d_bn_act_nhwc_f32.
 *   rpg_resnet_forward_bf16:     tensors = per conv {w_ohwi bf16 (stem Cin padded 3 -> 8), scale f32, shift f32}, then
 *                                fc weight bf16 [feat][512], fc bias f32, then OPTIONALLY the wpack of the fused bf16 stem
 *                                below (64-channel stems; without it: re-layout + generic convolution + max-pool kernels);
 *                                x_nchw fp32 -> feat fp32 [n][feat_dim].
 *   rpg_stem_conv7x7s2_bn_relu_maxpool_bf16: the stem of the bf16 encoder in one kernel (same op chain and shapes as
 *                                rpg_stem_conv7x7s2_bn_relu_maxpool_f32; torchvision conv1 / bn1 / relu / maxpool reached from
 *                                posenet.py:1037): x_nchw fp32 [n][3][h][w] -> y bf16 [n][hp][wp][64]; inputs and weights
 *                                rounded to bf16, fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 scale / shift / ReLU /
 *                                max, one bf16 rounding at the store.
 *                                wpack_bf16: two operand images back to back (23,552 bf16).  [11][2][64][8]: element
 *                                [s][nf][l][j] = W[ch = 32 nf + (l & 31)][c][kh][kw = j] with (c, kh) = divmod(2 s + (l >> 5), 7);
 *                                zero for j == 7 and for the 22nd (c, kh) row (the tile kernel of rounds 3-5,
 *                                RPG_TUNE_FUSED_STEM = 3).  Then [2][3][4][64][8]: element [nf][c][j][l][t] =
 *                                W[ch = 32 nf + (l & 31)][c][kh = 2 j + (l >> 5)][kw = t - 1]; zero for t == 0 and kh == 7 (the
 *                                strip-march kernel of round 6, the default).  relpose-gnn_amd/params.py pack_stem_bf16
 *                                builds both.                                                                             */
int rpg_stem_conv7x7s2_bn_relu_maxpool_bf16(const float* x_nchw, const void* wpack_bf16, const float* scale, const float* shift,
                                            void* y_nhwc_bf16, int n, int h, int w, void* stream);
/* ... and on node images that are ALREADY bf16 [n][3][h][w] (the reference's fp32 `data.x`, posenet.py:1034-1035, rounded to bf16
 * on the host so that the host-to-device copy is half the size): the kernel above rounds its fp32 input to bf16 first thing, so the
 * result is bit-identical. */
int rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16(const void* x_nchw_bf16, const void* wpack_bf16, const float* scale, const float* shift,
                                                  void* y_nhwc_bf16, int n, int h, int w, void* stream);
int rpg_conv2d_bn_act_nhwc_bf16(const void* x, const void* w_ohwi, const float* scale, const float* shift,
                                const void* residual, void* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                                int stride, int pad, int relu, int out_f32, void* stream);
size_t rpg_resnet_bf16_workspace_bytes(int n, int h, int w, const int* planes);
int rpg_resnet_forward_bf16(const void* const* tensors, int n_tensors, const int* blocks, const int* planes, int feat_dim,
                            const float* x_nchw, int n, int h, int w, float* feat, void* workspace,
                            size_t workspace_bytes, void* stream);
/* the same forward on bf16 node images (see rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16).  Works with either stem: the fused
 * kernel (wpack as the optional last tensor, RPG_TUNE_FUSED_STEM on) or the three-kernel stem, whose re-layout pass reads the
 * bf16 pixels as they are (round 4; it used to return RPG_ERR_BAD_ARG, so a tuning knob could break a caller that stages bf16).
 * Non-finite pixels: the fused stem's padded K slots (8th kernel column, 22nd (channel, row) pair) multiply a ZERO weight with a
 * real neighbouring pixel, so an Inf / NaN pixel just outside a 7x7 window yields NaN where the reference's conv2d stays finite;
 * finite inputs (every image) are unaffected. */
int rpg_resnet_forward_bf16_xbf16(const void* const* tensors, int n_tensors, const int* blocks, const int* planes, int feat_dim,
                                  const void* x_nchw_bf16, int n, int h, int w, float* feat, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* GNN primitives                                                                                */
/* ------------------------------------------------------------------------------------------- */

/* Index preparation for one edge list given as its two rows src[e] (message source) and dst[e] (target) -- for a
 * contiguous edge_index [2][E] pass edge_index and edge_index + E; for a column slice [e0, e1) of a batched edge_index
 * pass edge_index + e0 and edge_index + E + e0 with node_offset = first node id of the slice (ids are stored
 * relative to the whole batch, the kernels work on ids - node_offset in [0, n)):
 *   ends [4][e]    int64: sanitised source, sanitised target, min(s,t), max(s,t)
 *                  (the last two are compute_edge_features' gather indices, posenet.py:1014-1017)
 *   rowptr [n+1], perm [e]   CSR of edges grouped by TARGET node, edge ids ascending inside a
 *                  group (the order torch_scatter's CPU kernel accumulates in)
 *   cursor [n]     scratch
 *   status [1]     += number of edges with an endpoint outside [0, n) (the caller zeroes it; a
 *                  counter that several calls share accumulates); such edges are left out of the
 *                  CSR and their endpoints clamped, so no later kernel reads out of bounds.  The
 *                  host mirror raises IndexError when it reads a non-zero count back.
 * Single workgroup; e and n up to 2^20.                                                          */
int rpg_graph_prepare(const int64_t* src, const int64_t* dst, int64_t node_offset, int e, int n, int64_t* ends,
                      int32_t* rowptr, int32_t* cursor, int32_t* perm, int32_t* status, void* stream);

/* torch_cluster.knn_graph(x, k, batch, loop=False, flow='source_to_target') (posenet.py:1043-1050): for every node
 * the k nearest OTHER nodes of its graph by squared Euclidean distance (k+1 nearest including itself by
 * (distance, index), self match dropped).  x [n][d]; batch [n] int64 graph id per node, nodes of a graph contiguous,
 * NULL = one graph.  edge_index: capacity [2][n*(k+1)], row stride n*(k+1); the first *total columns are valid:
 * row 0 = neighbour (source), row 1 = query node (target), grouped by target in node order, nearest first.
 * cand [n][k+1], cnt [n] scratch; total [1]; status [1] += graphs larger than 2048 nodes (unsupported).
 * k <= 64.                                                                                                   */
int rpg_knn_graph_f32(const float* x, const int64_t* batch, int n, int d, int k, int64_t* edge_index,
                      int32_t* cand, int32_t* cnt, int32_t* total, int32_t* status, void* stream);

/* compute_edge_features (posenet.py:999-1019): out[e] = [x[min(s,t)], x[max(s,t)]], [e][2d]. */
int rpg_edge_concat_gather_f32(const float* x, const int64_t* edge_index, int e, int d, float* out, void* stream);

/* out[m][n_out] = act( cat_k( a_k[idx_k[m]] ) @ W^T + bias ) (+ residual): nn.Linear over a row-wise
 * concatenation of up to three gathered sources that is never materialised
 * (my_gnn_layer.py:238, :305, :310; posenet.py:1053-1055).  idx_k == NULL means row m itself.
 *   a_k [rows_k][ld_k] uses its first width_k columns, width_k % 4 == 0, ld_k % 4 == 0
 *   W   [n_out][sum width_k] (PyTorch Linear layout), bias [n_out] or NULL
 *   residual [m][n_out] or NULL (added before the activation)                                   */
int rpg_linear_gather_f32(int n_src, const float* const* a, const int64_t* const* idx, const int* ld,
                          const int* width, const float* weight, const float* bias, const float* residual,
                          float* out, int m, int n_out, int relu, void* stream);

/* The same Linear with the epilogue the composite GNN forward uses for the split formulation of the concatenated-input
 * Linears (W [x_a, x_b, e] = W_a x_a + W_b x_b + W_e e, node terms computed once per node; my_gnn_layer.py:236-239,304-311
 * -- the reference materialises the torch.cat): the residual rows may be GATHERED,
 *   out[r] = act( cat_k(a_k[idx_k[r]]) W^T + bias + residual[(res_idx ? res_idx[r] : r) * ldr + :]
 *                                                 + residual2[res2_idx[r] * ldr + :] ),
 * and out_relu (or NULL) receives max(out, 0) as a second tensor of the same shape (the caller-side ReLU of
 * posenet.py:1064-1065 next to the un-activated value).  rows[k]: row count of a gathered source k (its index range; 0 /
 * NULL for plain sources); ldr >= n_out: row pitch of residual / residual2 (floats); res_idx == NULL and residual2 == NULL
 * = rpg_linear_gather_f32 (ldr 0 or n_out); residual2 needs res_idx AND res2_idx.  RPG_ERR_BAD_ARG: residual2 without both
 * index arrays, index arrays without residual, ldr < n_out, mis-aligned out_relu.                                        */
int rpg_linear_gather_ex_f32(int n_src, const float* const* a, const int64_t* const* idx, const int* ld,
                             const int* width, const long* rows, const float* weight, const float* bias,
                             const float* residual, const int64_t* res_idx, const float* residual2,
                             const int64_t* res2_idx, int ldr, float* out, float* out_relu, int m, int n_out, int relu,
                             void* stream);

/* AttentionBlock core (att.py:20-31) on rows: gtp [r][3c] = [g | theta | phi] projections,
 * y[r][i] = sum_j softmax_j(phi_i * theta_j) * g_j.                                            */
int rpg_attention_rows_f32(const float* gtp, int r, int c, float* y, void* stream);

/* torch_scatter.scatter(msg, target, dim=0, dim_size=n, reduce='mean') through the CSR of
 * rpg_graph_prepare: out[v] = sum_{p in rowptr[v]..rowptr[v+1]} msg[perm[p]] / max(count,1).
 * msg [e][d], d % 4 == 0.  (my_gnn_layer.py:279,301)                                                        */
int rpg_scatter_mean_f32(const float* msg, const int32_t* rowptr, const int32_t* perm, int n, int e, int d,
                         float* out, void* stream);

/* AttentionBlock core + mean aggregation fused (att.py:20-31, my_gnn_layer.py:279,301,304-307).  att = W y + b + msg is
 * linear in (y, msg), so mean_e(att) = W mean_e(y) + (b + mean_e(msg)): this entry point produces the two means per
 * target node through the CSR of rpg_graph_prepare, in ascending edge order,
 *   ybar[v][i] = mean_{e -> v} y_e[i]   (y_e = the attention row of gtp[e], as rpg_attention_rows_f32)       [n][c]
 *   mbar[v][:] = mean_{e -> v} msg[e][:] (+ bias[:] if the node has an incoming edge; bias may be NULL)       [n][d]
 * (mbar without bias is bit-exact rpg_scatter_mean_f32 of msg; nodes without incoming edges give zeros) and the composite
 * forward then runs att.W on n rows: agg = ybar W^T + mbar.  gtp [e][3c], msg [e][d]; c % 4 == 0, d % 4 == 0,
 * d <= 1024 * ceil(c / 64).                                                                                   */
int rpg_attention_aggregate_f32(const float* gtp, const float* msg, const int32_t* rowptr, const int32_t* perm,
                                const float* bias, int n, int e, int c, int d, float* ybar, float* mbar, void* stream);

/* Two 3-output Linear heads on the same rows, concatenated: out[r][0:3] = x W1^T + b1,
 * out[r][3:6] = x W2^T + b2 (posenet.py:1077-1091).  w6 [6][d] = cat(W1, W2), b6 [6].           */
int rpg_pose_heads_f32(const float* x, const float* w6, const float* b6, int r, int d, float* out, void* stream);

/* Everything after the encoder for use_gnn=True, use_AP=True, knn<=0 (posenet.py:1052-1091).
 * `tensors` HOST array of device pointers, order in params.py: 22 tensors (reference formulation) or 26 (adds the
 * node/edge column blocks of proj_edge, edge_mlp.0 and mlp.0 as separate matrices, which lets the node terms be
 * computed once per node and gathered in the edge GEMM epilogues).
 * feat [n][d], edge rows src[e] / dst[e] with node_offset (see rpg_graph_prepare) -> abs_pose [n][6], rel_pose [e][6].
 * node_out [n][d] / edge_out [e][d]: optional (NULL to skip) copies of the final ReLU'd node and
 * edge features, i.e. the inputs of the heads; the host mirror uses them to apply the reference's
 * always-on F.dropout (posenet.py:1073-1075) before calling rpg_pose_heads_f32 itself.
 * status: device int32, see rpg_graph_prepare.                                                  */
size_t rpg_gnn_workspace_bytes(int n, int e, int d);
int rpg_gnn_forward_f32(const float* const* tensors, int n_tensors, const float* feat, const int64_t* src,
                        const int64_t* dst, int64_t node_offset, int n, int e, int d, int gnn_recursion,
                        float* abs_pose, float* rel_pose,
                        float* node_out, float* edge_out, int32_t* status, void* workspace,
                        size_t workspace_bytes, void* stream);

/* The same forward with the Linears on the bf16 matrix pipe (BASELINE configs[2]/[4] "bf16 activations"): a Linear's
 * input is rounded to bf16 on the fly, its weight is read from weights_bf16, accumulation / bias / residual / output
 * stay fp32, and everything that is not a GEMM (attention rows, scatter-mean, heads) is the fp32 kernel of above.
 * Needs the 26-tensor (split) table and d % 64 == 0.  weights_bf16: HOST array of 10 device pointers to bf16 [out][in]
 * matrices, in this order (params.pack_gnn_bf16): tensors[22] (proj node blocks), tensors[23] (edge_mlp.0 | mlp.0 node
 * blocks), tensors[24] (edge_mlp.0 edge block), edge_mlp.2, tensors[25] (mlp.0 edge block), mlp.2, att.{g|theta|phi},
 * att.W, mlp_updating.0, mlp_updating.2.  Not part of the fp32 parity claim: tolerance in tests/test_hip_bf16.py.   */
int rpg_gnn_forward_bf16(const float* const* tensors, int n_tensors, const void* const* weights_bf16, int n_bf16,
                         const float* feat, const int64_t* src, const int64_t* dst, int64_t node_offset, int n, int e, int d,
                         int gnn_recursion, float* abs_pose, float* rel_pose, float* node_out, float* edge_out,
                         int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* One 64-channel identity BasicBlock of the bf16 encoder as a single kernel (round 5): y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x),
 * both convolutions 3x3 / stride 1 / pad 1, 64 -> 64 channels, x / y bf16 NHWC [n][h][w][64] (y must not alias x), weights bf16
 * [64][3][3][64] (OHWI), folded BatchNorm scale / shift fp32 [64] (16-byte aligned).  Replaces the two aten conv2d + batch_norm +
 * relu (+ add) sequences of torchvision's BasicBlock.forward (reference call site: modules/posenet.py:1037) for ResNet34's
 * layer 1.  The intermediate activation is rounded to bf16 exactly where the two-launch path stores it: outputs are bit-identical
 * to rpg_conv2d_bn_act_nhwc_bf16 called twice.  Maps up to 62 pixels wide as they are, up to 120 (round 6: the 64 x 86 maps of the
 * 256 x 341 evaluation shape, dataset_7Scenes_multi.py:341,434) as two column strips per image whose two columns next to the cut are
 * recomputed; RPG_ERR_BAD_ARG for shapes it does not take (w < 4, w > 120, a strip's activations beyond 2^31 bytes: the caller then
 * uses two convolution calls).  No workspace, no allocation, no synchronisation. */
int rpg_basicblock64_bf16(const void* x, const void* w1_ohwi, const float* scale1, const float* shift1, const void* w2_ohwi,
                          const float* scale2, const float* shift2, void* y, int n, int h, int w, void* stream);

/* Building blocks of the above.  rpg_f32_to_bf16: dst[r][col_off + c] = bf16(src[r][c]) for c < cols (cols, col_off,
 * ld_dst % 8 == 0).  rpg_linear_bf16: out[m][n_out] (fp32) = act(a[m][k] (bf16) * weight[n_out][k]^T (bf16) + bias +
 * residual[(res_idx ? res_idx[r] : r) * ldr + :] + residual2[res2_idx[r] * ldr + :]); k % 8 == 0, n_out % 4 == 0.
 * Replaces nn.Linear (+ the gathered node terms of the split formulation), my_gnn_layer.py:236-239,304-311.         */
int rpg_f32_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, int col_off, long rows, int cols, void* stream);
int rpg_linear_bf16(const void* a, const void* weight, const float* bias, const float* residual, const int64_t* res_idx,
                    const float* residual2, const int64_t* res2_idx, int ldr, float* out, int m, int k, int n_out,
                    int relu, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline).  */
/* ------------------------------------------------------------------------------------------- */
#define RPG_TIMER_CONV 0          /* direct implicit-GEMM convolutions (stem, strided 3x3, 1x1)     */
#define RPG_TIMER_LINEAR 1        /* gathered Linear GEMMs                                         */
#define RPG_TIMER_SCATTER 2       /* scatter-mean                                                  */
#define RPG_TIMER_ATTENTION 3     /* attention rows                                                */
#define RPG_TIMER_CONV_WINO 4     /* Winograd F(4,3) 3x3/stride-1 convolutions (work = direct-conv FLOP) */
#define RPG_TIMER_ATT_AGG 5       /* fused attention rows + mean aggregation (work = algorithmic bytes)   */
#define RPG_TIMER_COUNT 6
/* enable != 0: every launch of the listed kernel classes is bracketed by hipEventRecord. */
int rpg_timing_enable(int enable);
/* Synchronises, sums the elapsed time of all bracketed launches since the last read.
 * HOST outputs, arrays of RPG_TIMER_COUNT: total milliseconds, launches, algorithmic work
 * (FLOP for CONV/LINEAR/ATTENTION, bytes for SCATTER).                                          */
int rpg_timing_read(double* ms, long long* launches, double* work);
/* The same plus executed[k]: the FLOP the matrix pipe really issues for those launches (whole padded
 * tiles; for the Winograd kernels workgroups x K steps x 48 MFMAs x waves x 4096, i.e. about half of
 * the algorithmic direct-convolution FLOP).  executed may be NULL.                               */
int rpg_timing_read_ex(double* ms, long long* launches, double* work, double* executed);

/* Measurement aid for the bf16 roofline (no counterpart in the reference, which has no bf16 path and no
 * benchmark harness -- /root/reference/python/niantic/testing/test.py:192-211 is the only caller): one launch of `workgroups` x 8
 * waves, each issuing iters x 16 v_mfma_f32_32x32x16_bf16 on registers only (no LDS / L2 / HBM traffic), operands taken from
 * `operands` (device, 65,536 x 16 bytes of bf16 values chosen by the caller: the sustained clock of the matrix pipe depends on
 * them -- on MI355X 2.50 PFLOP/s on zeros, 1.8 on random data, 2.1 on ReLU-like data).  The caller times the launch on `stream`;
 * FLOP = workgroups * 8 * iters * 16 * 32768.  `sink`: one float of device memory (never written in practice).
 * RPG_ERR_BAD_ARG: null / misaligned operands, iters outside 1 .. 2^24, workgroups outside 1 .. 65536. */
int rpg_probe_mfma_bf16(const void* operands, long iters, int workgroups, float* sink, void* stream);

/* Tuning knobs of the f32 MFMA tile engine (benchmarking aid; defaults are the tuned choice).
 *   RPG_TUNE_TILE      -1 automatic (default) | 0: 128x128 | 1: 256x64 | 2: 64x64 | 3: 128x64 workgroup tile
 *   RPG_TUNE_BK        K-step: 0 automatic (default: 32 for the 128x128 tile, else 16) | 16 | 32
 *   RPG_TUNE_EPILOGUE  1: LDS-transposed 16-byte epilogue (default) | 0: direct 4-byte epilogue
 *   RPG_TUNE_STREAMK   1: stream-K pass for the tiles that do not fill a round of resident workgroups
 *                      (default; partial tiles go to the caller's workspace / the scratch pool) | 0: off  */
#define RPG_TUNE_TILE 0
#define RPG_TUNE_BK 1
#define RPG_TUNE_EPILOGUE 2
#define RPG_TUNE_STREAMK 3
#define RPG_TUNE_GNN_SPLIT 5      /* 1: per-node precompute of the split concatenated-input Linears (default, needs
                                     the 26-tensor table) | 0: reference formulation (gathered 3-source GEMMs) */
#define RPG_TUNE_BF16_BK 6        /* K step of the bf16 convolution kernel: 32 (default) | 64 */
#define RPG_TUNE_BF16_FAST 9      /* 1: interleaved buffer-load bf16 conv kernel where Cin % 64 == 0 (default) | 0: general kernel */
#define RPG_TUNE_GNN_FUSE_AGG 13  /* 1: attention rows + mean aggregation in one kernel, att.W on node rows (default) | 0: per-edge
                                     attention rows, att.W on edge rows, separate scatter-mean (the reference's order) */
#define RPG_TUNE_WAVES8 11        /* 1: 8-wave workgroups (two waves per SIMD) for the 128x128 / 128x64 tiles of the f32 tile engine where
                                     the buffer-load path applies (default) | 0: always 4-wave workgroups */
#define RPG_TUNE_FUSED_STEM 10    /* bit 0: one-kernel stem (conv7x7 + BN + ReLU + max-pool) where its operands are given (default 1) | 0: three kernels.
                                     bf16 stem only: bit 1 = the tile kernel of rounds 3-5 instead of the strip-march kernel of round 6; bit 5 = the
                                     strip-march kernel with one 32-channel half per wave and three waves per SIMD (default: both halves in one wave,
                                     two waves per SIMD; the register-weight and four-wave variants were measured slower and removed); value >> 8 =
                                     pooled rows per band
                                     (0, the default: by the launch's size).  fp32 stem: bit 7 = the strip-march kernel (round 6; opt-in: 6-12 % faster
                                     than the tile kernel, every fp32 bar holds, but it re-rolls the rounding noise of the noise-floor ratio test) */
#define RPG_TUNE_WINO_SPLIT 8     /* 1: split-K tail + fix-up for the 8-wave Winograd kernel (default) | 0: whole tiles only | n >= 2: as 1, and a part of a
                                     tile gets at least n K steps in the one-workgroup-per-tile form (default 3: one 8-node graph 1.50 ms per forward, 1.63 with 4) */
#define RPG_TUNE_FAST_LOADER 7    /* 1: buffer-load loaders + interleaved main loop where eligible (default) | 0: general loaders */
#define RPG_TUNE_WINOGRAD 4       /* 0: always the direct kernel | 1: use u_wino43 where given, kernel by size (default) |
                                     2 / 3: as 1 but always the 4-wave single-image / the 8-wave Winograd kernel | n >= 16: as 1 with the
                                     Winograd kernels from n blocks of 64 tiles x 64 channels per layer up (default 16; larger: the small layers of
                                     a small batch go to the direct kernel, which reads half the weight bytes) */
#define RPG_TUNE_WINO_SHORT 12    /* retired in round 3 with the short-K Winograd kernel it selected (measured: no gain); the key is
                                     still accepted (values >= 0) and ignored */
#define RPG_TUNE_WINO_PERSIST 14  /* 1: launches with more 8-wave tiles than CUs run the persistent kernel (one workgroup per CU walks its
                                     tiles, loads pipelined across tiles; needs Cin % 16 == 0) (default) | 2: also launches of at most one
                                     tile per CU (measured equal) | 0: one workgroup per tile */
#define RPG_TUNE_BF16_TILE 15     /* tile of the interleaved bf16 convolution kernel: -1 by shape (default) | 0: 64x64 | 1: 128x128 | 2: 256x64 | 3: 128x64 */
#define RPG_TUNE_BF16_DMA 16      /* the LDS-DMA bf16 convolution kernel (buffer_load ... lds, counted vmcnt, up to 256 x 256 tiles on 8 waves):
                                     0: off | 1: by shape (default) | 10 + i: configuration i wherever eligible (experiments) */
#define RPG_TUNE_BF16_PATCH 17    /* the patch kernel of the bf16 encoder's 3x3 / stride-1 convolutions (input patch resident in LDS, nine taps
                                     read it at shifted slots; Cin % 64 == 0): 0: off | 1: for more than 64 output channels on >= 8192
                                     pixels (default) | 2: on any eligible size | 3: as 2 with the 256 x 128 tile for every width;
                                     + 10: always three weight stages (without: four where the LDS allows -- the next step's weight fragments
                                     are then read before the barrier) */
#define RPG_TUNE_BF16_WS64 18     /* probe builds only (-DRPG_PROBE_WS64, tools/probes/conv3x3_bf16_ws64.inc: the weights-stationary layer-1
                                     experiment of round 3, correct and not faster): 1 / 2 select it.  The product library accepts 0 and
                                     returns RPG_ERR_BAD_ARG for anything else */
/* HOST: fp32 -> bf16 (round to nearest even, NaN -> quiet NaN: what the device's conversion and torch's .bfloat16() do) of a
 * contiguous host array; thread-safe, called by evaluate_stream's staging threads for the bf16 encoder's node images. */
int rpg_host_f32_to_bf16(const float* src, void* dst_bf16, size_t count);
/* HOST: the same with the instruction set named (round 6: the conversion runs on 8 / 16 lanes where the CPU has AVX2 / AVX-512F,
 * chosen once at run time by rpg_host_f32_to_bf16; every variant is the same integer arithmetic, bit-identical).  isa: 0 = what
 * rpg_host_f32_to_bf16 uses on this CPU | 1 = scalar | 2 = AVX2 | 3 = AVX-512F.  RPG_ERR_BAD_ARG for an unknown index or an
 * instruction set this CPU lacks (tests compare every available variant with the scalar one). */
int rpg_host_f32_to_bf16_isa(const float* src, void* dst_bf16, size_t count, int isa);

#define RPG_TUNE_SK_MIN_ITS 19    /* least K steps a stream-K workgroup of the fp32 GEMM engine gets (default 8) */
#define RPG_TUNE_INKERNEL_FIXUP 20  /* experiment of round 4, OFF by default (measured slower on MI355X): the partial tiles of a
                                     split launch are combined by the LAST-ARRIVING workgroup of each tile inside the producing kernel (per-tile
                                     arrival counter in the scratch block, agent-scope slab stores / loads, slabs summed in ascending k order:
                                     same bits whoever comes last) instead of by a separate fix-up launch.  Bit 0: stream-K tiles of the fp32 GEMM
                                     engine; bit 1: Winograd tail tiles of up to 8 parts (wino43_conv8_kernel only); + 4: up to 32 parts */
#define RPG_TUNE_BF16_CHUNK 21     /* bf16 encoder: runs of identity blocks on activation tensors of >= M MB are walked depth-first, `value & 4095`
                                     images at a time (their intermediates then come back from the Infinity Cache instead of HBM); M = value >> 12
                                     (0: 64 MB); 0 = off */
#define RPG_TUNE_WINO2D 22         /* probe builds only (-DRPG_PROBE_WINO2D: the nested 2-D Winograd F(4x2, 3x3) kernel, 3 instead of 4.5 multiplies per
                                     output -- correct and 13-34 % slower, profiles/r4_wino2d_nested_kernel.txt): 1 by shape | 2 wherever
                                     eligible.  The product library accepts 0 and returns RPG_ERR_BAD_ARG for anything else */
#define RPG_TUNE_BF16_LEAN_EPI 23   /* bf16 convolutions (bf16 output, optional bf16 residual): 1 (default) the branch-free epilogue of round 4 (raw buffer
                                     accesses with out-of-range offsets instead of a branch per row, 32-bit offsets, residual as a template
                                     parameter: -7..-12 % on the convolution kernels) | 0: the general epilogue of rounds 1-3 */
#define RPG_TUNE_BF16_LINEAR_DMA 24 /* bf16 GNN Linears on the edge rows (>= 192 tiles of 128 x 128): 0 the interleaved buffer-load kernel | 10 + i:
                                     configuration i of the LDS-DMA convolution kernel (a Linear is a 1 x 1 convolution over an m-pixel image) */
#define RPG_TUNE_BF16_PERSIST 25    /* bf16 3x3 / stride-1 convolutions with <= 64 output channels (layer 1) on more than one round of tiles: 1 = the persistent
                                     form of the patch kernel (one workgroup per CU walks its tiles; the next tile's first patch chunk and weights are
                                     loaded during the current tile's last chunk and epilogue).  The kernel is 10 % (in the model) to 15 % (stand-alone)
                                     faster; the default two-stream step is 2 % SLOWER with it (resident workgroups leave the other stream no gaps), so
                                     the default is 0 = one workgroup per tile.  For single-stream use */
#define RPG_TUNE_FOLD_K 26          /* fp32 Linears (rpg_linear_gather_f32 and the GNN composite): two-level accumulation -- the v_mfma_f32_32x32x2_f32
                                     accumulator is folded into a second one and restarted every `value` of K (a multiple of 64; default 256; 0 = one
                                     sequential chain over all of K as in rounds 1-4).  Summation order only: cuts the rounding noise of the K = 2048..4096
                                     GNN Linears (reference: addmm in my_gnn_layer.py:236-239,304-311 through oneDNN's blocked sums) */
#define RPG_TUNE_BF16_FUSE_BLOCK 27  /* bf16 encoder: 64-channel identity BasicBlocks (ResNet34 layer 1) run as ONE kernel, conv1 + BN + ReLU + conv2 +
                                     BN + identity + ReLU with the intermediate activation in LDS (rpg_basicblock64_bf16; bit-identical to the two
                                     convolution launches): 3 (default) = persistent form (one workgroup per CU walks the tiles, the next tile's
                                     first loads under the current tile's last steps and epilogue; launches of more than one round of tiles) |
                                     1 = one tile per workgroup (rounds 5-6a) | 5 = one tile per workgroup in the two-group schedule (waves 4-7 one
                                     barrier behind waves 0-3: load / compute phases alternate on every SIMD; experiment, not faster) | 0 = two launches */
#define RPG_TUNE_BF16_TAIL 28        /* bf16 3x3 / stride-1 convolutions on more than one round of tiles: the rows beyond the last FULL round of workgroups
                                     go to a second launch of the patch kernel with smaller tiles instead of a mostly empty round of full-size ones
                                     (49 * 2^k pixels: 3.06 / 1.53 rounds at 512 images); same arithmetic per output, bit-identical.  Bit 0 (default
                                     on): 512 x 128 -> 256 x 128 tiles (layer 2: -4..5 %); bit 1 (off): 256 x 256 -> 160 x 256 (layer 3: measured
                                     no gain); bit 2 (off, round 6): layer 2's tail on 128 x 128 tiles instead of 256 x 128 (measured: no gain,
                                     0.350 vs 0.351); 0 = always one launch */
#define RPG_TUNE_LIN112 29           /* fp32 Linears with a plain A operand, M % 112 == 0, N % 64 == 0, K % 32 == 0 and at least one 112 x 64 tile per CU (the
                                     GNN's edge GEMMs: M = 56 edges x graphs = 7 * 2^k rows): bit 0 = the exact-fit kernel on v_mfma_f32_16x16x4_f32
                                     (no stream-K split, no fix-up launch), bit 1 = its eight-wave form (the two k halves of a step on two wave
                                     groups) for launches of about one tile per CU; default 3; 0 = the 32x32x2 tile engine */
#define RPG_TUNE_FIXUP_PRIO 30       /* experiment of round 6, OFF by default: 1 = the fix-up launches of split tiles (Winograd tail, stream-K) run on a
                                     high-priority companion of the launch stream (event hand-off there and back), so that under two concurrent
                                     streams they are dispatched ahead of the other stream's queued convolution workgroups */
#define RPG_TUNE_BF16_PAIR 31        /* bf16 encoder: 1 (default) = the 3x3 / stride-2 convolution and the 1x1 / stride-2 shortcut of a down-sampling BasicBlock
                                     run as ONE launch of the LDS-DMA kernel (their tiles side by side in the grid; same arithmetic) | 0 = two launches */
int rpg_set_tuning(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* RELPOSE_GNN_HIP_H */
