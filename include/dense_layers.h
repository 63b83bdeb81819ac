/*
 * dense_layers.h -- C ABI of the fp32-accurate dense layers on the bf16 matrix cores (libgs_rasterizer_hip.so): the trunk of the node
 * network the dynamic mapping loop trains (utils/time_utils.py:327-476 DeformNetwork: eight layers of width 256 on every (node, time sample)
 * row of an iteration; utils/slam_backend.py:361-371 evaluates it per keyframe).
 *
 * An fp32 value is carried as three bf16 terms (hi + mid + lo = the 24 bits of its significand, by truncation: the remainders are exact) and a
 * product as its six cross terms of weight >= 2^-16, accumulated in fp32 by v_mfma_f32_16x16x32_bf16; what is dropped is below 2e-7 of |x w|
 * per product -- the results are fp32 GEMM results for every test of this repository -- at up to 2.7x the fp32 matrix rate.
 * Non-finite inputs: a +-inf operand yields NaN where an fp32 GEMM yields +-inf (inf - inf in the split).
 *
 * All pointers are DEVICE pointers; fp32 matrices are row-major with explicit row strides (in floats). Returns 0 or a negative GSR_ERR_* code.
 */
#ifndef DENSE_LAYERS_H_INCLUDED
#define DENSE_LAYERS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The split form of a weight matrix W [N, K] (nn.Linear.weight: [out, in]; columns k0 .. k0 + K of a wider matrix with row stride ldw):
 * `planes` receives three zero-padded bf16 planes [round_up(N, 128)][round_up(K, 32)] -- the weight of  Y = X W^T  -- or, with transposed != 0,
 * the planes [round_up(K, 128)][round_up(N, 32)] of W^T -- the weight of the input gradient  dX = G W.  gsr_dense_planes_size(rows, cols) bytes
 * for a weight of `rows` outputs and `cols` inputs. To be redone whenever W changes (once per optimizer step). */
size_t gsr_dense_planes_size(int rows, int cols);
int gsr_dense_split(int N, int K, const float* W, int ldw, int k0, int transposed, void* planes, void* stream);

/* Y [M, N] = act(X [M, K] Wt + bias): planes = the split weight with N outputs and K inputs (gsr_dense_split), bias [N] or NULL, relu != 0:
 * max(., 0). gate [M, K] or NULL: X is read as X * (gate > 0) -- the ReLU mask of the layer whose output `gate` is, so that the input-gradient
 * product consumes the upstream cotangent directly. */
int gsr_dense_forward(int M, int N, int K, const float* X, int ldx, const float* gate, int ldgate, const void* planes, const float* bias, int relu,
                      float* Y, int ldy, void* stream);

/* The input gradient of one layer handed straight to the ReLU of the layer below, with that layer's bias gradient:
 *   dX [M, N] = (G [M, K] W) * (mask > 0),   dbias [N] = column sums of dX
 * for W^T's planes (gsr_dense_split(..., transposed = 1) of the [K, N] weight), mask [M, N] = the OUTPUT of the layer below (post-ReLU; NULL: no
 * mask), dbias NULL: no column sums. Replaces utils/time_utils.py:428-452's autograd chain  mm -> threshold_backward -> sum(0)  per layer: one
 * pass over the rows instead of three. The column sums are added in a fixed order (per lane, per wave, per block, then over the row blocks):
 * deterministic. They need N % 4 == 0 and 16-byte aligned rows of dX / mask. workspace: gsr_dense_backward_input_workspace_size(M, N) bytes. */
size_t gsr_dense_backward_input_workspace_size(int M, int N);
int gsr_dense_backward_input(int M, int N, int K, const float* G, int ldg, const void* planes_t, const float* mask, int ldmask, float* dX, int lddx,
                             float* dbias, char* workspace, void* stream);

/* Several products of 256 output columns in a row on the SAME rows, in ONE launch: op l + 1 reads what op l wrote -- the node network's forward
 * (X = the previous layer's output, relu, bias) or its input-gradient chain (X = the previous op's dX, planes = the layer's transposed planes,
 * mask = the layer below's output, dbias = that layer's bias gradient). Each op is  Y [M, 256] = act(X [M, K] Wt + bias) * (mask > 0)  with
 * dbias [256] = column sums of Y when not NULL. A block of the launch owns its rows through all ops (rows are independent); no operand may be
 * written by a LATER op of the same call. All row pointers 16-byte aligned, ld* % 4 == 0. At most 8 ops. Against one launch per product this
 * saves the launch gap, the wait for every block's output stores, and most of the first operand fetch per boundary.
 * workspace: gsr_dense_chain_workspace_size(M, 256, count) bytes (only read when some op has dbias). */
typedef struct gsr_dense_chain_op {
    const float* X; int32_t ldx; int32_t K;
    const void* planes; const float* bias; int32_t relu;
    float* Y; int32_t ldy;
    const float* mask; int32_t ldmask;
    float* dbias;
} gsr_dense_chain_op;
size_t gsr_dense_chain_workspace_size(int M, int N, int count);
int gsr_dense_chain(int M, int N, int count, const gsr_dense_chain_op* ops, char* workspace, void* stream);

/* Several weights split in ONE launch (a network's layers in both orientations, once per optimizer step): at most 24 items, each with the
 * arguments of gsr_dense_split. */
typedef struct gsr_dense_split_item {
    const float* W;
    void* planes;
    int32_t N, K, ldw, k0, transposed;
} gsr_dense_split_item;
int gsr_dense_split_many(int count, const gsr_dense_split_item* items, void* stream);

/* dW [N, K] = G^T X  for G [M, N] (optionally gated like X above: gate [M, N]) and X [M, K]; the rows are cut into slices whose partial products
 * are added in a fixed order (deterministic). workspace: gsr_dense_wgrad_workspace_size(M, N, K) bytes. */
size_t gsr_dense_wgrad_workspace_size(int M, int N, int K);
int gsr_dense_wgrad(int M, int N, int K, const float* G, int ldg, const float* gate, int ldgate, const float* X, int ldx, float* dW, int lddw,
                    char* workspace, void* stream);

/* Several weight gradients over the SAME M rows in one launch (the backward pass of utils/time_utils.py:428-452's eight layers + heads leaves
 * every layer's G and input in memory): dW_i [N_i, K_i] = G_i^T X_i, at most 12 items. All 128 x 128 result tiles x row slices are resident at
 * once; the slices' partial tiles are added in a fixed order (deterministic). Rows of G / X that are 16-byte aligned with N, K, ldg, ldx % 4 == 0
 * are read as float4, others (the heads' [M, 14] cotangent) element-wise. workspace: gsr_dense_wgrad_many_workspace_size(M, count, items) bytes. */
typedef struct gsr_dense_wgrad_item {
    const float* G;
    const float* X;
    float* dW;
    int32_t ldg, ldx, lddw, N, K;
} gsr_dense_wgrad_item;
size_t gsr_dense_wgrad_many_workspace_size(int M, int count, const gsr_dense_wgrad_item* items);
int gsr_dense_wgrad_many(int M, int count, const gsr_dense_wgrad_item* items, char* workspace, void* stream);

/* ---- the node network's trunk, forward, as ONE launch ------------------------------------------------------------------------------------
 * utils/time_utils.py:428-452 with the shipped structure: eight layers y = relu(x W^T + b) of width 256 on the embedding emb [R, E] (E <= 96),
 * the embedding re-injected behind layer 4 (layer 5's weight is [256, E + 256] and reads [emb | h]), then all heads as one linear layer of
 * n_head_outputs <= 16 columns. A 64-row tile's activations stay in LDS across the layers; weights come from L2 as the planes of
 * gsr_dense_split:
 *   planes[0]      layer 0's weight [256, E]                      planes[1..4]  layers 1..4 [256, 256]
 *   planes[5], [6] layer 5's weight, columns [0, E) and [E, E + 256)            planes[7], [8]  layers 6, 7
 *   planes[9]      the heads' weight [n_head_outputs, 256]
 * planes[0] and planes[5] are read with a FIXED K of 96 (three 32-column steps, plane rows 96 apart): gsr_dense_split pads K to the next
 * multiple of 32 only, so for E <= 64 split a copy of the weight whose columns [E, 96) are zero (K = 96).
 * bias[0..7] the layers' (256 floats, 16-byte aligned), bias[8] the heads'. outs[l] receives layer l's output (post-ReLU, fp32, row stride
 * ldo[l] floats, 16-byte aligned rows: the backward pass reads them; outs[4] may point INTO a [R, E + 256] buffer at column E so that layer
 * 5's input exists as one matrix), heads [R, n_head_outputs]. */
typedef struct gsr_trunk {
    int32_t E;
    int32_t n_head_outputs;
    const void* planes[10];
    const float* bias[9];
} gsr_trunk;
int gsr_trunk_forward(const gsr_trunk* trunk, int R, const float* emb, float* const* outs, const int* ldo, float* heads, void* stream);

#ifdef __cplusplus
}
#endif
#endif
