/*
 * control_nodes.h -- C ABI of the SC-GS control-node warp (libgs_rasterizer_hip.so), SURVEY.md 8(f) rank 3, second half.
 *
 * The SLAM back-end moves its dynamic Gaussians with a sparse set of control nodes (utils/slam_backend.py:361-371 ->
 * gaussian_splatting/scene/deform_model.py:33 -> utils/time_utils.py ControlNodeWarp.forward :1192-1296): every Gaussian finds
 * its K nearest nodes (pytorch3d.ops.knn_points, :998), weights them with a Gaussian RBF of the node radius times a per-node
 * weight, normalised over the K (:1000-1006), and blends the nodes' translation / rotation / scale predictions (:1206-1258).
 * pytorch3d is an un-vendored CUDA dependency of the reference that does not exist on ROCm; this header replaces it on this
 * path and fuses the ~20 gather / elementwise launches around it (and their autograd twins) into one launch per direction.
 *
 * All pointers are DEVICE pointers, fp32 unless stated, contiguous.  Returns 0 or a negative GSR_ERR_* code (gs_rasterizer.h).
 */
#ifndef CONTROL_NODES_H_INCLUDED
#define CONTROL_NODES_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_KNN_MAX_K 32
#define GSR_KNN_MAX_DIM 32
#define GSR_BLEND_MAX_K 8
#define GSR_NODE_RADIUS_IS_LOG 1     /* node_radius holds the raw parameter _node_radius; radius = exp(raw)   (:893-894) */
#define GSR_NODE_WEIGHT_IS_LOGIT 2   /* node_weight holds the raw parameter _node_weight; weight = sigmoid(raw) (:897-898) */

/* pytorch3d.ops.knn_points(p1[None], p2[None], K=K) for one batch element (the only form the reference calls,
 * utils/time_utils.py:998,1028,1099,1109,1183; utils/deform_utils.py:49,74,87): for each of the n rows of p1 [n, D] the K rows of
 * p2 [m, D] with the smallest SQUARED Euclidean distance sum_d (p1 - p2)^2, ascending; among equal distances the lower index
 * first.  dist2 [n, K] fp32, idx [n, K] int64.  If m < K the missing entries are dist2 = 0, idx = 0 (pytorch3d's padding).
 * 1 <= K <= 32, 1 <= D <= 32. */
int gsr_knn_points(int64_t n, int64_t m, int D, int K, const float* p1, const float* p2, float* dist2, int64_t* idx, void* stream);
/* The batched form, pytorch3d.ops.knn_points(p1 [B, n, D], p2 [B, m, D], K) with equal lengths (utils/deform_utils.py:74 calls it on
 * [B, Nv, 3] node positions): B independent searches in one call, outputs [B, n, K]. Same results as B calls of gsr_knn_points. */
int gsr_knn_points_batch(int64_t B, int64_t n, int64_t m, int D, int K, const float* p1, const float* p2, float* dist2, int64_t* idx, void* stream);

typedef struct gsr_node_blend {
    int64_t n;                    /* Gaussians */
    int32_t m;                    /* control nodes */
    int32_t K;                    /* neighbours per Gaussian, 1 .. GSR_BLEND_MAX_K (self.K, arguments/__init__.py: 3) */
    int32_t local_frame;          /* :1206-1212: rotate the offset to the node by the node's local frame before translating */
    int32_t rot_as_residual;      /* d_rot_as_res (:1254-1256); 0 = the absolute form with the (1,0,0,0) bias (:1218-1231) */
    int32_t node_stride;          /* floats per row of `nodes` (3 + hyper_dim); the first 3 are the position */
    int32_t flags;                /* GSR_NODE_RADIUS_IS_LOG | GSR_NODE_WEIGHT_IS_LOGIT: apply the activation (and its chain rule) in the kernels */
    const float* x;               /* [n, 3]   Gaussian positions (detached in the reference, :1196) */
    const float* motion_mask;     /* [n]      or NULL (= 1) */
    const float* nodes;           /* [m, node_stride] */
    const float* node_radius;     /* [m]      exp(_node_radius) (:893-894), or the raw parameter with GSR_NODE_RADIUS_IS_LOG */
    const float* node_weight;     /* [m]      sigmoid(_node_weight) (:897-898) or the raw parameter with GSR_NODE_WEIGHT_IS_LOGIT; NULL = with_node_weight False */
    const float* node_trans;      /* [m, 3]   node_attrs['d_xyz']; NULL = weights only (cal_nn_weight on its own) */
    const float* node_rot;        /* [m, 4]   node_attrs['d_rotation'] */
    const float* node_scale;      /* [m, 3]   node_attrs['d_scaling'] */
    const float* node_frame;      /* [m, 9]   quaternion_to_matrix(node_attrs['local_rotation'] + (1,0,0,0)), row-major; local_frame only */
    const float* node_local_rotation; /* [m, 4] node_attrs['local_rotation'] itself (:1207): when not NULL it replaces node_frame -- the bias and
                                     quaternion_to_matrix (:115-133) are applied in the kernels, and backward returns the quaternion's gradient */
    int32_t attr_stride;          /* 0: node_trans / node_rot / node_scale / node_local_rotation are packed [m, 3 | 4] arrays. Otherwise the floats between
                                     two nodes' rows of ALL four: they are column ranges of one [m, attr_stride] matrix -- the node network's heads
                                     as one linear layer produces them, [d_xyz | d_rotation | d_scaling | local_rotation] (no copies); batches
                                     are [B, m, attr_stride]. node_frame must be NULL then. */
    int32_t grad_stride;          /* the same for the four attribute gradients of the backward calls (0: packed) */
} gsr_node_blend;

/* Forward.  nn_weight / nn_dist [n, K] fp32 and nn_idx [n, K] int64 are the three results of cal_nn_weight (:981-1011) and are
 * also what the backward call reads; d_xyz [n,3], d_rotation [n,4], d_scaling [n,3] may be NULL when node_trans is NULL. */
int gsr_node_blend_forward(const gsr_node_blend* a, float* nn_weight, float* nn_dist, int64_t* nn_idx, float* d_xyz, float* d_rotation,
                           float* d_scaling, void* stream);

/* Backward for the cotangents g_xyz [n,3], g_rotation [n,4], g_scaling [n,3] (each may be NULL = 0) and, optionally, a direct
 * cotangent of nn_weight (g_nn_weight [n,K] or NULL).  WRITES the node gradients (any may be NULL to skip):
 *   g_node_trans [m,3], g_node_rot [m,4], g_node_scale [m,3], g_node_frame [m,9] (or [m,4], the gradient of node_local_rotation,
 *   when that was given), g_node_radius [m], g_node_weight [m] (of the raw parameters when the flags say the inputs are raw).
 * x and the node positions receive no gradient (both are detached on this path, :993,1196).  Deterministic: block partials in
 * `workspace` (gsr_node_blend_workspace_size bytes) summed in a fixed order. */
size_t gsr_node_blend_workspace_size(int64_t n, int32_t m);
int gsr_node_blend_backward(const gsr_node_blend* a, const float* nn_weight, const float* nn_dist, const int64_t* nn_idx,
                            const float* g_xyz, const float* g_rotation, const float* g_scaling, const float* g_nn_weight,
                            float* g_node_trans, float* g_node_rot, float* g_node_scale, float* g_node_frame, float* g_node_radius,
                            float* g_node_weight, char* workspace, void* stream);

/* B blends of the SAME Gaussians and nodes with B sets of node attributes -- the time samples of one mapping iteration
 * (utils/slam_backend.py:361-373 calls the warp once per view and flow partner) -- in one launch per stage. node_trans / node_rot /
 * node_scale / node_local_rotation (or node_frame) of the descriptor hold [B, m, .]; d_xyz [B, n, 3], d_rotation [B, n, 4], d_scaling
 * [B, n, 3]; nn_weight / nn_dist / nn_idx are those of a single blend (they do not depend on the attributes). Backward: cotangents
 * [B, n, .] (g_nn_weight must be NULL for B > 1), node gradients [B, m, .] including g_node_radius / g_node_weight [B, m] -- the
 * radius and weight are shared by the B blends, the caller sums their rows. B = 1 is gsr_node_blend_forward / _backward. */
int gsr_node_blend_forward_batch(const gsr_node_blend* a, int B, float* nn_weight, float* nn_dist, int64_t* nn_idx, float* d_xyz, float* d_rotation,
                                 float* d_scaling, void* stream);
size_t gsr_node_blend_workspace_size_batch(int64_t n, int32_t m, int B);
int gsr_node_blend_backward_batch(const gsr_node_blend* a, int B, const float* nn_weight, const float* nn_dist, const int64_t* nn_idx,
                                  const float* g_xyz, const float* g_rotation, const float* g_scaling, const float* g_nn_weight,
                                  float* g_node_trans, float* g_node_rot, float* g_node_scale, float* g_node_frame, float* g_node_radius,
                                  float* g_node_weight, char* workspace, void* stream);

/* ---- deterministic scatter-add through an index array (round 4) ------------------------------------------------------------------------
 * The backward pass of a gather out[b][e][:] = table[b][idx[s(b)][e]][:] is a scatter-add; torch's scatter_add / index_put backward and any
 * float-atomic kernel add in hardware order, i.e. not bit-reproducibly. These two calls do it in a FIXED order:
 *   gsr_index_csr     for S index sets idx [S, E] (int64, values in [0, Nv)): per set and target v the positions e with idx[e] == v, in
 *                     increasing e, laid out in `workspace` (gsr_index_csr_workspace_size bytes); one wave per (set, target);
 *   gsr_segment_sum   out[b][v][c] = sum over those positions, in that order, of g[b][e][c]   (g [B, E, C], out [B, Nv, C]; set_of_b [B]
 *                     int32 names the index set of batch element b, NULL = set 0 for all).
 * Used by the node blend's backward (a Gaussian's gradient goes to its K nearest nodes; gsr_node_blend_backward* takes this route for K <= 4
 * by itself) and by the gathers of the ARAP / elastic node regularisers (utils/deform_utils.py:35-42, utils/time_utils.py:1160-1165). */
/* The node network's input for n time samples x M nodes (DeformNetwork's two embedders, utils/time_utils.py:208-273,428-436: include_input,
 * frequencies 2^0 .. 2^(F-1), order x, sin f0, cos f0, sin f1, ...): out [n * M, 3 (1 + 2 Fx) + (1 + 2 Ft)] row-major, row i * M + m =
 * [embed(nodes[m]) | embed(times[i])]. nodes [M, node_stride >= 3], times [n]; all on the device; workspace (device):
 * gsr_node_embedding_workspace_size bytes. Two launches: every sin / cos once, then the broadcast. */
size_t gsr_node_embedding_workspace_size(int n, int M, int Fx, int Ft);
int gsr_node_embedding(int n, int M, int Fx, int Ft, const float* nodes, int node_stride, const float* times, float* out, char* workspace, void* stream);

/* The non-GEMM half of a trunk layer's backward pass, y = relu(x W^T + b) on `rows` x `cols` row-major fp32 matrices (the node network,
 * utils/time_utils.py:327-470: eight such layers on ~50 000 rows per mapping iteration): G = dY . [Y > 0] and dbias[c] = sum over rows of
 * G[r][c], in ONE pass over dY and Y plus a few-microsecond finalisation; the sums are formed in a fixed order (bit-reproducible).
 * cols in {64, 128, 256, 512, 1024}; all buffers 16-byte aligned; workspace: gsr_relu_backward_bias_workspace_size(rows, cols) bytes. */
size_t gsr_relu_backward_bias_workspace_size(int rows, int cols);
int gsr_relu_backward_bias(int rows, int cols, const float* dY, const float* Y, float* G, float* dbias, char* workspace, void* stream);

size_t gsr_index_csr_workspace_size(int S, int E, int Nv);
int gsr_index_csr(int S, int E, int Nv, const int64_t* idx, char* workspace, void* stream);
int gsr_segment_sum(int B, int S, int E, int C, int Nv, const float* g, const char* csr_workspace, const int* set_of_b, float* out, void* stream);

/* Several small sums in ONE launch: dst[j] = src[0][j] + src[1][j] + src[2][j] + src[3][j] for j < count, per item (NULL sources are skipped;
 * no source: zeros), at most 64 items. The rows of the control-node warp's output (utils/time_utils.py:1192-1258, one per time sample) feed up to
 * three rasterizer calls of a dynamic mapping iteration (utils/slam_backend.py:357-509: the keyframe's render and the two flow renders); this
 * forms every row of the stacked gradient at once where autograd adds the calls' gradients pairwise (control_nodes.fan_out). Fixed order. */
typedef struct gsr_multi_add_item {
    float* dst;
    const float* src[4];
    int32_t count;
} gsr_multi_add_item;
int gsr_multi_add(int count, const gsr_multi_add_item* items, void* stream);

#ifdef __cplusplus
}
#endif
#endif
