/*
 * deformation_field.h -- C ABI of the HexPlane feature field of the 4D-Gaussians deformation network
 * (libgs_rasterizer_hip.so), SURVEY.md 8(f) rank 3.
 *
 * Replaces, with ONE forward and ONE backward launch, the tensor program of
 *   gaussian_splatting/utils/hexplane.py:19-22   normalize_aabb          p -> clamp((p - aabb[0]) * 2 / (aabb[1] - aabb[0]) - 1, -1, 1)
 *   gaussian_splatting/utils/hexplane.py:23-50   grid_sample_wrapper     F.grid_sample(bilinear, border, align_corners=True)
 *   gaussian_splatting/utils/hexplane.py:81-112  interpolate_ms_features product over the 6 coordinate planes, concat over levels
 *   gaussian_splatting/utils/hexplane.py:162-188 HexPlaneField.get_density / forward
 * (24 grid_sample launches + 20 products + 1 concat forward, and their autograd twins, per call of
 *  utils/deformation.py:71-87 Deformation.query_time).
 *
 * Geometry.  The field has `num_levels` resolution levels; level l has six planes in itertools.combinations(range(4), 2) order
 *   0: (x,y)  1: (x,z)  2: (x,t)  3: (y,z)  4: (y,t)  5: (z,t)
 * plane (c0,c1) is the reference parameter of logical shape [1, C, res[l][c1], res[l][c0]] (hexplane.py:66-68): the FIRST
 * coordinate indexes the width.  C = feat_dim channels (a multiple of 4, at most 64).  Two memory layouts are accepted:
 *   channels_last = 1   [H][W][C]  (what torch calls channels_last for the logical [1,C,H,W]; one 16*C/4-byte run per texel --
 *                                   the layout this library is designed for)
 *   channels_last = 0   [C][H][W]  (the reference's contiguous layout; works, every corner fetch is C scattered 4-byte loads)
 * Output features are [n, num_levels * C] with level l at columns [l*C, (l+1)*C).
 *
 * All pointers are DEVICE pointers to fp32 unless stated; the descriptor itself lives in host memory and is copied by value
 * into the kernel arguments.  Returns 0 or a negative GSR_ERR_* code (gs_rasterizer.h).
 */
#ifndef DEFORMATION_FIELD_H_INCLUDED
#define DEFORMATION_FIELD_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_HEXPLANE_MAX_LEVELS 8

typedef struct gsr_hexplane_level {
    const float* planes[6];      /* parameters, layout above */
    float* grad_planes[6];       /* backward only: dL/dplane, same layout, ACCUMULATED into (caller zeroes or keeps .grad); an
                                    entry may be NULL to skip that plane */
    int32_t res[4];              /* resolution along x, y, z, t at this level (hexplane.py:139-145: spatial ones times multires) */
} gsr_hexplane_level;

typedef struct gsr_hexplane_field {
    int32_t num_levels;          /* 1 .. GSR_HEXPLANE_MAX_LEVELS */
    int32_t feat_dim;            /* C */
    int32_t channels_last;       /* memory layout of every plane and gradient plane */
    int32_t reserved;
    const float* aabb;           /* 6 floats: aabb[0] (xyz) then aabb[1] (xyz), the reference's HexPlaneField.aabb parameter; NULL = xyz
                                    is already normalised (interpolate_ms_features called directly: no scaling, no clamp) */
    gsr_hexplane_level levels[GSR_HEXPLANE_MAX_LEVELS];
} gsr_hexplane_field;

/* features[n, L*C] = HexPlaneField.forward(xyz, time).  xyz row i = xyz + i * xyz_stride (floats; 3 used -- the reference passes
 * the first three columns of a [n, 63] positional embedding, utils/deformation.py:78), time likewise (1 used). */
int gsr_hexplane_forward(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, const float* time,
                         int64_t time_stride, float* features, void* stream);

/* Backward of the above for the cotangent dL_dfeatures [n, L*C]: accumulates into field->levels[l].grad_planes[p] and, when
 * dL_dxyz is not NULL, WRITES dL_dxyz [n, 3] (contiguous) -- including the zero the reference produces for a coordinate clamped
 * by normalize_aabb or sitting on the sampler's border (GridSampler.h clip_coordinates_set_grad).  Time receives no gradient
 * (the reference builds it with torch.tensor(...).repeat, gaussian_renderer/__init__.py:112). */
int gsr_hexplane_backward(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, const float* time,
                          int64_t time_stride, const float* dL_dfeatures, float* dL_dxyz, char* workspace, void* stream);
/* `workspace` (gsr_hexplane_backward_workspace_size bytes, contents undefined on entry) selects the sorted algorithm: the points
 * are counting-sorted per plane family by the Morton code of their finest-level cell, dL/dsample is staged in the workspace in
 * sorted order, and runs of points that share a cell are summed in registers before ONE set of four atomics per run
 * (channels-last planes, resolutions <= 1024) -- the fast path for large n, and the one whose cost falls rather than rises when
 * many points share texels.  workspace NULL = one float atomic per (point, corner), no extra memory.
 *
 * Ordered mode (gsr_set_option("hex_ordered", 1), the default; both sorted entry points): the plane gradients are BITWISE reproducible -- run
 * to run and under any permutation of the points. Every product dL/dsample x corner weight is rounded once to a power-of-two quantum
 * (2^-40 of the call's largest |dL/dsample|; 2^-(62 - ceil(log2(4 n))) beyond 1 M points) and from there on only integers are added
 * (registers, LDS and 64-bit integer atomics in the workspace); a last pass converts the sums and ADDS them to grad_planes, one owner per
 * texel. Against the float-atomic mode (0: rounds 1-5) the sums differ by rounding only. The workspace grows by 8 bytes per texel of the
 * planes: read the size AFTER setting the option and do not change the option between the size query and the call. The unsorted
 * path (workspace NULL) keeps its float atomics. */
size_t gsr_hexplane_backward_workspace_size(const gsr_hexplane_field* field, int64_t n);

/* ---- the views of one mapping iteration at once ---------------------------------------------------------------------------------
 * The mapping back-end calls the field once per keyframe of an iteration with the SAME positions and one time per keyframe
 * (gaussian_renderer/__init__.py:112,149-157; utils/slam_backend.py:357,526,657). Per level the field is the product of three planes that
 * depend on the position only (xy, xz, yz) and three that depend on one coordinate and the time (hexplane.py:93-103). These calls take V <=
 * GSR_HEXPLANE_MAX_VIEWS times (HOST pointer, copied into the kernel arguments) and
 *   forward:  gather the spatial planes once per point, the time planes per view; features [V][n][L*C], view v bit-identical to
 *             gsr_hexplane_forward(xyz, time = times[v]);
 *   backward: dL_dfeatures [V][n][L*C]; ONE counting sort of the points for all views, dL/dsample of the spatial planes summed over the
 *             views in registers before the (single) spatial scatter, the time planes' per view; accumulates into grad_planes like V
 *             gsr_hexplane_backward calls, WRITES dL_dxyz [n,3] = the sum over the views. Needs channels-last planes, resolutions
 *             <= 1024 and a workspace of gsr_hexplane_backward_views_workspace_size() bytes (0 = unsupported geometry: call view by view).
 *             Ordered mode (above): the spatial planes as there; the time families keep integer column sums per view and the last pass
 *             applies each view's two time-row weights, views in their order. */
#define GSR_HEXPLANE_MAX_VIEWS 12
int gsr_hexplane_forward_views(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, int V, const float* times,
                               float* features, void* stream);
size_t gsr_hexplane_backward_views_workspace_size(const gsr_hexplane_field* field, int64_t n, int V);
/* view_mask (device, [n] or NULL): bit v of view_mask[i] clear = row (v, i) of dL_dfeatures is ZERO and is not read (it may be uninitialised
 * memory: gsr_deform_mlp_backward_rows does not write the rows it was not given); points whose bits are all clear are left out of the sort.
 * NULL = every row is read. gsr_row_mask below builds the mask from the network's [V][n][10] cotangent. */
int gsr_hexplane_backward_views(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, int V, const float* times,
                                const float* dL_dfeatures, const uint32_t* view_mask, float* dL_dxyz, char* workspace, void* stream);

/* Which rows of a batched cotangent g [V][n][width] are not zero -- in a mapping iteration most (view, Gaussian) pairs receive no gradient at
 * all (outside the frustum, or behind saturated pixels: 63 % at BASELINE config #3). view_mask[i] bit v = row (v, i) has a non-zero element;
 * rows[] = the flat row indices v n + i of those rows in ascending order, n_rows[0] (DEVICE memory) their number; rows must hold V n
 * entries. Deterministic. V <= 32, V n < 2^31; workspace: gsr_row_mask_workspace_size(V, n) bytes. */
size_t gsr_row_mask_workspace_size(int V, int64_t n);
int gsr_row_mask(int V, int64_t n, int width, const float* g, uint32_t* view_mask, int32_t* rows, int32_t* n_rows, char* workspace, void* stream);

/* ---- weight gradient of the deformation MLP's dense layers ----------------------------------------------------------------
 * utils/deformation.py:58-70 builds the network from nn.Linear layers (width 64, inputs <= 128) applied to every point; their
 * weight gradients are GEMMs with a tiny output and a reduction over all n points, the shape vendor GEMMs handle worst.
 *   dW[out_dim][in_dim] = dYᵀ X        (WRITTEN, row-major like nn.Linear.weight.grad)
 *   db[out_dim]         = column sums of dY   (WRITTEN; may be NULL)
 * X is [n, in_dim] with row stride x_stride floats, dY is [n, out_dim] with row stride dy_stride.  in_dim <= 128 per call.
 * The result is deterministic (block partials in `workspace`, summed in a fixed order). */
size_t gsr_linear_wgrad_workspace_size(int64_t n, int in_dim, int out_dim);
int gsr_linear_wgrad(int64_t n, int in_dim, int out_dim, const float* x, int64_t x_stride, const float* dy, int64_t dy_stride,
                     float* dW, float* db, char* workspace, void* stream);

/* ---- the deformation MLP, fused ---------------------------------------------------------------------------------------------
 * utils/deformation.py:58-70,101-149 with the shipped flags (defor_depth 1, width 64, position / scale / rotation heads):
 *   h0 = F W0^T + b0;  u_j = relu(h0) W1j^T + b1j;  o_j = relu(u_j) W2j^T + b2j,  j = 0 (3 outputs), 1 (3), 2 (4)
 * Weights are nn.Linear tensors as they are ([out, in] row-major).  in_dim must be a multiple of 16, at most 128; width is 64. */
typedef struct gsr_deform_mlp {
    const float* W0; const float* b0;              /* [64, in_dim], [64]                    feature_out.0 */
    const float* W1[3]; const float* b1[3];        /* [64, 64], [64]                        {pos,scales,rotations}_deform.1 */
    const float* W2[3]; const float* b2[3];        /* [3|3|4, 64], [3|3|4]                  {pos,scales,rotations}_deform.3 */
    int32_t in_dim;
    int32_t reserved;
} gsr_deform_mlp;

/* features [n, in_dim] -> out [n, 10] = (dx 3, ds 3, dr 4).  Nothing else is kept: the backward call recomputes the activations. */
int gsr_deform_mlp_forward(const gsr_deform_mlp* mlp, int64_t n, const float* features, float* out, void* stream);
/* dout [n, 10] -> dfeatures [n, in_dim] and ALL parameter gradients, written into one flat array `grads` of
 * gsr_deform_mlp_grad_count(in_dim) floats laid out as
 *     W0 [64][in_dim] | b0 [64] | for j = 0, 1, 2:  W1j [64][64] | b1j [64] | W2j [o_j][64] | b2j [o_j]      (o = 3, 3, 4)
 * (row-major like nn.Linear.weight.grad).  Deterministic: block partials in `workspace`, summed in a fixed order. */
size_t gsr_deform_mlp_grad_count(int in_dim);
size_t gsr_deform_mlp_workspace_size(int in_dim);
int gsr_deform_mlp_backward(const gsr_deform_mlp* mlp, int64_t n, const float* features, const float* dout, float* dfeatures,
                            float* grads, char* workspace, void* stream);
/* The same over a LIST of rows (gsr_row_mask): only rows[0 .. n_rows[0]) of features / dout are read and only those rows of dfeatures are
 * WRITTEN -- the rows of a zero cotangent contribute nothing to any gradient and their dfeatures would be zero; n_rows is read on the device
 * (no host synchronisation). rows = n_rows = NULL: every row, as gsr_deform_mlp_backward. n = number of rows of the arrays. */
int gsr_deform_mlp_backward_rows(const gsr_deform_mlp* mlp, int64_t n, const float* features, const float* dout, float* dfeatures,
                                 float* grads, char* workspace, const int32_t* rows, const int32_t* n_rows, void* stream);
#ifdef __cplusplus
}
#endif
#endif
