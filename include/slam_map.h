/*
 * slam_map.h -- C ABI of the per-Gaussian / per-camera work of the SLAM loop around the rasterizer (libgs_rasterizer_hip.so),
 * seeding Gaussians from an RGB-D keyframe, densify / clone / split / prune with optimizer-state surgery,
 * and the camera-pose update of the tracking / mapping loops. Host control flow (keyframe selection, windows, queues) stays in
 * Python (4dgs-slam_amd/slam/), as in the reference. All pointers are DEVICE pointers unless stated; fp32 / int32; contiguous.
 * Functions return 0 or a negative GSR_ERR_* code (gs_rasterizer.h); gsr_last_error() has the text.
 */
#ifndef SLAM_MAP_H_INCLUDED
#define SLAM_MAP_H_INCLUDED

#include <stddef.h>
#include "gs_rasterizer.h"   /* gsr_alloc_fn, gsr_raw_inputs (gsr_track_step) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- seeding: GaussianModel.create_pcd_from_image_and_depth, gaussian_splatting/scene/gaussian_model.py:185-255 -----------------
 * (Open3D's create_from_rgbd_image + the attribute initialisation), for n already selected pixels pix[i] = v * width + u whose
 * depth is valid (the random down-sampling of :215 stays with the caller):
 *   z = depth[pix];  p_cam = ((u - cx) z / fx, (v - cy) z / fy, z);  xyz = R^T (p_cam - T)           (extrinsic = W2C = [R | T])
 *   rgb = floor(clamp(exp(a) * image + b, 0, 1) * 255) / 255          (:186-188: the byte image Open3D turns back into floats)
 *   features_dc = (rgb - 0.5) / 0.28209479177387814                   (RGB2SH, gaussian_splatting/utils/sh_utils.py:121-122)
 *   log_scales  = log(sqrt(max(distCUDA2(xyz), 1e-7) * point_size))   (:235-242; scale_dim 1 = isotropic model, 3 = repeated)
 *   rotations = (1, 0, 0, 0);  logit_opacity = inverse_sigmoid(0.5) = 0                                            (:244-253)
 * R [3,3] row-major and T [3] are the camera's device tensors; exposure_a / exposure_b (1 float each) may be NULL (a = b = 0).
 * The 3-nearest-neighbour distances come from the same spatial-hash kernels as gsr_knn_mean_dist2 (simple_knn.h). */
size_t gsr_seed_workspace_size(int n);
int gsr_seed_from_rgbd(int n, const int* pix, int width, int height, const float* depth, const float* image, const float* exposure_a,
                       const float* exposure_b, float fx, float fy, float cx, float cy, const float* R, const float* T, float point_size,
                       int scale_dim, float* xyz, float* features_dc, float* log_scales, float* rotations, float* logit_opacity,
                       char* workspace, void* stream);

/* ---- densification: GaussianModel.densify_and_prune, gaussian_model.py:866-971 -------------------------------------------------
 * Step 1, gsr_densify_select: the per-Gaussian decisions of densify_and_clone (:920-951), densify_and_split (:866-918) and the
 * prune mask (:953-971), evaluated for the P Gaussians the call starts with:
 *   g = xyz_gradient_accum / denom (NaN -> 0);  s = max_k exp(log_scales[k]);  o = sigmoid(logit_opacity)
 *   clone  = g >= grad_threshold && s <= dense_scale            split = g >= grad_threshold && s > dense_scale
 *   pruned(original / clone) = o < min_opacity || (big_scale > 0 && s > big_scale)
 *   pruned(children)         = o < min_opacity || (big_scale > 0 && s / 1.6 > big_scale)      (children: scale / (0.8 * 2), :885-887)
 * dense_scale = percent_dense * extent, big_scale = 0.1 * extent when the caller passes a max_screen_size, else <= 0 (:961-969; the
 * screen-size test itself never fires in the reference because densification_postfix zeroes max_radii2D first, :857).
 * flags int32[4][P] (0 / 1): [0] original survives (not split, not pruned), [1] a clone survives, [2] selected for splitting
 * (consumes two rows of the caller's normal samples), [3] its two children survive.
 * Step 2 (caller): exclusive prefix sums of the four flag rows -> offsets int32[4][P], totals n_keep, n_clone, n_split, n_child.
 * Step 3, gsr_densify_apply: writes every output tensor in ONE launch. Output rows, in the reference's order:
 *   [0, n_keep) surviving originals | [n_keep, +n_clone) clones | then the first samples of the surviving split parents
 *   [.., +n_child) | then their second samples [.., +n_child).
 * Each gsr_densify_tensor describes one [P, width] fp32 (or any 4-byte) tensor and its [n_out, width] destination:
 *   GSR_DENSIFY_COPY   rows are copied (parameters, per-Gaussian bookkeeping)
 *   GSR_DENSIFY_STATE  optimizer moments: copied for surviving originals, ZERO for clones and children (cat_tensors_to_optimizer, :812-830)
 *   GSR_DENSIFY_XYZ    like COPY, children get  xyz + Rq(normalize(raw_rot)) (noise * exp(log_scales))           (:877-883)
 *   GSR_DENSIFY_SCALE  like COPY, children get  log(exp(log_scale) / 1.6)                                          (:885-887)
 * noise float[2 * n_split, 3]: standard normal samples in the reference's order (row off_split[i] for the first child of parent i,
 * n_split + off_split[i] for the second) -- torch.normal(mean=0, std=stds) draws exactly those (:875-876). */
enum { GSR_DENSIFY_COPY = 0, GSR_DENSIFY_STATE = 1, GSR_DENSIFY_XYZ = 2, GSR_DENSIFY_SCALE = 3 };
typedef struct gsr_densify_tensor { const float* src; float* dst; int width; int kind; } gsr_densify_tensor;
#define GSR_DENSIFY_MAX_TENSORS 32
int gsr_densify_select(int P, const float* xyz_gradient_accum, const float* denom, const float* log_scales, int scale_dim,
                       const float* logit_opacity, float grad_threshold, float dense_scale, float min_opacity, float big_scale,
                       int* flags, void* stream);
int gsr_densify_apply(int P, const int* flags, const int* offsets, int n_keep, int n_clone, int n_split, int n_child, int ntensors,
                      const gsr_densify_tensor* tensors /* host array */, const float* xyz, const float* log_scales, int scale_dim,
                      const float* raw_rotations, const float* noise, void* stream);

/* ---- pruning by a mask: GaussianModel.prune_points, gaussian_model.py:786-810 -- the same apply step with flags[0] = !mask and no
 * clones / children; kept for symmetry: gsr_densify_apply(P, flags, offsets, n_keep, 0, 0, 0, ...). */

/* ---- camera step of the tracking / mapping loops -------------------------------------------------------------------------------
 * One launch, no host synchronisation, capturable in a hipGraph (every operand lives in device memory, including the Adam step):
 *  (1) Adam (torch.optim.Adam single-tensor arithmetic, no weight decay) on up to 4 small tensors of the camera -- cam_rot_delta[3],
 *      cam_trans_delta[3], exposure_a[1], exposure_b[1] (utils/slam_frontend.py:346-376, utils/slam_backend.py:948-992) -- whose
 *      gradients the rasterizer / loss kernels left in device memory; `step` (1 float, device) is incremented;
 *  (2) update_pose (utils/pose_utils.py:80-97): tau = [trans_delta | rot_delta], [R | T] <- SE3_exp(tau) [R | T], deltas <- 0,
 *      converged[0] = |tau| < threshold;
 *  (3) the matrices the rasterizer reads (utils/camera_utils.py:124-148): viewmatrix = W2C^T, full_proj = viewmatrix @ projmatrix
 *      (projmatrix = P^T as the callers keep it), campos = -R^T T.
 * A NULL gradient pointer skips that tensor's Adam update; do_pose = 0 skips (2) (then (3) just refreshes the matrices). */
typedef struct gsr_camera_step {
    float* rot_delta; const float* g_rot_delta; float* trans_delta; const float* g_trans_delta;
    float* exposure_a; const float* g_exposure_a; float* exposure_b; const float* g_exposure_b;
    float* exp_avg;       /* [8] first moments: rot 3, trans 3, a, b */
    float* exp_avg_sq;    /* [8] */
    float* step;          /* [1] */
    float lr_rot, lr_trans, lr_exposure, beta1, beta2, eps;
    float* R; float* T;   /* [3,3] row-major, [3] */
    const float* projmatrix;   /* [4,4] = P^T */
    float* viewmatrix; float* full_proj; float* campos;   /* outputs [4,4], [4,4], [3] */
    int* converged;       /* [1] */
    float converged_threshold;
    int do_pose;
    int latch;            /* 1: a launch that finds *converged already set does nothing. The tracking loop breaks as soon as update_pose
                             reports convergence (utils/slam_frontend.py:441-442); a host that polls the flag only every few iterations
                             (or replays a hipGraph) gets the same final pose this way. The caller clears *converged per frame. */
} gsr_camera_step;
int gsr_camera_step_launch(const gsr_camera_step* s, void* stream);

/* Several camera steps in ONE launch (block k = steps[k]); n <= GSR_CAMERA_STEPS_MAX. The window keyframes of a mapping iteration
 * (utils/slam_backend.py:748-755, :1213-1222) are stepped together. Same arithmetic per camera as gsr_camera_step_launch. */
#define GSR_CAMERA_STEPS_MAX 12
int gsr_camera_steps_launch(int n, const gsr_camera_step* steps /* host array */, void* stream);

/* ---- a tracking iteration in one call: utils/slam_frontend.py:405-448 (render -> get_loss_tracking -> loss.backward() -> pose_optimizer.step()
 * -> update_pose), seven launches (round 6; eleven before) ---------------------------------------------------------------
 * gsr_forward_raw's pipeline with the weighted L1 tracking loss's cotangents (utils/slam_utils.py:57-173, include/slam_losses.h:
 * gsr_l1_loss_backward's arithmetic, pixel for pixel the same bits) formed in the epilogue of the tile kernel from the pixel still in
 * registers; the backward pass in its pose-only mode (GSR_BACKWARD_POSE_ONLY); and ONE tail launch that finishes the pose-gradient sum
 * dL/dtau [rho | theta] and the two exposure-gradient sums in a fixed order and performs gsr_camera_step_launch's step with them (handed over
 * in LDS). Nothing passes through autograd or the host; capturable in a hipGraph (lazy mode, like gsr_forward_raw).
 *   camera: `step` -- its viewmatrix / full_proj / campos are the matrices the pass renders with (and the step refreshes), rot_delta /
 *     trans_delta are stepped always, exposure_a / exposure_b when given (both or neither; they also enter the loss as exp(a) I + b). The
 *     g_* fields are ignored. projmatrix_raw: the projection without the view (P^T), as gsr_backward_raw takes it.
 *   loss: alpha * mean_{3,H,W}(w_rgb |exp(a) I + b - gt_image|) + (1 - alpha) * mean_{H,W}(w_depth |D - gt_depth|); opacity_weights = 1
 *     (tracking): w_rgb *= rendered opacity, w_depth *= (opacity > opacity_depth_threshold). Weights may be NULL (= 1).
 *   workspace (gsr_track_workspace_size bytes, 16-byte aligned), as floats: dL_dimage [3 N] | dL_ddepth [N] | exposure partial sums [2 T] |
 *     dL_dtau [6] | dL_dexposure [2]  (N = width * height, T = 16 x 16 tiles) -- readable after the call (tests compare them with the
 *     autograd route: images and dL_dtau bit-identical, the exposure sums equal to rounding: per tile here, per 256 strided pixels there).
 * Returns what gsr_forward_raw returns (instances, or the speculative capacity in lazy mode); < 0: error. Outputs as gsr_forward_raw. */
typedef struct gsr_track_loss {
    const float* gt_image; const float* gt_depth; const float* w_rgb; const float* w_depth;
    float alpha; float opacity_depth_threshold; int opacity_weights;
} gsr_track_loss;
size_t gsr_track_workspace_size(int width, int height);
int gsr_track_step(gsr_alloc_fn geometry_alloc, void* geometry_user, gsr_alloc_fn binning_alloc, void* binning_user, gsr_alloc_fn image_alloc,
                   void* image_user, int P, int D, int M, const float* background, int width, int height, const gsr_raw_inputs* in,
                   float scale_modifier, const float* projmatrix_raw, float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                   float* out_opacity, int* radii, int* n_touched, const gsr_track_loss* loss, const gsr_camera_step* step, float* dL_dmean2D,
                   char* workspace, void* stream);

/* ---- a mapping iteration as ONE hipGraph (utils/slam_backend.py:1013-1224; slam/mapping_graph.py) ----------------------------------
 * A captured graph cannot take new host values per replay. What changes from one mapping iteration to the next -- WHICH two random
 * keyframes are rendered besides the window (:1031-1037) and the step-dependent Adam coefficients / learning rates -- therefore lives in
 * a device-side SCHEDULE: `rows` rows of `row_words` 32-bit words, written by the host once for a run of iterations.
 *   gsr_schedule_advance: current[0 .. row_words) = table[min(*counter, rows - 1)][..]; *counter += 1.       (one tiny launch)
 * The rest of the graph reads `current` (keyframe indices as int32, coefficients as float).
 *   gsr_slot_gather: for slot s < n_slots: e = table[index[s]] (a device array of gsr_keyframe_entry, one per candidate keyframe, holding
 *   DEVICE pointers to that keyframe's persistent buffers); copy its camera block (viewmatrix 16, full_proj 16, campos 3, exposure_a 1,
 *   exposure_b 1 floats) and its ground truth / loss-weight images (gt_image [3 * pixels], gt_depth, w_rgb, w_depth [pixels] each) into
 *   the slot's own persistent buffers dst[s]. The graph renders and scores the SLOTS; which keyframe fills them is an index write. */
typedef struct gsr_keyframe_entry {
    float* viewmatrix; float* full_proj; float* campos; float* exposure_a; float* exposure_b;
    float* gt_image; float* gt_depth; float* w_rgb; float* w_depth;
} gsr_keyframe_entry;
#define GSR_SLOTS_MAX 4
int gsr_schedule_advance(int* counter, const unsigned int* table, int row_words, int rows, unsigned int* current, void* stream);
int gsr_slot_gather(int n_slots, const gsr_keyframe_entry* table /* device */, const int* index /* device, n_slots */,
                    const gsr_keyframe_entry* dst /* host array, n_slots */, int pixels, void* stream);

/* ---- edge mask of a frame: Camera.compute_grad_mask, utils/camera_utils.py:205-233 (non-replica branch) with image_gradient /
 * image_gradient_mask of utils/slam_utils.py:5-39 ---------------------------------------------------------------------------------
 * image [3,H,W] -> gray = mean over channels; Scharr gradients (normalised by 16) of the reflect-padded gray image, zeroed where any of the
 * nine neighbours has |gray| <= eps (0.01 in the reference); intensity [H*W] = gradient magnitude; median[1] = torch.median(intensity)
 * (the lower of the two middle values); mask [H*W] (0 / 1) = intensity > median * edge_threshold. Three launches, no synchronisation. */
int gsr_edge_mask(const float* image, int height, int width, float edge_threshold, float eps, float* intensity, float* median,
                  unsigned char* mask, void* stream);

/* ---- best-fit rotations for the ARAP regulariser of the control nodes: estimate_rotation, utils/deform_utils.py:130-166 ----------------
 * S, R: n row-major 3x3 matrices. R = V U^T of S = U Sigma V^T, with the reference's reflection rule (the column of the smallest singular
 * value is flipped when det <= 0): always a proper rotation; S = 0 -> identity. One thread per matrix, double precision inside. */
int gsr_kabsch_rotations(int n, const float* S, float* R, void* stream);

/* The isotropic regulariser of the mapping loops (utils/slam_backend.py:653-655,:1189-1191): loss = 10 * mean over [P, 3] of |s - mean_k s|,
 * s = exp(raw_scales), raw_scales [P, 3] (the model's _scaling). Forward: per-block sums + an ordered finalisation (workspace:
 * gsr_isotropic_loss_workspace_size(P) bytes); backward: d_raw_scales [P, 3] for the upstream gradient g_loss (one device float). */
size_t gsr_isotropic_loss_workspace_size(int P);
int gsr_isotropic_loss_forward(int P, const float* raw_scales, float* loss, char* workspace, void* stream);
int gsr_isotropic_loss_backward(int P, const float* raw_scales, const float* g_loss, float* d_raw_scales, void* stream);

/* The node graph's two regularisers, one launch each way (round 4). The caller gathers the neighbours' positions (a gather whose backward is an
 * ordered scatter: gsr_index_csr + gsr_segment_sum) and reduces the per-node results; everything per (view, sample, node) happens here.
 *   ARAP (cal_arap_error, utils/deform_utils.py:177-205, no edge weights): p [V][T][M][3] node positions at T time samples of V views,
 *     nb [V][T][M][K][3] their K neighbours' positions, keep [V][M][K] 0 / 1 (cal_connectivity_from_points' radius rule).
 *     E_t[k] = (p_t - nb_t[k]) keep[k]; S = sum_k E_0[k] E_t[k]^T, zeroed when no edge changed in some coordinate (:147-149); R = V U^T of S's
 *     SVD with the reflection rule (gsr_kabsch_rotations' arithmetic); partial[v][t-1][m] = sum_k keep[k] |E_t[k] - R E_0[k]|^2.
 *     R [V][T-1][M][9] is kept for the backward pass, where it is a constant (:190-204).  g_partial: cotangent of partial.
 *   Elastic (ControlNodeWarp.elastic_loss, utils/time_utils.py:1160-1165): x [V][M][T][3], nb [V][M][K][T][3] ->
 *     ratio[v][m][k] = var_t |nb_t - x_t| / (the same value, detached, + 1e-5), unbiased variance, 2 <= T <= 16. */
int gsr_arap_forward(int V, int T, int M, int K, const float* p, const float* nb, const float* keep, float* R, float* partial, void* stream);
int gsr_arap_backward(int V, int T, int M, int K, const float* p, const float* nb, const float* keep, const float* R, const float* g_partial,
                      float* dp, float* dnb, void* stream);
int gsr_elastic_forward(int V, int M, int K, int T, const float* x, const float* nb, float* ratio, void* stream);
int gsr_elastic_backward(int V, int M, int K, int T, const float* x, const float* nb, const float* g_ratio, float* dx, float* dnb, void* stream);

#ifdef __cplusplus
}
#endif
#endif
