/*
 * slam_losses.h -- C ABI of the fused photometric + depth L1 loss (libgs_rasterizer_hip.so), SURVEY.md 8(f) rank 2.
 *
 * Replaces the tensor expression of the reference's mapping / tracking losses
 *   utils/slam_utils.py:252-364 (get_loss_mapping, get_loss_mapping_rgbd) and the same structure at :57-173, 200-250:
 *     L = alpha * mean_{3,H,W}( w_rgb * |exp(a) * I + b - I_gt| ) + (1 - alpha) * mean_{H,W}( w_d * |D - D_gt| )
 * where every mask of the reference (rgb boundary threshold, valid-depth range, motion masks, the x2 / x3 weighting of
 * dynamic regions) is folded by the caller into the per-pixel weights w_rgb, w_d (constants of the keyframe).
 * The backward call writes dL/dI [3,H,W] and dL/dD [1,H,W] -- the cotangents gsr_backward consumes -- already multiplied
 * by the upstream gradient, plus dL/d(a, b). All pointers are DEVICE pointers, fp32, contiguous.
 */
#ifndef SLAM_LOSSES_H_INCLUDED
#define SLAM_LOSSES_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of scratch both calls need (per-block partial sums) */
size_t gsr_l1_loss_workspace_size(void);

/* loss[0] = L. w_rgb / w_depth [H*W] may be NULL (= 1); exposure_a / exposure_b (1 float each) may be NULL (a = b = 0,
 * the `initialization` branch, slam_utils.py:253-254). opacity [H*W] (may be NULL) is the RENDERED opacity of the tracking loss
 * (slam_utils.py:104,125-137): w_rgb is multiplied by it and w_depth is kept only where it exceeds opacity_depth_threshold; it is
 * treated as a constant weight (the rasterizer's backward ignores the opacity cotangent, DGR/diff_gaussian_rasterization/__init__.py:108).
 * Returns 0 or a negative GSR_ERR_* code (gs_rasterizer.h). */
int gsr_l1_loss_forward(int width, int height, const float* image, const float* depth, const float* gt_image, const float* gt_depth,
                        const float* w_rgb, const float* w_depth, const float* exposure_a, const float* exposure_b, float alpha,
                        const float* opacity, float opacity_depth_threshold, float* loss, char* workspace, void* stream);

/* upstream: device pointer to dLoss_total/dL (NULL = 1). dL_dexposure[2] = (dL/da, dL/db), may be NULL. */
int gsr_l1_loss_backward(int width, int height, const float* image, const float* depth, const float* gt_image, const float* gt_depth,
                         const float* w_rgb, const float* w_depth, const float* exposure_a, const float* exposure_b, float alpha,
                         const float* opacity, float opacity_depth_threshold, const float* upstream, float* dL_dimage, float* dL_ddepth, float* dL_dexposure, char* workspace, void* stream);

/* ---- masked L1 against a constant target (the optical-flow terms of the dynamic mapping loop, utils/slam_backend.py:479-509) ------
 *   loss[0] = scale * sum_terms mean_{c < channels, p}( | target[c,p] - image[c,p] * mask[p] | )
 * image [image_channels,H,W] is a rendering (render_flow: u, v, dynamic mask -> channels = 2 of image_channels = 3), target
 * [channels,H,W] and mask [H,W] (0 / 1 floats) are constants of a keyframe pair (the flow already multiplied by the mask). Up to 4
 * terms per call: a keyframe's two flow directions are one forward (2 launches) and one backward launch. The backward call writes
 * every term's dL_dimage [image_channels,H,W] (zeros in the channels beyond `channels`), multiplied by upstream[0] (NULL = 1).
 * workspace: gsr_l1_loss_workspace_size() bytes. */
typedef struct gsr_masked_l1_term { const float* image; const float* target; const float* mask; float* dL_dimage; } gsr_masked_l1_term;
int gsr_masked_l1_forward(int n_terms, const gsr_masked_l1_term* terms, int width, int height, int channels, int image_channels, float scale,
                          float* loss, char* workspace, void* stream);
int gsr_masked_l1_backward(int n_terms, const gsr_masked_l1_term* terms, int width, int height, int channels, int image_channels, float scale,
                           const float* upstream, void* stream);

/* ---- fused SSIM ("optional SSIM") ---------------------------------------------------------------
 * gaussian_splatting/utils/loss_utils.py:46-111 (ssim, size_average=True, window 11, sigma 1.5, zero padding, per channel) as the
 * mapping / colour-refinement losses use it (utils/slam_backend.py:636,824-832): ssim_mean[0] = mean of the SSIM map of img1
 * (the rendering, [C,H,W]) against img2 (ground truth). mask [H*W] bytes (may be NULL): both images are zeroed where it is 0
 * (loss_utils.py:66-68). The workspace (gsr_ssim_workspace_size bytes) carries the derivative maps from forward to backward. */
size_t gsr_ssim_workspace_size(int width, int height, int channels);
int gsr_ssim_forward(int width, int height, int channels, const float* img1, const float* img2, const unsigned char* mask,
                     float* ssim_mean, char* workspace, void* stream);
/* dL_dimg1 [C,H,W] = upstream[0] (NULL = 1) * d ssim_mean / d img1. img2 receives no gradient. */
int gsr_ssim_backward(int width, int height, int channels, const float* img1, const float* img2, const unsigned char* mask,
                      const float* upstream, float* dL_dimg1, char* workspace, void* stream);

/* ---- fused Adam step: all parameter tensors of the Gaussian model in one launch -----------------
 * Replaces optimizer.step() of scene/gaussian_model.py:447 (torch.optim.Adam(lr=0.0, eps=1e-15) over the six groups of :404-434)
 * with the arithmetic of torch.optim.Adam's single-tensor path (no amsgrad, no weight decay, no maximize). At most 32 segments. */
typedef struct gsr_adam_segment {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;   /* device, n floats each */
    unsigned long long n;
    float lr, beta2, eps;                                                  /* beta2 as the fp32 multiplier of v */
    double beta1_d, beta2_d;                                               /* betas in double: 1 - beta and the bias corrections are
                                                                              evaluated in double, as torch.optim.Adam does */
    int step;                                                              /* step count AFTER this step (>= 1): bias corrections */
} gsr_adam_segment;
int gsr_adam_step(int nseg, const gsr_adam_segment* segs, void* stream);
/* The two step-dependent coefficients of a segment as gsr_adam_step evaluates them on the host (double arithmetic like torch):
 * out[0] = (float)(lr / (1 - beta1^step)), out[1] = (float)(1 / sqrt(1 - beta2^step)). */
void gsr_adam_coefficients(double lr, double beta1, double beta2, int step, float out[2]);
/* gsr_adam_step with those two coefficients read from DEVICE memory (coefficients[2 * k], [2 * k + 1] for segment k) when the kernel
 * runs; segs[k].lr / .step are ignored. Lets a hipGraph that contains the optimizer step be replayed for many iterations: the host (or
 * gsr_schedule_advance, slam_map.h) rewrites 2 * nseg floats per iteration instead of re-capturing. Same arithmetic, bit for bit. */
int gsr_adam_step_scheduled(int nseg, const gsr_adam_segment* segs, const float* coefficients, void* stream);
/* gsr_adam_step for an optimizer whose step COUNTS live on the device (torch.optim.Adam(capturable=True): one float32 scalar per parameter,
 * step_counts[k] for segment k; segments may share one): a first tiny launch advances the counts and evaluates the two coefficients of every
 * segment into `coefficients` (device, 2 * nseg floats), then gsr_adam_step_scheduled. Recorded in a hipGraph the step stays correct on
 * replay. The `step` field of the segments is ignored. */
int gsr_adam_step_device_count(int nseg, const gsr_adam_segment* segs, float* const* step_counts, float* coefficients, void* stream);

/* ---- densification statistics of one rendered view in one launch ----------------------------------------------------------
 * utils/slam_backend.py:712-720 + scene/gaussian_model.py:973-977 (add_densification_stats): for every Gaussian with radii > 0
 *   max_radii2D = max(max_radii2D, radii);  xyz_gradient_accum += |grad_mean2D[:, :2]|;  denom += 1.
 * radii int32[P] and grad_mean2D float[P,3] (viewspace_points.grad) are the rasterizer's outputs; the three state arrays are
 * float[P] (the reference keeps them as [P] and [P,1] float tensors). No host synchronisation. */
int gsr_densification_stats(int P, const int* radii, const float* grad_mean2D, float* max_radii2D, float* xyz_gradient_accum, float* denom,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif
