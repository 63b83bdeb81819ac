"""Test infrastructure (never imported by the product path): the weighted L1 loss of include/slam_losses.h as a plain torch
expression on whatever device the tensors live on, used to check 4dgs-slam_amd/slam_losses.py against the golden vectors of
the reference's get_loss_mapping (utils/slam_utils.py:252-364; tests/golden/make_golden_loss.py)."""
import torch


def weighted_l1_loss_reference(image, depth, gt_image, gt_depth, w_rgb=None, w_depth=None, exposure_a=None, exposure_b=None, alpha=0.95,
                               opacity=None, opacity_depth_threshold=0.95):
    image_ab = image if exposure_a is None else torch.exp(exposure_a) * image + exposure_b
    w_rgb = torch.ones_like(depth) if w_rgb is None else w_rgb.view(*depth.shape)
    w_depth = torch.ones_like(depth) if w_depth is None else w_depth.view(*depth.shape)
    if opacity is not None:          # tracking loss: rendered opacity as a weight (no gradient reaches the rasterizer through it)
        w_rgb = w_rgb * opacity.detach().view(*depth.shape)
        w_depth = w_depth * (opacity.detach() > opacity_depth_threshold).view(*depth.shape)
    return alpha * (w_rgb * torch.abs(image_ab - gt_image)).mean() + (1 - alpha) * (w_depth * torch.abs(depth - gt_depth)).mean()
