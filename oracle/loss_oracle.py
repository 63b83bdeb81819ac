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


def ssim_reference(img1, img2, mask=None):
    """SSIM exactly as the 3DGS code base defines it (11x11 Gaussian window, sigma 1.5, zero padding, per channel, mean): the
    restatement checked against the golden vectors of the reference's own loss_utils.ssim."""
    import torch.nn.functional as F
    C = img1.shape[-3]
    g = torch.tensor([float(__import__("math").exp(-((x - 5) ** 2) / (2 * 1.5 ** 2))) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    window = (g @ g.t()).float()[None, None].expand(C, 1, 11, 11).contiguous().to(img1)
    if mask is not None:
        img1, img2 = torch.where(mask.unsqueeze(0), img1, 0.0), torch.where(mask.unsqueeze(0), img2, 0.0)
    conv = lambda t: F.conv2d(t, window, padding=5, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()
