#!/usr/bin/env python
"""lambda_dssim * (1 - ssim(image, gt)) forward + backward at 3x480x640: the torch expression of the 3DGS code base vs the fused kernels."""
import json, math, os, sys, time
import torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "4dgs-slam_amd"))
from slam_losses import ssim as fused_ssim

def torch_ssim(img1, img2):
    C = img1.shape[-3]
    g = torch.tensor([math.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    w = (g @ g.t()).float()[None, None].expand(C, 1, 11, 11).contiguous().to(img1)
    conv = lambda t: F.conv2d(t, w, padding=5, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

img1 = torch.rand(3, 480, 640, device="cuda", requires_grad=True)
img2 = torch.rand(3, 480, 640, device="cuda")
out = {}
for name, fn in (("torch_ssim", torch_ssim), ("fused_ssim", fused_ssim)):
    def step():
        img1.grad = None
        (0.2 * (1.0 - fn(img1, img2))).backward()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize()
    out[name + "_us_fwd_bwd"] = (time.perf_counter() - t0) / 100 * 1e6
print(json.dumps(out))
