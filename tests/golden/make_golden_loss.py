#!/usr/bin/env python
"""Generates tests/golden/golden_loss.npz by calling the reference's OWN utils/slam_utils.py:get_loss_mapping (imported from
/root/reference, authoring container only) on seeded CPU tensors, for the flag combinations utils/slam_backend.py uses.
Fixtures are data only: inputs, the loss value, and autograd gradients w.r.t. image, depth and the exposure parameters."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self          # the reference calls .cuda() on the ground-truth image
import utils.slam_utils as ref                            # noqa: E402

H, W = 48, 64
rng = np.random.default_rng(7)
cases = {
    "plain": dict(),
    "alpha_half": dict(alpha=0.5),
    "initialization": dict(initialization=True),
    "rm_dynamic_motion": dict(rm_dynamic=True),
    "rm_dynamic_motion_and_mask": dict(rm_dynamic=True, use_mask=True),
    "dynamic_motion": dict(dynamic=True),
    "dynamic_mask": dict(dynamic=True, use_mask=True, rm_dynamic=True),
}
config = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
out = {"H": H, "W": W, "cases": np.array(list(cases)), "cfg_alpha": 0.9, "cfg_thr": 0.01}
gt_image = rng.uniform(0, 1, size=(3, H, W)).astype(np.float32)
gt_image[:, :4, :6] = 0.0                                 # below the rgb boundary threshold
gt_depth = rng.uniform(0.3, 5.0, size=(H, W)).astype(np.float32)
gt_depth[10:14, 20:30] = 0.0                              # invalid depth
motion = rng.uniform(size=(H, W)) > 0.3
mask = rng.uniform(size=(H, W)) > 0.5
out.update(gt_image=gt_image, gt_depth=gt_depth, motion_mask=motion, mask=mask)
for name, kw in cases.items():
    kw = dict(kw)
    image = torch.tensor(rng.uniform(0, 1, size=(3, H, W)).astype(np.float32), requires_grad=True)
    depth = torch.tensor(rng.uniform(0.2, 5.0, size=(1, H, W)).astype(np.float32), requires_grad=True)
    a = torch.nn.Parameter(torch.tensor([0.07]))
    b = torch.nn.Parameter(torch.tensor([-0.03]))
    vp = types.SimpleNamespace(original_image=torch.tensor(gt_image), depth=gt_depth, exposure_a=a, exposure_b=b,
                               motion_mask=torch.tensor(motion), uid=3)
    use_mask = kw.pop("use_mask", False)
    loss = ref.get_loss_mapping(config, image, depth, vp, None, mask=torch.tensor(mask) if use_mask else None, **kw)
    loss.backward()
    out[f"{name}/image"], out[f"{name}/depth"] = image.detach().numpy(), depth.detach().numpy()
    out[f"{name}/loss"] = loss.item()
    out[f"{name}/g_image"], out[f"{name}/g_depth"] = image.grad.numpy(), depth.grad.numpy()
    out[f"{name}/g_a"] = a.grad.numpy() if a.grad is not None else np.zeros(0, np.float32)
    out[f"{name}/g_b"] = b.grad.numpy() if b.grad is not None else np.zeros(0, np.float32)
    out[f"{name}/flags"] = np.array([kw.get("initialization", False), kw.get("rm_dynamic", False), kw.get("dynamic", False), use_mask,
                                     "alpha" in kw])
    out[f"{name}/alpha"] = kw.get("alpha", -1.0)
# ---- tracking loss (utils/slam_utils.py:57-173); it creates config["Results"]["save_dir"]/tracking as a side effect
import tempfile                                            # noqa: E402
tcfg = {"Training": dict(config["Training"]), "Results": {"save_dir": tempfile.mkdtemp()}}
grad_mask = rng.uniform(size=(1, H, W)) > 0.4
out["grad_mask"] = grad_mask
tcases = {"track_plain": dict(), "track_rm_dynamic": dict(rm_dynamic=True), "track_mask": dict(use_mask=True),
          "track_rm_dynamic_mask": dict(rm_dynamic=True, use_mask=True)}
out["tracking_cases"] = np.array(list(tcases))
for name, kw in tcases.items():
    kw = dict(kw)
    image = torch.tensor(rng.uniform(0, 1, size=(3, H, W)).astype(np.float32), requires_grad=True)
    depth = torch.tensor(rng.uniform(0.2, 5.0, size=(1, H, W)).astype(np.float32), requires_grad=True)
    opacity = torch.tensor(rng.uniform(0.6, 1.0, size=(1, H, W)).astype(np.float32))
    a = torch.nn.Parameter(torch.tensor([0.07]))
    b = torch.nn.Parameter(torch.tensor([-0.03]))
    vp = types.SimpleNamespace(original_image=torch.tensor(gt_image), depth=gt_depth, exposure_a=a, exposure_b=b,
                               motion_mask=torch.tensor(motion), grad_mask=torch.tensor(grad_mask), uid=3)
    use_mask = kw.pop("use_mask", False)
    loss = ref.get_loss_tracking(tcfg, image, depth, opacity, vp, mask=torch.tensor(mask) if use_mask else None, **kw)
    loss.backward()
    out[f"{name}/image"], out[f"{name}/depth"], out[f"{name}/opacity"] = image.detach().numpy(), depth.detach().numpy(), opacity.numpy()
    out[f"{name}/loss"] = loss.item()
    out[f"{name}/g_image"], out[f"{name}/g_depth"] = image.grad.numpy(), depth.grad.numpy()
    out[f"{name}/g_a"], out[f"{name}/g_b"] = a.grad.numpy(), b.grad.numpy()
    out[f"{name}/flags"] = np.array([kw.get("rm_dynamic", False), use_mask])
# ---- SSIM (gaussian_splatting/utils/loss_utils.py:63-111); the module imports cv2 at the top (absent here, unused by ssim)
if "cv2" not in sys.modules:
    try:
        import cv2  # noqa: F401
    except Exception:
        sys.modules["cv2"] = types.ModuleType("cv2")
import gaussian_splatting.utils.loss_utils as lu            # noqa: E402
for name, use_mask, hw in (("ssim_plain", False, (H, W)), ("ssim_mask", True, (H, W)), ("ssim_ragged", False, (37, 53))):
    hh, ww = hw
    img1 = torch.tensor(rng.uniform(0, 1, size=(3, hh, ww)).astype(np.float32), requires_grad=True)
    img2 = torch.tensor(np.clip(img1.detach().numpy() + rng.normal(scale=0.1, size=(3, hh, ww)), 0, 1).astype(np.float32))
    mk = torch.tensor(rng.uniform(size=(hh, ww)) > 0.3) if use_mask else None
    val = lu.ssim(img1, img2, mask=mk)
    val.backward()
    out[f"{name}/img1"], out[f"{name}/img2"], out[f"{name}/value"], out[f"{name}/g_img1"] = img1.detach().numpy(), img2.numpy(), val.item(), img1.grad.numpy()
    out[f"{name}/mask"] = mk.numpy() if mk is not None else np.zeros(0, bool)
out["ssim_cases"] = np.array(["ssim_plain", "ssim_mask", "ssim_ragged"])
np.savez_compressed(os.path.join(HERE, "golden_loss.npz"), **out)
print("wrote golden_loss.npz", {k: out[f"{k}/loss"] for k in cases})
