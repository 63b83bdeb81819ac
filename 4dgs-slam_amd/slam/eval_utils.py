"""Trajectory and rendering evaluation -- the counterpart of the reference's ``utils/eval_utils.py``: Horn alignment + ATE
(align :160-200, evaluate_ate :203-219, the evo APE-RMSE of evaluate_evo :106-150), eval_ate (:221-297), eval_rendering (:300-428:
masked PSNR, SSIM, depth L1; LPIPS needs the AlexNet weights and is left out), save_gaussians (:431-440). No evo / wandb / cv2."""
import json
import os

import numpy as np
import torch

import slam_losses
from gaussian_renderer import render


def align(model, data):
    """Horn's closed-form rigid alignment of two 3xn point sets (utils/eval_utils.py:160-200): returns (rot 3x3, trans 3x1,
    per-point translational error after alignment [n])."""
    model, data = np.asarray(model, np.float64), np.asarray(data, np.float64)
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = np.zeros((3, 3))
    for c in range(model.shape[1]):
        W += np.outer(mz[:, c], dz[:, c])
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    err = rot @ model + trans - data
    return rot, trans, np.sqrt(np.sum(err * err, 0))


def evaluate_ate(gt_traj, est_traj):
    """utils/eval_utils.py:203-219: mean translational error after aligning the GROUND TRUTH onto the estimate (the reference passes
    (gt, est) into align(model, data)); poses are 4x4 camera-to-world matrices."""
    gt = np.stack([np.asarray(p)[:3, 3] for p in gt_traj]).T
    est = np.stack([np.asarray(p)[:3, 3] for p in est_traj]).T
    _, _, err = align(gt, est)
    return float(err.mean())


def ate_rmse(gt_traj, est_traj):
    """What evaluate_evo reports (:106-128): RMSE of the translation part after aligning the estimate onto the ground truth
    (evo's align_trajectory without scale = the same Horn / Umeyama rigid fit)."""
    gt = np.stack([np.asarray(p)[:3, 3] for p in gt_traj]).T
    est = np.stack([np.asarray(p)[:3, 3] for p in est_traj]).T
    _, _, err = align(est, gt)
    return float(np.sqrt(np.mean(err * err)))


def _pose_c2w(R, T):
    m = np.eye(4)
    m[:3, :3] = R.detach().cpu().numpy()
    m[:3, 3] = T.detach().cpu().numpy()
    return np.linalg.inv(m)


def eval_ate(frames, kf_ids, save_dir=None, iterations=0, final=False, monocular=False):
    """utils/eval_utils.py:221-297 without the plots: trajectory json + ATE. `frames` = {frame id: Camera}. Returns the RMSE; with
    final=True every tracked frame counts (:238-249), otherwise the keyframes."""
    ids = sorted(frames.keys()) if final else list(kf_ids)
    est = [_pose_c2w(frames[i].R, frames[i].T) for i in ids]
    gt = [_pose_c2w(frames[i].R_gt, frames[i].T_gt) for i in ids]
    out = {"trj_id": ids, "ate_rmse": ate_rmse(gt, est) if len(ids) >= 3 else 0.0, "ate_mean": evaluate_ate(gt, est) if len(ids) >= 3 else 0.0}
    if save_dir:
        plot_dir = os.path.join(save_dir, "plot")
        os.makedirs(plot_dir, exist_ok=True)
        label = "final" if final else "{:04}".format(iterations)
        with open(os.path.join(plot_dir, f"trj_{label}.json"), "w", encoding="utf-8") as f:
            json.dump({**out, "trj_est": [p.tolist() for p in est], "trj_gt": [p.tolist() for p in gt]}, f, indent=1)
    return out["ate_rmse"]


def psnr(img1, img2):
    """gaussian_splatting/utils/image_utils.py:19-21."""
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


@torch.no_grad()
def eval_rendering(frames, gaussians, dataset, save_dir, pipe, background, kf_indices=(), iteration="final", deltas_for=None, interval=1):
    """utils/eval_utils.py:300-428: render every frame at its estimated pose and compare with the sensor image: PSNR over the valid
    pixels (:371-381), SSIM, depth L1 (:383-388). `deltas_for(frame)` -> (dx, ds, dr) supplies the dynamic subset's deformation."""
    psnrs, ssims, depths = [], [], []
    end_idx = len(frames) - 1 if len(frames) > 1 else 1
    for idx in range(0, end_idx, interval):
        frame = frames[idx]
        gt_image, gt_depth, _, motion_mask = dataset[idx]
        dx, ds, dr = deltas_for(frame) if deltas_for is not None else (0, 0, 0)
        pkg = render(frame, gaussians, pipe, background, dynamic=False, dx=dx, ds=ds, dr=dr)
        image = torch.clamp(pkg["render"], 0.0, 1.0)
        valid_depth = torch.as_tensor(gt_depth > 0, device=image.device)
        mask = gt_image > 0
        if deltas_for is None and motion_mask is not None:                        # static map: the moving region is not evaluated (:372-376)
            mask = mask & motion_mask.view(1, *valid_depth.shape) & valid_depth[None]
            valid_depth = valid_depth & motion_mask
        else:
            mask = mask & valid_depth[None]
        psnrs.append(float(psnr(image[mask].unsqueeze(0), gt_image[mask].unsqueeze(0))))
        ssims.append(float(slam_losses.ssim(image, gt_image)))
        l1 = torch.abs(torch.as_tensor(gt_depth, device=image.device)[None] - pkg["depth"]) * valid_depth[None]
        depths.append(float(l1.sum() / (valid_depth.sum() + 1e-7)))
    out = {"mean_psnr": float(np.mean(psnrs)), "mean_ssim": float(np.mean(ssims)), "l1_depth": float(np.mean(depths)), "frames": len(psnrs)}
    if save_dir:
        d = os.path.join(save_dir, "psnr", str(iteration))
        os.makedirs(d, exist_ok=True)
        json.dump(out, open(os.path.join(d, "final_result.json"), "w", encoding="utf-8"), indent=1)
    return out


def save_gaussians(gaussians, name, iteration, final=False):
    """utils/eval_utils.py:431-440."""
    if name is None:
        return
    path = os.path.join(name, "point_cloud/final" if final else "point_cloud/iteration_{}".format(str(iteration)))
    gaussians.save_ply(os.path.join(path, "point_cloud.ply"))
