"""Keyframe selection and the sliding keyframe window of the tracking front-end, as batched device-side statistics.

Behaviour being reproduced (decisions only -- the golden vectors of tests/golden/golden_slam.npz were recorded from the reference's
``FrontEnd.is_keyframe`` / ``add_to_window``, utils/slam_frontend.py:472-562):

  * a tracked frame becomes a keyframe when it has moved far enough from the newest keyframe (relative to the scene's median depth),
    or when it has moved a little AND shares too few visible Gaussians with it (intersection over union of the two visibility sets);
  * a new keyframe goes to the front of the window; of the older entries (everything behind the two newest) the LAST one whose
    Szymkiewicz-Simpson overlap with the new keyframe is at or below the cut-off is dropped; if the window is still over its size,
    the entry that is far from the new keyframe yet crowded by the others -- largest sqrt(distance to the new keyframe) x sum of
    inverse distances to the other old entries -- is dropped as well.

How it is computed here: the visibility sets of the window are stacked into one [K, P] matrix, so every overlap count is one masked
reduction; "translation of T_i T_j^-1" is the distance between the two camera centres (a rotation does not change a norm), so all
pairwise distances are one ``cdist`` of the centres; both eviction decisions are taken on the device and reach the host as ONE
two-element transfer. The reference walks the window with Python loops and synchronises on every pair (``.item()``, 4x4 inverses).
"""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class KeyframeThresholds:
    translation: float          # kf_translation: keyframe if moved further than this x median depth
    min_translation: float      # kf_min_translation: ... or further than this and the overlap is low
    overlap: float              # kf_overlap: intersection over union below which a frame is "new"
    cutoff: float               # kf_cutoff: overlap coefficient at or below which an old window entry is dropped
    window_size: int

    @classmethod
    def from_config(cls, config):
        t = config["Training"]
        return cls(float(t["kf_translation"]), float(t["kf_min_translation"]), float(t["kf_overlap"]), float(t.get("kf_cutoff", 0.4)),
                   int(t.get("window_size", 8)))


def camera_centres(cameras):
    """World-space centres c = -R^T T of world-to-camera poses, stacked [K, 3]."""
    R = torch.stack([torch.as_tensor(c.R, dtype=torch.float32) for c in cameras])
    T = torch.stack([torch.as_tensor(c.T, dtype=torch.float32) for c in cameras])
    return -torch.einsum("kji,kj->ki", R, T)


def visibility_matrix(visibility, ids, like):
    """[K, P] boolean matrix of the visibility sets of keyframes `ids` (on the device of `like`)."""
    return torch.stack([visibility[k].to(like.device) != 0 for k in ids]) if ids else torch.zeros((0, like.shape[0]), dtype=torch.bool, device=like.device)


def overlap_counts(current, others):
    """|current AND other_k|, |other_k| for every row k of `others` [K, P], and |current|: three reductions, no host round trip."""
    cur = current != 0
    inter = (others & cur).sum(dim=1)
    return inter, others.sum(dim=1), cur.sum()


def frame_statistics(cur_cam, ref_cam, cur_visibility, ref_visibility):
    """(intersection over union of the two visibility sets, distance between the two cameras) as one 2-element device tensor."""
    ref = visibility_matrix({0: ref_visibility}, [0], cur_visibility)
    inter, n_ref, n_cur = overlap_counts(cur_visibility, ref)
    iou = inter[0] / (n_ref[0] + n_cur - inter[0])
    c = camera_centres([cur_cam, ref_cam]).to(iou.device)
    return torch.stack([iou.to(torch.float32), (c[0] - c[1]).norm()])


def keyframe_decision(stats, median_depth, thr: KeyframeThresholds):
    """The reference's is_keyframe on (iou, distance) -- a 0-dim boolean tensor (the caller decides when to synchronise)."""
    iou, dist = stats[0], stats[1]
    far = dist > thr.translation * median_depth
    moved = dist > thr.min_translation * median_depth
    return far | (moved & (iou < thr.overlap))


def window_evictions(new_cam, old_cams, new_visibility, old_visibility, cutoff, window_size, keep_newest=2):
    """Which entries leave the window when `new_cam` is pushed to its front.

    old_cams / old_visibility: the window before the push, newest first ([K] cameras, [K, P] matrix). The first `keep_newest - 1`
    of them (with the new keyframe: the `keep_newest` newest) are never dropped. Returns a device tensor (low_overlap, crowded) of
    indices into the OLD window, -1 where nothing is dropped; `crowded` already accounts for the entry `low_overlap` removed."""
    K = len(old_cams)
    dev = new_visibility.device
    none = torch.full((), -1, dtype=torch.long, device=dev)
    first = keep_newest - 1                                  # first old index that may be dropped
    if K <= first:
        return torch.stack([none, none])
    idx = torch.arange(K, device=dev)
    candidate = idx >= first
    inter, n_old, n_new = overlap_counts(new_visibility, old_visibility)
    coeff = inter / torch.minimum(n_old, n_new)              # overlap coefficient (0 / 0 -> nan: compares false, like the reference)
    weak = candidate & (coeff <= cutoff)
    low = torch.where(weak.any(), (idx * weak).max(), none)  # the LAST weak entry
    alive = candidate & (idx != low)
    over = (K + 1 - (low >= 0).long()) > window_size         # window still too long after the first eviction?
    centres = camera_centres(list(old_cams) + [new_cam]).to(dev)
    pair = torch.cdist(centres[:K], centres[:K])             # distances among the old entries
    inv = 1.0 / (pair + 1e-6)
    inv = inv * (alive[None, :] & alive[:, None] & ~torch.eye(K, dtype=torch.bool, device=dev))
    score = (centres[:K] - centres[K]).norm(dim=1).sqrt() * inv.sum(dim=1)
    score = torch.where(alive, score, torch.full_like(score, -float("inf")))
    crowded = torch.where(over & alive.any(), score.argmax(), none)
    return torch.stack([low, crowded])


# ---- co-visible older keyframes of the dynamic mapping loop: Camera.keyframe_selection_overlap (utils/camera_utils.py:319-365) ------------
def backprojected_points(depth, R, T, fx, fy, cx, cy):
    """World-space points of every pixel with depth > 0 of a keyframe at pose [R | T] (world-to-camera), as get_pointcloud builds them
    (utils/camera_utils.py:236-265) -- INCLUDING its duplicate filter: the points are rounded to 0.1 mm, their absolute values compared,
    and every point whose rounded |x|, |y|, |z| coincide with another point's (or with the origin's) is dropped, all copies of it (the
    reference's comment says "remove points at camera origin"; torch.unique(return_counts) + isin does more than that). [N, 3] on the
    depth map's device."""
    depth = torch.as_tensor(depth, dtype=torch.float32)
    dev = depth.device
    v, u = torch.where(depth > 0)
    z = depth[v, u]
    cam = torch.stack(((u - cx) / fx * z, (v - cy) / fy * z, z), dim=-1)
    R = torch.as_tensor(R, dtype=torch.float32, device=dev)
    T = torch.as_tensor(T, dtype=torch.float32, device=dev)
    pts = (cam - T) @ R                                          # c2w = [R^T | -R^T T]:  R^T (p - T), row-vector form
    A = torch.abs(torch.round(pts, decimals=4))
    _, idx, counts = torch.cat([A, torch.zeros((1, 3), device=dev)], dim=0).unique(dim=0, return_inverse=True, return_counts=True)
    keep = counts[idx[:A.shape[0]]] == 1
    return pts[keep]


OVERLAP_GROUP = 16      # candidate keyframes per batched projection of overlap_fractions


def overlap_fractions(points, Rs, Ts, fx, fy, cx, cy, width, height, edge=20):
    """For K candidate keyframes with poses Rs [K,3,3], Ts [K,3]: the fraction of `points` [N,3] that project inside the image minus an
    `edge`-pixel border, in front of the camera (:339-355) -- a [K] float32 device tensor. The candidates are projected in groups of
    `OVERLAP_GROUP` (the [k, N, 3] camera-frame points and their [k, N] temporaries are the peak: ~40 bytes per point and candidate, i.e.
    0.2 GB per group at 307 200 points; the reference loops per keyframe -- O(N) -- and a run of a few hundred keyframes must not need
    gigabytes here), the group results concatenated on the device: still one host transfer for the caller."""
    fractions = []
    for lo in range(0, Rs.shape[0], OVERLAP_GROUP):
        pc = torch.einsum("kij,nj->kni", Rs[lo:lo + OVERLAP_GROUP], points) + Ts[lo:lo + OVERLAP_GROUP, None, :]   # [k, N, 3]
        z = pc[..., 2] + 1e-5
        px, py = (fx * pc[..., 0] + cx * pc[..., 2]) / z, (fy * pc[..., 1] + cy * pc[..., 2]) / z
        inside = (px < width - edge) & (px > edge) & (py < height - edge) & (py > edge) & (z > 0)
        fractions.append(inside.sum(dim=1) / points.shape[0])
    return fractions[0] if len(fractions) == 1 else torch.cat(fractions)


def keyframe_selection_overlap(newest, viewpoints, time, intrinsics, pose_window=3, permutation=None):
    """Camera.keyframe_selection_overlap(dataset, viewpoints, time) of keyframe `newest` (utils/camera_utils.py:319-365): the keyframes with
    id < `time`, ordered by the fraction of `newest`'s depth samples they see (descending, ties in the dictionary's order), those with a
    fraction of zero dropped, then a random permutation of that list cut to 8 - pose_window entries. The reference permutes with numpy's
    global generator (``np.random.permutation``); here the default draw comes from torch's CPU generator like every other random choice of
    the back-end (the ranks of a sharded run are seeded alike through torch.manual_seed) -- pass ``permutation=np.random.permutation`` for
    the reference's stream. intrinsics = (fx, fy, cx, cy, width, height).
    All candidates are projected in one batched operation on the device of the depth map; ONE host transfer (their fractions)."""
    fx, fy, cx, cy, width, height = intrinsics
    ids = [k for k in viewpoints if k < time]
    if not ids:
        return []
    depth = newest.depth_device() if hasattr(newest, "depth_device") else torch.as_tensor(newest.depth, dtype=torch.float32)
    pts = backprojected_points(depth, newest.R, newest.T, fx, fy, cx, cy)
    if pts.shape[0] == 0:
        return []
    dev = pts.device
    Rs = torch.stack([torch.as_tensor(viewpoints[k].R, dtype=torch.float32).to(dev) for k in ids])
    Ts = torch.stack([torch.as_tensor(viewpoints[k].T, dtype=torch.float32).to(dev) for k in ids])
    frac = overlap_fractions(pts, Rs, Ts, fx, fy, cx, cy, width, height).tolist()
    order = sorted(range(len(ids)), key=lambda i: frac[i], reverse=True)          # (stable: ties keep the dictionary's order, like sorted() there)
    selected = np.array([ids[i] for i in order if frac[i] > 0.0])
    if permutation is None:
        permutation = lambda a: a[torch.randperm(len(a)).numpy()]
    return [int(k) for k in permutation(selected)[:8 - pose_window]]
