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
