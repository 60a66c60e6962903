"""View sharding for multi-GPU solves (one process per GPU).

Views are independent given the intrinsics, so each rank keeps a contiguous
range of views balanced by corner-observation count; the solver drivers
(solver.py) all-reduce only the small global records.  No reference
counterpart: the reference calibrates in one process
(controller/controller_internal_calibration.py:371-402).
"""
from __future__ import annotations

import numpy as np


def shard_ranges(n_per_view, world):
    """Contiguous view ranges [(lo, hi)] * world with near-equal observation
    counts (prefix-sum split)."""
    n = np.asarray(n_per_view, dtype=np.int64)
    N = n.shape[0]
    if world <= 1:
        return [(0, N)]
    csum = np.concatenate([[0], np.cumsum(n)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        k = min(max(k, cuts[-1]), N)
        cuts.append(k)
    cuts.append(N)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_lists(object_points, image_points, rank, world):
    """This rank's slice of cv2-style view lists."""
    n = [len(o) for o in object_points]
    lo, hi = shard_ranges(n, world)[rank]
    return list(object_points[lo:hi]), list(image_points[lo:hi]), (lo, hi)


def shard_rig_lists(object_points, image_points, rank, world):
    """This rank's slice of rig inputs (objectPoints[c][p]): board poses are
    split into contiguous ranges balanced by the observations all cameras hold
    of them, so every camera's view of a pose stays on one rank."""
    P = len(object_points[0])
    n = np.zeros(P, dtype=np.int64)
    for cam in object_points:
        for p, o in enumerate(cam):
            if o is not None:
                n[p] += len(o)
    lo, hi = shard_ranges(n, world)[rank]
    return ([list(cam[lo:hi]) for cam in object_points], [list(cam[lo:hi]) for cam in image_points], (lo, hi))


def shard_rig_cams(cams, n_poses, rank, world):
    """The packed form of :func:`shard_rig_lists`: ``cams`` = per camera dict(pose_idx (sorted, numpy),
    view_ptr, obj, img) as rig.calibrate_rig(cams=...) takes them.  Returns (cams of this rank with
    pose indices rebased to its range, poses in the range, (lo, hi))."""
    n = np.zeros(int(n_poses), dtype=np.int64)
    for c in cams:
        pi = np.asarray(c["pose_idx"], dtype=np.int64)
        vp = np.asarray(c["view_ptr"].cpu() if hasattr(c["view_ptr"], "cpu") else c["view_ptr"], dtype=np.int64)
        if pi.size > 1 and np.any(np.diff(pi) <= 0):
            raise ValueError("pose_idx must be strictly increasing per camera")
        n[pi] += np.diff(vp)
    lo, hi = shard_ranges(n, world)[rank]
    out = []
    for c in cams:
        pi = np.asarray(c["pose_idx"], dtype=np.int64)
        vp = c["view_ptr"]
        a, b = int(np.searchsorted(pi, lo, side="left")), int(np.searchsorted(pi, hi, side="left"))
        p0, p1 = int(vp[a]), int(vp[b])
        out.append(dict(pose_idx=pi[a:b] - lo, view_ptr=vp[a:b + 1] - vp[a], obj=c["obj"][p0:p1], img=c["img"][p0:p1]))
    return out, hi - lo, (lo, hi)
