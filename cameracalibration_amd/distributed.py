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
