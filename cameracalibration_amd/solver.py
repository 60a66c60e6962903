"""Solver drivers: the control flow of cv2.calibrateCamera's CvLevMarq loop and
of cv2.fisheye.calibrate's damped Gauss-Newton loop, with every numeric stage
executed by libcamcal.so on the device (see engine.py / include/camcal.h).

The host only sequences stages and reads back a handful of scalars per
iteration (cost, step norms) to take the accept / reject / stop decisions that
``CvLevMarq::updateAlt`` and ``cv::fisheye::calibrate`` take
[OpenCV 4.10 compat_ptsetreg.cpp, fisheye.cpp; SURVEY.md 8(c)-appendix].

Multi-GPU: views are sharded across ranks (one process per GPU).  The only
exchanged data are the small global records -- ``[U|b|cost]``, ``[C|d]`` and the
step norms -- summed with one ``torch.distributed.all_reduce`` each (RCCL on
GPUs, gloo in the CPU tests).  Every rank then solves the reduced system
redundantly and back-substitutes its own poses.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

DBL_EPSILON = float(np.finfo(np.float64).eps)


class Comm:
    """Sum-all-reduce of small device records; identity for a single rank."""

    def __init__(self, group=None, enabled=None):
        if enabled is None:
            enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.enabled = bool(enabled)
        self.group = group

    def allreduce(self, t: torch.Tensor):
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def total(self, x: float, device) -> float:
        if not self.enabled:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())


def lm_pinhole(E, mask, max_iter, eps, comm=None, log=None):
    """CvLevMarq::updateAlt driven over the Schur-eliminated system.

    Expects ``E.intr[0]`` / ``E.pose[0]`` to hold the starting point.  On
    return buffer 0 holds the result.  Returns (errNorm2, iterations).

    Stage use per accepted iteration: evaluation B at the current point (W and
    the Schur complement for the current lambda), reduced solve, back-
    substitution, evaluation A at the trial point (cost, U, b and the trial's
    V, g).  A rejected trial only repeats the cheap stored-block Schur kernel."""
    comm = comm or Comm(enabled=False)
    max_iter = int(min(max(max_iter, 1), 1000))
    eps = max(float(eps), DBL_EPSILON)
    if not comm.enabled and log is None:
        return E.solve_pinhole(mask, max_iter, eps)       # same loop, sequenced by the library
    lam_lg10 = -3
    cur, tri = 0, 1
    cost_idx = E.glob_len - 1
    E.intr_normal(cur, cur, cur, with_normal=True, blocks=0)
    comm.allreduce(E.glob[cur])
    err = float(E.glob[cur][cost_idx].item())
    iters = 0

    def step(lam, fresh):
        if fresh:
            E.cross_schur(cur, cur, lam, store_w=True)
        else:
            E.schur_complement(lam)
        comm.allreduce(E.schur)
        E.solve_reduced(cur, mask, lam, -1.0, cur, tri)
        E.backsub(lam, cur, tri)
        comm.allreduce(E.norms)
        E.intr_normal(tri, tri, tri, with_normal=True, blocks=1)
        comm.allreduce(E.glob[tri])
        # one device->host readback per trial step
        rec = torch.cat([E.glob[tri][cost_idx:cost_idx + 1], E.stats[:5], E.norms]).cpu().numpy()
        return float(rec[0]), rec[1:6], rec[6:8]

    while True:
        prev_err = err
        err, stats, norms = step(10.0 ** lam_lg10, True)
        while err > prev_err:
            lam_lg10 += 1
            if lam_lg10 > 16:
                break
            err, stats, norms = step(10.0 ** lam_lg10, False)
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        rel = math.sqrt(stats[0] + norms[0]) / math.sqrt(stats[1] + norms[1])
        # accept: trial becomes current (buffers swap roles)
        E.intr[cur], E.intr[tri] = E.intr[tri], E.intr[cur]
        E.pose[cur], E.pose[tri] = E.pose[tri], E.pose[cur]
        E.glob[cur].copy_(E.glob[tri])
        E.swap_blocks()
        if log is not None:
            log.append(dict(iter=iters, err=err, rel=rel, lam_lg10=lam_lg10))
        if iters >= max_iter or rel < eps:
            return err, iters


def gn_fisheye(E, mask, criteria, recompute_extrinsic=True, comm=None, log=None):
    """cv::fisheye::calibrate's outer loop.  ``E.intr[0]`` holds the guess and
    ``E.pose[0]`` the per-view poses already initialised and refined
    (CalibrateExtrinsics).  Returns (cost, iterations)."""
    comm = comm or Comm(enabled=False)
    ctype, maxc, eps = criteria
    if not comm.enabled and log is None:
        return E.solve_fisheye(mask, (ctype, maxc, eps), recompute_extrinsic)   # same loop, in the library
    change, it = 1.0, 0
    cost_idx = E.glob_len - 1
    while True:
        if (ctype == 1 and it >= maxc) or (ctype == 2 and change <= eps) or \
           (ctype == 3 and (change <= eps or it >= maxc)):
            break
        alpha2 = 1.0 - 0.6 ** (it + 1.0)
        E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
        comm.allreduce(E.rec)
        E.solve_reduced(0, mask, 0.0, -alpha2, 0, 0)
        st = E.stats[:5].cpu().numpy()
        change = math.sqrt(st[2]) / math.sqrt(st[3])
        if recompute_extrinsic:
            E.pose_refine(0, 0, 20, 1e-10)
        it += 1
        if log is not None:
            log.append(dict(iter=it, change=change, intr=E.get_intr(0)))
    E.intr_normal(0, 0, 0, with_normal=False)
    comm.allreduce(E.glob[0][cost_idx:cost_idx + 1])
    return float(E.glob[0][cost_idx].item()), it
