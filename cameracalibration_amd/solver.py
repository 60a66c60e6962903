"""Solver drivers: the control flow of cv2.calibrateCamera's CvLevMarq loop and
of cv2.fisheye.calibrate's damped Gauss-Newton loop, with every numeric stage
executed by libcamcal.so on the device (see engine.py / include/camcal.h).

The host only sequences stages and reads back a handful of scalars per
iteration (cost, step norms) to take the accept / reject / stop decisions that
``CvLevMarq::updateAlt`` and ``cv::fisheye::calibrate`` take
[OpenCV 4.10 compat_ptsetreg.cpp, fisheye.cpp; SURVEY.md 8(c)-appendix].

Multi-GPU: views are sharded across ranks (one process per GPU).  The only
exchanged data are the small global records -- ``[U|b|cost|C|d]`` per fisheye
iteration; ``[C|d]`` and ``[U|b|cost|step norms]`` per pinhole trial step --
summed inside the library's loops through a ``camcal_comm`` (its own RCCL
communicator on GPUs, a torch.distributed callback over gloo in the CPU tests).
Every rank then solves the reduced system redundantly and back-substitutes its
own poses.  The loops below are the logging path (``_info={"log": True}``).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

DBL_EPSILON = float(np.finfo(np.float64).eps)


# library RCCL communicators, one per (process group, device): creating one costs a
# rendezvous, so repeated drop-in calls share it; they live until the process ends
_RCCL_COMMS = {}


def _all_reduce(t, group, op=None):
    """dist.all_reduce that is safe for device tensors on a non-NCCL group: gloo's own path for CUDA
    tensors hangs on this stack when the tensor's stream still has work queued (tools/gloo_cuda_probe.py),
    so such a tensor is staged through the host (the copy synchronises the stream)."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class Comm:
    """Sum-all-reduce of the small global records; identity for a single rank.

    The solver loops run inside the library (camcal_fisheye_gn / camcal_pinhole_lm) and
    call the communicator between their stages through a ``camcal_comm`` handle
    (include/camcal.h), made by :meth:`handle`:

    * ``backend="rccl"`` (default on ROCm devices with the nccl process group): the library's
      own RCCL communicator -- one ``ncclAllReduce`` on the compute stream per record, no
      Python and no host synchronisation inside an iteration.  Its unique id travels from
      rank 0 through ``torch.distributed``.
    * ``backend="torch"`` (gloo in the CPU tests; ``CAMCAL_COMM=torch`` forces it): a callback
      that runs ``torch.distributed.all_reduce`` on the engine's own tensor.

    The Python loops of this module (iteration log) use :meth:`allreduce` directly."""

    def __init__(self, group=None, enabled=None, backend=None):
        if enabled is None:
            enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.enabled = bool(enabled)
        self.group = group
        self.backend = backend or os.environ.get("CAMCAL_COMM") or None
        self._keep = {}            # id(engine) -> (weakref, callback, struct, pools): must outlive the calls
        self.kind = None           # what the last handle() returned: "rccl" | "torch" | None

    def allreduce(self, t: torch.Tensor):
        if self.enabled:
            _all_reduce(t, self.group)
        return t

    def total(self, x: float, device) -> float:
        if not self.enabled:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64)       # a host scalar: no device work to order
        t = _all_reduce(t.to(device) if dist.get_backend(self.group) == "nccl" else t, self.group)
        return float(t.item())

    # -- camcal_comm handles --------------------------------------------------------------
    def _rccl_handle(self, lib, device):
        # keyed by the ranks of the group (a group object's id() can be reused after garbage collection)
        key = (tuple(dist.get_process_group_ranks(self.group)) if self.group is not None else None,
               device.index if device.index is not None else torch.cuda.current_device())
        if key in _RCCL_COMMS:
            return _RCCL_COMMS[key][1]
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        ident = torch.zeros(128, dtype=torch.uint8)
        ok = torch.ones(1, dtype=torch.int32)
        if rank == 0:
            buf = (C.c_char * 128)()
            if lib.camcal_rccl_unique_id(C.addressof(buf)) != 0:
                ok[0] = 0
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        on_dev = dist.get_backend(self.group) == "nccl"
        pack = torch.cat([ok.view(torch.uint8), ident])
        pack = pack.to(device) if on_dev else pack
        dist.broadcast(pack, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        pack = pack.cpu()
        if int(pack[:4].view(torch.int32).item()) == 0:
            return None
        idb = bytes(pack[4:].numpy().tobytes())
        out = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.camcal_comm_rccl_create(idb, rank, world, C.byref(out))
        good = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device if on_dev else "cpu")
        dist.all_reduce(good, op=dist.ReduceOp.MIN, group=self.group)     # all ranks or none
        if int(good.item()) == 0:
            if rc == 0:
                lib.camcal_comm_destroy(out)
            return None
        _RCCL_COMMS[key] = (lib, out)
        return out

    def _torch_handle(self, E):
        """Callback communicator: the buffers the loops reduce are slices of tensors the
        engine owns (``E._comm_pools()``)."""
        pools = [(t.data_ptr(), t.numel() * 8, t) for t in E._comm_pools()]
        group = self.group

        def cb(ctx, buf, n, stream):
            try:
                for base, nbytes, t in pools:
                    off = int(buf) - base
                    if 0 <= off and off % 8 == 0 and off + 8 * n <= nbytes:
                        _all_reduce(t.view(-1)[off // 8: off // 8 + n], group)
                        return 0
                return -1
            except Exception:          # never unwind through the C frames
                return -5
        fn = _lib.ALLREDUCE_FN(cb)
        st = _lib.CommStruct(fn, None)
        # one handle per engine, kept only while the engine lives (a long-lived Comm used for many
        # solves must not pin every engine's device pool)
        import weakref
        key = id(E)
        self._keep = {k: v for k, v in self._keep.items() if v[0]() is not None}
        self._keep[key] = (weakref.ref(E), fn, st)
        return C.pointer(st)

    def handle(self, E):
        """``camcal_comm*`` for the library's loops on engine ``E`` (None for one rank)."""
        self.kind = None
        if not self.enabled:
            return None
        want = self.backend
        if want is None:
            want = "rccl" if (E.device.type == "cuda" and not E.lib.is_emulator
                              and dist.get_backend(self.group) == "nccl") else "torch"
        if want == "rccl":
            h = self._rccl_handle(E.lib, E.device)
            if h is not None:
                self.kind = "rccl"
                return h
            import warnings
            warnings.warn("camcal: RCCL communicator unavailable, using torch.distributed for the records")
        self.kind = "torch"
        return self._torch_handle(E)

    def close(self):
        """Drop the callback handles of this object (library RCCL communicators are shared
        and stay; :func:`destroy_rccl_comms` ends them -- collectively, on every rank)."""
        self._keep.clear()


def destroy_rccl_comms():
    for lib, h in _RCCL_COMMS.values():
        lib.camcal_comm_destroy(h)
    _RCCL_COMMS.clear()


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
    if log is None:
        return E.solve_pinhole(mask, max_iter, eps, comm=comm.handle(E))   # same loop, sequenced by the library
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
    if log is None:
        return E.solve_fisheye(mask, (ctype, maxc, eps), recompute_extrinsic, comm=comm.handle(E))   # same loop, in the library
    change, it = 1.0, 0
    cost_idx = E.glob_len - 1
    while True:
        if (ctype == 1 and it >= maxc) or (ctype == 2 and change <= eps) or \
           (ctype == 3 and (change <= eps or it >= maxc)):
            break
        alpha2 = 1.0 - 0.6 ** (it + 1.0)
        predict = recompute_extrinsic and E.PREDICT_POSES
        if predict:
            E.normal_eq_response(0, 0)
        else:
            E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
        comm.allreduce(E.rec)
        E.solve_reduced(0, mask, 0.0, -alpha2, 0, 0)
        st = E.stats[:5].cpu().numpy()
        change = math.sqrt(st[2]) / math.sqrt(st[3])
        if predict:
            E.pose_predict(0, -alpha2)
        if recompute_extrinsic:
            E.pose_refine(0, 0, 20, 1e-10)
        it += 1
        if log is not None:
            log.append(dict(iter=it, change=change, intr=E.get_intr(0)))
    E.intr_normal(0, 0, 0, with_normal=False)
    comm.allreduce(E.glob[0][cost_idx:cost_idx + 1])
    return float(E.glob[0][cost_idx].item()), it
