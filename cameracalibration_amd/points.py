"""Point-level cv2 entry points the reference calls either side of the solver
(SURVEY.md 8f rows N1-N4), computed by libcamcal.so on the device:

* :func:`projectPoints`, :func:`fisheye_projectPoints`, :func:`norm`  -- the
  reprojection-error loops (server/aruco/camera_calib.py:96-104,174-181;
  server/internal/IntrinsicCalibration/intrinsicCalib.py:137-146,229-238);
  :func:`reproj_errors` does a whole list of views in one launch.
* :func:`undistortPoints`, :func:`fisheye_undistortPoints`
  (camera_calib.py:241,252,261,281,292,301;
  controller/controller_internal_calibration.py:553,570).
* :func:`solvePnP` (camera_calib.py:243,283), :func:`solvePnP_batch`.
* :func:`initUndistortRectifyMap`, :func:`fisheye_initUndistortRectifyMap`
  (server/video/video_server.py:473,497; intrinsicCalib.py:159,251).
* :func:`Rodrigues`, :func:`checkRange` -- the two scalar helpers those call
  sites also use (host side).

Signatures, shapes and dtypes follow the cv2 4.10 Python binding.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import api as _api
from .api import _gather_views, error
from .engine import Engine

NORM_L2 = 4
INTER_LINEAR = 1
BORDER_CONSTANT = 0
CV_16SC2 = 11
CV_32FC1 = 5
SOLVEPNP_ITERATIVE = 0


def _ctx():
    lib, dev = _api._resolve()
    lib = lib if lib is not None else _lib.load()
    if dev.type != "cuda" and not lib.is_emulator:
        raise _lib.CamcalError("the HIP library needs a cuda (ROCm) device; there is no CPU path")
    return lib, dev


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None


def _call(lib, dev, name, *args):
    if dev.type == "cuda":
        with torch.cuda.device(dev):
            rc = getattr(lib, name)(*args)
    else:
        rc = getattr(lib, name)(*args)
    lib.check(rc, name)


def _intr(model, K, D):
    K = np.asarray(K, dtype=np.float64)
    if K.shape != (3, 3):
        raise error("cameraMatrix must be 3x3")
    d = np.zeros(0) if D is None else np.asarray(D, dtype=np.float64).ravel()
    if model == _lib.FISHEYE:
        if d.size != 4:
            raise error("D.total() == 4")
        if K[0, 1] != 0.0:
            raise error("a non-zero skew is not supported")
        dd = d
    else:
        if d.size not in (0, 4, 5):
            raise error("distCoeffs must hold 0, 4 or 5 coefficients (k1 k2 p1 p2 [k3])")
        dd = np.zeros(5)
        dd[:d.size] = d
    v = np.zeros(16)
    v[:4] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    v[4:4 + dd.size] = dd
    return v


def _to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)


# --- N1: projection and the reprojection error ---------------------------------
def _project(model, objectPoints, rvec, tvec, K, D):
    lib, dev = _ctx()
    op = np.asarray(objectPoints)
    if op.size % 3:
        raise error("objectPoints must be Nx3")
    out_dtype = np.float32 if op.dtype == np.float32 else np.float64
    n = op.size // 3
    r = np.asarray(rvec, dtype=np.float64).ravel()
    if r.size == 9:
        r = Rodrigues(r.reshape(3, 3))[0].ravel()
    pose = np.concatenate([r, np.asarray(tvec, dtype=np.float64).ravel()])
    if pose.size != 6:
        raise error("rvec / tvec must hold 3 values each")
    t_xyz = _to_dev(op.reshape(-1, 3), dev)
    t_pose, t_intr = _to_dev(pose, dev), _to_dev(_intr(model, K, D), dev)
    t_uv = torch.empty(n, 2, dtype=torch.float64, device=dev)
    _call(lib, dev, "camcal_project_points", model, n, t_xyz.data_ptr(), t_pose.data_ptr(), t_intr.data_ptr(),
          t_uv.data_ptr(), _stream(dev))
    return t_uv.cpu().numpy().astype(out_dtype).reshape(n, 1, 2)


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs, imagePoints=None, jacobian=None,
                  aspectRatio=0):
    """``cv2.projectPoints`` -> (imagePoints (n,1,2) in the depth of objectPoints, None);
    the Jacobian output is not provided (the reference discards it)."""
    return _project(_lib.PINHOLE, objectPoints, rvec, tvec, cameraMatrix, distCoeffs), None


def fisheye_projectPoints(objectPoints, rvec, tvec, K, D, imagePoints=None, alpha=0, jacobian=None):
    """``cv2.fisheye.projectPoints`` -> (imagePoints, None)."""
    if alpha != 0:
        raise error("a non-zero skew is not supported")
    return _project(_lib.FISHEYE, objectPoints, rvec, tvec, K, D), None


def norm(src1, src2=None, normType=NORM_L2):
    """``cv2.norm`` for NORM_L2: the element type's subtraction, double accumulation."""
    if normType != NORM_L2:
        raise error("only NORM_L2 is provided")
    a = np.asarray(src1)
    if src2 is None:
        return float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    b = np.asarray(src2)
    if a.size != b.size or a.dtype != b.dtype:
        raise error("The operation is neither 'array op array' (where arrays have the same size and type)")
    d = (a.reshape(-1) - b.reshape(-1)).astype(np.float64)
    return float(np.sqrt((d * d).sum()))


def checkRange(a):
    """``cv2.checkRange`` with its default bounds: every element finite."""
    return bool(np.all(np.isfinite(np.asarray(a, dtype=np.float64))))


def reproj_errors(model, objectPoints, imagePoints, rvecs, tvecs, K, D, f32_round=True):
    """The per-view acceptance numbers ``norm(project(board_i), corners_i) / n_i``
    of a whole list of views in one launch.  Returns (errors (N,), their mean)."""
    lib, dev = _api._resolve()
    mdl = {"pinhole": _lib.PINHOLE, "fisheye": _lib.FISHEYE}[model] if isinstance(model, str) else int(model)
    view_ptr, obj, img = _gather_views(objectPoints, imagePoints, min_points=1)
    E = Engine(mdl, view_ptr, obj, img, dev, lib=lib)
    E.set_intr(0, _intr(mdl, K, D))
    n_views = len(view_ptr) - 1
    if len(rvecs) < n_views or len(tvecs) < n_views:
        raise error("one rvec / tvec per view is needed")
    poses = np.array([np.concatenate([np.asarray(rvecs[k], dtype=np.float64).ravel(),
                                      np.asarray(tvecs[k], dtype=np.float64).ravel()]) for k in range(n_views)])
    E.set_poses(0, poses)
    err, mean = E.reproj_err(0, 0, f32_round)
    return err, mean


# --- N3: undistortPoints -----------------------------------------------------------
def _undistort(model, src, K, D, dst, R, P):
    lib, dev = _ctx()
    a = np.asarray(src)
    if a.size % 2:
        raise error("points must be Nx2")
    n = a.size // 2
    M = None
    if R is not None or P is not None:
        RR = np.eye(3)
        if R is not None:
            Rm = np.asarray(R, dtype=np.float64)
            if Rm.size == 3:
                Rm = Rodrigues(Rm.reshape(3, 1))[0]
            if Rm.shape != (3, 3):
                raise error("R must be 3x3 or a rotation vector")
            RR = Rm
        if P is not None:
            Pm = np.asarray(P, dtype=np.float64)
            if Pm.shape not in ((3, 3), (3, 4)):
                raise error("P must be 3x3 or 3x4")
            RR = Pm[:, :3] @ RR
        M = _to_dev(RR.reshape(-1), dev)
    t_uv = _to_dev(a.reshape(-1, 2), dev)
    t_intr = _to_dev(_intr(model, K, D), dev)
    t_out = torch.empty(n, 2, dtype=torch.float64, device=dev)
    _call(lib, dev, "camcal_undistort_points", model, n, t_uv.data_ptr(), t_intr.data_ptr(),
          M.data_ptr() if M is not None else None, t_out.data_ptr(), _stream(dev))
    out_dtype = a.dtype if a.dtype in (np.float32, np.float64) else np.float32
    shape = a.shape if a.ndim == 3 else (n, 1, 2)
    res = t_out.cpu().numpy().astype(out_dtype).reshape(shape)
    if isinstance(dst, np.ndarray) and dst.dtype == res.dtype and dst.size == res.size and dst.flags.writeable:
        dst[...] = res.reshape(dst.shape)
        return dst
    return res


def undistortPoints(src, cameraMatrix, distCoeffs, dst=None, R=None, P=None):
    """``cv2.undistortPoints`` (5 fixed-point sweeps, then R and P)."""
    return _undistort(_lib.PINHOLE, src, cameraMatrix, distCoeffs, dst, R, P)


def fisheye_undistortPoints(distorted, K, D, undistorted=None, R=None, P=None):
    """``cv2.fisheye.undistortPoints`` (Newton on theta; -1e6 where it fails)."""
    return _undistort(_lib.FISHEYE, distorted, K, D, undistorted, R, P)


# --- N2: solvePnP ----------------------------------------------------------------------
def solvePnP_batch(objectPoints, imagePoints, cameraMatrix, distCoeffs):
    """One planar-target pose per view for a list of views, all on the device
    (homography initialisation + 6-dof refinement on the reprojection error, as
    cv2.solvePnP's SOLVEPNP_ITERATIVE does for planar targets).
    Returns (ok (N,) bool, rvecs (N,3), tvecs (N,3))."""
    lib, dev = _api._resolve()
    view_ptr, obj, img = _gather_views(objectPoints, imagePoints)
    E = Engine(_lib.PINHOLE, view_ptr, obj, img, dev, lib=lib)
    E.set_intr(0, _intr(_lib.PINHOLE, cameraMatrix, distCoeffs))
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    poses = E.get_poses(0)
    status = E.slots_to_views(E.status)
    return status == 0, poses[:, :3].copy(), poses[:, 3:].copy()


def solvePnP(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None, useExtrinsicGuess=False,
             flags=SOLVEPNP_ITERATIVE):
    """``cv2.solvePnP`` (SOLVEPNP_ITERATIVE, planar target) -> (retval, rvec (3,1), tvec (3,1))."""
    if flags != SOLVEPNP_ITERATIVE or useExtrinsicGuess:
        raise error("only SOLVEPNP_ITERATIVE without an extrinsic guess is provided")
    ok, r, t = solvePnP_batch([objectPoints], [imagePoints], cameraMatrix, distCoeffs)
    return bool(ok[0]), r[0].reshape(3, 1), t[0].reshape(3, 1)


# --- N4: undistort / rectify maps -----------------------------------------------------
def _rectify_map(model, K, D, R, newK, size, m1type):
    lib, dev = _ctx()
    if m1type not in (CV_16SC2, CV_32FC1):
        raise error("m1type must be CV_16SC2 or CV_32FC1")
    w, h = int(size[0]), int(size[1])
    if w <= 0 or h <= 0:
        raise error("size must be positive")
    Rm = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    if Rm.size == 3:
        Rm = Rodrigues(Rm.reshape(3, 1))[0]
    Pm = np.asarray(K if newK is None else newK, dtype=np.float64)
    if Rm.shape != (3, 3) or Pm.shape not in ((3, 3), (3, 4)):
        raise error("R must be 3x3 and the new camera matrix 3x3 or 3x4")
    iR = np.linalg.inv(Pm[:, :3] @ Rm)
    t_intr, t_iR = _to_dev(_intr(model, K, D), dev), _to_dev(iR.reshape(-1), dev)
    if m1type == CV_32FC1:
        m1 = torch.empty(h, w, dtype=torch.float32, device=dev)
        m2 = torch.empty(h, w, dtype=torch.float32, device=dev)
    else:
        m1 = torch.empty(h, w, 2, dtype=torch.int16, device=dev)
        m2 = torch.empty(h, w, dtype=torch.int16, device=dev)      # bit pattern of uint16
    _call(lib, dev, "camcal_undistort_rectify_map", model, w, h, t_intr.data_ptr(), t_iR.data_ptr(),
          0 if m1type == CV_32FC1 else 1, m1.data_ptr(), m2.data_ptr(), _stream(dev))
    a1, a2 = m1.cpu().numpy(), m2.cpu().numpy()
    return a1, (a2 if m1type == CV_32FC1 else a2.view(np.uint16))


def initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type):
    """``cv2.initUndistortRectifyMap`` -> (map1, map2)."""
    return _rectify_map(_lib.PINHOLE, cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type)


def fisheye_initUndistortRectifyMap(K, D, R, P, size, m1type):
    """``cv2.fisheye.initUndistortRectifyMap`` -> (map1, map2)."""
    return _rectify_map(_lib.FISHEYE, K, D, R, P, size, m1type)


def remap(src, map1, map2, interpolation=INTER_LINEAR, dst=None, borderMode=BORDER_CONSTANT, borderValue=0):
    """``cv2.remap`` for 8-bit images, INTER_LINEAR, BORDER_CONSTANT, with the maps of
    :func:`initUndistortRectifyMap` (float32 pair or CV_16SC2 + CV_16UC1)."""
    lib, dev = _ctx()
    if interpolation != INTER_LINEAR or borderMode != BORDER_CONSTANT:
        raise error("only INTER_LINEAR with BORDER_CONSTANT is provided")
    img = np.asarray(src)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise error("only 8-bit images of 1..4 channels are provided")
    cn = 1 if img.ndim == 2 else img.shape[2]
    m1, m2 = np.asarray(map1), np.asarray(map2)
    if m1.dtype == np.float32 and m1.ndim == 2 and m2.dtype == np.float32 and m2.shape == m1.shape:
        mt = 0
        t1, t2 = torch.from_numpy(np.ascontiguousarray(m1)).to(dev), torch.from_numpy(np.ascontiguousarray(m2)).to(dev)
    elif m1.dtype == np.int16 and m1.ndim == 3 and m1.shape[2] == 2 and m2.dtype == np.uint16 \
            and m2.shape == m1.shape[:2]:
        mt = 1
        t1 = torch.from_numpy(np.ascontiguousarray(m1)).to(dev)
        t2 = torch.from_numpy(np.ascontiguousarray(m2).view(np.int16)).to(dev)
    else:
        raise error("maps must be a float32 pair or CV_16SC2 + CV_16UC1")
    bv = int(borderValue[0] if np.ndim(borderValue) else borderValue)
    dh, dw = m1.shape[:2]
    t_src = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
    t_dst = torch.empty((dh, dw) + ((cn,) if img.ndim == 3 else ()), dtype=torch.uint8, device=dev)
    _call(lib, dev, "camcal_remap_u8", t_src.data_ptr(), img.shape[1], img.shape[0], img.shape[1] * cn, cn,
          t1.data_ptr(), t2.data_ptr(), mt, dw, dh, t_dst.data_ptr(), dw * cn, bv, _stream(dev))
    return t_dst.cpu().numpy()


# --- host-side scalar helper ---------------------------------------------------------------
def Rodrigues(src):
    """``cv2.Rodrigues`` for one vector or matrix -> (dst, None).  Host side: a
    3-vector conversion at the call sites (camera_calib.py:255,295), not worth a
    launch; the Jacobian output is not provided."""
    a = np.asarray(src, dtype=np.float64)
    if a.size == 3:
        r = a.ravel()
        th = float(np.linalg.norm(r))
        if th < np.finfo(np.float64).eps:
            return np.eye(3), None
        n = r / th
        c, s = np.cos(th), np.sin(th)
        Kx = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
        return c * np.eye(3) + (1 - c) * np.outer(n, n) + s * Kx, None
    if a.shape != (3, 3):
        raise error("Input matrix must be 1x3, 3x1 or 3x3")
    U, _, Vt = np.linalg.svd(a)
    Rm = U @ Vt
    v = np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]])
    s = np.sqrt((v * v).sum() * 0.25)
    c = np.clip((np.trace(Rm) - 1) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros((3, 1)), None
        t = (Rm[0, 0] + 1) * 0.5
        x = np.sqrt(max(t, 0.0))
        t = (Rm[1, 1] + 1) * 0.5
        y = np.sqrt(max(t, 0.0)) * (-1.0 if Rm[0, 1] < 0 else 1.0)
        t = (Rm[2, 2] + 1) * 0.5
        z = np.sqrt(max(t, 0.0)) * (-1.0 if Rm[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (Rm[1, 2] > 0) != (y * z > 0):
            z = -z
        w = np.array([x, y, z])
        w *= th / np.linalg.norm(w)
        return w.reshape(3, 1), None
    return (v * (th / (2 * s))).reshape(3, 1), None
