"""Drop-in replacements for the two cv2 entry points the reference calls.

* :func:`calibrateCamera`      <- ``cv2.calibrateCamera``
  (reference: server/aruco/camera_calib.py:153-156,
  server/internal/IntrinsicCalibration/intrinsicCalib.py:194-197,223-226)
* :func:`fisheye_calibrate`    <- ``cv2.fisheye.calibrate``
  (reference: server/aruco/camera_calib.py:74-77,
  server/internal/IntrinsicCalibration/intrinsicCalib.py:102-105,131-134)

Same positional/keyword signature, same return tuple
``(retval, cameraMatrix, distCoeffs, rvecs, tvecs)``, K / D written in place
when the caller passes float64 arrays of the right shape (cv2 InputOutputArray
semantics), rvecs / tvecs returned as tuples of (3, 1) float64 arrays, invalid
input raises :class:`error` (cv2.error's stand-in).

Every number is computed by the HIP library; importing this module without the
built extension or calling it without a ROCm device raises.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib
from .engine import Engine, IncrementalEngine
from . import solver as _solver

DBL_EPSILON = float(np.finfo(np.float64).eps)

# --- constants (cv2 4.10 values) -------------------------------
TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_MAX_ITER = 1
TERM_CRITERIA_EPS = 2
CALIB_USE_INTRINSIC_GUESS = 1
CALIB_FIX_ASPECT_RATIO = 2
CALIB_FIX_PRINCIPAL_POINT = 4
CALIB_ZERO_TANGENT_DIST = 8
CALIB_FIX_FOCAL_LENGTH = 16
CALIB_FIX_K1 = 32
CALIB_FIX_K2 = 64
CALIB_FIX_K3 = 128
CALIB_FIX_K4 = 2048
CALIB_FIX_K5 = 4096
CALIB_FIX_K6 = 8192
CALIB_RATIONAL_MODEL = 16384
CALIB_THIN_PRISM_MODEL = 32768
CALIB_FIX_S1_S2_S3_S4 = 65536
CALIB_TILTED_MODEL = 262144
CALIB_FIX_TAUX_TAUY = 524288
CALIB_FIX_TANGENT_DIST = 2097152

FISHEYE_CALIB_USE_INTRINSIC_GUESS = 1
FISHEYE_CALIB_RECOMPUTE_EXTRINSIC = 2
FISHEYE_CALIB_CHECK_COND = 4
FISHEYE_CALIB_FIX_SKEW = 8
FISHEYE_CALIB_FIX_K1 = 16
FISHEYE_CALIB_FIX_K2 = 32
FISHEYE_CALIB_FIX_K3 = 64
FISHEYE_CALIB_FIX_K4 = 128
FISHEYE_CALIB_FIX_INTRINSIC = 256
FISHEYE_CALIB_FIX_PRINCIPAL_POINT = 512
FISHEYE_CALIB_ZERO_DISPARITY = 1024
FISHEYE_CALIB_FIX_FOCAL_LENGTH = 2048


class error(Exception):
    """Raised where cv2 would raise ``cv2.error``."""


_default_device = None


def set_device(device):
    """Select the GPU the drop-in calls run on (default: current cuda device)."""
    global _default_device
    _default_device = None if device is None else torch.device(device)


def _device():
    if _default_device is not None:
        return _default_device
    if not torch.cuda.is_available():
        raise _lib.CamcalError("no ROCm device visible: cameracalibration_amd has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _resolve():
    """(library, device) of a call: always the HIP build (``None`` = _lib.load()) on a ROCm device.
    The package holds no other route; the CPU tier of the tests replaces this function from the
    outside (tests/conftest.py: use_emulator) to run the host logic on its lane emulator."""
    return None, _device()


_NOT_PLANAR = ("only planar calibration targets with Z = 0 are supported "
               "(the reference's chessboard / ChArUco boards)")


def _gather_views(object_points, image_points, min_points=4, planar_check=True):
    """cv2-style lists of (n_i,1,3)/(n_i,1,2) arrays -> CSR float32 arrays.  ``planar_check=False``: the caller
    hands the arrays to camcal_calibrate, which makes that check itself (CAMCAL_E_UNSUPPORTED)."""
    try:
        nv = len(object_points)
        ni = len(image_points)
    except TypeError:
        raise error("objectPoints / imagePoints must be sequences of arrays")
    if nv == 0 or nv != ni:
        raise error("objectPoints and imagePoints must be non-empty and of equal length "
                    "(nimages > 0 && nimages == (int)imagePoints.total())")
    if isinstance(object_points, np.ndarray) and isinstance(image_points, np.ndarray) \
            and object_points.ndim >= 3 and image_points.ndim >= 3 \
            and object_points[0].size % 3 == 0 and image_points[0].size % 2 == 0 \
            and object_points[0].size // 3 == image_points[0].size // 2:
        # views stacked in one array each ((N,n,1,3) / (N,n,1,2) or (N,n,3) / (N,n,2)): equal counts, no gathering
        n = object_points[0].size // 3
        if n < min_points:
            raise error("every view needs at least %d points" % min_points)
        counts = np.full(nv, n, dtype=np.int64)
        objs = [np.ascontiguousarray(object_points, dtype=np.float32).reshape(-1, 3)]
        imgs = [np.ascontiguousarray(image_points, dtype=np.float32).reshape(-1, 2)]
    elif set(map(type, object_points)) == {np.ndarray} and set(map(type, image_points)) == {np.ndarray}:
        # the usual call: lists of ndarrays -- one concatenation each, no per-view conversions
        try:                                   # equal trailing shapes ((n,1,3) / (n,3)): one C call each
            cat_o = np.concatenate(object_points, axis=0)
            cat_i = np.concatenate(image_points, axis=0)
            # rows per view from the leading dimensions when a row is one point (len() is much
            # cheaper than .size over a list), else from the sizes
            if cat_o.ndim > 1 and cat_i.ndim > 1 and int(np.prod(cat_o.shape[1:])) == 3 \
                    and int(np.prod(cat_i.shape[1:])) == 2:
                counts = np.fromiter(map(len, object_points), dtype=np.int64, count=nv)
                ci = np.fromiter(map(len, image_points), dtype=np.int64, count=nv)
                so, si = counts * 3, ci * 2
            else:
                so = si = None
        except ValueError:                     # mixed layouts: flatten per view
            cat_o = np.concatenate([o.ravel() for o in object_points])
            cat_i = np.concatenate([i.ravel() for i in image_points])
            so = si = None
        if so is None:
            so = np.fromiter((o.size for o in object_points), dtype=np.int64, count=nv)
            si = np.fromiter((i.size for i in image_points), dtype=np.int64, count=nv)
        if np.any(so % 3) or np.any(si % 2):
            raise error("objectPoints must be Nx3 / imagePoints Nx2")
        counts = so // 3
        if np.any(counts != si // 2):
            raise error("number of object and image points must be equal in every view")
        if counts.min() < min_points:
            raise error("every view needs at least %d points" % min_points)
        objs = [cat_o.astype(np.float32, copy=False).reshape(-1, 3)]
        imgs = [cat_i.astype(np.float32, copy=False).reshape(-1, 2)]
    else:
        objs, imgs, counts = [], [], []
        for o, i in zip(object_points, image_points):
            o = np.asarray(o)
            i = np.asarray(i)
            if o.size % 3 or i.size % 2:
                raise error("objectPoints must be Nx3 / imagePoints Nx2")
            o = o.reshape(-1, 3).astype(np.float32, copy=False)
            i = i.reshape(-1, 2).astype(np.float32, copy=False)
            if o.shape[0] != i.shape[0]:
                raise error("number of object and image points must be equal in every view")
            if o.shape[0] < min_points:
                raise error("every view needs at least %d points" % min_points)
            objs.append(o)
            imgs.append(i)
            counts.append(o.shape[0])
    view_ptr = np.zeros(nv + 1, dtype=np.int64)
    view_ptr[1:] = np.cumsum(counts)
    obj = objs[0] if len(objs) == 1 else np.concatenate(objs, axis=0)
    img = imgs[0] if len(imgs) == 1 else np.concatenate(imgs, axis=0)
    if planar_check and obj[:, 2].any():
        raise error(_NOT_PLANAR)
    return view_ptr, obj, img


def _gather_collectively(object_points, image_points, comm, planar_check=True):
    """_gather_views; in a multi-rank call a shard that is rejected (a view with three corners, mismatched
    lists) makes EVERY rank raise, not only the one that holds it -- the others would otherwise wait in the
    first collective of the solve for a rank that has left."""
    if not comm.enabled:
        return _gather_views(object_points, image_points, planar_check=planar_check)
    exc = None
    try:
        out = _gather_views(object_points, image_points)
    except error as e:
        exc, out = e, None
    n_bad = comm.total(0.0 if exc is None else 1.0, _resolve()[1])
    if exc is not None:
        raise exc
    if n_bad > 0:
        raise error("the view lists of another rank were rejected")
    return out


def _write_back(dst, values):
    """cv2 InputOutputArray: write into the caller's array when it is a
    float64 ndarray of matching size, and return the array handed back."""
    values = np.asarray(values, dtype=np.float64)
    if isinstance(dst, np.ndarray) and dst.dtype == np.float64 and dst.size == values.size \
            and dst.flags.writeable:
        dst[...] = values.reshape(dst.shape)
        return dst
    return values


def _poses_to_tuples(poses):
    """(N, 6) -> cv2's tuples of (3, 1) arrays (each a contiguous slice of one block)."""
    n = poses.shape[0]
    rvecs = tuple(np.ascontiguousarray(poses[:, :3]).reshape(n, 3, 1))
    tvecs = tuple(np.ascontiguousarray(poses[:, 3:]).reshape(n, 3, 1))
    return rvecs, tvecs


def _raise_degenerate(E, cause=None, bad=None, comm=None, dev=None):
    """cv2 has no usable answer for a view whose points are collinear or coincident (its
    homography / SVD pose start is garbage and the solve returns non-finite values for the
    caller's checkRange); this module raises instead and names the views.  With a
    communicator the decision is collective (the count of such views is summed over the
    ranks first): every rank raises or none does, so none is left alone in a later collective."""
    if bad is None:
        bad = np.flatnonzero(E.slots_to_views(E.status) != 0)
    n_all = bad.size if comm is None else int(round(comm.total(bad.size, dev)))
    if bad.size:
        raise error("degenerate view(s) %s: the pose initialisation failed (collinear / coincident points); "
                    "remove them" % (bad[:16].tolist(),))
    if n_all:
        raise error("degenerate view(s) on another rank: the pose initialisation failed there")
    if cause is not None:
        raise error("the solve met a singular system or a non-finite cost (%s)" % cause)


def _pinhole_mask(flags, nd):
    mask = np.ones(9, dtype=bool)
    if flags & CALIB_FIX_FOCAL_LENGTH:
        mask[0] = mask[1] = False
    if flags & CALIB_FIX_PRINCIPAL_POINT:
        mask[2] = mask[3] = False
    if flags & (CALIB_ZERO_TANGENT_DIST | CALIB_FIX_TANGENT_DIST):
        mask[6] = mask[7] = False
    if flags & CALIB_FIX_K1:
        mask[4] = False
    if flags & CALIB_FIX_K2:
        mask[5] = False
    if flags & CALIB_FIX_K3:
        mask[8] = False
    return mask


def _prep_pinhole(imageSize, cameraMatrix, distCoeffs, flags, criteria):
    """Argument checks of cv2.calibrateCamera (cvCalibrateCamera2Internal) -> (D, intr0, mask, max_iter, eps)."""
    unsupported = (CALIB_FIX_ASPECT_RATIO | CALIB_RATIONAL_MODEL | CALIB_THIN_PRISM_MODEL |
                   CALIB_TILTED_MODEL)
    if flags & unsupported:
        raise error("RATIONAL / THIN_PRISM / TILTED models and FIX_ASPECT_RATIO are not supported")
    if not flags & CALIB_USE_INTRINSIC_GUESS:
        raise error("CALIB_USE_INTRINSIC_GUESS is required (cameraMatrix / distCoeffs must hold the guess)")
    if cameraMatrix is None or distCoeffs is None:
        raise error("CALIB_USE_INTRINSIC_GUESS needs cameraMatrix and distCoeffs")
    K = np.asarray(cameraMatrix, dtype=np.float64)
    D = np.asarray(distCoeffs, dtype=np.float64)
    if K.shape != (3, 3):
        raise error("cameraMatrix must be 3x3")
    if D.size not in (4, 5):
        raise error("distCoeffs must hold 4 or 5 coefficients")
    w, h = int(imageSize[0]), int(imageSize[1])
    if w <= 0 or h <= 0:
        raise error("image width and height must be positive")
    if K[0, 0] <= 0 or K[1, 1] <= 0:
        raise error("Focal length (fx and fy) must be positive")
    if K[0, 2] < 0 or K[0, 2] >= w or K[1, 2] < 0 or K[1, 2] >= h:
        raise error("Principal point must be within the image")
    if abs(K[0, 1]) > 1e-5:
        raise error("Non-zero skew is not supported by the function")
    if abs(K[1, 0]) > 1e-5 or abs(K[2, 0]) > 1e-5 or abs(K[2, 1]) > 1e-5 or abs(K[2, 2] - 1) > 1e-5:
        raise error("The intrinsic matrix must have [fx 0 cx; 0 fy cy; 0 0 1] shape")
    dist = np.zeros(5)
    dist[:D.size] = D.ravel()
    if flags & CALIB_ZERO_TANGENT_DIST:
        dist[2] = dist[3] = 0.0
    mask = _pinhole_mask(flags, D.size)
    ctype, maxc, eps = criteria
    if not ctype & TERM_CRITERIA_COUNT:
        maxc = 1000
    if not ctype & TERM_CRITERIA_EPS:
        eps = DBL_EPSILON
    return D, np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dist]), mask, maxc, eps


def _solve_pinhole(E, dev, D, mask, maxc, eps, comm, _info, cameraMatrix, distCoeffs):
    """CvLevMarq from intr[0] / pose[0] of ``E`` (poses initialised) -> the cv2 return tuple."""
    log = [] if (_info is not None and _info.get("log") is True) else None
    try:
        err, iters = _solver.lm_pinhole(E, mask, maxc, eps, comm=comm, log=log)
    except _lib.CamcalNumericError as exc:
        _raise_degenerate(E, exc)
        raise
    intr, poses, bad = E.read_solution()
    _raise_degenerate(E, bad=bad, comm=comm, dev=dev)
    n_total = comm.total(E.n_obs, dev)
    rms = float(np.sqrt(err / n_total))
    Kout = np.array([[intr[0], 0.0, intr[2]], [0.0, intr[1], intr[3]], [0.0, 0.0, 1.0]])
    if D.size == 5:
        Dout = intr[4:9].reshape(np.asarray(distCoeffs).shape)
    else:
        # cv2 pads a 4-coefficient guess to its 5-coefficient model (prepareDistCoeffs + rowRange(0, 5)),
        # estimates k3 unless CALIB_FIX_K3 and hands back a NEW 5-element array in the input's orientation
        Dout = intr[4:9].reshape((1, 5) if (D.ndim == 2 and D.shape[0] == 1) else (5, 1))
    if _info is not None:
        _info.update(iters=iters, log=log, engine=E)
    rv, tv = _poses_to_tuples(poses)
    return rms, _write_back(cameraMatrix, Kout), _write_back(distCoeffs, Dout), rv, tv


# --- the one-call path (camcal_calibrate / camcal_calibrate_many) --------------------------
# Device workspaces are kept between calls (a call allocates nothing once one of the right size
# exists): a free list per device, taken and given back under a lock, so concurrent callers
# each get their own.
_ws_lock = threading.Lock()
_ws_free = {}


def _take_workspace(dev, nbytes):
    key = str(dev)
    with _ws_lock:
        pool = _ws_free.setdefault(key, [])
        best = None
        for i, t in enumerate(pool):
            if t.numel() >= nbytes + 256 and (best is None or t.numel() < pool[best].numel()):
                best = i
        if best is not None:
            return pool.pop(best)
    return torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)


def _give_workspace(dev, t):
    with _ws_lock:
        pool = _ws_free.setdefault(str(dev), [])
        pool.append(t)
        pool.sort(key=lambda x: x.numel())
        del pool[:-8]                       # keep the eight largest


def _problem(lib, dev, model, view_ptr, obj, img, size, K9, D5, n_dist, flags, criteria, keep):
    """Fill a camcal_problem; ``keep`` collects every array the struct points at."""
    n_views = len(view_ptr) - 1
    nbytes = lib.camcal_workspace_len(model, n_views, view_ptr.ctypes.data)
    if nbytes < 0:
        lib.check(int(nbytes), "camcal_workspace_len")
    ws = _take_workspace(dev, nbytes)
    base = ws.data_ptr()
    aligned = (base + 255) & ~255
    rt = np.empty((n_views, 6))
    status = np.zeros(n_views, dtype=np.int32)
    obj = np.ascontiguousarray(obj, dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    ctype, maxc, eps = criteria
    keep.extend([ws, rt, status, obj, img, view_ptr, K9, D5])
    p = _lib.Problem(model, int(flags), int(size[0]), int(size[1]), int(ctype), int(maxc), float(eps), n_views,
                     view_ptr.ctypes.data, obj.ctypes.data, img.ctypes.data, K9.ctypes.data, D5.ctypes.data, int(n_dist),
                     int(Engine.LANES_PER_VIEW), rt.ctypes.data, status.ctypes.data, aligned, ws.numel() - (aligned - base),
                     0.0, 0, 0)
    return p, ws, rt, status


def _raise_status(rc, status, what):
    """A camcal_problem's status -> the exception the drop-in raises for it."""
    if rc == 0:
        return
    bad = np.flatnonzero(status != 0)
    if bad.size:
        raise error("degenerate view(s) %s: the pose initialisation failed (collinear / coincident points); "
                    "remove them" % (bad[:16].tolist(),))
    if rc == _lib.E_NUMERIC:
        raise error("the solve met a singular system or a non-finite cost (%s)" % what)
    if rc == _lib.E_UNSUPPORTED:      # the flags were checked before the call: what the library refuses is the target
        raise error(_NOT_PLANAR)
    raise _lib.CamcalStatus("%s failed with status %d" % (what, rc), rc)


def _stream_of(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None


class _on_device:
    """``with torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager costs
    tens of microseconds, a tenth of a production-sized call)."""

    def __init__(self, dev):
        self.ctx = None
        if dev.type == "cuda" and dev.index is not None and torch.cuda.current_device() != dev.index:
            self.ctx = torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _one_call(model, view_ptr, obj, img, size, K9, D5, n_dist, flags, criteria):
    """camcal_calibrate on the resolved library / device -> (rms, K9, D5, rt, iterations)."""
    lib, dev = _resolve()
    lib = lib if lib is not None else _lib.load()
    if dev.type != "cuda" and not lib.is_emulator:
        raise _lib.CamcalError("the HIP library needs a cuda (ROCm) device; there is no CPU path")
    keep = []
    p, ws, rt, status = _problem(lib, dev, model, view_ptr, obj, img, size, K9, D5, n_dist, flags, criteria, keep)
    rms, its = C.c_double(0.0), C.c_int32(0)
    try:
        with _on_device(dev):
            rc = lib.camcal_calibrate(p.model, p.n_views, p.view_ptr, p.obj, p.img, p.width, p.height, p.K, p.D, p.n_dist,
                                      p.rt, p.view_status, p.flags, p.crit_type, p.max_iter, p.eps, p.workspace,
                                      p.workspace_bytes, None, _stream_of(dev), C.byref(rms), C.byref(its))
    finally:
        _give_workspace(dev, ws)
    _raise_status(rc, status, "camcal_calibrate")
    return float(rms.value), K9, D5, rt, int(its.value)


def _pinhole_result(rms, K9, D5, rt, D, cameraMatrix, distCoeffs):
    Kout = K9.reshape(3, 3)
    if D.size == 5:
        Dout = D5.reshape(np.asarray(distCoeffs).shape)
    else:       # a 4-coefficient guess comes back as cv2's 5-coefficient model (see _solve_pinhole)
        Dout = D5.reshape((1, 5) if (D.ndim == 2 and D.shape[0] == 1) else (5, 1))
    rv, tv = _poses_to_tuples(rt)
    return rms, _write_back(cameraMatrix, Kout), _write_back(distCoeffs, Dout), rv, tv


def _fisheye_result(rms, K9, D4, rt, K, D):
    rv, tv = _poses_to_tuples(rt)
    return rms, _write_back(K, K9.reshape(3, 3)), _write_back(D, D4.reshape(np.asarray(D).shape)), rv, tv


def _k9(intr0):
    return np.array([intr0[0], 0.0, intr0[2], 0.0, intr0[1], intr0[3], 0.0, 0.0, 1.0])


def calibrate_many(calls):
    """Several independent calibrations in one call -- the reference solves its four cameras one after
    another (controller/controller_internal_calibration.py:375-399: ``cali_in`` loops over L, ML, MR, R and
    calls ``cv2.fisheye.calibrate`` / ``cv2.calibrateCamera`` for each).  ``calls``: a sequence of dicts with
    the arguments of the respective drop-in call, ``dict(model="fisheye" | "pinhole", objectPoints=...,
    imagePoints=..., imageSize=..., cameraMatrix=..., distCoeffs=..., flags=..., criteria=...)``.  Returns the
    list of the calls' return tuples ``(retval, K, D, rvecs, tvecs)``, each equal bit for bit to what the
    call alone returns; K / D are written in place as there.  The solves share the device: every camera's
    next iteration is enqueued before the host waits for any of them (camcal_calibrate_many)."""
    lib, dev = _resolve()
    lib = lib if lib is not None else _lib.load()
    if dev.type != "cuda" and not lib.is_emulator:
        raise _lib.CamcalError("the HIP library needs a cuda (ROCm) device; there is no CPU path")
    keep, items = [], []
    for c in calls:
        model = c["model"]
        flags = int(c.get("flags", 0))
        if model == "pinhole":
            crit = c.get("criteria", (TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 30, DBL_EPSILON))
            D, intr0, mask, maxc, eps = _prep_pinhole(c["imageSize"], c["cameraMatrix"], c["distCoeffs"], flags, crit)
            D5 = intr0[4:9].copy()
            nd, mid = D.size, _lib.PINHOLE
        elif model == "fisheye":
            crit = c.get("criteria", (TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 100, DBL_EPSILON))
            intr0, mask = _prep_fisheye(c["cameraMatrix"], c["distCoeffs"], flags)
            D = np.asarray(c["distCoeffs"], dtype=np.float64)
            D5 = np.asarray(D, dtype=np.float64).ravel().copy()       # the guess as given: the library applies FIX_Ki
            nd, mid = 4, _lib.FISHEYE
        else:
            raise error("model must be 'pinhole' or 'fisheye'")
        view_ptr, obj, img = _gather_views(c["objectPoints"], c["imagePoints"], planar_check=False)
        K9 = _k9(intr0)
        p, ws, rt, status = _problem(lib, dev, mid, view_ptr, obj, img, c["imageSize"], K9, D5, nd, flags, crit, keep)
        items.append((c, model, p, ws, rt, status, K9, D5, D))
    arr = (_lib.Problem * len(items))(*[it[2] for it in items])
    try:
        with _on_device(dev):
            lib.camcal_calibrate_many(arr, len(items), _stream_of(dev))
    finally:
        for it in items:
            _give_workspace(dev, it[3])
    out = []
    for k, (c, model, _, _, rt, status, K9, D5, D) in enumerate(items):
        _raise_status(int(arr[k].status), status, "camcal_calibrate_many (problem %d)" % k)
        if model == "pinhole":
            out.append(_pinhole_result(float(arr[k].rms), K9, D5, rt, D, c["cameraMatrix"], c["distCoeffs"]))
        else:
            out.append(_fisheye_result(float(arr[k].rms), K9, D5, rt, c["cameraMatrix"], c["distCoeffs"]))
    return out


def calibrateCamera(objectPoints, imagePoints, imageSize, cameraMatrix, distCoeffs, rvecs=None,
                    tvecs=None, flags=0, criteria=(TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 30, DBL_EPSILON),
                    *, _comm=None, _info=None):
    """``cv2.calibrateCamera`` on the MI355X (5-coefficient pinhole model).

    Supported: ``CALIB_USE_INTRINSIC_GUESS`` (required -- the only mode the
    reference uses) optionally with FIX_FOCAL_LENGTH / FIX_PRINCIPAL_POINT /
    ZERO_TANGENT_DIST / FIX_TANGENT_DIST / FIX_K1..K3.  Other model flags raise.
    A 4-coefficient ``distCoeffs`` is padded to 5 as cv2 does: k3 is estimated unless
    ``CALIB_FIX_K3`` and a new 5-element array comes back.

    Keyword-only extras (tests, tools): ``_info`` -- a dict that receives
    ``iters``, ``engine`` and ``log`` (the per-iteration records, collected only
    when the dict holds ``log=True``: asking for them keeps the loop in
    solver.py instead of the library); ``_comm`` -- a solver.Comm for a
    multi-rank solve.
    """
    flags = int(flags)
    D, intr0, mask, maxc, eps = _prep_pinhole(imageSize, cameraMatrix, distCoeffs, flags, criteria)
    comm = _comm or _solver.Comm()
    # (the one-call path leaves the planarity check to the library)
    view_ptr, obj, img = _gather_collectively(objectPoints, imagePoints, comm,
                                              planar_check=not (_info is None and not comm.enabled))
    if _info is None and not comm.enabled:
        # the plain call: everything behind one library entry point (camcal_calibrate)
        rms, K9, D5, rt, _ = _one_call(_lib.PINHOLE, view_ptr, obj, img, imageSize, _k9(intr0), intr0[4:9].copy(), D.size,
                                       flags, criteria)
        return _pinhole_result(rms, K9, D5, rt, D, cameraMatrix, distCoeffs)
    lib, dev = _resolve()
    E = Engine(_lib.PINHOLE, view_ptr, obj, img, dev, lib=lib)
    E.set_intr(0, intr0)
    # cvFindExtrinsicCameraParams2 per view: homography init + iterative refine
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    return _solve_pinhole(E, dev, D, mask, maxc, eps, comm, _info, cameraMatrix, distCoeffs)


def _prep_fisheye(K, D, flags):
    """Argument checks of cv2.fisheye.calibrate -> (intr0, mask)."""
    if flags & FISHEYE_CALIB_CHECK_COND:
        raise error("CALIB_CHECK_COND is not supported")
    if not flags & FISHEYE_CALIB_USE_INTRINSIC_GUESS:
        raise error("CALIB_USE_INTRINSIC_GUESS is required (K / D must hold the guess)")
    if not flags & FISHEYE_CALIB_FIX_SKEW:
        raise error("skew estimation is not supported: pass fisheye.CALIB_FIX_SKEW")
    if K is None or D is None:
        raise error("(!K.empty() && !D.empty()) || !(flags & CALIB_USE_INTRINSIC_GUESS)")
    Kin = np.asarray(K, dtype=np.float64)
    Din = np.asarray(D, dtype=np.float64)
    if Kin.shape != (3, 3):
        raise error("K.size() == Size(3,3)")
    if Din.size != 4:
        raise error("D.total() == 4")
    if Kin[0, 1] != 0.0:
        raise error("a non-zero fixed skew is not supported")
    mask = np.ones(8, dtype=bool)
    if flags & FISHEYE_CALIB_FIX_FOCAL_LENGTH:
        mask[0] = mask[1] = False
    if flags & FISHEYE_CALIB_FIX_PRINCIPAL_POINT:
        mask[2] = mask[3] = False
    k = Din.ravel().copy()
    for j, f in enumerate((FISHEYE_CALIB_FIX_K1, FISHEYE_CALIB_FIX_K2, FISHEYE_CALIB_FIX_K3,
                           FISHEYE_CALIB_FIX_K4)):
        if flags & f:
            mask[4 + j] = False
            k[j] = 0.0
    return np.concatenate([[Kin[0, 0], Kin[1, 1], Kin[0, 2], Kin[1, 2]], k]), mask


def _solve_fisheye(E, dev, mask, flags, criteria, comm, _info, K, D):
    """cv::fisheye::calibrate's loop from intr[0] / pose[0] of ``E`` -> the cv2 return tuple."""
    log = [] if (_info is not None and _info.get("log") is True) else None
    try:
        cost, iters = _solver.gn_fisheye(E, mask, tuple(criteria),
                                         recompute_extrinsic=bool(flags & FISHEYE_CALIB_RECOMPUTE_EXTRINSIC),
                                         comm=comm, log=log)
    except _lib.CamcalNumericError as exc:
        _raise_degenerate(E, exc)
        raise
    intr, poses, bad = E.read_solution()
    _raise_degenerate(E, bad=bad, comm=comm, dev=dev)
    n_total = comm.total(E.n_obs, dev)
    rms = float(np.sqrt(cost / n_total))
    Kout = np.array([[intr[0], 0.0, intr[2]], [0.0, intr[1], intr[3]], [0.0, 0.0, 1.0]])
    Dout = intr[4:8].reshape(np.asarray(D).shape)
    if _info is not None:
        _info.update(iters=iters, log=log, engine=E)
    rv, tv = _poses_to_tuples(poses)
    return rms, _write_back(K, Kout), _write_back(D, Dout), rv, tv


def fisheye_calibrate(objectPoints, imagePoints, image_size, K, D, rvecs=None, tvecs=None, flags=0,
                      criteria=(TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 100, DBL_EPSILON),
                      *, _comm=None, _info=None):
    """``cv2.fisheye.calibrate`` on the MI355X (Kannala-Brandt, k1..k4).

    Supported flag set: ``CALIB_USE_INTRINSIC_GUESS | CALIB_FIX_SKEW`` (both
    required, as the reference always passes them) with optional
    ``CALIB_RECOMPUTE_EXTRINSIC``, ``CALIB_FIX_K1..K4``,
    ``CALIB_FIX_PRINCIPAL_POINT``, ``CALIB_FIX_FOCAL_LENGTH``.
    """
    flags = int(flags)
    intr0, mask = _prep_fisheye(K, D, flags)
    comm = _comm or _solver.Comm()
    # (the one-call path leaves the planarity check to the library)
    view_ptr, obj, img = _gather_collectively(objectPoints, imagePoints, comm,
                                              planar_check=not (_info is None and not comm.enabled))
    if _info is None and not comm.enabled:
        D4 = np.asarray(D, dtype=np.float64).ravel().copy()
        rms, K9, D4, rt, _ = _one_call(_lib.FISHEYE, view_ptr, obj, img, image_size, _k9(intr0), D4, 4, flags, criteria)
        return _fisheye_result(rms, K9, D4, rt, K, D)
    lib, dev = _resolve()
    E = Engine(_lib.FISHEYE, view_ptr, obj, img, dev, lib=lib)
    E.set_intr(0, intr0)
    # CalibrateExtrinsics: InitExtrinsics + ComputeExtrinsicRefine per view
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    return _solve_fisheye(E, dev, mask, flags, criteria, comm, _info, K, D)


class CalibrationSession:
    """The two drop-in calls for a caller whose view lists GROW between calls.

    The reference's capture loops call the solver after every accepted frame with the
    same lists, one view longer (intrinsicCalib.py:538-564 -> ``Fisheye/Normal.update``
    :86-94, 100 iterations first, then 10-iteration refinements :129-135).  For lists of
    ``ONE_SHOT_BELOW`` corner observations or more a session
    keeps the packed views and the solver state on the device: a call whose lists start
    with the very arrays of the previous call (checked by identity) uploads and packs only
    the new views, initialises only their poses and warm-starts the others from the
    previous result; any other call starts over.  Same signatures, same return tuples and
    -- to the refinement tolerance -- the same numbers as the one-shot functions
    (``tests/test_hostlogic.py::test_session_*``)."""

    # Lists with fewer corner observations than this go through the one-call path, from scratch, as cv2 does:
    # at production sizes (the reference keeps <= 25 views of <= 88 corners per camera) a fresh one-call solve
    # (0.4 - 0.5 ms at 10 ... 80 views) is faster than keeping the device state up to date from Python
    # (0.6 - 0.7 ms: tools/gpu_session.py); what a session saves -- the upload, packing and pose initialisation
    # of the views already held -- only outweighs that for large growing sets.
    ONE_SHOT_BELOW = 100_000

    def __init__(self, model):
        if model not in ("pinhole", "fisheye"):
            raise error("model must be 'pinhole' or 'fisheye'")
        self.model = model
        self._E = None
        self._held = []                 # (object array, image array) of every view on the device

    def _small(self, objectPoints, _info):
        """Production-sized lists: the plain one-call solve (any device state is dropped)."""
        if _info is not None or self.ONE_SHOT_BELOW <= 0:
            return False
        try:
            n = sum(map(len, objectPoints))
        except TypeError:
            return False
        if n >= self.ONE_SHOT_BELOW:
            return False
        self._E, self._held = None, []
        return True

    def _sync(self, objectPoints, imagePoints):
        """Bring the engine to the given lists; returns the slots of the views added."""
        lib, dev = _resolve()
        nh = len(self._held)
        same = (self._E is not None and len(objectPoints) >= nh and len(imagePoints) == len(objectPoints) and
                all(objectPoints[j] is self._held[j][0] and imagePoints[j] is self._held[j][1] for j in range(nh)))
        if not same:
            self._E = IncrementalEngine(_lib.PINHOLE if self.model == "pinhole" else _lib.FISHEYE, dev, lib=lib)
            self._held, nh = [], 0
        new_o, new_i = list(objectPoints[nh:]), list(imagePoints[nh:])
        if len(objectPoints) == 0 or len(objectPoints) != len(imagePoints):
            _gather_views(objectPoints, imagePoints)          # raises like the one-shot call
        slots = np.zeros(0, dtype=np.int64)
        if new_o:
            vp, obj, img = _gather_views(new_o, new_i)
            slots = self._E.append([obj[vp[j]:vp[j + 1]] for j in range(len(new_o))],
                                   [img[vp[j]:vp[j + 1]] for j in range(len(new_o))])
            self._held.extend(zip(new_o, new_i))
        return self._E, dev, slots

    def calibrateCamera(self, objectPoints, imagePoints, imageSize, cameraMatrix, distCoeffs, rvecs=None, tvecs=None,
                        flags=0, criteria=(TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 30, DBL_EPSILON), *, _info=None):
        if self.model != "pinhole":
            raise error("this session holds a fisheye camera")
        if self._small(objectPoints, _info):
            return calibrateCamera(objectPoints, imagePoints, imageSize, cameraMatrix, distCoeffs, rvecs, tvecs, flags, criteria)
        flags = int(flags)
        D, intr0, mask, maxc, eps = _prep_pinhole(imageSize, cameraMatrix, distCoeffs, flags, criteria)
        E, dev, slots = self._sync(objectPoints, imagePoints)
        E.set_intr(0, intr0)
        E.init_new_poses(slots)
        E.pose_refine(0, 0, 20, 1e-10)
        return _solve_pinhole(E, dev, D, mask, maxc, eps, _solver.Comm(enabled=False), _info, cameraMatrix, distCoeffs)

    def calibrate(self, objectPoints, imagePoints, image_size, K, D, rvecs=None, tvecs=None, flags=0,
                  criteria=(TERM_CRITERIA_COUNT + TERM_CRITERIA_EPS, 100, DBL_EPSILON), *, _info=None):
        if self.model != "fisheye":
            raise error("this session holds a pinhole camera")
        if self._small(objectPoints, _info):
            return fisheye_calibrate(objectPoints, imagePoints, image_size, K, D, rvecs, tvecs, flags, criteria)
        flags = int(flags)
        intr0, mask = _prep_fisheye(K, D, flags)
        E, dev, slots = self._sync(objectPoints, imagePoints)
        E.set_intr(0, intr0)
        E.init_new_poses(slots)
        E.pose_refine(0, 0, 20, 1e-10)
        return _solve_fisheye(E, dev, mask, flags, criteria, _solver.Comm(enabled=False), _info, K, D)
