"""Camera models of OpenCV 4.10 restated in numpy fp64 (oracle, test-only).

PARITY UNPINNED (see oracle/__init__.py): restates the published OpenCV
algorithms recalled in SURVEY.md 8(c)-appendix; cv2 itself is not available.

What each function follows (OpenCV 4.10, modules/calib3d/src):

* ``rodrigues``            -> ``cvRodrigues2`` (calibration.cpp), vector->matrix
                              branch incl. the 3x9 derivative
* ``rodrigues_inv``        -> ``cvRodrigues2`` matrix->vector branch (SVD
                              orthonormalisation first)
* ``project_pinhole``      -> ``cvProjectPoints2Internal`` (calibration.cpp)
* ``project_fisheye``      -> ``cv::fisheye::projectPoints`` (fisheye.cpp)
* ``undistort_pinhole``    -> ``cvUndistortPointsInternal`` (undistort.dispatch.cpp)
* ``undistort_fisheye``    -> ``cv::fisheye::undistortPoints`` (fisheye.cpp)

Reference call sites that consume these through cv2.calibrateCamera /
cv2.fisheye.calibrate / cv2.(fisheye.)projectPoints:
``server/aruco/camera_calib.py:74-77,97-99,153-156,175-177`` and
``server/internal/IntrinsicCalibration/intrinsicCalib.py:102-105,143-144,194-197,235-236``.

All functions broadcast over leading batch dimensions: points ``(..., n, 3)``,
``r``/``t`` ``(..., 3)``.  Jacobian column order used throughout the oracle:

* pinhole: ``[fx fy cx cy k1 k2 p1 p2 k3 | r0 r1 r2 | t0 t1 t2]``   (15)
* fisheye: ``[fx fy cx cy k1 k2 k3 k4    | r0 r1 r2 | t0 t1 t2]``   (14)
"""
from __future__ import annotations

import numpy as np

DBL_EPSILON = np.finfo(np.float64).eps

_GEN = np.array(  # d[r]x / dr_i flattened row-major, as in cvRodrigues2
    [[0, 0, 0, 0, 0, -1, 0, 1, 0],
     [0, 0, 1, 0, 0, 0, -1, 0, 0],
     [0, -1, 0, 1, 0, 0, 0, 0, 0]], dtype=np.float64)


def rodrigues(r):
    """Rotation vector -> (R (...,3,3), dRdr (...,3,9)); row i of dRdr is
    d vec(R) / d r_i (row-major vec), exactly the layout of cvRodrigues2."""
    r = np.asarray(r, dtype=np.float64)
    batch = r.shape[:-1]
    r2 = r.reshape(-1, 3)
    B = r2.shape[0]
    R = np.empty((B, 3, 3))
    J = np.empty((B, 3, 9))
    theta = np.sqrt((r2 * r2).sum(-1))
    eye = np.eye(3)
    for b in range(B):
        th = theta[b]
        if th < DBL_EPSILON:
            R[b] = eye
            J[b] = _GEN
            continue
        c, s = np.cos(th), np.sin(th)
        c1 = 1.0 - c
        it = 1.0 / th
        n = r2[b] * it
        rx, ry, rz = n
        rrt = np.outer(n, n)
        r_x = np.array([[0, -rz, ry], [rz, 0, -rx], [-ry, rx, 0]])
        R[b] = c * eye + c1 * rrt + s * r_x
        drrt = np.array(
            [[rx + rx, ry, rz, ry, 0, 0, rz, 0, 0],
             [0, rx, 0, rx, ry + ry, rz, 0, rz, 0],
             [0, 0, rx, 0, 0, ry, rx, ry, rz + rz]])
        for i in range(3):
            ri = n[i]
            a0 = -s * ri
            a1 = (s - 2 * c1 * it) * ri
            a2 = c1 * it
            a3 = (c - s * it) * ri
            a4 = s * it
            J[b, i] = (a0 * eye.ravel() + a1 * rrt.ravel() + a2 * drrt[i]
                       + a3 * r_x.ravel() + a4 * _GEN[i])
    return R.reshape(batch + (3, 3)), J.reshape(batch + (3, 9))


def rodrigues_inv(R):
    """Rotation matrix (3,3) -> rotation vector, following cvRodrigues2's
    matrix branch: nearest rotation by SVD, then the axis/angle extraction
    with its small-sin special case."""
    R = np.asarray(R, dtype=np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rv = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((rv * rv).sum() * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5
    c = min(max(c, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        x = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        y = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        z = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        v *= theta / np.linalg.norm(v)
        return v
    return rv * (theta / (2 * s))


def _pose_chain(X, r, t):
    """Y = R X + t and dY/d[r,t] (..., n, 3, 6) via cvRodrigues2's dRdr."""
    X = np.asarray(X, dtype=np.float64)
    R, dRdr = rodrigues(r)
    t = np.asarray(t, dtype=np.float64)
    Y = np.einsum('...ij,...nj->...ni', R, X) + t[..., None, :]
    dR = dRdr.reshape(dRdr.shape[:-2] + (3, 3, 3))          # [j][row][col]
    dYdr = np.einsum('...jik,...nk->...nij', dR, X)          # (..., n, 3, 3)
    dYdt = np.broadcast_to(np.eye(3), dYdr.shape)
    return Y, np.concatenate([dYdr, dYdt], axis=-1)


def project_pinhole(X, r, t, K4, dist, jac=False):
    """cvProjectPoints2Internal.  K4 = (fx, fy, cx, cy); dist = up to 12
    coefficients (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4), zero padded; the
    tilt model is not restated (the reference never enables it).

    Returns pts (..., n, 2) and, if ``jac``, d pts / d[fx fy cx cy k1 k2 p1 p2
    k3 | r | t]  (..., n, 2, 15)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    k = np.zeros(12)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:min(12, d.size)] = d[:12]
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k
    Y, dY = _pose_chain(X, r, t)
    Z = Y[..., 2]
    z = np.where(Z != 0, 1.0 / np.where(Z != 0, Z, 1.0), 1.0)
    x = Y[..., 0] * z
    y = Y[..., 1] * z
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    a1 = 2 * x * y
    a2 = r2 + 2 * x * x
    a3 = r2 + 2 * y * y
    cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6
    icdist2 = 1.0 / (1 + k4 * r2 + k5 * r4 + k6 * r6)
    xd = x * cdist * icdist2 + p1 * a1 + p2 * a2 + s1 * r2 + s2 * r4
    yd = y * cdist * icdist2 + p1 * a3 + p2 * a1 + s3 * r2 + s4 * r4
    pts = np.stack([xd * fx + cx, yd * fy + cy], axis=-1)
    if not jac:
        return pts
    J = np.zeros(pts.shape + (15,))
    J[..., 0, 0] = xd
    J[..., 1, 1] = yd
    J[..., 0, 2] = 1.0
    J[..., 1, 3] = 1.0
    J[..., 0, 4] = fx * x * icdist2 * r2
    J[..., 1, 4] = fy * y * icdist2 * r2
    J[..., 0, 5] = fx * x * icdist2 * r4
    J[..., 1, 5] = fy * y * icdist2 * r4
    J[..., 0, 6] = fx * a1
    J[..., 1, 6] = fy * a3
    J[..., 0, 7] = fx * a2
    J[..., 1, 7] = fy * a1
    J[..., 0, 8] = fx * x * icdist2 * r6
    J[..., 1, 8] = fy * y * icdist2 * r6
    # d(xd,yd)/d(x,y)
    g = (k1 + 2 * k2 * r2 + 3 * k3 * r4) * icdist2 \
        - cdist * icdist2 * icdist2 * (k4 + 2 * k5 * r2 + 3 * k6 * r4)
    ci = cdist * icdist2
    dxdx = ci + 2 * x * x * g + 2 * p1 * y + 6 * p2 * x + 2 * x * (s1 + 2 * s2 * r2)
    dxdy = 2 * x * y * g + 2 * p1 * x + 2 * p2 * y + 2 * y * (s1 + 2 * s2 * r2)
    dydx = 2 * x * y * g + 2 * p1 * x + 2 * p2 * y + 2 * x * (s3 + 2 * s4 * r2)
    dydy = ci + 2 * y * y * g + 6 * p1 * y + 2 * p2 * x + 2 * y * (s3 + 2 * s4 * r2)
    # d(x,y)/dY = z [1 0 -x; 0 1 -y]
    dxdY = np.stack([z, np.zeros_like(z), -x * z], axis=-1)
    dydY = np.stack([np.zeros_like(z), z, -y * z], axis=-1)
    gu = fx * (dxdx[..., None] * dxdY + dxdy[..., None] * dydY)   # (..., n, 3)
    gv = fy * (dydx[..., None] * dxdY + dydy[..., None] * dydY)
    J[..., 0, 9:] = np.einsum('...ni,...nij->...nj', gu, dY)
    J[..., 1, 9:] = np.einsum('...ni,...nij->...nj', gv, dY)
    return pts, J


def project_fisheye(X, r, t, K4, dist, alpha=0.0, jac=False):
    """cv::fisheye::projectPoints (Kannala-Brandt).  Returns pts and, if
    ``jac``, d pts / d[fx fy cx cy k1..k4 | r | t] (..., n, 2, 14) plus the
    skew column d pts / d alpha (..., n, 2) as a third return value."""
    fx, fy, cx, cy = [float(v) for v in K4]
    k1, k2, k3, k4 = [float(v) for v in np.asarray(dist, dtype=np.float64).ravel()[:4]]
    Y, dY = _pose_chain(X, r, t)
    iz = 1.0 / Y[..., 2]
    x = Y[..., 0] * iz
    y = Y[..., 1] * iz
    r2 = x * x + y * y
    rr = np.sqrt(r2)
    th = np.arctan(rr)
    th2 = th * th
    th3 = th2 * th
    th5 = th3 * th2
    th7 = th5 * th2
    th9 = th7 * th2
    thd = th + k1 * th3 + k2 * th5 + k3 * th7 + k4 * th9
    big = rr > 1e-8
    inv_r = np.where(big, 1.0 / np.where(big, rr, 1.0), 1.0)
    cdist = np.where(big, thd * inv_r, 1.0)
    xd = x * cdist
    yd = y * cdist
    xs = xd + alpha * yd
    pts = np.stack([xs * fx + cx, yd * fy + cy], axis=-1)
    if not jac:
        return pts
    J = np.zeros(pts.shape + (14,))
    J[..., 0, 0] = xs
    J[..., 1, 1] = yd
    J[..., 0, 2] = 1.0
    J[..., 1, 3] = 1.0
    for j, thp in enumerate((th3, th5, th7, th9)):
        dxk = x * thp * inv_r
        dyk = y * thp * inv_r
        J[..., 0, 4 + j] = fx * (dxk + alpha * dyk)
        J[..., 1, 4 + j] = fy * dyk
    dthd = 1 + 3 * k1 * th2 + 5 * k2 * th2 * th2 + 7 * k3 * th2 ** 3 + 9 * k4 * th2 ** 4
    dth_dr = 1.0 / (1 + r2)
    # s(r) = thd / r ; q = (ds/dr) / r
    q = np.where(big, (dthd * dth_dr - cdist) * inv_r * inv_r, 0.0)
    dxdx = cdist + x * x * q
    dxdy = x * y * q
    dydy = cdist + y * y * q
    dxdY = np.stack([iz, np.zeros_like(iz), -x * iz], axis=-1)
    dydY = np.stack([np.zeros_like(iz), iz, -y * iz], axis=-1)
    gx = dxdx[..., None] * dxdY + dxdy[..., None] * dydY
    gy = dxdy[..., None] * dxdY + dydy[..., None] * dydY
    gu = fx * (gx + alpha * gy)
    gv = fy * gy
    J[..., 0, 8:] = np.einsum('...ni,...nij->...nj', gu, dY)
    J[..., 1, 8:] = np.einsum('...ni,...nij->...nj', gv, dY)
    dalpha = np.stack([fx * yd, np.zeros_like(yd)], axis=-1)
    return pts, J, dalpha


def undistort_pinhole(uv, K4, dist, iters=5):
    """cvUndistortPointsInternal with identity R/P: pixel -> normalised,
    fixed-point iteration (5 sweeps, the OpenCV default)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    k = np.zeros(12)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:min(12, d.size)] = d[:12]
    uv = np.asarray(uv, dtype=np.float64)
    x0 = (uv[..., 0] - cx) / fx
    y0 = (uv[..., 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        bad = icdist < 0
        dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
        dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
        xn = (x0 - dX) * icdist
        yn = (y0 - dY) * icdist
        x = np.where(bad, x0, xn)
        y = np.where(bad, y0, yn)
    return np.stack([x, y], axis=-1)


def undistort_fisheye(uv, K4, dist, alpha=0.0, max_iter=10, eps=1e-8):
    """cv::fisheye::undistortPoints with identity R/P (as used by
    internal::NormalizePixels): Newton on theta, clamp to +-pi/2, non-converged
    or sign-flipped points -> (-1e6, -1e6)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    k1, k2, k3, k4 = [float(v) for v in np.asarray(dist, dtype=np.float64).ravel()[:4]]
    uv = np.asarray(uv, dtype=np.float64)
    py = (uv[..., 1] - cy) / fy
    px = (uv[..., 0] - cx) / fx - alpha * py
    thd = np.sqrt(px * px + py * py)
    thd = np.clip(thd, -np.pi / 2, np.pi / 2)
    th = thd.copy()
    conv = thd <= 1e-8
    active = ~conv
    for _ in range(max_iter):
        t2 = th * th
        t4 = t2 * t2
        t6 = t4 * t2
        t8 = t4 * t4
        fix = (th * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8) - thd) / \
              (1 + 3 * k1 * t2 + 5 * k2 * t4 + 7 * k3 * t6 + 9 * k4 * t8)
        th = np.where(active, th - fix, th)
        newly = active & (np.abs(fix) < eps)
        conv = conv | newly
        active = active & ~newly
    scale = np.where(thd > 1e-8, np.tan(th) / np.where(thd > 1e-8, thd, 1.0), 1.0)
    flipped = ((thd < 0) & (th > 0)) | ((thd > 0) & (th < 0))
    ok = conv & ~flipped
    out = np.stack([px * scale, py * scale], axis=-1)
    out[~ok] = -1000000.0
    return out


def undistort_points(model, uv, K4, dist, R=None, P=None):
    """cv2.undistortPoints / cv2.fisheye.undistortPoints with the optional
    rectification R and new camera matrix P applied afterwards (RR = P[:, :3] R;
    reference call sites: camera_calib.py:241,252,261,281,292,301 and
    controller_internal_calibration.py:553,570, which passes a camera matrix as R)."""
    if model == "pinhole":
        xy = undistort_pinhole(uv, K4, dist)
        bad = np.zeros(xy.shape[:-1], dtype=bool)
    else:
        xy = undistort_fisheye(uv, K4, dist)
        bad = xy[..., 0] == -1000000.0
    if R is None and P is None:
        return xy
    RR = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    if P is not None:
        RR = np.asarray(P, dtype=np.float64)[:, :3] @ RR
    h = np.concatenate([xy, np.ones(xy.shape[:-1] + (1,))], axis=-1) @ RR.T
    out = h[..., :2] / h[..., 2:3]
    out[bad] = -1000000.0
    return out


def rectify_map(model, K4, dist, R, newK, size, fixed_point=False):
    """cv2.initUndistortRectifyMap / cv2.fisheye.initUndistortRectifyMap
    (video_server.py:473,497; intrinsicCalib.py:159,251): for every destination
    pixel the source pixel; CV_32FC1 pair, or CV_16SC2 + CV_16UC1 fixed point
    (INTER_BITS = 5, cvRound)."""
    w, h = size
    fx, fy, cx, cy = [float(v) for v in K4]
    iR = np.linalg.inv(np.asarray(newK, dtype=np.float64)[:, :3] @ (np.eye(3) if R is None else np.asarray(R)))
    j, i = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    _x = j * iR[0, 0] + i * iR[0, 1] + iR[0, 2]
    _y = j * iR[1, 0] + i * iR[1, 1] + iR[1, 2]
    _w = j * iR[2, 0] + i * iR[2, 1] + iR[2, 2]
    x, y = _x / _w, _y / _w
    d = np.asarray(dist, dtype=np.float64).ravel()
    if model == "fisheye":
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        t2 = th * th
        thd = th * (1 + d[0] * t2 + d[1] * t2 ** 2 + d[2] * t2 ** 3 + d[3] * t2 ** 4)
        sc = np.where(r == 0, 1.0, thd / np.where(r == 0, 1.0, r))
        u, v = fx * x * sc + cx, fy * y * sc + cy
    else:
        k = np.zeros(5)
        k[:d.size] = d[:5]
        r2 = x * x + y * y
        kr = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
        u = fx * (x * kr + k[2] * 2 * x * y + k[3] * (r2 + 2 * x * x)) + cx
        v = fy * (y * kr + k[2] * (r2 + 2 * y * y) + k[3] * 2 * x * y) + cy
    if not fixed_point:
        return u.astype(np.float32), v.astype(np.float32)
    iu = np.rint(u * 32).astype(np.int64)
    iv = np.rint(v * 32).astype(np.int64)
    m1 = np.stack([iu >> 5, iv >> 5], axis=-1).astype(np.int16)
    m2 = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
    return m1, m2


def remap_bilinear_u8(src, map1, map2, border_value=0):
    """cv2.remap(src, map1, map2, INTER_LINEAR, borderMode=BORDER_CONSTANT) for uint8
    images (video_server.py:504): 1/32-pixel fixed-point source position, the four
    neighbours weighted with cv2's 15-bit bilinear table -- whose entries
    (32-fx)(32-fy)*32 ... are exact, so the table's sum correction never acts --
    and rounded by (sum + 2^14) >> 15; neighbours outside the image read the
    border value."""
    src = np.asarray(src)
    img = src if src.ndim == 3 else src[..., None]
    sh, sw, cn = img.shape
    m1, m2 = np.asarray(map1), np.asarray(map2)
    if m1.dtype == np.float32:
        sx = np.rint(m1.astype(np.float64) * 32).astype(np.int64)
        sy = np.rint(m2.astype(np.float64) * 32).astype(np.int64)
        ix, iy = np.clip(sx >> 5, -32768, 32767), np.clip(sy >> 5, -32768, 32767)
        fx, fy = sx & 31, sy & 31
    else:
        ix, iy = m1[..., 0].astype(np.int64), m1[..., 1].astype(np.int64)
        a = m2.astype(np.int64) & 1023
        fx, fy = a & 31, a >> 5
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.zeros(ix.shape + (cn,), dtype=np.int64)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        x, y = ix + dx, iy + dy
        ok = (x >= 0) & (x < sw) & (y >= 0) & (y < sh)
        px = img[np.clip(y, 0, sh - 1), np.clip(x, 0, sw - 1)].astype(np.int64)
        px = np.where(ok[..., None], px, border_value)
        acc += px * wk[..., None]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out if src.ndim == 3 else out[..., 0]
