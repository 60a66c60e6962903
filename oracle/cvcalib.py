"""cv2.calibrateCamera / cv2.fisheye.calibrate restated in numpy fp64 (oracle).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
PARITY UNPINNED (oracle/__init__.py): OpenCV 4.10 is absent; this follows the
published algorithm as recalled in SURVEY.md 8(c)-appendix.

Reference call sites being restated (what the tool asks cv2 to do):

* ``server/aruco/camera_calib.py:153-156`` / ``intrinsicCalib.py:194-197,223-226``
  ``cv2.calibrateCamera(obj, img, size, K, D, flags=USE_INTRINSIC_GUESS,
  criteria=(EPS+COUNT, 1000|100|10, 1e-6))``          -> :func:`calibrate_camera`
* ``server/aruco/camera_calib.py:74-77`` / ``intrinsicCalib.py:102-105,131-134``
  ``cv2.fisheye.calibrate(obj, img, size, K, D, flags=FIX_SKEW|RECOMPUTE_EXTRINSIC|
  USE_INTRINSIC_GUESS, criteria=...)``                -> :func:`fisheye_calibrate`

OpenCV functions followed: ``cvCalibrateCamera2Internal``,
``cvFindExtrinsicCameraParams2`` (calibration.cpp), ``CvLevMarq::update /
updateAlt / step`` (compat_ptsetreg.cpp), ``cv::fisheye::calibrate``,
``internal::ComputeJacobians / CalibrateExtrinsics / InitExtrinsics /
ComputeHomography / ComputeExtrinsicRefine / NormalizePixels /
EstimateUncertainties`` (fisheye.cpp).

Two linear-algebra back ends per solver:

* ``solver="dense"``: builds the dense (Nintr + 6 N)^2 normal matrix exactly as
  cv2 does and factors it (SVD for CvLevMarq::step, LU for fisheye) -- O(N^3),
  for small N; this is the cv2-faithful statement.
* ``solver="schur"``: eliminates every (damped) 6x6 pose block first; the same
  step up to round-off at O(N).  This is the form the HIP path computes, and
  the tests prove dense == schur on small problems.

Pose initialisation keeps cv2's structure (undistort -> planar homography ->
decomposition -> iterative refinement) but its homography refinement is a
plain Gauss-Newton rather than cv::LMSolver; both converge to the same
homography minimiser, so initial poses agree to the refinement tolerance, not
bit for bit (documented in DESIGN.md).
"""
from __future__ import annotations

import numpy as np

from . import cvmodel as M

DBL_EPSILON = np.finfo(np.float64).eps
FLT_EPSILON = float(np.finfo(np.float32).eps)

# --- flag values (cv2 4.10) -------------------------------------------------
CALIB_USE_INTRINSIC_GUESS = 1
CALIB_FIX_ASPECT_RATIO = 2
CALIB_FIX_PRINCIPAL_POINT = 4
CALIB_ZERO_TANGENT_DIST = 8
CALIB_FIX_FOCAL_LENGTH = 16
CALIB_FIX_K1, CALIB_FIX_K2, CALIB_FIX_K3 = 32, 64, 128
CALIB_FIX_TANGENT_DIST = 2097152
FISHEYE_USE_INTRINSIC_GUESS = 1
FISHEYE_RECOMPUTE_EXTRINSIC = 2
FISHEYE_CHECK_COND = 4
FISHEYE_FIX_SKEW = 8
FISHEYE_FIX_K1, FISHEYE_FIX_K2, FISHEYE_FIX_K3, FISHEYE_FIX_K4 = 16, 32, 64, 128
FISHEYE_FIX_PRINCIPAL_POINT = 512
FISHEYE_FIX_FOCAL_LENGTH = 2048
TERM_COUNT, TERM_EPS = 1, 2


class OracleError(Exception):
    """Stands in for cv2.error in the oracle."""


# --- view bookkeeping --------------------------------------------------------
class Views:
    """Ragged list of views, grouped by corner count so that the model
    functions can be called batched.  Points are taken as given (float32 from
    the callers) and widened to float64 before any arithmetic, like cv2."""

    def __init__(self, object_points, image_points):
        if len(object_points) != len(image_points) or len(object_points) == 0:
            raise OracleError("objectPoints / imagePoints count mismatch or empty")
        self.obj, self.img = [], []
        for o, i in zip(object_points, image_points):
            o = np.asarray(o).reshape(-1, 3).astype(np.float64)
            i = np.asarray(i).reshape(-1, 2).astype(np.float64)
            if o.shape[0] != i.shape[0]:
                raise OracleError("per-view point count mismatch")
            if o.shape[0] < 4:
                raise OracleError("a view needs at least 4 points")
            self.obj.append(o)
            self.img.append(i)
        self.n = np.array([o.shape[0] for o in self.obj])
        self.N = len(self.obj)
        self.total = int(self.n.sum())
        self.groups = []
        for n in np.unique(self.n):
            idx = np.nonzero(self.n == n)[0]
            self.groups.append((idx,
                                np.stack([self.obj[k] for k in idx]),
                                np.stack([self.img[k] for k in idx])))


def _project(model, X, r, t, intr, jac):
    if model == "pinhole":
        out = M.project_pinhole(X, r, t, intr[:4], intr[4:], jac=jac)
        return out if jac else (out,)
    out = M.project_fisheye(X, r, t, intr[:4], intr[4:8], alpha=0.0, jac=jac)
    return out[:2] if jac else (out,)


def n_intr(model):
    return 9 if model == "pinhole" else 8


def normal_blocks(model, views, intr, poses):
    """One normal-equation evaluation with e = proj - obs.

    Returns dict: U (NI,NI), b (NI), cost, V (N,6,6), W (N,NI,6), g (N,6)."""
    NI = n_intr(model)
    N = views.N
    V = np.empty((N, 6, 6))
    W = np.empty((N, NI, 6))
    g = np.empty((N, 6))
    U = np.zeros((NI, NI))
    b = np.zeros(NI)
    cost = 0.0
    for idx, X, uv in views.groups:
        p, J = _project(model, X, poses[idx, :3], poses[idx, 3:], intr, True)
        e = (p - uv).reshape(len(idx), -1)                 # (B, 2n)
        J = J.reshape(len(idx), -1, NI + 6)                # (B, 2n, NI+6)
        A, Bm = J[..., :NI], J[..., NI:]
        V[idx] = np.einsum('bki,bkj->bij', Bm, Bm)
        W[idx] = np.einsum('bki,bkj->bij', A, Bm)
        g[idx] = np.einsum('bki,bk->bi', Bm, e)
        U += np.einsum('bki,bkj->ij', A, A)
        b += np.einsum('bki,bk->i', A, e)
        cost += float((e * e).sum())
    return dict(U=U, b=b, cost=cost, V=V, W=W, g=g)


def reproj_cost(model, views, intr, poses):
    """sum ||proj - obs||^2 over every observation."""
    cost = 0.0
    for idx, X, uv in views.groups:
        (p,) = _project(model, X, poses[idx, :3], poses[idx, 3:], intr, False)
        cost += float(((p - uv) ** 2).sum())
    return cost


def per_view_reproj_err(model, views, intr, poses):
    """The tool's acceptance number: ||proj - obs||_2 / n per view
    (camera_calib.py:96-104,174-181; intrinsicCalib.py:137-146,229-238)."""
    out = np.empty(views.N)
    for idx, X, uv in views.groups:
        (p,) = _project(model, X, poses[idx, :3], poses[idx, 3:], intr, False)
        out[idx] = np.sqrt(((p - uv) ** 2).sum(axis=(1, 2))) / X.shape[1]
    return out


def per_view_reproj_err_f32(model, views_obj, views_img, intr, poses):
    """The same number computed the way the tool's loop does with float32 corner
    arrays: cv2.projectPoints returns float32 for float32 object points, and
    cv2.norm subtracts in float32 and accumulates in double."""
    out = np.empty(len(views_obj))
    for k, (o, i) in enumerate(zip(views_obj, views_img)):
        X = np.asarray(o).reshape(1, -1, 3).astype(np.float64)
        (p,) = _project(model, X, poses[k:k + 1, :3], poses[k:k + 1, 3:], intr, False)
        d = (p[0].astype(np.float32) - np.asarray(i, dtype=np.float32).reshape(-1, 2)).astype(np.float64)
        out[k] = np.sqrt((d * d).sum()) / X.shape[1]
    return out


# --- small dense helpers -----------------------------------------------------
def _svd_solve(A, b):
    """cv::solve(A, b, DECOMP_SVD): pseudo-inverse with OpenCV's threshold
    (2 eps * sum of singular values)."""
    U, w, Vt = np.linalg.svd(A)
    thr = 2 * DBL_EPSILON * w.sum()
    winv = np.where(w > thr, 1.0 / np.where(w > thr, w, 1.0), 0.0)
    return Vt.T @ (winv * (U.T @ b))


# --- CvLevMarq (compat_ptsetreg.cpp) ----------------------------------------
def _levmarq_alt(p0, eval_j, eval_e, step, max_iter, eps, log=None):
    """CvLevMarq::updateAlt state machine.  ``eval_j(p)`` -> (sys, errNorm2),
    ``eval_e(p)`` -> errNorm2, ``step(sys, prev, lam)`` -> new p.
    Returns (p, errNorm2_last, iters)."""
    max_iter = int(min(max(max_iter, 1), 1000))
    eps = max(float(eps), DBL_EPSILON)
    lam_lg10 = -3
    p = p0.copy()
    iters = 0
    sys, err = eval_j(p)
    while True:
        prev = p.copy()
        prev_err = err
        p = step(sys, prev, 10.0 ** lam_lg10)
        err = eval_e(p)
        while err > prev_err:
            lam_lg10 += 1
            if lam_lg10 > 16:
                break
            p = step(sys, prev, 10.0 ** lam_lg10)
            err = eval_e(p)
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        rel = np.linalg.norm(p - prev) / np.linalg.norm(prev)
        if log is not None:
            log.append(dict(iter=iters, err=err, rel=rel, lam_lg10=lam_lg10))
        if iters >= max_iter or rel < eps:
            return p, err, iters
        sys, err2 = eval_j(p)
        # errNorm at p was just computed by eval_e; eval_j recomputes the same
        err = err2


# --- homography + pose initialisation ---------------------------------------
def _homography(Mxy, m):
    """Planar homography m ~ H [X Y 1]: normalised DLT (smallest eigenvector of
    LtL, cf. HomographyEstimatorCallback::runKernel / ComputeHomography)
    followed by Gauss-Newton on the transfer error with h33 = 1."""
    n = Mxy.shape[0]
    cM, cm = Mxy.mean(0), m.mean(0)
    sM = np.abs(Mxy - cM).mean(0)
    sm = np.abs(m - cm).mean(0)
    sM = np.where(sM > DBL_EPSILON, sM, 1.0)
    sm = np.where(sm > DBL_EPSILON, sm, 1.0)
    X = (Mxy - cM) / sM
    x = (m - cm) / sm
    L = np.zeros((2 * n, 9))
    L[0::2, 0:2] = X
    L[0::2, 2] = 1
    L[0::2, 6:8] = -x[:, 0:1] * X
    L[0::2, 8] = -x[:, 0]
    L[1::2, 3:5] = X
    L[1::2, 5] = 1
    L[1::2, 6:8] = -x[:, 1:2] * X
    L[1::2, 8] = -x[:, 1]
    w, v = np.linalg.eigh(L.T @ L)
    H0 = v[:, 0].reshape(3, 3)
    Tm_inv = np.array([[sm[0], 0, cm[0]], [0, sm[1], cm[1]], [0, 0, 1]])
    TM = np.array([[1 / sM[0], 0, -cM[0] / sM[0]], [0, 1 / sM[1], -cM[1] / sM[1]], [0, 0, 1]])
    H = Tm_inv @ H0 @ TM
    H = H / H[2, 2]
    if n > 4:
        Mh = np.concatenate([Mxy, np.ones((n, 1))], axis=1)
        for _ in range(10):
            q = Mh @ H.T
            iw = 1.0 / q[:, 2]
            pr = q[:, :2] * iw[:, None]
            err = (pr - m).reshape(-1)
            J = np.zeros((2 * n, 8))
            J[0::2, 0:3] = Mh * iw[:, None]
            J[1::2, 3:6] = Mh * iw[:, None]
            J[0::2, 6:8] = -pr[:, 0:1] * Mxy * iw[:, None]
            J[1::2, 6:8] = -pr[:, 1:2] * Mxy * iw[:, None]
            d = np.linalg.solve(J.T @ J, J.T @ err)
            h = H.reshape(-1)[:8] - d
            H = np.append(h, 1.0).reshape(3, 3)
    return H


def _planar_frame(obj):
    """Shared by both pose initialisers: PCA frame of the object points.
    Returns (Rt (3,3), Tt (3,), is_planar)."""
    mean = obj.mean(0)
    C = (obj - mean).T @ (obj - mean)
    U, w, Vt = np.linalg.svd(C)
    planar = w[2] / w[1] < 1e-3 if w[1] > 0 else True
    R = Vt.copy()
    if R[0, 2] ** 2 + R[1, 2] ** 2 < 1e-10:
        R = np.eye(3)
    if np.linalg.det(R) < 0:
        R = -R
    return R, -R @ mean, planar


def find_extrinsics_pinhole(obj, img, intr, max_iter=20):
    """cvFindExtrinsicCameraParams2, planar branch + CvLevMarq::update refine
    (6 params, <= 20 iterations, eps = FLT_EPSILON)."""
    mn = M.undistort_pinhole(img, intr[:4], intr[4:])
    Rt, Tt, planar = _planar_frame(obj)
    if not planar:
        raise OracleError("non-planar calibration targets are not restated")
    Mxy = (obj @ Rt.T + Tt)[:, :2]
    H = _homography(Mxy, mn)
    if np.all(np.isfinite(H)):
        h1, h2, h3 = H[:, 0].copy(), H[:, 1].copy(), H[:, 2].copy()
        n1, n2 = np.linalg.norm(h1), np.linalg.norm(h2)
        h1 /= max(n1, DBL_EPSILON)
        h2 /= max(n2, DBL_EPSILON)
        t = h3 * (2.0 / max(n1 + n2, DBL_EPSILON))
        R0 = np.stack([h1, h2, np.cross(h1, h2)], axis=1)
        r = M.rodrigues_inv(R0)
        R0, _ = M.rodrigues(r)
        t = R0 @ Tt + t
        R = R0 @ Rt
    else:
        R, t = np.eye(3), np.zeros(3)
    r = M.rodrigues_inv(R)
    p0 = np.concatenate([r, t])
    X = obj[None]

    def ev_j(p):
        pts, J = M.project_pinhole(X, p[None, :3], p[None, 3:], intr[:4], intr[4:], jac=True)
        e = (pts[0] - img).reshape(-1)
        Jp = J[0].reshape(-1, 15)[:, 9:]
        return (Jp.T @ Jp, Jp.T @ e), float(e @ e)

    def ev_e(p):
        pts = M.project_pinhole(X, p[None, :3], p[None, 3:], intr[:4], intr[4:])
        e = (pts[0] - img).reshape(-1)
        return float(e @ e)

    def step(sys, prev, lam):
        A = sys[0].copy()
        A[np.diag_indices(6)] *= 1.0 + lam
        return prev - _svd_solve(A, sys[1])

    p, _, _ = _levmarq_alt(p0, ev_j, ev_e, step, max_iter, FLT_EPSILON)
    return p


def init_extrinsics_fisheye(obj, img, intr):
    """internal::InitExtrinsics: NormalizePixels -> PCA frame -> homography ->
    orthonormalise (u1, u2 - (u1.u2)u1, u1 x u2) -> Rodrigues."""
    mn = M.undistort_fisheye(img, intr[:4], intr[4:8])
    Rt, Tt, _ = _planar_frame(obj)
    # cv2 tests only |R(0:2,2)| < 1e-6 (same predicate as the squared form above)
    Mxy = (obj @ Rt.T + Tt)[:, :2]
    H = _homography(Mxy, mn)
    sc = 0.5 * (np.linalg.norm(H[:, 0]) + np.linalg.norm(H[:, 1]))
    H = H / sc
    u1 = H[:, 0] / np.linalg.norm(H[:, 0])
    u2 = H[:, 1] - (u1 @ H[:, 1]) * u1
    u2 /= np.linalg.norm(u2)
    RRR = np.stack([u1, u2, np.cross(u1, u2)], axis=1)
    om = M.rodrigues_inv(RRR)
    Rc, _ = M.rodrigues(om)
    T = H[:, 2] + Rc @ Tt
    Rc = Rc @ Rt
    return np.concatenate([M.rodrigues_inv(Rc), T])


def extrinsic_refine_fisheye(obj, img, pose, intr, max_iter=20, thresh_cond=1e6):
    """internal::ComputeExtrinsicRefine: Gauss-Newton on the 6 pose parameters,
    stop when ||d|| / ||x + d|| <= 1e-10, when cond(J) > thresh_cond, or after
    ``max_iter`` steps.  Returns (pose, iterations, cond_tripped)."""
    x = np.asarray(pose, dtype=np.float64).copy()
    X = obj[None]
    change, it, tripped = 1.0, 0, False
    while change > 1e-10 and it < max_iter:
        pts, J, _ = M.project_fisheye(X, x[None, :3], x[None, 3:], intr[:4], intr[4:8], jac=True)
        ex = (img - pts[0]).reshape(-1)
        Jp = J[0].reshape(-1, 14)[:, 8:]
        w = np.linalg.svd(Jp, compute_uv=False)
        if w[0] / w[5] > thresh_cond:
            change, tripped = 0.0, True
            continue
        d = _svd_solve(Jp.T @ Jp, Jp.T @ ex)
        xn = x + d
        change = np.linalg.norm(d) / np.linalg.norm(xn)
        x = xn
        it += 1
    return x, it, tripped


def refine_poses_fisheye_batched(views, poses, intr, max_iter=20):
    """Batched ComputeExtrinsicRefine over all views, warm-started from
    ``poses`` (used by the schur driver and as the checker of the HIP
    pose-refine kernel).  Same stopping rule per view; the cond(J) guard is
    evaluated from the eigenvalues of J^T J."""
    out = poses.copy()
    # a warm start carries the previous Rodrigues vector; cv2 derives each start from a rotation
    # matrix (InitExtrinsics), i.e. inside |r| <= pi: bring the vector back into that ball first
    th = np.linalg.norm(out[:, :3], axis=1)
    far = th > np.pi
    if far.any():
        k = th[far] - 2.0 * np.pi * np.floor((th[far] + np.pi) / (2.0 * np.pi))
        out[far, :3] *= (k / th[far])[:, None]
    iters = np.zeros(views.N, dtype=np.int64)
    for idx, X, uv in views.groups:
        x = out[idx].copy()
        active = np.ones(len(idx), dtype=bool)
        nit = np.zeros(len(idx), dtype=np.int64)
        for _ in range(max_iter):
            if not active.any():
                break
            a = np.nonzero(active)[0]
            pts, J, _ = M.project_fisheye(X[a], x[a, :3], x[a, 3:], intr[:4], intr[4:8], jac=True)
            ex = (uv[a] - pts).reshape(len(a), -1)
            Jp = J.reshape(len(a), -1, 14)[..., 8:]
            JtJ = np.einsum('bki,bkj->bij', Jp, Jp)
            Jte = np.einsum('bki,bk->bi', Jp, ex)
            ev = np.linalg.eigvalsh(JtJ)
            bad = np.sqrt(np.abs(ev[:, -1] / ev[:, 0])) > 1e6
            d = np.linalg.solve(JtJ, Jte[..., None])[..., 0]
            d[bad] = 0.0
            xn = x[a] + d
            change = np.linalg.norm(d, axis=1) / np.linalg.norm(xn, axis=1)
            x[a] = xn
            nit[a] += (~bad).astype(np.int64)
            done = (change <= 1e-10) | bad
            active[a[done]] = False
        out[idx] = x
        iters[idx] = nit
    return out, iters


# --- cv2.calibrateCamera -----------------------------------------------------
def _pinhole_mask(flags):
    mask = np.ones(9, dtype=bool)           # fx fy cx cy k1 k2 p1 p2 k3
    if flags & CALIB_FIX_ASPECT_RATIO:
        raise OracleError("CALIB_FIX_ASPECT_RATIO is not restated")
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


def check_pinhole_guess(K, image_size):
    """Validation performed by cvCalibrateCamera2Internal on the guess."""
    w, h = image_size
    if K[0, 0] <= 0 or K[1, 1] <= 0:
        raise OracleError("Focal length (fx and fy) must be positive")
    if K[0, 2] < 0 or K[0, 2] >= w or K[1, 2] < 0 or K[1, 2] >= h:
        raise OracleError("Principal point must be within the image")
    if abs(K[0, 1]) > 1e-5:
        raise OracleError("Non-zero skew is not supported by the function")
    if abs(K[1, 0]) > 1e-5 or abs(K[2, 0]) > 1e-5 or abs(K[2, 1]) > 1e-5 or abs(K[2, 2] - 1) > 1e-5:
        raise OracleError("The intrinsic matrix must have [fx 0 cx; 0 fy cy; 0 0 1] shape")


def _schur_step(blocks, lam, mask, damp):
    """Intrinsic step dK and per-view steps d_i of the (damped) normal
    equations  [U' W; W^T V'] [dK; d] = [b; g]  by pose-block elimination."""
    U, b, V, W, g = blocks["U"], blocks["b"], blocks["V"], blocks["W"], blocks["g"]
    Vd = V.copy()
    if damp:
        di = np.arange(6)
        Vd[:, di, di] *= 1.0 + lam
    Y = np.linalg.solve(Vd, np.concatenate([W.transpose(0, 2, 1), g[..., None]], axis=2))
    VinvWt, Vinvg = Y[..., :-1], Y[..., -1]
    C = np.einsum('bik,bkj->ij', W, VinvWt)
    d = np.einsum('bik,bk->i', W, Vinvg)
    Ud = U.copy()
    if damp:
        Ud[np.diag_indices_from(Ud)] *= 1.0 + lam
    S = (Ud - C)[np.ix_(mask, mask)]
    rhs = (b - d)[mask]
    dK = np.zeros(U.shape[0])
    dK[mask] = np.linalg.solve(S, rhs)
    di_ = Vinvg - np.einsum('bij,j->bi', VinvWt, dK)
    return dK, di_


def _dense_system(blocks, mask):
    NI = blocks["U"].shape[0]
    N = blocks["V"].shape[0]
    free = np.nonzero(mask)[0]
    nf = len(free)
    A = np.zeros((nf + 6 * N, nf + 6 * N))
    rhs = np.zeros(nf + 6 * N)
    A[:nf, :nf] = blocks["U"][np.ix_(free, free)]
    rhs[:nf] = blocks["b"][free]
    for i in range(N):
        s = slice(nf + 6 * i, nf + 6 * i + 6)
        A[s, s] = blocks["V"][i]
        A[:nf, s] = blocks["W"][i][free]
        A[s, :nf] = blocks["W"][i][free].T
        rhs[s] = blocks["g"][i]
    return A, rhs, free


def calibrate_camera(object_points, image_points, image_size, K, D, flags=0,
                     criteria=(TERM_COUNT + TERM_EPS, 30, DBL_EPSILON), solver="schur",
                     init_poses=None, log=None):
    """cv2.calibrateCamera restated (5-coefficient model, USE_INTRINSIC_GUESS).

    Returns (rms, K (3,3), D (like input), rvecs (N,3), tvecs (N,3), info)."""
    if not flags & CALIB_USE_INTRINSIC_GUESS:
        raise OracleError("only CALIB_USE_INTRINSIC_GUESS calls are restated")
    views = Views(object_points, image_points)
    K = np.asarray(K, dtype=np.float64)
    check_pinhole_guess(K, image_size)
    Din = np.asarray(D, dtype=np.float64)
    if Din.size not in (4, 5):
        raise OracleError("only 4/5 distortion coefficients are restated")
    dist = np.zeros(5)
    dist[:Din.size] = Din.ravel()
    mask = _pinhole_mask(flags)
    # a 4-coefficient guess: calibrateCameraRO pads it (prepareDistCoeffs) and keeps rowRange(0, 5), so k3
    # is estimated unless CALIB_FIX_K3 and five coefficients are returned [cv-recall]
    if flags & CALIB_ZERO_TANGENT_DIST:
        dist[2] = dist[3] = 0.0
    intr0 = np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dist])
    _, maxc, eps = criteria
    if init_poses is None:
        poses0 = np.stack([find_extrinsics_pinhole(o, i, intr0) for o, i in zip(views.obj, views.img)])
    else:
        poses0 = np.asarray(init_poses, dtype=np.float64).copy()
    N = views.N
    # cv2's parameter vector carries 18 intrinsic slots (9 unused ones are 0)
    pad = np.zeros(9)

    def unpack(p):
        return p[:9], p[18:].reshape(N, 6)

    def ev_j(p):
        intr, poses = unpack(p)
        blk = normal_blocks("pinhole", views, intr, poses)
        return blk, blk["cost"]

    def ev_e(p):
        intr, poses = unpack(p)
        return reproj_cost("pinhole", views, intr, poses)

    def step(blk, prev, lam):
        p = prev.copy()
        if solver == "dense":
            A, rhs, free = _dense_system(blk, mask)
            A[np.diag_indices_from(A)] *= 1.0 + lam
            sol = _svd_solve(A, rhs)
            nf = len(free)
            p[free] -= sol[:nf]
            p[18:] -= sol[nf:]
        else:
            dK, di = _schur_step(blk, lam, mask, damp=True)
            p[:9] -= dK
            p[18:] -= di.reshape(-1)
        return p

    p0 = np.concatenate([intr0, pad, poses0.reshape(-1)])
    p, err, iters = _levmarq_alt(p0, ev_j, ev_e, step, maxc, eps, log=log)
    intr, poses = unpack(p)
    rms = np.sqrt(err / views.total)
    Kout = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    if Din.size == 5:
        Dout = intr[4:9].reshape(Din.shape)
    else:
        Dout = intr[4:9].reshape((1, 5) if (Din.ndim == 2 and Din.shape[0] == 1) else (5, 1))
    return rms, Kout, Dout, poses[:, :3].copy(), poses[:, 3:].copy(), dict(iters=iters, init_poses=poses0)


# --- cv2.fisheye.calibrate ---------------------------------------------------
def _fisheye_mask(flags):
    mask = np.ones(8, dtype=bool)            # fx fy cx cy k1 k2 k3 k4
    if flags & FISHEYE_FIX_FOCAL_LENGTH:
        mask[0] = mask[1] = False
    if flags & FISHEYE_FIX_PRINCIPAL_POINT:
        mask[2] = mask[3] = False
    for j, f in enumerate((FISHEYE_FIX_K1, FISHEYE_FIX_K2, FISHEYE_FIX_K3, FISHEYE_FIX_K4)):
        if flags & f:
            mask[4 + j] = False
    return mask


def calibrate_extrinsics_fisheye(views, intr):
    """internal::CalibrateExtrinsics: from-scratch InitExtrinsics followed by
    ComputeExtrinsicRefine for every view."""
    poses = np.empty((views.N, 6))
    for k, (o, i) in enumerate(zip(views.obj, views.img)):
        p0 = init_extrinsics_fisheye(o, i, intr)
        poses[k], _, _ = extrinsic_refine_fisheye(o, i, p0, intr)
    return poses


def fisheye_calibrate(object_points, image_points, image_size, K, D, flags=0,
                      criteria=(TERM_COUNT + TERM_EPS, 100, DBL_EPSILON), solver="schur",
                      reinit_poses=None, log=None):
    """cv2.fisheye.calibrate restated for FIX_SKEW | USE_INTRINSIC_GUESS calls.

    ``reinit_poses``: True  = cv2-faithful (InitExtrinsics from scratch at
    every outer iteration); False = warm start ComputeExtrinsicRefine from the
    previous poses (what the HIP path does; the refinement converges to the
    same per-view optimum).  Default: True for ``dense``, False for ``schur``.

    Returns (rms, K, D, rvecs (N,3), tvecs (N,3), info)."""
    if not flags & FISHEYE_USE_INTRINSIC_GUESS:
        raise OracleError("only CALIB_USE_INTRINSIC_GUESS calls are restated")
    if not flags & FISHEYE_FIX_SKEW:
        raise OracleError("skew estimation is not restated (the tool always fixes it)")
    if flags & FISHEYE_CHECK_COND:
        raise OracleError("CALIB_CHECK_COND is not restated")
    views = Views(object_points, image_points)
    K = np.asarray(K, dtype=np.float64)
    Din = np.asarray(D, dtype=np.float64)
    if Din.size != 4:
        raise OracleError("D.total() == 4")
    if K[0, 1] != 0:
        raise OracleError("non-zero fixed skew is not restated")
    mask = _fisheye_mask(flags)
    k = Din.ravel().copy()
    k[~mask[4:]] = 0.0                                       # FIX_Ki zeroes the guess
    intr = np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], k])
    recompute = bool(flags & FISHEYE_RECOMPUTE_EXTRINSIC)
    if reinit_poses is None:
        reinit_poses = solver == "dense"
    ctype, maxc, eps = criteria
    poses = calibrate_extrinsics_fisheye(views, intr)
    change, it = 1.0, 0
    while True:
        if (ctype == 1 and it >= maxc) or (ctype == 2 and change <= eps) or \
           (ctype == 3 and (change <= eps or it >= maxc)):
            break
        alpha2 = 1.0 - 0.6 ** (it + 1.0)
        blk = normal_blocks("fisheye", views, intr, poses)
        if solver == "dense":
            A, rhs, free = _dense_system(blk, mask)
            G = -np.linalg.solve(A, rhs)                    # e = proj - obs here
            dK = np.zeros(8)
            dK[free] = G[:len(free)]
        else:
            dK, _ = _schur_step(blk, 0.0, mask, damp=False)
            dK = -dK
        new = intr + alpha2 * dK
        change = np.linalg.norm(new[:4] - intr[:4]) / np.linalg.norm(new[:4])
        intr = new
        if recompute:
            if reinit_poses:
                poses = calibrate_extrinsics_fisheye(views, intr)
            else:
                poses, _ = refine_poses_fisheye_batched(views, poses, intr)
        it += 1
        if log is not None:
            log.append(dict(iter=it, change=change, intr=intr.copy()))
    cost = reproj_cost("fisheye", views, intr, poses)
    rms = np.sqrt(cost / views.total)
    Kout = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    Dout = intr[4:8].reshape(Din.shape)
    return rms, Kout, Dout, poses[:, :3].copy(), poses[:, 3:].copy(), dict(iters=it)
