"""Multi-camera rig (BASELINE.json config 4): CPU restatement of the joint
problem.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED, and necessarily so: the reference never optimises its
cameras jointly -- ``cali_in`` calibrates them one after another
(controller/controller_internal_calibration.py:375-399) and ``cali_ex`` stores
one solvePnP pose per camera side by side (:454-538).  SURVEY.md 8a defines C4
as a build-side capability:

    residual = proj_c( R_c (R_p X + t_p) + t_c ) - obs,      T_0 = identity

with the reference's two camera models (cv2.calibrateCamera's pinhole model,
intrinsicCalib.py:194-197; cv2.fisheye's Kannala-Brandt model, :102-105) and
cv2's parametrisation (Rodrigues vectors, additive updates, CvLevMarq).  This
file states it densely and independently of the product's Schur/tangent-frame
formulation: explicit Jacobian columns through cvRodrigues2's dR/dr, one dense
(JtJ, Jte) system, CvLevMarq::updateAlt on it.  It is anchored by (i) finite
differences, (ii) scipy.optimize.least_squares on the same residual, (iii) the
single-camera oracle for n_cams == 1 and (iv) ground truth of synthetic rigs
(tests/test_rig_*.py).
"""
from __future__ import annotations

import numpy as np

from . import cvmodel as M
from . import cvcalib as CC


def n_glob(model):
    return CC.n_intr(model) + 6


def _proj(model, Y, intr, jac):
    """Projection of camera-frame points (pose = identity) and d/dY."""
    z3 = np.zeros(3)
    if model == "pinhole":
        out = M.project_pinhole(Y, z3, z3, intr[:4], intr[4:], jac=jac)
    else:
        out = M.project_fisheye(Y, z3, z3, intr[:4], intr[4:8], jac=jac)
    if not jac:
        return out
    NI = CC.n_intr(model)
    p, J = out[0], out[1]
    return p, J[..., :NI], J[..., NI + 3:NI + 6]          # d/d intr, d/dY (translation columns)


class RigObs:
    """cams[c][p] = (X (n,3), uv (n,2)) or None.  Points widened to float64."""

    def __init__(self, object_points, image_points):
        self.n_cams = len(object_points)
        self.P = len(object_points[0])
        self.items = []                                    # (c, p, X, uv)
        for c in range(self.n_cams):
            if len(object_points[c]) != self.P or len(image_points[c]) != self.P:
                raise CC.OracleError("every camera needs one entry per board pose")
            for p in range(self.P):
                o, i = object_points[c][p], image_points[c][p]
                if o is None or len(o) == 0:
                    continue
                X = np.asarray(o).reshape(-1, 3).astype(np.float64)
                uv = np.asarray(i).reshape(-1, 2).astype(np.float64)
                if X.shape[0] != uv.shape[0] or X.shape[0] < 4:
                    raise CC.OracleError("bad view")
                self.items.append((c, p, X, uv))
        self.total = sum(it[2].shape[0] for it in self.items)


def residual_jacobian(model, obs, intr, ext, poses, jac=True):
    """Stacked residual e (2*total) and dense Jacobian over
    [ (intr_c, r_c, t_c) for c ] + [ (r_p, t_p) for p ]."""
    NI, G = CC.n_intr(model), n_glob(model)
    npar = obs.n_cams * G + 6 * obs.P
    e = np.empty(2 * obs.total)
    J = np.zeros((2 * obs.total, npar)) if jac else None
    row = 0
    for c, p, X, uv in obs.items:
        Rc, dRc = M.rodrigues(ext[c, :3])
        Rp, dRp = M.rodrigues(poses[p, :3])
        Xb = X @ Rp.T + poses[p, 3:]                       # board -> rig frame
        Y = Xb @ Rc.T + ext[c, 3:]                         # rig -> camera frame
        n = X.shape[0]
        sl = slice(row, row + 2 * n)
        if not jac:
            e[sl] = (_proj(model, Y, intr[c], False) - uv).reshape(-1)
            row += 2 * n
            continue
        pr, Ji, Gy = _proj(model, Y, intr[c], True)        # (n,2), (n,2,NI), (n,2,3)
        e[sl] = (pr - uv).reshape(-1)
        dRc3 = dRc.reshape(3, 3, 3)                        # [j][row][col] = d R[row,col] / d r_j
        dRp3 = dRp.reshape(3, 3, 3)
        dY_drc = np.einsum('jik,nk->nij', dRc3, Xb)
        dY_drp = np.einsum('il,jlk,nk->nij', Rc, dRp3, X)
        g0 = c * G
        J[sl, g0:g0 + NI] = Ji.reshape(2 * n, NI)
        J[sl, g0 + NI:g0 + NI + 3] = np.einsum('nai,nij->naj', Gy, dY_drc).reshape(2 * n, 3)
        J[sl, g0 + NI + 3:g0 + G] = Gy.reshape(2 * n, 3)
        q0 = obs.n_cams * G + 6 * p
        J[sl, q0:q0 + 3] = np.einsum('nai,nij->naj', Gy, dY_drp).reshape(2 * n, 3)
        J[sl, q0 + 3:q0 + 6] = np.einsum('nai,ij->naj', Gy, Rc).reshape(2 * n, 3)
        row += 2 * n
    return (e, J) if jac else e


def pack(intr, ext, poses):
    return np.concatenate([np.concatenate([intr, ext], axis=1).reshape(-1), poses.reshape(-1)])


def unpack(model, n_cams, x):
    NI, G = CC.n_intr(model), n_glob(model)
    g = x[:n_cams * G].reshape(n_cams, G)
    return g[:, :NI].copy(), g[:, NI:].copy(), x[n_cams * G:].reshape(-1, 6).copy()


def free_mask(model, n_cams, P, intr_mask=None):
    """Camera 0 is the rig frame: its extrinsic is fixed."""
    NI, G = CC.n_intr(model), n_glob(model)
    m = np.ones(n_cams * G + 6 * P, dtype=bool)
    m[NI:G] = False
    if intr_mask is not None:
        for c in range(n_cams):
            m[c * G:c * G + NI] = np.asarray(intr_mask, dtype=bool)
    return m


def compose(ext, pose):
    """(r, t) of T_c o T_p."""
    Rc, _ = M.rodrigues(ext[:3])
    Rp, _ = M.rodrigues(pose[:3])
    return np.concatenate([M.rodrigues_inv(Rc @ Rp), Rc @ pose[3:] + ext[3:]])


def _inv_T(R, t):
    return R.T, -R.T @ t


def _proj_so3(Msum):
    U, _, Vt = np.linalg.svd(Msum)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = U @ np.diag([1, 1, -1.0]) @ Vt
    return R


def init_rig(model, obs, intr):
    """Starting point: per-camera poses at the guessed intrinsics (the
    single-camera initialisers of cvcalib), camera-to-rig transforms from the
    chordal mean of T_pc o T_pk^-1 over shared poses (cameras attached to the
    rig breadth-first from camera 0, always through the known camera with the
    most shared poses), board poses from the lowest-numbered camera seeing them."""
    nC, P = obs.n_cams, obs.P
    cam_pose = {}
    for c, p, X, uv in obs.items:
        if model == "pinhole":
            cam_pose[(c, p)] = CC.find_extrinsics_pinhole(X, uv, intr[c])
        else:
            x0 = CC.init_extrinsics_fisheye(X, uv, intr[c])
            cam_pose[(c, p)] = CC.extrinsic_refine_fisheye(X, uv, x0, intr[c])[0]
    Rt = {k: (M.rodrigues(v[:3])[0], v[3:]) for k, v in cam_pose.items()}
    cnt = np.zeros((nC, nC), dtype=np.int64)
    Rs = np.zeros((nC, nC, 3, 3))
    ts = np.zeros((nC, nC, 3))
    for p in range(P):
        seen = [c for c in range(nC) if (c, p) in Rt]
        for c in seen:
            for k in seen:
                if c == k:
                    continue
                Rc_, tc_ = Rt[(c, p)]
                Rk_, tk_ = Rt[(k, p)]
                Rck = Rc_ @ Rk_.T
                cnt[c, k] += 1
                Rs[c, k] += Rck
                ts[c, k] += tc_ - Rck @ tk_
    ext = np.zeros((nC, 6))
    extRt = {0: (np.eye(3), np.zeros(3))}
    while len(extRt) < nC:
        best = None
        for c in range(nC):
            if c in extRt:
                continue
            for k in sorted(extRt):
                if cnt[c, k] > 0 and (best is None or cnt[c, k] > best[0]):
                    best = (cnt[c, k], c, k)
        if best is None:
            raise CC.OracleError("a camera shares no board pose with the rest of the rig")
        _, c, k = best
        Rck = _proj_so3(Rs[c, k])
        tck = ts[c, k] / cnt[c, k]
        Rk_, tk_ = extRt[k]
        extRt[c] = (Rck @ Rk_, Rck @ tk_ + tck)
        ext[c] = np.concatenate([M.rodrigues_inv(extRt[c][0]), extRt[c][1]])
    poses = np.zeros((P, 6))
    poses[:, 5] = 1.0
    for p in range(P):
        for c in range(nC):
            if (c, p) in Rt:
                Rc_, tc_ = extRt[c]
                Rpc, tpc = Rt[(c, p)]
                Ri, ti = _inv_T(Rc_, tc_)
                poses[p] = np.concatenate([M.rodrigues_inv(Ri @ Rpc), Ri @ tpc + ti])
                break
    return ext, poses, cam_pose


def rig_calibrate(model, object_points, image_points, intr0, max_iter=30, eps=1e-9, intr_mask=None,
                  init=None, log=None):
    """Joint CvLevMarq over every camera's intrinsics, the camera-to-rig
    transforms of cameras 1.. and one pose per board placement.

    Returns (rms, intr (nC,NI), ext (nC,6), poses (P,6), info)."""
    obs = RigObs(object_points, image_points)
    intr0 = np.asarray(intr0, dtype=np.float64)
    if init is None:
        ext0, poses0, _ = init_rig(model, obs, intr0)
    else:
        ext0, poses0 = init
    nC, P = obs.n_cams, obs.P
    seen = np.zeros(P, dtype=bool)
    for _, p, _, _ in obs.items:
        seen[p] = True
    mask = free_mask(model, nC, P, intr_mask)
    mask[nC * n_glob(model):] &= np.repeat(seen, 6)
    x0 = pack(intr0, ext0, poses0)

    def ev_j(x):
        e, J = residual_jacobian(model, obs, *unpack(model, nC, x))
        Jf = J[:, mask]
        return (Jf.T @ Jf, Jf.T @ e), float(e @ e)

    def ev_e(x):
        e = residual_jacobian(model, obs, *unpack(model, nC, x), jac=False)
        return float(e @ e)

    def step(sys, prev, lam):
        A = sys[0].copy()
        A[np.diag_indices(A.shape[0])] *= 1.0 + lam
        out = prev.copy()
        out[mask] -= CC._svd_solve(A, sys[1])
        return out

    x, err, iters = CC._levmarq_alt(x0, ev_j, ev_e, step, max_iter, eps, log=log)
    intr, ext, poses = unpack(model, nC, x)
    return np.sqrt(err / obs.total), intr, ext, poses, dict(iters=iters, cost=err, n_obs=obs.total,
                                                             init=(ext0, poses0))
