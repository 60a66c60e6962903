"""Multi-camera rig calibration (BASELINE.json config 4, SURVEY.md 8a last row).

The reference calibrates its cameras one after another
(controller/controller_internal_calibration.py:375-399) and never optimises
them jointly; SURVEY.md defines the joint problem as a capability of this build:

    residual = proj_c( R_c (R_p X + t_p) + t_c ) - obs,       T_0 = identity

unknowns: the intrinsics of every camera, the camera-from-rig transforms of
cameras 1.., one 6-dof pose per board placement.  The camera models and the
optimiser are the reference's (cv2's Kannala-Brandt / pinhole models, Rodrigues
vectors with additive updates, the CvLevMarq control flow of
cv2.calibrateCamera), so that with one camera the solver IS the single-camera
path (tests/test_rig_*.py).

Device mapping: every (board pose, camera) pair is an ordinary view whose pose
is the composed transform; the per-camera :class:`Engine` evaluates it with the
same single-pass kernel as the single-camera solvers (local tangent frame).  A
lane-per-view kernel applies the point-independent 6x6 chain rule to the rig
parameters, a lane-per-pose kernel eliminates the pose blocks
(``camcal_rig_*`` in include/camcal.h).  The reduced system over the
n_cams*(NI+6) <= 64 globals is formed with one library GEMM (Y^T Y) and
solved by a one-wavefront Cholesky kernel; the host only sequences stages and
reads back the cost and step norms of each trial.

Multi-GPU: board poses are sharded across ranks (all cameras' views of a pose
stay together); one all-reduce of the record [per-camera blocks | C | d] per
evaluation plus one of the step norms.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .engine import Engine
from .solver import Comm, DBL_EPSILON
from .synth import _rot_from_rvec, _rvec_from_rot


def _tri_index(n):
    """(n, n) -> index into the packed upper triangle used by the library."""
    idx = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(n):
            a, b = (i, j) if i <= j else (j, i)
            idx[i, j] = a * n - (a * (a - 1)) // 2 + (b - a)
    return idx


class RigEngine:
    """Device state of one rank's share of the board poses.

    cams: list (one per camera) of dict(pose_idx (n_c,), view_ptr (n_c+1,),
    obj (T_c,3), img (T_c,2)); ``n_poses`` = poses held by this rank."""

    def __init__(self, model, cams, n_poses, device, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.model = int(model)
        self.device = dev = torch.device(device)
        self.n_cams = len(cams)
        self.P = int(max(n_poses, 1))
        self.n_poses = int(n_poses)
        self.E = [Engine(model, c["view_ptr"], c["obj"], c["img"], dev, lib=self.lib) for c in cams]
        self.NI = self.E[0].NI
        self.G = self.NI + 6
        self.nG = self.n_cams * self.G
        if self.nG > 64:
            raise ValueError("at most 64 global parameters (4 cameras) are supported by the reduced solve")
        self.NV = self.E[0].glob_len
        self.NX = self.NI * 6 + 27
        self.n_obs = sum(e.n_obs for e in self.E)
        f64 = dict(dtype=torch.float64, device=dev)
        self.slot_pose = []
        seen = np.zeros(self.P, dtype=bool)
        for e, c in zip(self.E, cams):
            pi = np.asarray(c["pose_idx"], dtype=np.int64)
            if pi.size and (pi.min() < 0 or pi.max() >= self.P):
                raise ValueError("pose index out of range")
            if np.unique(pi).size != pi.size:
                raise ValueError("a camera sees a board pose at most once")
            seen[pi] = True
            sv = e.layout["slot_view"]
            sp = np.full(e.S, -1, dtype=np.int32)
            sp[sv >= 0] = pi[sv[sv >= 0]]
            self.slot_pose.append(torch.from_numpy(sp).to(dev))
        self.seen = seen
        nC, P, G = self.n_cams, self.P, self.G
        self.intr = [torch.zeros(nC, 16, **f64), torch.zeros(nC, 16, **f64)]
        self.ext = [torch.zeros(nC, 6, **f64), torch.zeros(nC, 6, **f64)]
        self.rig_pose = [torch.zeros(6, P, **f64), torch.zeros(6, P, **f64)]
        for b in range(2):
            self.rig_pose[b][5] = 1.0
        self.Vp = [torch.zeros(nC, 21, P, **f64) for _ in range(2)]
        self.gp = [torch.zeros(nC, 6, P, **f64) for _ in range(2)]
        self.Wp = [torch.zeros(nC, G * 6, P, **f64) for _ in range(2)]
        # per-camera records [U|b|cost | U_ie|U_ee|b_e] followed by [C|d]: one all-reduce
        self.rec_len = nC * (self.NV + self.NX)
        self.rec = [torch.zeros(self.rec_len + self.nG * self.nG + self.nG, **f64) for _ in range(2)]
        # Y^T Y is a (nG x 6P) x (6P x nG) product with a tiny output: split the long
        # dimension into chunks so the library GEMM (batched) fills the device, then sum
        self.chunk = 256
        rows = (6 * P + self.chunk - 1) // self.chunk * self.chunk
        self.Y = torch.zeros(rows, self.nG, **f64)          # rows beyond 6P stay zero
        self.z = torch.zeros(rows, **f64)
        self.dG = torch.zeros(self.nG, **f64)
        self.small = torch.zeros(4, **f64)                   # [gstats | norms]: read back together
        self.gstats = self.small[:2]
        self.norms = self.small[2:]
        self._cost_idx = torch.tensor([c * (self.NV + self.NX) + self.NV - 1 for c in range(nC)],
                                      dtype=torch.long, device=dev)
        self.scratch = torch.zeros(int(self.lib.camcal_scratch_len(self.model, 1)), **f64)
        self._tri_i = torch.from_numpy(_tri_index(self.NI)).to(dev)
        self._tri_6 = torch.from_numpy(_tri_index(6)).to(dev)

    # -- plumbing --------------------------------------------------------------
    def _call(self, name, *args):
        self.E[0]._call(name, *args)

    def _stream(self):
        return self.E[0]._stream()

    def cam_rec(self, b, c):
        o = c * (self.NV + self.NX)
        return self.rec[b][o:o + self.NV], self.rec[b][o + self.NV:o + self.NV + self.NX]

    def cd(self, b):
        n = self.nG
        return self.rec[b][self.rec_len:self.rec_len + n * n].view(n, n), self.rec[b][self.rec_len + n * n:]

    def set_params(self, b, intr, ext, poses=None):
        ti = torch.zeros(self.n_cams, 16, dtype=torch.float64)
        ti[:, :self.NI] = torch.as_tensor(np.asarray(intr, dtype=np.float64)[:, :self.NI])
        self.intr[b].copy_(ti.to(self.device))
        self.ext[b].copy_(torch.as_tensor(np.asarray(ext, dtype=np.float64)).to(self.device))
        if poses is not None:
            p = np.zeros((self.P, 6))
            p[:, 5] = 1.0
            p[:self.n_poses] = np.asarray(poses, dtype=np.float64)
            self.rig_pose[b].copy_(torch.from_numpy(np.ascontiguousarray(p.T)).to(self.device))

    def get_params(self, b):
        return (self.intr[b][:, :self.NI].cpu().numpy().copy(), self.ext[b].cpu().numpy().copy(),
                self.rig_pose[b].cpu().numpy().T[:self.n_poses].copy())

    # -- stages ------------------------------------------------------------------
    def evaluate(self, b):
        """Blocks of the whole rig at parameter buffer ``b`` -> rec[b], Vp/gp/Wp[b]."""
        st = self._stream()
        for c, e in enumerate(self.E):
            glob, rext = self.cam_rec(b, c)
            ext_c = self.ext[b][c]
            intr_c = self.intr[b][c]
            self._call("camcal_rig_compose", e._tp(), self.slot_pose[c].data_ptr(), self.rig_pose[b].data_ptr(),
                       self.P, ext_c.data_ptr(), e.pose[0].data_ptr(), st)
            self._call("camcal_rig_local_blocks", self.model, e._tp(), intr_c.data_ptr(), e.pose[0].data_ptr(),
                       e.V.data_ptr(), e.W.data_ptr(), e.g.data_ptr(), glob.data_ptr(), e.scratch.data_ptr(), st)
            self._call("camcal_rig_assemble", self.model, e._tp(), self.slot_pose[c].data_ptr(),
                       self.rig_pose[b].data_ptr(), self.P, ext_c.data_ptr(), e.V.data_ptr(), e.W.data_ptr(),
                       e.g.data_ptr(), self.Vp[b][c].data_ptr(), self.gp[b][c].data_ptr(),
                       self.Wp[b][c].data_ptr(), rext.data_ptr(), e.scratch.data_ptr(), st)

    def eliminate(self, b, lam):
        """Pose elimination at damping ``lam`` -> [C|d] of rec[b]."""
        self._call("camcal_rig_pose_schur", self.model, self.n_cams, self.P, self.Vp[b].data_ptr(),
                   self.gp[b].data_ptr(), self.Wp[b].data_ptr(), float(lam), self.Y.data_ptr(),
                   self.z.data_ptr(), self._stream())
        Cm, d = self.cd(b)
        Yc = self.Y.view(-1, self.chunk, self.nG)
        Yt = Yc.transpose(1, 2)
        Cm.copy_(torch.matmul(Yt, Yc).sum(0))                 # batched library GEMM (rocBLAS fp64)
        d.copy_(torch.matmul(Yt, self.z.view(-1, self.chunk, 1)).sum(0).view(-1))

    def cost(self, b):
        return self.rec[b][self._cost_idx].sum()

    def trial_scalars(self, b):
        """(cost, [||dG||^2, ||globals||^2], [||d poses||^2, ||poses||^2]) of a trial step: one read-back."""
        out = torch.cat([self.rec[b][self._cost_idx], self.small]).cpu().numpy()
        n = self.n_cams
        return float(out[:n].sum()), out[n:n + 2], out[n + 2:n + 4]

    def reduced_system(self, b, lam):
        """(M, rhs) over the n_cams*(NI+6) globals, diag(M) damped like CvLevMarq."""
        NI, G, nC = self.NI, self.G, self.n_cams
        nt = NI * (NI + 1) // 2
        M = torch.zeros(self.nG, self.nG, dtype=torch.float64, device=self.device)
        rhs = torch.zeros(self.nG, dtype=torch.float64, device=self.device)
        for c in range(nC):
            glob, rx = self.cam_rec(b, c)
            o = c * G
            M[o:o + NI, o:o + NI] = glob[:nt][self._tri_i]
            Uie = rx[:NI * 6].view(NI, 6)
            M[o:o + NI, o + NI:o + G] = Uie
            M[o + NI:o + G, o:o + NI] = Uie.t()
            M[o + NI:o + G, o + NI:o + G] = rx[NI * 6:NI * 6 + 21][self._tri_6]
            rhs[o:o + NI] = glob[nt:nt + NI]
            rhs[o + NI:o + G] = rx[NI * 6 + 21:]
        M.diagonal().mul_(1.0 + lam)
        Cm, d = self.cd(b)
        return M - Cm, rhs - d

    def solve(self, b, lam, free, out):
        """dG = M^-1 rhs on the free globals (``free``: boolean mask over the
        n_cams*(NI+6) globals); parameters[out] = parameters[b] - dG.  Returns
        (||dG||^2, ||globals||^2) as a 2-vector on the device."""
        m = bytes(bytearray(int(bool(x)) for x in free))
        self._call("camcal_rig_solve_reduced", self.model, self.n_cams, self.rec[b].data_ptr(), m, float(lam),
                   self.intr[b].data_ptr(), self.ext[b].data_ptr(), self.intr[out].data_ptr(),
                   self.ext[out].data_ptr(), self.dG.data_ptr(), self.gstats.data_ptr(), self._stream())
        return self.gstats

    def backsub(self, b, lam, out):
        self._call("camcal_rig_backsub", self.model, self.n_cams, self.P, self.Vp[b].data_ptr(),
                   self.gp[b].data_ptr(), self.Wp[b].data_ptr(), float(lam), self.dG.data_ptr(),
                   self.rig_pose[b].data_ptr(), self.rig_pose[out].data_ptr(), self.norms.data_ptr(),
                   self.scratch.data_ptr(), self._stream())

    # -- initialisation ----------------------------------------------------------
    def init_from_guess(self, comm):
        """Per-camera pose initialisation at the guessed intrinsics (the
        single-camera kernels), camera-from-rig transforms from the chordal mean
        of T_pc o T_pk^-1 over shared poses, board poses from the
        lowest-numbered camera that sees them.  Writes buffer 0."""
        nC, P = self.n_cams, self.P
        dv = self.device                  # the batched 3x3 algebra stays where the poses are
        f64 = dict(dtype=torch.float64, device=dv)
        Rpc = torch.zeros(nC, P, 3, 3, **f64)
        tpc = torch.zeros(nC, P, 3, **f64)
        vis = torch.zeros(nC, P, dtype=torch.bool, device=dv)
        for c, e in enumerate(self.E):
            e.intr[0].copy_(self.intr[0][c])
            e.pose_init(0, 0)
            e.pose_refine(0, 0, 20, 1e-10)
            sp = self.slot_pose[c]
            ok = sp >= 0
            pose = e.pose[0].t()[ok]
            idx = sp[ok].long()
            Rpc[c, idx] = _rot_from_rvec(pose[:, :3])
            tpc[c, idx] = pose[:, 3:]
            vis[c, idx] = True
        sums = torch.zeros(nC, nC, 13, **f64)
        for c in range(nC):
            for k in range(nC):
                if c == k:
                    continue
                both = (vis[c] & vis[k]).to(torch.float64)           # weights instead of a gather
                Rck = Rpc[c] @ Rpc[k].transpose(1, 2)
                tck = tpc[c] - torch.einsum("pij,pj->pi", Rck, tpc[k])
                sums[c, k, :9] = (Rck * both[:, None, None]).sum(0).reshape(-1)
                sums[c, k, 9:12] = (tck * both[:, None]).sum(0)
                sums[c, k, 12] = both.sum()
        sums = comm.allreduce(sums).cpu()
        extR = {0: (torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64))}
        while len(extR) < nC:
            best = None
            for c in range(nC):
                if c in extR:
                    continue
                for k in sorted(extR):
                    n = float(sums[c, k, 12])
                    if n > 0 and (best is None or n > best[0]):
                        best = (n, c, k)
            if best is None:
                raise ValueError("a camera shares no board pose with the rest of the rig")
            n, c, k = best
            U, _, Vt = torch.linalg.svd(sums[c, k, :9].view(3, 3))
            R = U @ Vt
            if torch.linalg.det(R) < 0:
                R = U @ torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)) @ Vt
            t = sums[c, k, 9:12] / n
            Rk, tk = extR[k]
            extR[c] = (R @ Rk, R @ tk + t)
        ext = torch.zeros(nC, 6, dtype=torch.float64)
        for c in range(1, nC):
            ext[c, :3] = _rvec_from_rot(extR[c][0][None])[0]
            ext[c, 3:] = extR[c][1]
        # board poses from the lowest-numbered camera that sees them: all candidates at once
        first = torch.full((P,), nC, dtype=torch.long, device=dv)
        for c in range(nC - 1, -1, -1):
            first = torch.where(vis[c], torch.full_like(first, c), first)
        Rall = torch.stack([extR[c][0] for c in range(nC)]).to(dv)      # (nC, 3, 3)
        tall = torch.stack([extR[c][1] for c in range(nC)]).to(dv)      # (nC, 3)
        seen = first < nC
        cam = first.clamp_max(nC - 1)
        ar = torch.arange(P, device=dv)
        Rp = Rall[cam].transpose(1, 2) @ Rpc[cam, ar]                    # R_c^T R_pc
        tp = torch.einsum("pji,pj->pi", Rall[cam], tpc[cam, ar] - tall[cam])   # R_c^T (t_pc - t_c)
        poses = torch.zeros(P, 6, **f64)
        poses[:, 5] = 1.0
        if bool(seen.any()):
            cand = torch.cat([_rvec_from_rot(Rp), tp], dim=1)
            poses = torch.where(seen[:, None], cand, poses)
        self.ext[0].copy_(ext.to(dv))
        self.rig_pose[0].copy_(poses.t().contiguous())


def lm_rig(R, free, max_iter, eps, comm=None, log=None):
    """CvLevMarq::updateAlt over the rig (same control flow as
    solver.lm_pinhole).  Buffer 0 holds the starting point and, on return, the
    result.  Returns (errNorm2, iterations)."""
    comm = comm or Comm(enabled=False)
    max_iter = int(min(max(max_iter, 1), 1000))
    eps = max(float(eps), DBL_EPSILON)
    free = np.asarray(free, dtype=bool)
    lam_lg10 = -3
    cur, tri = 0, 1
    R.evaluate(cur)
    comm.allreduce(R.rec[cur][:R.rec_len])
    err = float(R.cost(cur).item()) if comm.enabled else R.trial_scalars(cur)[0]
    iters = 0

    def step(lam):
        R.eliminate(cur, lam)
        comm.allreduce(R.rec[cur][R.rec_len:])
        gn = R.solve(cur, lam, free, tri)
        R.backsub(cur, lam, tri)
        comm.allreduce(R.norms)
        R.evaluate(tri)
        comm.allreduce(R.rec[tri][:R.rec_len])
        if comm.enabled:
            out = torch.cat([R.cost(tri).reshape(1), gn, R.norms]).cpu().numpy()
            return float(out[0]), out[1:3], out[3:5]
        return R.trial_scalars(tri)

    while True:
        prev_err = err
        err, gn, pn = step(10.0 ** lam_lg10)
        while not err <= prev_err:
            lam_lg10 += 1
            if lam_lg10 > 16:
                break
            err, gn, pn = step(10.0 ** lam_lg10)
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        rel = math.sqrt(gn[0] + pn[0]) / math.sqrt(gn[1] + pn[1])
        for buf in (R.intr, R.ext, R.rig_pose, R.Vp, R.gp, R.Wp, R.rec):
            buf[cur], buf[tri] = buf[tri], buf[cur]
        if log is not None:
            log.append(dict(iter=iters, err=err, rel=rel, lam_lg10=lam_lg10))
        if iters >= max_iter or rel < eps:
            return err, iters


def _gather_rig(object_points, image_points):
    nC = len(object_points)
    if nC == 0 or nC != len(image_points):
        raise ValueError("objectPoints / imagePoints: one list per camera")
    P = len(object_points[0])
    cams = []
    for c in range(nC):
        if len(object_points[c]) != P or len(image_points[c]) != P:
            raise ValueError("every camera needs one entry (or None) per board pose")
        idx, objs, imgs, counts = [], [], [], []
        for p in range(P):
            o, i = object_points[c][p], image_points[c][p]
            if o is None or i is None or len(o) == 0:
                continue
            o = np.asarray(o).reshape(-1, 3).astype(np.float32, copy=False)
            i = np.asarray(i).reshape(-1, 2).astype(np.float32, copy=False)
            if o.shape[0] != i.shape[0]:
                raise ValueError("number of object and image points must be equal in every view")
            if o.shape[0] < 4:
                raise ValueError("every view needs at least 4 points")
            if np.any(o[:, 2] != 0.0):
                raise ValueError("only planar calibration targets with Z = 0 are supported")
            idx.append(p)
            objs.append(o)
            imgs.append(i)
            counts.append(o.shape[0])
        vp = np.zeros(len(idx) + 1, dtype=np.int64)
        vp[1:] = np.cumsum(counts)
        cams.append(dict(pose_idx=np.asarray(idx, dtype=np.int64), view_ptr=vp,
                         obj=np.concatenate(objs) if objs else np.zeros((0, 3), np.float32),
                         img=np.concatenate(imgs) if imgs else np.zeros((0, 2), np.float32)))
    return cams, P


def calibrate_rig(objectPoints, imagePoints, imageSize, cameraMatrices, distCoeffs, model="fisheye",
                  fix_intrinsics=None, criteria=(3, 30, 1e-9), *, cams=None, n_poses=None,
                  _lib_override=None, _device_override=None, _comm=None, _info=None, _init=None):
    """Joint calibration of a camera rig observing shared board poses.

    objectPoints[c][p] / imagePoints[c][p]: (n,1,3) / (n,1,2) float32 arrays as
    for cv2, or None where camera c does not see pose p (alternatively pass the
    packed per-camera dicts as ``cams`` with ``n_poses``).  cameraMatrices /
    distCoeffs: per-camera guesses (fisheye: 4 coefficients, pinhole: 5).
    ``fix_intrinsics``: optional boolean mask (NI,) of intrinsics to hold.
    criteria as cv2: (type, max_iter, eps), eps on ||step|| / ||parameters||.

    Returns (rms, Ks, Ds, Ts, rvecs, tvecs): Ts[c] = (rvec, tvec) of
    camera-from-rig (camera 0: zeros), rvecs/tvecs the board poses in the rig
    frame (tuples of (3,1) arrays; poses no camera sees keep (0,0,0 | 0,0,1))."""
    from .api import _resolve
    mdl = {"pinhole": _lib.PINHOLE, "fisheye": _lib.FISHEYE}[model]
    if cams is None:
        cams, n_poses = _gather_rig(objectPoints, imagePoints)
    nC = len(cams)
    if len(cameraMatrices) != nC or len(distCoeffs) != nC:
        raise ValueError("one cameraMatrix / distCoeffs per camera")
    nd = 5 if model == "pinhole" else 4
    intr0 = np.zeros((nC, 4 + nd))
    for c in range(nC):
        K = np.asarray(cameraMatrices[c], dtype=np.float64)
        D = np.asarray(distCoeffs[c], dtype=np.float64).ravel()
        if K.shape != (3, 3) or D.size != nd:
            raise ValueError("cameraMatrix must be 3x3 and distCoeffs hold %d coefficients" % nd)
        intr0[c] = np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], D])
    _lib_override, dev = _resolve(_lib_override, _device_override)
    R = RigEngine(mdl, cams, n_poses, dev, lib=_lib_override)
    comm = _comm or Comm()
    R.set_params(0, intr0, np.zeros((nC, 6)))
    if _init is None:
        R.init_from_guess(comm)
    else:
        R.set_params(0, intr0, _init[0], _init[1])
    NI, G = R.NI, R.G
    free = np.ones((nC, G), dtype=bool)
    free[0, NI:] = False
    if fix_intrinsics is not None:
        free[:, :NI] &= ~np.asarray(fix_intrinsics, dtype=bool)[None, :]
    ctype, maxc, eps = criteria
    if not ctype & 1:
        maxc = 1000
    if not ctype & 2:
        eps = DBL_EPSILON
    log = [] if _info is not None else None
    init_state = R.get_params(0) if _info is not None else None
    err, iters = lm_rig(R, free.reshape(-1), maxc, eps, comm=comm, log=log)
    n_total = comm.total(R.n_obs, dev)
    rms = float(np.sqrt(err / n_total))
    intr, ext, poses = R.get_params(0)
    Ks = [np.array([[v[0], 0.0, v[2]], [0.0, v[1], v[3]], [0.0, 0.0, 1.0]]) for v in intr]
    Ds = [v[4:].reshape(-1, 1).copy() for v in intr]
    Ts = [(e[:3].reshape(3, 1).copy(), e[3:].reshape(3, 1).copy()) for e in ext]
    npz = poses.shape[0]
    rvecs = tuple(np.ascontiguousarray(poses[:, :3]).reshape(npz, 3, 1))
    tvecs = tuple(np.ascontiguousarray(poses[:, 3:]).reshape(npz, 3, 1))
    if _info is not None:
        _info.update(iters=iters, log=log, engine=R, init=init_state, cost=err, n_obs=n_total)
    return rms, Ks, Ds, Ts, rvecs, tvecs
