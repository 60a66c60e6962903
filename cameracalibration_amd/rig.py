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


def _tri_index(n):
    """(n, n) -> index into the packed upper triangle used by the library."""
    idx = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(n):
            a, b = (i, j) if i <= j else (j, i)
            idx[i, j] = a * n - (a * (a - 1)) // 2 + (b - a)
    return idx


class RigEngine:
    """Device state of one rank's share of the board poses: the buffers of a
    ``camcal_rig`` (include/camcal.h) and thin wrappers over its stage calls.

    cams: list (one per camera) of dict(pose_idx (n_c,), view_ptr (n_c+1,),
    obj (T_c,3), img (T_c,2)); ``n_poses`` = poses held by this rank."""

    def __init__(self, model, cams, n_poses, device, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.model = int(model)
        self.device = dev = torch.device(device)
        self.n_cams = len(cams)
        if not 1 <= self.n_cams <= _lib.RIG_MAX_CAMS:
            raise ValueError("1 to %d cameras are supported" % _lib.RIG_MAX_CAMS)
        self.P = int(max(n_poses, 1))
        self.n_poses = int(n_poses)
        self.E = [Engine(model, c["view_ptr"], c["obj"], c["img"], dev, lib=self.lib) for c in cams]
        self.NI = self.E[0].NI
        self.G = self.NI + 6
        self.nG = self.n_cams * self.G
        if self.nG > 63:
            raise ValueError("at most 63 global parameters (4 cameras) are supported by the reduced solve")
        self.NV = self.E[0].glob_len
        self.NX = self.NI * 6 + 27
        self.n_obs = sum(e.n_obs for e in self.E)
        f64 = dict(dtype=torch.float64, device=dev)
        sps = []
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
            sps.append(sp)
        self.seen = seen
        t_sp = torch.from_numpy(np.concatenate(sps)).to(dev)          # one upload for all cameras
        self.slot_pose, o = [], 0
        for e in self.E:
            self.slot_pose.append(t_sp[o:o + e.S])
            o += e.S
        nC, P, G, n = self.n_cams, self.P, self.G, self.nG
        # rec[b] = [per-camera records | cost | norms | gstats | C | d] (include/camcal.h)
        self.rec_len = nC * (self.NV + self.NX)
        self.o_cost, self.o_norms, self.o_gstats = self.rec_len, self.rec_len + 1, self.rec_len + 3
        self.o_C = self.rec_len + 5
        total = int(self.lib.camcal_rig_rec_len(self.model, nC))
        assert total == self.o_C + n * n + n
        nt = np.asarray([e.layout["n_tiles"] for e in self.E], dtype=np.int32)
        n_scratch = int(self.lib.camcal_rig_scratch_len(self.model, nC, nt.ctypes.data, P))
        # one zeroed allocation, carved up: the pose blocks of (camera, pose) pairs without a
        # view must be zero, everything else is written before it is read
        sizes = [("intr0", nC * 16), ("intr1", nC * 16), ("ext0", nC * 6), ("ext1", nC * 6), ("rec0", total), ("rec1", total),
                 ("dG", 64), ("pose0", 6 * P), ("pose1", 6 * P), ("z", 6 * P),
                 ("Vp0", nC * 21 * P), ("Vp1", nC * 21 * P), ("gp0", nC * 6 * P), ("gp1", nC * 6 * P),
                 ("Wp0", nC * G * 6 * P), ("Wp1", nC * G * 6 * P), ("Yt", n * 6 * P), ("scratch", n_scratch)]
        pool = torch.zeros(sum(k for _, k in sizes), **f64)
        part, off = {}, 0
        for name, k in sizes:
            part[name] = pool[off:off + k]
            off += k
        self._pool = pool
        self.intr = [part["intr0"].view(nC, 16), part["intr1"].view(nC, 16)]
        self.ext = [part["ext0"].view(nC, 6), part["ext1"].view(nC, 6)]
        self.rig_pose = [part["pose0"].view(6, P), part["pose1"].view(6, P)]
        pool[part["pose0"].storage_offset():part["pose1"].storage_offset() + 6 * P].view(2, 6, P)[:, 5] = 1.0
        self.Vp = [part["Vp0"].view(nC, 21, P), part["Vp1"].view(nC, 21, P)]
        self.gp = [part["gp0"].view(nC, 6, P), part["gp1"].view(nC, 6, P)]
        self.Wp = [part["Wp0"].view(nC, G * 6, P), part["Wp1"].view(nC, G * 6, P)]
        self.rec = [part["rec0"], part["rec1"]]
        self.Yt = part["Yt"].view(n * 6, P)
        self.z = part["z"].view(6, P)
        self.dG = part["dG"]
        self.scratch = part["scratch"]
        S_all = sum(e.S for e in self.E)
        ints = torch.zeros(S_all + nC * P, dtype=torch.int32, device=dev)     # pose_slot: filled by camcal_rig_init
        self.status = ints[:S_all]
        self.pose_slot = [ints[S_all + c * P:S_all + (c + 1) * P] for c in range(nC)]
        self._tri = None
        self._fill_struct()

    def _fill_struct(self):
        r = _lib.Rig()
        r.model, r.n_cams, r.P = self.model, self.n_cams, self.P
        for c, e in enumerate(self.E):
            k = r.cam[c]
            k.tiles = e.tiles
            k.slot_pose = self.slot_pose[c].data_ptr()
            k.pose_slot = self.pose_slot[c].data_ptr()
            k.pose = e.pose[0].data_ptr()
            k.V, k.W, k.g = e.Vs[0].data_ptr(), e.W.data_ptr(), e.gs[0].data_ptr()
        for b in range(2):
            r.intr[b], r.ext[b], r.rig_pose[b] = self.intr[b].data_ptr(), self.ext[b].data_ptr(), self.rig_pose[b].data_ptr()
            r.Vp[b], r.gp[b], r.Wp[b] = self.Vp[b].data_ptr(), self.gp[b].data_ptr(), self.Wp[b].data_ptr()
            r.rec[b] = self.rec[b].data_ptr()
        r.Yt, r.z, r.dG = self.Yt.data_ptr(), self.z.data_ptr(), self.dG.data_ptr()
        r.scratch, r.status = self.scratch.data_ptr(), self.status.data_ptr()
        self.c_rig = r

    def _swap(self):
        """Buffer 1 becomes buffer 0 (the library reported its result there)."""
        for buf in (self.intr, self.ext, self.rig_pose, self.Vp, self.gp, self.Wp, self.rec):
            buf.reverse()
        self._fill_struct()

    # -- plumbing --------------------------------------------------------------
    def _call(self, name, *args):
        self.E[0]._call(name, *args)

    def _stream(self):
        return self.E[0]._stream()

    def _rp(self):
        return C.byref(self.c_rig)

    def cam_rec(self, b, c):
        o = c * (self.NV + self.NX)
        return self.rec[b][o:o + self.NV], self.rec[b][o + self.NV:o + self.NV + self.NX]

    def cd(self, b):
        n = self.nG
        return self.rec[b][self.o_C:self.o_C + n * n].view(n, n), self.rec[b][self.o_C + n * n:]

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
        self._call("camcal_rig_evaluate", self._rp(), int(b), self._stream())

    def eliminate(self, b, lam):
        """Pose elimination at damping ``lam`` -> [C|d] of rec[b] (Y^T Y on the matrix cores)."""
        self._call("camcal_rig_eliminate", self._rp(), int(b), float(lam), self._stream())

    def cost(self, b):
        return self.rec[b][self.o_cost]

    def trial_scalars(self, b):
        """(cost, [||dG||^2, ||globals||^2], [||d poses||^2, ||poses||^2]) of a trial step: one read-back."""
        out = self.rec[b][self.o_cost:self.o_cost + 5].cpu().numpy()
        return float(out[0]), out[3:5], out[1:3]

    def reduced_system(self, b, lam):
        """(M, rhs) over the n_cams*(NI+6) globals, diag(M) damped like CvLevMarq."""
        NI, G, nC = self.NI, self.G, self.n_cams
        nt = NI * (NI + 1) // 2
        if self._tri is None:
            self._tri = (torch.from_numpy(_tri_index(NI)).to(self.device), torch.from_numpy(_tri_index(6)).to(self.device))
        M = torch.zeros(self.nG, self.nG, dtype=torch.float64, device=self.device)
        rhs = torch.zeros(self.nG, dtype=torch.float64, device=self.device)
        for c in range(nC):
            glob, rx = self.cam_rec(b, c)
            o = c * G
            M[o:o + NI, o:o + NI] = glob[:nt][self._tri[0]]
            Uie = rx[:NI * 6].view(NI, 6)
            M[o:o + NI, o + NI:o + G] = Uie
            M[o + NI:o + G, o:o + NI] = Uie.t()
            M[o + NI:o + G, o + NI:o + G] = rx[NI * 6:NI * 6 + 21][self._tri[1]]
            rhs[o:o + NI] = glob[nt:nt + NI]
            rhs[o + NI:o + G] = rx[NI * 6 + 21:]
        M.diagonal().mul_(1.0 + lam)
        Cm, d = self.cd(b)
        return M - Cm, rhs - d

    def solve(self, b, lam, free, out):
        """dG = M^-1 rhs on the free globals (``free``: boolean mask over the
        n_cams*(NI+6) globals); parameters[out] = parameters[b] - dG; gstats of rec[out]."""
        m = bytes(bytearray(int(bool(x)) for x in free))
        self._call("camcal_rig_solve", self._rp(), int(b), m, float(lam), int(out), self._stream())

    def backsub(self, b, lam, out):
        self._call("camcal_rig_backsub", self._rp(), int(b), float(lam), int(out), self._stream())

    # -- initialisation ----------------------------------------------------------
    def init_from_guess(self, comm):
        """camcal_rig_init: per-camera pose initialisation at the guessed intrinsics,
        camera-from-rig transforms from the chordal mean of T_pc o T_pk^-1 over shared
        poses, board poses from the lowest-numbered camera that sees them.  Writes buffer 0."""
        try:
            self._call("camcal_rig_init", self._rp(), comm.handle(self), self._stream())
        except _lib.CamcalStatus as exc:
            if exc.code == _lib.E_DISJOINT:
                raise ValueError("a camera shares no board pose with the rest of the rig") from exc
            if exc.code == _lib.E_PEER:
                raise ValueError("the rig initialisation failed on another rank") from exc
            raise

    def _comm_pools(self):
        """Tensors a torch.distributed callback communicator may be asked to reduce parts of."""
        return [self._pool]


def lm_rig(R, free, max_iter, eps, comm=None, log=None):
    """CvLevMarq::updateAlt over the rig (same control flow as
    solver.lm_pinhole).  Buffer 0 holds the starting point and, on return, the
    result.  Returns (errNorm2, iterations).  Sequenced by the library
    (camcal_rig_lm) unless the iteration log is asked for."""
    comm = comm or Comm(enabled=False)
    max_iter = int(min(max(max_iter, 1), 1000))
    eps = max(float(eps), DBL_EPSILON)
    free = np.asarray(free, dtype=bool)
    if log is None:
        m = bytes(bytearray(int(bool(x)) for x in free))
        err, its, cur = C.c_double(0.0), C.c_int32(0), C.c_int32(0)
        R._call("camcal_rig_lm", R._rp(), m, max_iter, eps, comm.handle(R), R._stream(), C.byref(err), C.byref(its),
                C.byref(cur))
        if cur.value == 1:
            R._swap()
        return float(err.value), int(its.value)
    lam_lg10 = -3
    cur, tri = 0, 1
    R.evaluate(cur)
    comm.allreduce(R.rec[cur][:R.o_cost + 1])
    err = float(R.cost(cur).item())
    iters = 0

    def step(lam):
        R.eliminate(cur, lam)
        comm.allreduce(R.rec[cur][R.o_C:])
        R.solve(cur, lam, free, tri)
        R.backsub(cur, lam, tri)
        R.evaluate(tri)
        comm.allreduce(R.rec[tri][:R.o_gstats])
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
        cur, tri = tri, cur
        if log is not None:
            log.append(dict(iter=iters, err=err, rel=rel, lam_lg10=lam_lg10))
        if iters >= max_iter or rel < eps:
            if cur == 1:
                R._swap()
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
                  _comm=None, _info=None, _init=None):
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
    from . import api as _api
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
    lib, dev = _api._resolve()
    R = RigEngine(mdl, cams, n_poses, dev, lib=lib)
    comm = _comm or Comm()
    R.set_params(0, intr0, np.zeros((nC, 6)))
    if _init is None:
        R.init_from_guess(comm)
        # a collective decision: the ranks that hold no degenerate view must not walk into the loop's all-reduce alone
        n_bad = int(R.status.count_nonzero().item())
        if comm.total(n_bad, dev) > 0:
            raise ValueError("a view is degenerate (collinear / coincident points): its pose initialisation failed"
                             + ("" if n_bad else " (on another rank)"))
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
    log = [] if (_info is not None and _info.get("log") is True) else None
    init_state = R.get_params(0) if (_info is not None and _info.get("init") is True) else None
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
