"""Seeded synthetic calibration problems (BASELINE.json configs C1..C5).

There is no network and the reference ships no corner data (its inputs are
git-ignored), so benchmarks and tests run on synthetic views of
the reference's own boards: truth intrinsics are the example results in
``configs/internal/external_cfg.json:37-71`` and the guesses are the
hard-coded ones of ``intrinsicCalib.py:81-84,172-176`` (values restated in
BASELINE.md section 1).  Generation is torch-only so the 1M-view stress set can
be produced on the GPU in seconds; it has its own forward projection and does
not touch the solver.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

PINHOLE_TRUE_K = (1365.77, 1367.35, 959.51, 556.16)
PINHOLE_TRUE_D = (-0.40958, 0.23505, -1.70e-4, -9.53e-4, -0.08694)
PINHOLE_GUESS_K = (1345.159861, 1344.626518, 973.776987, 541.407299)
PINHOLE_GUESS_D = (-0.404256259818931, 0.21922420928359876, -0.00016181802429342013,
                   -7.126641144546051e-05, -0.06924570764420157)
FISHEYE_TRUE_K = (1066.58, 1066.31, 1478.65, 791.19)
FISHEYE_TRUE_D = (-0.02334, 0.007086, -0.005574, 0.000994)
FISHEYE2_TRUE_K = (1100.0336809276707, 1059.556842631112, 1500.1849532258357, 867.406824905331)   # left_calib
FISHEYE2_TRUE_D = (-0.021524158595833925, 0.00484506003924255, -0.003996810007141137, 0.0006446938071167349)
FISHEYE_GUESS_K = (1062.267560, 1062.212564, 1501.057278, 867.887492)
FISHEYE_GUESS_D = (-0.018123981144576858, -0.0026101267970621823, 0.0006215782316954739,
                   -0.00030905667520437564)


@dataclass
class Problem:
    model: str                     # "pinhole" | "fisheye"
    image_size: tuple
    K_true: tuple
    D_true: tuple
    K_guess: np.ndarray            # (3, 3)
    D_guess: np.ndarray            # (nd, 1)
    view_ptr: np.ndarray           # (N+1,) int64
    obj: torch.Tensor              # (T, 3) float32
    img: torch.Tensor              # (T, 2) float32
    poses_true: torch.Tensor       # (N, 6) float64
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def n_views(self):
        return int(self.view_ptr.shape[0] - 1)

    @property
    def n_obs(self):
        return int(self.view_ptr[-1])

    def as_lists(self):
        """cv2-style inputs: lists of (n,1,3) / (n,1,2) float32 arrays."""
        o = self.obj.cpu().numpy()
        i = self.img.cpu().numpy()
        vp = self.view_ptr
        objs = [o[vp[k]:vp[k + 1]].reshape(-1, 1, 3).copy() for k in range(self.n_views)]
        imgs = [i[vp[k]:vp[k + 1]].reshape(-1, 1, 2).copy() for k in range(self.n_views)]
        return objs, imgs

    def intr_true(self):
        return np.array(list(self.K_true) + list(self.D_true), dtype=np.float64)

    def intr_guess(self):
        K = self.K_guess
        return np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], self.D_guess.ravel()])


def board_points(cols, rows, square, dtype=torch.float64, device="cpu"):
    """cols x rows inner corners, row-major, Z = 0 (intrinsicCalib.py:266-268 style)."""
    ys, xs = torch.meshgrid(torch.arange(rows, dtype=dtype, device=device),
                            torch.arange(cols, dtype=dtype, device=device), indexing="ij")
    return torch.stack([xs.reshape(-1) * square, ys.reshape(-1) * square,
                        torch.zeros(rows * cols, dtype=dtype, device=device)], dim=1)


def _rot_from_rvec(r):
    th = r.norm(dim=-1, keepdim=True).clamp_min(1e-300)
    n = r / th
    c, s = torch.cos(th)[..., None], torch.sin(th)[..., None]
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    zero = torch.zeros_like(x)
    Kx = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=-1).reshape(r.shape[:-1] + (3, 3))
    eye = torch.eye(3, dtype=r.dtype, device=r.device).expand(r.shape[:-1] + (3, 3))
    nn = n[..., :, None] * n[..., None, :]
    return c * eye + (1 - c) * nn + s * Kx


def _rvec_from_rot(R):
    """Batched log map via the unit quaternion (robust near pi)."""
    m = R
    tr = m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]
    qw = torch.sqrt((1 + tr).clamp_min(0)) / 2
    qx = torch.sqrt((1 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2]).clamp_min(0)) / 2
    qy = torch.sqrt((1 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2]).clamp_min(0)) / 2
    qz = torch.sqrt((1 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2]).clamp_min(0)) / 2
    q = torch.stack([qw, qx, qy, qz], dim=-1)
    big = q.argmax(dim=-1)
    # rebuild the other components from the largest one
    d = [m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]]
    s_ = [m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1]]
    out = torch.empty_like(q)
    for b in range(4):
        sel = big == b
        if not sel.any():
            continue
        qb = q[sel][:, b]
        f = 1.0 / (4 * qb)
        if b == 0:
            comp = [qb, d[0][sel] * f, d[1][sel] * f, d[2][sel] * f]
        elif b == 1:
            comp = [d[0][sel] * f, qb, s_[0][sel] * f, s_[1][sel] * f]
        elif b == 2:
            comp = [d[1][sel] * f, s_[0][sel] * f, qb, s_[2][sel] * f]
        else:
            comp = [d[2][sel] * f, s_[1][sel] * f, s_[2][sel] * f, qb]
        out[sel] = torch.stack(comp, dim=-1)
    sign = torch.where(out[..., 0] < 0, -1.0, 1.0).to(out.dtype)
    out = out * sign[..., None]
    vn = out[..., 1:].norm(dim=-1)
    ang = 2 * torch.atan2(vn, out[..., 0])
    sc = torch.where(vn > 1e-12, ang / vn.clamp_min(1e-300), torch.full_like(vn, 2.0))
    return out[..., 1:] * sc[..., None]


def project(model, X, rvec, tvec, K4, D):
    """Forward projection (no derivatives).  X (..., n, 3), rvec/tvec (..., 3).
    Returns pixels (..., n, 2), camera-frame depth, and the model's radial
    quantity (pinhole r^2, fisheye theta)."""
    R = _rot_from_rvec(rvec)
    Y = torch.einsum("...ij,...nj->...ni", R, X) + tvec[..., None, :]
    fx, fy, cx, cy = K4
    z = Y[..., 2]
    x = Y[..., 0] / z
    y = Y[..., 1] / z
    r2 = x * x + y * y
    if model == "pinhole":
        k1, k2, p1, p2, k3 = D
        cd = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
        xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        rad = r2
    else:
        k1, k2, k3, k4 = D
        r = torch.sqrt(r2)
        th = torch.atan(r)
        th2 = th * th
        thd = th * (1 + th2 * (k1 + th2 * (k2 + th2 * (k3 + th2 * k4))))
        sc = torch.where(r > 1e-8, thd / r.clamp_min(1e-300), torch.ones_like(r))
        xd, yd = x * sc, y * sc
        rad = th
    return torch.stack([xd * fx + cx, yd * fy + cy], dim=-1), z, rad


def _axis_rot(axis, ang):
    c, s = torch.cos(ang), torch.sin(ang)
    o, z = torch.ones_like(ang), torch.zeros_like(ang)
    if axis == 0:
        m = [o, z, z, z, c, -s, z, s, c]
    elif axis == 1:
        m = [c, z, s, z, o, z, -s, z, c]
    else:
        m = [c, -s, z, s, c, z, z, z, o]
    return torch.stack(m, dim=-1).reshape(ang.shape + (3, 3))


def sample_poses(model, board, K4, D, size, n_views, gen, device="cpu", max_tilt_deg=35.0,
                 max_incidence_deg=70.0, margin=8.0, z_range=(600.0, 1400.0)):
    """Rejection-sample poses whose every corner lands inside the frame."""
    w, h = size
    fx, fy, cx, cy = K4
    dt = torch.float64
    centre_b = board.mean(dim=0)
    out = []
    have = 0
    max_inc = math.radians(max_incidence_deg)
    while have < n_views:
        B = int(min(max(4 * (n_views - have), 4096), 1 << 20))
        u = torch.rand(B, generator=gen, device=device, dtype=dt) * (0.8 * w) + 0.1 * w
        v = torch.rand(B, generator=gen, device=device, dtype=dt) * (0.8 * h) + 0.1 * h
        z = torch.rand(B, generator=gen, device=device, dtype=dt) * (z_range[1] - z_range[0]) + z_range[0]
        ax = (torch.rand(B, generator=gen, device=device, dtype=dt) * 2 - 1) * math.radians(max_tilt_deg)
        ay = (torch.rand(B, generator=gen, device=device, dtype=dt) * 2 - 1) * math.radians(max_tilt_deg)
        roll = (torch.rand(B, generator=gen, device=device, dtype=dt) * 2 - 1) * math.pi
        x = (u - cx) / fx
        y = (v - cy) / fy
        R = _axis_rot(0, ax) @ _axis_rot(1, ay) @ _axis_rot(2, roll)
        if model == "fisheye":
            thd = torch.sqrt(x * x + y * y).clamp_min(1e-12)
            th = thd.clamp_max(max_inc)
            tanth = torch.tan(th)
            centre = torch.stack([tanth * x / thd * z, tanth * y / thd * z, z], dim=-1)
            # face the board towards the camera centre
            d = centre / centre.norm(dim=-1, keepdim=True)
            zax = torch.tensor([0.0, 0.0, 1.0], dtype=dt, device=device).expand_as(d)
            axis = torch.cross(zax, d, dim=-1)
            sn = axis.norm(dim=-1, keepdim=True)
            ang = torch.atan2(sn[..., 0], d[..., 2])
            rface = axis / sn.clamp_min(1e-12) * ang[..., None]
            R = _rot_from_rvec(rface) @ R
        else:
            centre = torch.stack([x * z, y * z, z], dim=-1)
        t = centre - torch.einsum("bij,j->bi", R, centre_b)
        rvec = _rvec_from_rot(R)
        px, depth, rad = project(model, board.expand(B, -1, -1), rvec, t, K4, D)
        ok = (depth > 50.0).all(dim=1)
        ok &= (px[..., 0] > margin).all(dim=1) & (px[..., 0] < w - margin).all(dim=1)
        ok &= (px[..., 1] > margin).all(dim=1) & (px[..., 1] < h - margin).all(dim=1)
        if model == "fisheye":
            ok &= (rad < max_inc).all(dim=1)
        else:
            k1, k2, p1, p2, k3 = D
            mono = 1 + rad * (3 * k1 + rad * (5 * k2 + rad * 7 * k3))
            ok &= (mono > 0.3).all(dim=1)
        ok &= torch.isfinite(px).all(dim=2).all(dim=1)
        sel = torch.nonzero(ok)[:, 0]
        out.append(torch.cat([rvec[sel], t[sel]], dim=1))
        have += int(sel.numel())
    return torch.cat(out, dim=0)[:n_views]


def make_problem(model, n_views, board, size, K_true, D_true, K_guess, D_guess, seed, noise=0.2,
                 device="cpu", ragged=None, name="", max_incidence_deg=70.0):
    """Generic generator.  ``ragged=(nmin, nmax)`` keeps a random subset of the
    board's corners per view (ChArUco partial visibility)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    board = board.to(device=device, dtype=torch.float64)
    poses = sample_poses(model, board, K_true, D_true, size, n_views, gen, device=device,
                         max_incidence_deg=max_incidence_deg)
    N, nb = n_views, board.shape[0]
    chunks_obj, chunks_img, counts = [], [], []
    step = 1 << 16
    for a in range(0, N, step):
        b = min(a + step, N)
        px, _, _ = project(model, board.expand(b - a, -1, -1), poses[a:b, :3], poses[a:b, 3:], K_true, D_true)
        if noise > 0:
            px = px + noise * torch.randn(px.shape, generator=gen, device=device, dtype=torch.float64)
        if ragged is None:
            chunks_obj.append(board.expand(b - a, -1, -1).reshape(-1, 3).to(torch.float32))
            chunks_img.append(px.reshape(-1, 2).to(torch.float32))
            counts.append(torch.full((b - a,), nb, dtype=torch.int64, device=device))
        else:
            nmin, nmax = ragged
            keep_n = torch.randint(nmin, nmax + 1, (b - a,), generator=gen, device=device)
            rank = torch.rand(b - a, nb, generator=gen, device=device).argsort(dim=1).argsort(dim=1)
            keep = rank < keep_n[:, None]
            chunks_obj.append(board.expand(b - a, -1, -1)[keep].to(torch.float32))
            chunks_img.append(px[keep].to(torch.float32))
            counts.append(keep_n.to(torch.int64))
    counts = torch.cat(counts).cpu().numpy()
    view_ptr = np.zeros(N + 1, dtype=np.int64)
    view_ptr[1:] = np.cumsum(counts)
    Kg = np.array([[K_guess[0], 0, K_guess[2]], [0, K_guess[1], K_guess[3]], [0, 0, 1.0]])
    Dg = np.array(D_guess, dtype=np.float64).reshape(-1, 1)
    return Problem(model=model, image_size=tuple(size), K_true=tuple(K_true), D_true=tuple(D_true),
                   K_guess=Kg, D_guess=Dg, view_ptr=view_ptr, obj=torch.cat(chunks_obj),
                   img=torch.cat(chunks_img), poses_true=poses, name=name,
                   meta=dict(seed=seed, noise=noise, board=(int(nb),), ragged=ragged))


@dataclass
class RigProblem:
    """Config C4: cameras sharing board poses.  Per camera c the views it sees:
    ``pose_idx[c]`` (n_views_c,) board-pose index of each view, CSR ``view_ptr[c]``,
    ``obj[c]`` / ``img[c]`` float32 points."""
    model: str
    image_size: tuple
    n_poses: int
    intr_true: np.ndarray          # (n_cams, NI)
    intr_guess: np.ndarray         # (n_cams, NI)
    ext_true: np.ndarray           # (n_cams, 6)  camera-from-rig (rvec, tvec); camera 0 = 0
    poses_true: np.ndarray         # (P, 6)       rig-from-board
    pose_idx: list
    view_ptr: list
    obj: list
    img: list
    meta: dict = field(default_factory=dict)

    @property
    def n_cams(self):
        return len(self.pose_idx)

    @property
    def n_obs(self):
        return int(sum(int(v[-1]) for v in self.view_ptr))

    def as_lists(self):
        """objectPoints[c][p] / imagePoints[c][p]: (n,1,3)/(n,1,2) float32 or None."""
        objs, imgs = [], []
        for c in range(self.n_cams):
            o, i, vp = self.obj[c].cpu().numpy(), self.img[c].cpu().numpy(), self.view_ptr[c]
            lo = [None] * self.n_poses
            li = [None] * self.n_poses
            for k, p in enumerate(self.pose_idx[c]):
                lo[p] = o[vp[k]:vp[k + 1]].reshape(-1, 1, 3).copy()
                li[p] = i[vp[k]:vp[k + 1]].reshape(-1, 1, 2).copy()
            objs.append(lo)
            imgs.append(li)
        return objs, imgs

    def guess_matrices(self):
        Ks, Ds = [], []
        for g in self.intr_guess:
            Ks.append(np.array([[g[0], 0, g[2]], [0, g[1], g[3]], [0, 0, 1.0]]))
            Ds.append(np.array(g[4:], dtype=np.float64).reshape(-1, 1))
        return Ks, Ds


def make_rig(n_poses, n_cams=4, board=None, size=(2960, 1664), seed=4, noise=0.2, device="cpu",
             yaw_deg=(0.0, 35.0, -35.0, 70.0), baseline=120.0, max_incidence_deg=70.0, margin=8.0):
    """Fisheye rig: cameras fanned about the rig's Y axis, board poses sampled
    in the frame of one camera (round robin) and kept in every camera whose
    image holds all corners."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    dt = torch.float64
    board = (board if board is not None else board_points(11, 8, 25.0)).to(device=device, dtype=dt)
    nb = board.shape[0]
    # truths: the two fisheye entries of configs/internal/external_cfg.json (left_calib, right_calib)
    # and +-2 % perturbations of them
    left = np.array(list(FISHEYE2_TRUE_K) + list(FISHEYE2_TRUE_D))
    right = np.array(list(FISHEYE_TRUE_K) + list(FISHEYE_TRUE_D))
    pert = np.array([.02, .02, -.01, .015, .02, -.02, .02, -.02])
    truths = [left, right, left * (1 + pert), right * (1 - pert)]
    intr_true = np.stack([truths[c % 4] for c in range(n_cams)])
    guess = np.array(list(FISHEYE_GUESS_K) + list(FISHEYE_GUESS_D))
    intr_guess = np.stack([guess for _ in range(n_cams)])
    ext = torch.zeros(n_cams, 6, dtype=dt, device=device)
    for c in range(1, n_cams):
        yaw = math.radians(yaw_deg[c % len(yaw_deg)]) * (1 + c // len(yaw_deg))
        ext[c, 1] = yaw
        ext[c, 0] = 0.03 * c                                      # small pitch so no axis is special
        ext[c, 3] = -baseline * c * math.copysign(1.0, yaw if yaw != 0 else 1.0)
        ext[c, 4] = 5.0 * c
        ext[c, 5] = 10.0 * c
    Rc = _rot_from_rvec(ext[:, :3])
    tc = ext[:, 3:]
    # poses in the frame of camera (p mod n_cams), moved to the rig frame
    per = (n_poses + n_cams - 1) // n_cams
    poses = torch.empty(n_poses, 6, dtype=dt, device=device)
    for c in range(n_cams):
        idx = torch.arange(c, n_poses, n_cams, device=device)
        if idx.numel() == 0:
            continue
        pc = sample_poses("fisheye", board, tuple(intr_true[c, :4]), tuple(intr_true[c, 4:]), size, per, gen,
                          device=device, max_incidence_deg=max_incidence_deg, margin=margin)[:idx.numel()]
        Rpc = _rot_from_rvec(pc[:, :3])
        Rp = Rc[c].T @ Rpc
        tp = (pc[:, 3:] - tc[c]) @ Rc[c]                        # R_c^T (t_pc - t_c)
        poses[idx] = torch.cat([_rvec_from_rot(Rp), tp], dim=1)
    Rp = _rot_from_rvec(poses[:, :3])
    max_inc = math.radians(max_incidence_deg)
    w, h = size
    pose_idx, view_ptr, objs, imgs = [], [], [], []
    for c in range(n_cams):
        R = Rc[c] @ Rp
        t = poses[:, 3:] @ Rc[c].T + tc[c]
        px, depth, rad = project("fisheye", board.expand(n_poses, -1, -1), _rvec_from_rot(R), t,
                                 tuple(intr_true[c, :4]), tuple(intr_true[c, 4:]))
        ok = (depth > 50.0).all(dim=1) & (rad < max_inc).all(dim=1)
        ok &= (px[..., 0] > margin).all(dim=1) & (px[..., 0] < w - margin).all(dim=1)
        ok &= (px[..., 1] > margin).all(dim=1) & (px[..., 1] < h - margin).all(dim=1)
        ok &= torch.isfinite(px).all(dim=2).all(dim=1)
        sel = torch.nonzero(ok)[:, 0]
        pxs = px[sel]
        if noise > 0:
            pxs = pxs + noise * torch.randn(pxs.shape, generator=gen, device=device, dtype=dt)
        pose_idx.append(sel.cpu().numpy().astype(np.int64))
        vp = np.arange(sel.numel() + 1, dtype=np.int64) * nb
        view_ptr.append(vp)
        objs.append(board.expand(sel.numel(), -1, -1).reshape(-1, 3).to(torch.float32))
        imgs.append(pxs.reshape(-1, 2).to(torch.float32))
    return RigProblem(model="fisheye", image_size=tuple(size), n_poses=n_poses, intr_true=intr_true,
                      intr_guess=intr_guess, ext_true=ext.cpu().numpy(), poses_true=poses.cpu().numpy(),
                      pose_idx=pose_idx, view_ptr=view_ptr, obj=objs, img=imgs,
                      meta=dict(seed=seed, noise=noise, board=(int(nb),)))


def config(name, device="cpu", n_views=None, noise=None, seed=None):
    """BASELINE.json configs by name: C1, C2, C3, C5 return a :class:`Problem`,
    C4 (``n_views`` = board poses) a :class:`RigProblem`."""
    name = name.upper()
    if name == "C1":
        return make_problem("pinhole", n_views or 20, board_points(9, 6, 25.0), (1920, 1080),
                            PINHOLE_TRUE_K, PINHOLE_TRUE_D, PINHOLE_GUESS_K, PINHOLE_GUESS_D,
                            seed if seed is not None else 1, 0.2 if noise is None else noise, device,
                            name="C1")
    if name == "C2":
        return make_problem("fisheye", n_views or 500, board_points(11, 8, 25.0), (2960, 1664),
                            FISHEYE_TRUE_K, FISHEYE_TRUE_D, FISHEYE_GUESS_K, FISHEYE_GUESS_D,
                            seed if seed is not None else 2, 0.2 if noise is None else noise, device,
                            name="C2")
    if name == "C3":
        return make_problem("fisheye", n_views or 100000, board_points(9, 7, 50.0), (2960, 1664),
                            FISHEYE_TRUE_K, FISHEYE_TRUE_D, FISHEYE_GUESS_K, FISHEYE_GUESS_D,
                            seed if seed is not None else 3, 0.2 if noise is None else noise, device,
                            ragged=(13, 63), name="C3")
    if name == "C4":
        return make_rig(n_views or 10000, 4, seed=seed if seed is not None else 4,
                        noise=0.2 if noise is None else noise, device=device)
    if name == "C5":
        return make_problem("fisheye", n_views or 1000000, board_points(11, 8, 25.0), (2960, 1664),
                            FISHEYE_TRUE_K, FISHEYE_TRUE_D, FISHEYE_GUESS_K, FISHEYE_GUESS_D,
                            seed if seed is not None else 5, 0.2 if noise is None else noise, device,
                            name="C5")
    if name == "C1R":      # ragged pinhole views (the ChArUco partial boards of C3 through the pinhole model)
        return make_problem("pinhole", n_views or 2000, board_points(9, 7, 50.0), (1920, 1080),
                            PINHOLE_TRUE_K, PINHOLE_TRUE_D, PINHOLE_GUESS_K, PINHOLE_GUESS_D,
                            seed if seed is not None else 7, 0.2 if noise is None else noise, device,
                            ragged=(13, 63), name="C1R")
    if name == "C5P":      # the stress set through the pinhole model (same board and view count as C5)
        return make_problem("pinhole", n_views or 1000000, board_points(11, 8, 25.0), (1920, 1080),
                            PINHOLE_TRUE_K, PINHOLE_TRUE_D, PINHOLE_GUESS_K, PINHOLE_GUESS_D,
                            seed if seed is not None else 6, 0.2 if noise is None else noise, device,
                            name="C5P")
    raise ValueError(f"unknown config {name}")
