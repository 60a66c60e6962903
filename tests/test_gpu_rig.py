"""Config C4 (multi-camera rig) on the MI355X through the C ABI: blocks and the
whole solve against the dense oracle at oracle-sized problems, the 10k-pose
configuration through ground truth and size-independent properties.

Parity unpinned by construction (no joint solver in the reference, see
oracle/rig.py); tolerances: rms 1e-6 px, K rel 1e-6 (BASELINE.json north star)."""
import time

import numpy as np
import pytest
import torch

from cameracalibration_amd import _lib, rig, synth
from oracle import rig as OR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lib = _lib.load()
    assert not lib.is_emulator
    return torch.device("cuda", 0)


def small_rig(n_poses=16, n_cams=4, seed=7, noise=0.2):
    return synth.make_rig(n_poses, n_cams, board=synth.board_points(6, 5, 40.0), seed=seed, noise=noise)


def test_rig_blocks_match_dense_jacobian_gpu(dev):
    R = small_rig(70, 3, seed=3)                      # > 64 views per camera: two tiles
    objs, imgs = R.as_lists()
    cams, P = rig._gather_rig(objs, imgs)
    E = rig.RigEngine(_lib.FISHEYE, cams, P, dev)
    rng = np.random.default_rng(1)
    intr = R.intr_true * (1 + 1e-3 * rng.standard_normal(R.intr_true.shape))
    ext = R.ext_true + 1e-3 * rng.standard_normal(R.ext_true.shape)
    poses = R.poses_true + 1e-3 * rng.standard_normal(R.poses_true.shape)
    E.set_params(0, intr, ext, poses)
    E.evaluate(0)
    obs = OR.RigObs(objs, imgs)
    e, J = OR.residual_jacobian("fisheye", obs, intr, ext, poses)
    H, b = J.T @ J, J.T @ e
    nG = E.nG
    M, rhs = E.reduced_system(0, 0.0)
    M, rhs = M.cpu().numpy(), rhs.cpu().numpy()
    assert np.abs(M - H[:nG, :nG]).max() < 1e-11 * np.abs(H[:nG, :nG]).max()
    assert np.abs(rhs - b[:nG]).max() < 1e-10 * np.abs(b).max()
    assert abs(float(E.cost(0)) - e @ e) < 1e-11 * (e @ e)
    tri = rig._tri_index(6)
    Vp, gp, Wp = E.Vp[0].cpu().numpy(), E.gp[0].cpu().numpy(), E.Wp[0].cpu().numpy()
    for p in range(P):
        q = nG + 6 * p
        blk = H[q:q + 6, q:q + 6]
        assert np.abs(Vp[:, :, p].sum(0)[tri] - blk).max() < 1e-11 * np.abs(blk).max()
        assert np.abs(gp[:, :, p].sum(0) - b[q:q + 6]).max() < 1e-10 * np.abs(b).max()
        for c in range(3):
            ref = H[c * E.G:(c + 1) * E.G, q:q + 6]
            assert np.abs(Wp[c, :, p].reshape(E.G, 6) - ref).max() < 1e-11 * max(np.abs(ref).max(), 1.0)
    # pose elimination: C = sum W V^-1 W^T, d = sum W V^-1 g for a damped system
    lam = 1e-2
    E.eliminate(0, lam)
    Cm, d = E.cd(0)
    Cref = np.zeros((nG, nG))
    dref = np.zeros(nG)
    for p in range(P):
        q = nG + 6 * p
        V = H[q:q + 6, q:q + 6].copy()
        V[np.diag_indices(6)] *= 1 + lam
        W = H[:nG, q:q + 6]
        Cref += W @ np.linalg.solve(V, W.T)
        dref += W @ np.linalg.solve(V, b[q:q + 6])
    assert np.abs(Cm.cpu().numpy() - Cref).max() < 1e-9 * np.abs(Cref).max()
    assert np.abs(d.cpu().numpy() - dref).max() < 1e-9 * np.abs(dref).max()


def test_rig_solve_matches_oracle_gpu(dev):
    R = small_rig()
    objs, imgs = R.as_lists()
    Ks, Ds = R.guess_matrices()
    info = {"init": True}
    out = rig.calibrate_rig(objs, imgs, R.image_size, Ks, Ds, criteria=(3, 50, 1e-7),
                            _info=info)
    rms, intr, ext, poses, oi = OR.rig_calibrate("fisheye", objs, imgs, R.intr_guess, max_iter=50, eps=1e-7)
    assert np.allclose(info["init"][1], oi["init"][0], atol=1e-8)
    assert info["iters"] == oi["iters"]
    assert abs(out[0] - rms) < 1e-6                                   # north-star tolerance
    i2, e2, p2 = info["engine"].get_params(0)
    assert np.allclose(i2[:, :4], intr[:, :4], rtol=1e-6)
    assert np.allclose(i2[:, 4:], intr[:, 4:], rtol=1e-6, atol=1e-9)
    assert np.allclose(e2, ext, atol=1e-7) and np.allclose(p2, poses, atol=1e-6)


def test_rig_golden_gpu(dev):
    from tests.test_rig_cpu import load_rig_golden, check_rig_golden
    z, cams, Ks, Ds = load_rig_golden()
    info = {}
    out = rig.calibrate_rig(None, None, tuple(z["image_size"]), Ks, Ds, criteria=(3, 40, 1e-7), cams=cams,
                            n_poses=int(z["n_poses"]), _info=info)
    check_rig_golden(z, out, info)


def test_single_camera_rig_is_calibrate_camera_gpu(dev):
    import cameracalibration_amd as cc
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    info, ref_info = {}, {}
    out = rig.calibrate_rig([objs], [imgs], P.image_size, [P.K_guess], [P.D_guess], model="pinhole",
                            criteria=(3, 100, 1e-6), _info=info)
    ref = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                             criteria=(3, 100, 1e-6), _info=ref_info)
    assert info["iters"] == ref_info["iters"]
    assert abs(out[0] - ref[0]) < 1e-9
    assert np.allclose(out[1][0], ref[1], rtol=1e-8) and np.allclose(out[2][0], ref[2], rtol=1e-6, atol=1e-10)


def test_config4_full_size_recovers_truth(dev):
    """4 cameras x 10k board poses x 88 corners: noise-limited recovery of every
    camera's intrinsics and of the rig geometry; cost at the solution equals the
    dof-corrected noise level; a second solve from the solution stops at once."""
    R = synth.config("C4", device=dev)
    cams = [dict(pose_idx=R.pose_idx[c], view_ptr=R.view_ptr[c], obj=R.obj[c], img=R.img[c]) for c in range(4)]
    Ks, Ds = R.guess_matrices()
    info = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rig.calibrate_rig(None, None, R.image_size, Ks, Ds, criteria=(3, 30, 1e-7), cams=cams,
                            n_poses=R.n_poses, _info=info)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_obs = R.n_obs
    n_seen = int(info["engine"].seen.sum())
    dof = 2 * n_obs - 6 * n_seen - 4 * 8 - 3 * 6
    expect = 0.2 * np.sqrt(dof / n_obs)
    print(f"\nC4: {n_obs} obs, {n_seen} poses, rms {out[0]:.6f} (expected {expect:.6f}), "
          f"{info['iters']} LM iterations, {el:.2f} s")
    assert abs(out[0] - expect) < 2e-3
    intr, ext, poses = info["engine"].get_params(0)
    assert np.abs(intr[:, :4] / R.intr_true[:, :4] - 1).max() < 1e-4
    assert np.abs(intr[:, 4:] - R.intr_true[:, 4:]).max() < 2e-4
    assert np.abs(ext[:, :3] - R.ext_true[:, :3]).max() < 2e-5
    assert np.abs(ext[:, 3:] - R.ext_true[:, 3:]).max() < 0.05            # mm
    seen = info["engine"].seen
    Re = synth._rot_from_rvec(torch.from_numpy(poses[seen, :3]))        # rotation vectors near pi flip sign:
    Rt = synth._rot_from_rvec(torch.from_numpy(R.poses_true[seen, :3]))  # compare the rotations themselves
    ang = torch.linalg.matrix_norm(Re @ Rt.transpose(1, 2) - torch.eye(3, dtype=torch.float64))
    # (a handful of distant, near-frontal boards sit in the planar two-fold pose ambiguity)
    print(f"pose rotation error: median {float(ang.median()):.2e}, max {float(ang.max()):.2e}, "
          f"{int((ang >= 0.05).sum())} of {ang.numel()} above 0.05")
    assert float((ang < 0.05).double().mean()) > 0.998 and float(ang.median()) < 2e-3
    # idempotence: restarting at the solution takes one iteration and moves nothing
    K2 = [k.copy() for k in out[1]]
    D2 = [d.copy() for d in out[2]]
    info2 = {}
    out2 = rig.calibrate_rig(None, None, R.image_size, K2, D2, criteria=(3, 30, 1e-7), cams=cams,
                             n_poses=R.n_poses, _info=info2, _init=(ext, poses))
    assert info2["iters"] <= 2 and abs(out2[0] - out[0]) < 1e-8
    assert np.allclose(np.stack(out2[1]), np.stack(out[1]), rtol=1e-8)
