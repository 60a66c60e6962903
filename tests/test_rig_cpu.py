"""Config C4 (multi-camera rig) on CPU: the dense oracle against finite
differences, scipy and ground truth; the product's drivers (rig.py) over the
CPU lane emulator against the oracle; n_cams == 1 against the single-camera
solver; two gloo ranks against one.

Parity unpinned by construction: the reference has no joint solver
(controller_internal_calibration.py:375-399), see oracle/rig.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cameracalibration_amd import rig, synth
from oracle import cvcalib as CC
from oracle import rig as OR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.conftest import use_emulator  # noqa: E402
EMU = os.path.join(ROOT, "tests", "hostemu", "libcamcal_hostemu.so")
CPU = torch.device("cpu")


def small_rig(n_poses=16, n_cams=4, seed=7, noise=0.2):
    return synth.make_rig(n_poses, n_cams, board=synth.board_points(6, 5, 40.0), seed=seed, noise=noise)


def test_oracle_jacobian_fd():
    R = small_rig(8, 3)
    objs, imgs = R.as_lists()
    obs = OR.RigObs(objs, imgs)
    x = OR.pack(R.intr_true, R.ext_true, R.poses_true)
    e, J = OR.residual_jacobian("fisheye", obs, *OR.unpack("fisheye", 3, x))
    rng = np.random.default_rng(0)
    for _ in range(3):
        d = rng.standard_normal(x.size) * np.maximum(np.abs(x), 1e-3)
        h = 1e-6
        ep = OR.residual_jacobian("fisheye", obs, *OR.unpack("fisheye", 3, x + h * d), jac=False)
        em = OR.residual_jacobian("fisheye", obs, *OR.unpack("fisheye", 3, x - h * d), jac=False)
        fd, an = (ep - em) / (2 * h), J @ d
        assert np.abs(fd - an).max() < 1e-7 * np.abs(an).max()


def test_oracle_vs_scipy_and_truth():
    from scipy.optimize import least_squares
    R = small_rig(12, 3, seed=11)
    objs, imgs = R.as_lists()
    rms, intr, ext, poses, info = OR.rig_calibrate("fisheye", objs, imgs, R.intr_guess, max_iter=60, eps=1e-12)
    obs = OR.RigObs(objs, imgs)
    mask = OR.free_mask("fisheye", 3, 12)
    x0 = OR.pack(R.intr_guess, *info["init"])

    def fun(xf):
        x = x0.copy()
        x[mask] = xf
        return OR.residual_jacobian("fisheye", obs, *OR.unpack("fisheye", 3, x), jac=False)

    def jac(xf):
        x = x0.copy()
        x[mask] = xf
        return OR.residual_jacobian("fisheye", obs, *OR.unpack("fisheye", 3, x))[1][:, mask]

    sol = least_squares(fun, x0[mask], jac=jac, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale="jac")
    assert abs(2 * sol.cost - info["cost"]) < 1e-8 * info["cost"]
    xs = x0.copy()
    xs[mask] = sol.x
    i2, e2, p2 = OR.unpack("fisheye", 3, xs)
    assert np.allclose(i2[:, :4], intr[:, :4], rtol=1e-7)
    assert np.allclose(e2, ext, atol=1e-5) and np.allclose(p2, poses, atol=1e-4)
    # ground truth: noise-limited recovery
    assert 0.15 < rms < 0.3
    assert np.abs(intr[:, :4] / R.intr_true[:, :4] - 1).max() < 5e-3
    assert np.abs(ext[:, :3] - R.ext_true[:, :3]).max() < 3e-3
    assert np.abs(ext[:, 3:] - R.ext_true[:, 3:]).max() < 2.0


def test_emulated_rig_matches_oracle(emu):
    R = small_rig()
    objs, imgs = R.as_lists()
    Ks, Ds = R.guess_matrices()
    info = {"init": True}
    out = rig.calibrate_rig(objs, imgs, R.image_size, Ks, Ds, criteria=(3, 50, 1e-7), _info=info)
    log = []
    rms, intr, ext, poses, oi = OR.rig_calibrate("fisheye", objs, imgs, R.intr_guess, max_iter=50, eps=1e-7,
                                                 log=log)
    assert np.allclose(info["init"][1], oi["init"][0], atol=1e-9)
    assert np.allclose(info["init"][2], oi["init"][1], atol=1e-8)
    assert info["iters"] == oi["iters"]
    assert abs(out[0] - rms) < 1e-9
    i2, e2, p2 = info["engine"].get_params(0)
    assert np.allclose(i2[:, :4], intr[:, :4], rtol=1e-9)
    assert np.allclose(i2[:, 4:], intr[:, 4:], rtol=1e-6, atol=1e-10)
    assert np.allclose(e2, ext, atol=1e-8) and np.allclose(p2, poses, atol=1e-7)
    # output format
    assert len(out[1]) == 4 and out[1][0].shape == (3, 3) and out[2][0].shape == (4, 1)
    assert np.all(out[3][0][0] == 0) and np.all(out[3][0][1] == 0)
    assert len(out[4]) == 16 and out[4][0].shape == (3, 1)


def test_rig_blocks_match_dense_jacobian(emu):
    """Every block the kernels produce against J^T J of the oracle's dense Jacobian."""
    from cameracalibration_amd import _lib
    R = small_rig(10, 3, seed=3)
    objs, imgs = R.as_lists()
    cams, P = rig._gather_rig(objs, imgs)
    E = rig.RigEngine(_lib.FISHEYE, cams, P, CPU, lib=emu)
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
    M, rhs = E.reduced_system(0, 0.0)              # C, d of buffer 0 are still zero
    assert np.allclose(M.numpy(), H[:nG, :nG], rtol=1e-10, atol=1e-6 * np.abs(H[:nG, :nG]).max() * 1e-6)
    assert np.allclose(rhs.numpy(), b[:nG], rtol=1e-9, atol=1e-9 * np.abs(b).max())
    assert abs(float(E.cost(0)) - e @ e) < 1e-10 * (e @ e)
    tri = rig._tri_index(6)
    for p in range(P):
        Vp = sum(E.Vp[0][c, :, p].numpy()[tri] for c in range(3))
        q = nG + 6 * p
        sc = np.abs(H[q:q + 6, q:q + 6]).max()
        assert np.allclose(Vp, H[q:q + 6, q:q + 6], atol=1e-11 * sc)
        gp = sum(E.gp[0][c, :, p].numpy() for c in range(3))
        assert np.allclose(gp, b[q:q + 6], atol=1e-10 * np.abs(b).max())
        for c in range(3):
            Wp = E.Wp[0][c, :, p].numpy().reshape(E.G, 6)
            ref = H[c * E.G:(c + 1) * E.G, q:q + 6]
            assert np.allclose(Wp, ref, atol=1e-11 * max(np.abs(ref).max(), 1.0))


def test_single_camera_rig_is_calibrate_camera(emu):
    """n_cams == 1, pinhole: the rig solver walks the same CvLevMarq path as
    cv2.calibrateCamera's restatement and as the product's single-camera driver."""
    import cameracalibration_amd as cc
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    info = {}
    out = rig.calibrate_rig([objs], [imgs], P.image_size, [P.K_guess], [P.D_guess], model="pinhole",
                            criteria=(3, 100, 1e-6), _info=info)
    ref_info = {}
    ref = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                             criteria=(3, 100, 1e-6), _info=ref_info)
    assert info["iters"] == ref_info["iters"]
    assert abs(out[0] - ref[0]) < 1e-9
    assert np.allclose(out[1][0], ref[1], rtol=1e-8) and np.allclose(out[2][0], ref[2], rtol=1e-6, atol=1e-10)
    assert np.allclose(np.hstack(out[4]), np.hstack(ref[3]), atol=1e-7)
    assert np.allclose(np.hstack(out[5]), np.hstack(ref[4]), atol=1e-5)
    orc = CC.calibrate_camera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                              criteria=(3, 100, 1e-6))
    assert abs(out[0] - orc[0]) < 1e-7 and np.allclose(out[1][0], orc[1], rtol=1e-6)


def test_single_camera_fisheye_rig_reaches_fisheye_minimum(emu):
    """n_cams == 1, fisheye: cv2.fisheye.calibrate's damped Gauss-Newton and the
    rig's LM minimise the same cost; run to convergence they meet."""
    import cameracalibration_amd as cc
    P = synth.config("C2", n_views=40)
    objs, imgs = P.as_lists()
    out = rig.calibrate_rig([objs], [imgs], P.image_size, [P.K_guess], [P.D_guess], criteria=(3, 100, 1e-12))
    fl = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
    ref = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl,
                               criteria=(3, 300, 1e-13))
    assert abs(out[0] - ref[0]) < 1e-8
    assert np.allclose(out[1][0], ref[1], rtol=1e-6) and np.allclose(out[2][0], ref[2], rtol=1e-3, atol=1e-7)


def load_rig_golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "rig", "rig_fisheye_3x12.npz"), allow_pickle=False)
    cams = [dict(pose_idx=z[f"pose_idx{c}"], view_ptr=z[f"view_ptr{c}"], obj=z[f"obj{c}"], img=z[f"img{c}"])
            for c in range(3)]
    Ks = [np.array([[g[0], 0, g[2]], [0, g[1], g[3]], [0, 0, 1.0]]) for g in z["intr_guess"]]
    Ds = [g[4:].reshape(-1, 1) for g in z["intr_guess"]]
    return z, cams, Ks, Ds


def check_rig_golden(z, out, info):
    assert info["iters"] == int(z["iters"])
    assert abs(out[0] - float(z["rms"])) < 1e-8
    i2, e2, p2 = info["engine"].get_params(0)
    assert np.allclose(i2[:, :4], z["intr"][:, :4], rtol=1e-8)
    assert np.allclose(i2[:, 4:], z["intr"][:, 4:], rtol=1e-6, atol=1e-9)
    assert np.allclose(e2, z["ext"], atol=1e-7) and np.allclose(p2, z["poses"], atol=1e-6)


def test_rig_golden_on_emulator(emu):
    z, cams, Ks, Ds = load_rig_golden()
    info = {}
    out = rig.calibrate_rig(None, None, tuple(z["image_size"]), Ks, Ds, criteria=(3, 40, 1e-7), cams=cams,
                            n_poses=int(z["n_poses"]), _info=info)
    check_rig_golden(z, out, info)


def test_rig_rejects_bad_input(emu):
    R = small_rig(6, 2)
    objs, imgs = R.as_lists()
    Ks, Ds = R.guess_matrices()
    with pytest.raises(ValueError):
        rig.calibrate_rig(objs, imgs[:1], R.image_size, Ks, Ds)
    with pytest.raises(ValueError):
        rig.calibrate_rig(objs, imgs, R.image_size, Ks[:1], Ds)
    # camera 1 sharing no pose with camera 0
    o2 = [objs[0], [None] * 6]
    i2 = [imgs[0], [None] * 6]
    with pytest.raises(ValueError):
        rig.calibrate_rig(o2, i2, R.image_size, Ks, Ds)


# --- two ranks over gloo -----------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cameracalibration_amd import _lib
    from cameracalibration_amd.distributed import shard_rig_lists
    from cameracalibration_amd.solver import Comm
    emu = _lib.load(EMU)
    from cameracalibration_amd import api
    use_emulator(emu)      # the workers run on the CPU lane emulator
    R = small_rig(24, 3, seed=5)
    objs, imgs = R.as_lists()
    o, i, (lo, hi) = shard_rig_lists(objs, imgs, rank, world)
    Ks, Ds = R.guess_matrices()
    r = rig.calibrate_rig(o, i, R.image_size, Ks, Ds, criteria=(3, 40, 1e-9), _comm=Comm())
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), rms=r[0], K=np.stack(r[1]), D=np.stack(r[2]),
             T=np.stack([np.vstack(t) for t in r[3]]), rv=np.hstack(r[4]).T, tv=np.hstack(r[5]).T, lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_rank_rig_equals_single_rank(emu, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = small_rig(24, 3, seed=5)
    objs, imgs = R.as_lists()
    Ks, Ds = R.guess_matrices()
    ref = rig.calibrate_rig(objs, imgs, R.image_size, Ks, Ds, criteria=(3, 40, 1e-9))
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == 24
    for p in parts:
        assert abs(float(p["rms"]) - ref[0]) < 1e-9
        assert np.allclose(p["K"], np.stack(ref[1]), rtol=1e-9)
        # k1..k4 are weakly determined on 24 small boards: summation order shows at 1e-6 relative
        assert np.allclose(p["D"], np.stack(ref[2]), rtol=1e-4, atol=1e-8)
        assert np.allclose(p["T"], np.stack([np.vstack(t) for t in ref[3]]), atol=1e-8)
    rv = np.concatenate([p["rv"] for p in parts])
    tv = np.concatenate([p["tv"] for p in parts])
    assert np.allclose(rv, np.hstack(ref[4]).T, atol=1e-8) and np.allclose(tv, np.hstack(ref[5]).T, atol=1e-6)


def test_library_sequenced_rig_loop_equals_the_python_loop(emu):
    """camcal_rig_lm (default) against the same control flow sequenced by rig.lm_rig (chosen by asking
    for the iteration log): identical stage calls, identical results."""
    R = small_rig()
    objs, imgs = R.as_lists()
    Ks, Ds = R.guess_matrices()
    res = []
    for info in ({}, {"log": True}):
        out = rig.calibrate_rig(objs, imgs, R.image_size, Ks, Ds, criteria=(3, 50, 1e-7), _info=info)
        res.append((out, info))
    (a, ia), (b, ib) = res
    assert ia["log"] is None and len(ib["log"]) == ib["iters"] == ia["iters"]
    assert a[0] == b[0]
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert np.array_equal(x, y)
    assert np.array_equal(np.hstack(a[4]), np.hstack(b[4]))


def test_packed_rig_shards_partition_the_views():
    """distributed.shard_rig_cams (the packed form bench.py's sharded C4 solve uses): the shards of
    1, 2 and 3 ranks cover every view exactly once, in order, with rebased pose indices and CSR offsets."""
    from cameracalibration_amd.distributed import shard_rig_cams
    R = synth.config("C4", n_views=200)
    cams = [dict(pose_idx=R.pose_idx[c], view_ptr=R.view_ptr[c], obj=R.obj[c], img=R.img[c]) for c in range(R.n_cams)]
    for world in (1, 2, 3):
        parts = [shard_rig_cams(cams, R.n_poses, r, world) for r in range(world)]
        assert parts[0][2][0] == 0 and parts[-1][2][1] == R.n_poses
        assert all(parts[r][2][1] == parts[r + 1][2][0] for r in range(world - 1))
        for c in range(R.n_cams):
            pi = np.concatenate([p[0][c]["pose_idx"] + p[2][0] for p in parts])
            assert np.array_equal(pi, R.pose_idx[c])
            for key in ("obj", "img"):
                got = np.concatenate([np.asarray(p[0][c][key]) for p in parts])
                assert np.array_equal(got, np.asarray(getattr(R, key)[c]))
            n = np.concatenate([np.diff(np.asarray(p[0][c]["view_ptr"])) for p in parts])
            assert np.array_equal(n, np.diff(np.asarray(R.view_ptr[c])))
            for p in parts:
                assert int(p[0][c]["view_ptr"][0]) == 0 and int(p[0][c]["view_ptr"][-1]) == len(p[0][c]["obj"])
                assert p[0][c]["pose_idx"].size == 0 or (0 <= p[0][c]["pose_idx"].min()
                                                          and p[0][c]["pose_idx"].max() < p[1])
