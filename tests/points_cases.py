"""Shared bodies of the point-level parity tests (N1-N4): run with the CPU lane
emulator in the CPU tier and with the HIP library in the GPU tier."""
import numpy as np
import pytest

import cameracalibration_amd as cc
from cameracalibration_amd import synth, wrappers
from oracle import cvcalib as O
from oracle import cvmodel as M

FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS


def Kmat(k4):
    return np.array([[k4[0], 0, k4[2]], [0, k4[1], k4[3]], [0, 0, 1.0]])


def case_project_points():
    rng = np.random.default_rng(0)
    X = np.concatenate([rng.uniform(-300, 300, (50, 2)), rng.uniform(-40, 40, (50, 1))], axis=1)
    rvec, tvec = np.array([0.2, -0.3, 0.5]), np.array([20.0, -30.0, 900.0])
    for model, K4, D, fn in (("pinhole", synth.PINHOLE_TRUE_K, synth.PINHOLE_TRUE_D, cc.projectPoints),
                             ("fisheye", synth.FISHEYE_TRUE_K, synth.FISHEYE_TRUE_D, cc.fisheye.projectPoints)):
        ref = (M.project_pinhole if model == "pinhole" else M.project_fisheye)(X, rvec, tvec, K4, D)
        got, jac = fn(X.reshape(-1, 1, 3), rvec.reshape(3, 1), tvec.reshape(3, 1), Kmat(K4), np.array(D))
        assert jac is None and got.shape == (50, 1, 2) and got.dtype == np.float64
        assert np.abs(got.reshape(-1, 2) - ref).max() < 1e-9
        got32, _ = fn(X.astype(np.float32).reshape(-1, 1, 3), rvec, tvec, Kmat(K4), np.array(D))
        assert got32.dtype == np.float32
        ref32 = (M.project_pinhole if model == "pinhole" else M.project_fisheye)(
            X.astype(np.float32).astype(np.float64), rvec, tvec, K4, D)
        assert np.abs(got32.reshape(-1, 2) - ref32).max() < 2e-4
    # pinhole with 4 coefficients (the "normal" cameras of configs/fg/inter_cfg.json)
    got, _ = cc.projectPoints(X.reshape(-1, 1, 3), rvec, tvec, Kmat(synth.PINHOLE_TRUE_K),
                              np.array(synth.PINHOLE_TRUE_D[:4]))
    ref = M.project_pinhole(X, rvec, tvec, synth.PINHOLE_TRUE_K, synth.PINHOLE_TRUE_D[:4])
    assert np.abs(got.reshape(-1, 2) - ref).max() < 1e-9


def case_reproj_errors():
    for name, model in (("C1", "pinhole"), ("C3", "fisheye")):
        P = synth.config(name, n_views=70)
        objs, imgs = P.as_lists()
        intr = P.intr_true() * (1 + 1e-4)
        poses = P.poses_true.numpy()
        rv, tv = [p[:3].reshape(3, 1) for p in poses], [p[3:].reshape(3, 1) for p in poses]
        nd = 5 if model == "pinhole" else 4
        err, mean = cc.reproj_errors(model, objs, imgs, rv, tv, Kmat(intr[:4]), intr[4:4 + nd])
        ref = O.per_view_reproj_err_f32(model, objs, imgs, intr, poses)
        assert np.abs(err - ref).max() < 1e-6 and abs(mean - ref.mean()) < 1e-7
        err64, _ = cc.reproj_errors(model, objs, imgs, rv, tv, Kmat(intr[:4]), intr[4:4 + nd], f32_round=False)
        ref64 = O.per_view_reproj_err(model, O.Views(objs, imgs), intr, poses)
        assert np.abs(err64 - ref64).max() < 1e-11
        # the reference's own loop, call by call
        pp = cc.projectPoints if model == "pinhole" else cc.fisheye.projectPoints
        for k in (0, 17, 69):
            re, _ = pp(objs[k], rv[k], tv[k], Kmat(intr[:4]), intr[4:4 + nd])
            e = cc.norm(re, imgs[k], cc.NORM_L2) / len(re)
            assert abs(e - err[k]) < 1e-12 * max(1.0, e) + 1e-12


def case_undistort_points():
    rng = np.random.default_rng(1)
    Rm = M.rodrigues(np.array([0.05, -0.1, 0.02]))[0]
    for model, K4, D, fn, size in (("pinhole", synth.PINHOLE_TRUE_K, synth.PINHOLE_TRUE_D, cc.undistortPoints, (1920, 1080)),
                                   ("fisheye", synth.FISHEYE_TRUE_K, synth.FISHEYE_TRUE_D, cc.fisheye.undistortPoints,
                                    (2960, 1664))):
        uv = np.stack([rng.uniform(0, size[0], 200), rng.uniform(0, size[1], 200)], axis=1)
        K = Kmat(K4)
        P2 = K.copy()
        P2[0, 0] *= 0.6
        P2[1, 1] *= 0.6
        for R, Pm in ((None, None), (None, K), (None, P2), (Rm, None), (Rm, P2), (np.multiply(K, [[0.6, 1, 1], [1, 0.6, 1], [1, 1, 1]]), None)):
            ref = M.undistort_points(model, uv, K4, D, R, Pm)
            src = uv.reshape(-1, 1, 2)
            got = fn(src, K, np.array(D), R=R, P=Pm)
            assert got.shape == (200, 1, 2) and got.dtype == np.float64
            sc = max(1.0, np.abs(ref[ref > -1e5]).max())
            assert np.abs(got.reshape(-1, 2) - ref).max() < 1e-9 * sc
        # in-place float32 form used by calib_ex_* (dst is the source array)
        src32 = uv.astype(np.float32).reshape(-1, 1, 2)
        ref = M.undistort_points(model, src32.reshape(-1, 2).astype(np.float64), K4, D, None, K)
        out = fn(src32, K, np.array(D), src32, P=K)
        assert out is src32 and np.allclose(src32.reshape(-1, 2), ref, rtol=3e-7, atol=5e-4)
    # a fisheye pixel far outside the model's range does not converge -> (-1e6, -1e6)
    bad = cc.fisheye.undistortPoints(np.array([[[1e5, 1e5]]]), Kmat(synth.FISHEYE_TRUE_K),
                                     np.array([0.5, -3.0, 0.0, 0.0]))
    ref = M.undistort_fisheye(np.array([[1e5, 1e5]]), synth.FISHEYE_TRUE_K, [0.5, -3.0, 0.0, 0.0])
    assert np.allclose(bad.reshape(-1, 2), ref)


def case_solve_pnp():
    P = synth.config("C1", n_views=12, seed=9)
    objs, imgs = P.as_lists()
    K, D = Kmat(P.K_true), np.array(P.D_true)
    intr = P.intr_true()
    ok, rv, tv = cc.solvePnP_batch(objs, imgs, K, D)
    assert ok.all()
    for k in range(12):
        ref = O.find_extrinsics_pinhole(objs[k].reshape(-1, 3).astype(np.float64),
                                        imgs[k].reshape(-1, 2).astype(np.float64), intr)
        assert np.abs(rv[k] - ref[:3]).max() < 1e-6 and np.abs(tv[k] - ref[3:]).max() < 1e-4
    ret, r1, t1 = cc.solvePnP(objs[3], imgs[3], K, D)
    assert ret is True and r1.shape == (3, 1) and t1.shape == (3, 1)
    assert np.allclose(r1.ravel(), rv[3]) and np.allclose(t1.ravel(), tv[3])
    # the calib_ex flow: undistort to an ideal pinhole, then PnP with zero distortion
    flat = cc.undistortPoints(imgs[3].copy(), K, D, P=K)
    ret, r2, t2 = cc.solvePnP(objs[3], flat, K, np.zeros_like(D))
    assert np.abs(r2.ravel() - P.poses_true[3, :3].numpy()).max() < 5e-3


def case_rectify_maps():
    for model, K4, D, fn, size in (("pinhole", synth.PINHOLE_TRUE_K, synth.PINHOLE_TRUE_D, cc.initUndistortRectifyMap, (240, 135)),
                                   ("fisheye", synth.FISHEYE_TRUE_K, synth.FISHEYE_TRUE_D,
                                    cc.fisheye.initUndistortRectifyMap, (185, 104))):
        K = Kmat(K4)
        newK = K.copy()
        newK[0, 0] *= 0.5
        newK[1, 1] *= 0.5
        newK[0, 2], newK[1, 2] = size[0] / 2, size[1] / 2
        for R in (np.eye(3), M.rodrigues(np.array([0.02, 0.01, -0.03]))[0]):
            m1, m2 = fn(K, np.array(D), R, newK, size, cc.CV_32FC1)
            r1, r2 = M.rectify_map(model, K4, D, R, newK, size)
            assert m1.shape == (size[1], size[0]) and m1.dtype == np.float32
            assert np.abs(m1 - r1).max() < 2e-3 and np.abs(m2 - r2).max() < 2e-3
            f1, f2 = fn(K, np.array(D), R, newK, size, cc.CV_16SC2)
            g1, g2 = M.rectify_map(model, K4, D, R, newK, size, fixed_point=True)
            assert f1.shape == (size[1], size[0], 2) and f1.dtype == np.int16 and f2.dtype == np.uint16
            # fixed point: identical except where u*32 sits within rounding noise of a half
            full = f1.astype(np.int64) * 32 + np.stack([f2 & 31, f2 >> 5], axis=-1)
            gfull = g1.astype(np.int64) * 32 + np.stack([g2 & 31, g2 >> 5], axis=-1)
            assert np.abs(full - gfull).max() <= 1 and (full != gfull).mean() < 1e-3


def case_wrappers():
    # Camera_Cali.calib_in_fisheye / _normal against the oracle run through the same steps
    for kind, cfg, nv in (("fisheye", "C2", 24), ("pinhole", "C1", 12)):
        P = synth.config(cfg, n_views=nv)
        objs, imgs = P.as_lists()
        cal = wrappers.Camera_Cali()
        K_before = cal.init_camera_mat.copy()
        data = cal.calib_in(objs, imgs, P.image_size, "fisheye" if kind == "fisheye" else "normal")
        o_tr, o_te, i_tr, i_te = wrappers.split_points(objs, imgs, test_size=0.1, random_state=42)
        assert all(len(t) == int(len(o) * 0.1) for t, o in zip(o_te, objs))
        if kind == "fisheye":
            ref = O.fisheye_calibrate(o_tr, i_tr, P.image_size, np.array(wrappers.FISHEYE_INIT_K),
                                      np.array(wrappers.FISHEYE_INIT_D), flags=FE, criteria=(3, 1000, 1e-6))
            assert data.camera_mat is cal.init_camera_mat          # the aliasing the callers live with
            assert not np.allclose(cal.init_camera_mat, K_before)
        else:
            ref = O.calibrate_camera(o_tr, i_tr, P.image_size, np.array(wrappers.NORMAL_INIT_K),
                                     np.array(wrappers.NORMAL_INIT_D), flags=1, criteria=(3, 1000, 1e-6))
            assert data.camera_mat is not cal.init_camera_mat_normal
            assert np.array_equal(cal.init_camera_mat_normal, np.array(wrappers.NORMAL_INIT_K))
        assert abs(data.ok - ref[0]) < 1e-6 and np.allclose(data.camera_mat, ref[1], rtol=1e-6)
        intr = np.concatenate([[ref[1][0, 0], ref[1][1, 1], ref[1][0, 2], ref[1][1, 2]], ref[2].ravel()])
        poses = np.hstack([ref[3], ref[4]])
        e = O.per_view_reproj_err_f32(kind, o_te, i_te, intr, poses).mean()
        assert abs(data.reproj_err - e) < 1e-5
        assert len(data.rvecs) == nv and data.rvecs[0].shape == (3, 1)
    few = wrappers.Camera_Cali().calib_in(objs[:3], imgs[:3], P.image_size, "normal")
    assert few.camera_mat is None and few.ok is False
    # the incremental classes: init (100 iterations) then refine (10) on a grown list
    P = synth.config("C2", n_views=16)
    objs, imgs = P.as_lists()
    fe = wrappers.Fisheye(frame_width=296, frame_height=166)
    fe.board = list(objs[:10])
    fe.update(list(imgs[:10]), P.image_size)
    assert fe.inited and fe.data.type == "FISHEYE" and fe.data.ok and len(fe.data.reproj_err) == 10
    ref = O.fisheye_calibrate(objs[:10], imgs[:10], P.image_size, np.array(wrappers.FISHEYE_INIT_K),
                              np.array(wrappers.FISHEYE_INIT_D), flags=FE, criteria=(3, 100, 1e-6))
    assert np.allclose(fe.data.camera_mat, ref[1], rtol=1e-6)
    assert fe.data.map1.shape == (166, 296, 2) and fe.data.map2.dtype == np.uint16
    K1 = fe.data.camera_mat.copy()
    fe.board = list(objs)
    fe.update(list(imgs), P.image_size)
    ref2 = O.fisheye_calibrate(objs, imgs, P.image_size, K1, ref[2], flags=FE, criteria=(3, 10, 1e-6))
    assert np.allclose(fe.data.camera_mat, ref2[1], rtol=1e-6) and len(fe.data.reproj_err) == 16
    intr = np.concatenate([[ref2[1][0, 0], ref2[1][1, 1], ref2[1][0, 2], ref2[1][1, 2]], ref2[2].ravel()])
    e = O.per_view_reproj_err_f32("fisheye", objs, imgs, intr, np.hstack([ref2[3], ref2[4]]))
    assert np.abs(np.array(fe.data.reproj_err) - e).max() < 1e-5
    P = synth.config("C1", n_views=10)
    objs, imgs = P.as_lists()
    nm = wrappers.Normal(frame_width=192, frame_height=108)
    nm.board = list(objs)
    nm.update(list(imgs), P.image_size)
    assert nm.data.type == "NORMAL" and nm.data.ok and nm.data.map1.shape == (108, 192, 2)
    # calib_ex: pose of one shared board, fisheye and pinhole flows
    for kind, cfg in (("fisheye", "C2"), ("normal", "C1")):
        P = synth.config(cfg, n_views=4, noise=0.0)
        objs, imgs = P.as_lists()
        K, D = Kmat(P.K_true), np.array(P.D_true).reshape(-1, 1)
        ids = list(range(len(objs[1])))
        ret, rv, tv, pd, pdp = wrappers.Camera_Cali().calib_ex(objs[1].copy(), imgs[1].copy(), ids, K, D,
                                                             check_mode=True, camera_type=kind)
        assert ret and np.abs(rv.ravel() - P.poses_true[1, :3].numpy()).max() < 1e-3
        assert np.abs(tv.ravel() - P.poses_true[1, 3:].numpy()).max() < 0.5
        assert len(pd) == len(ids) and len(pdp) == len(ids)
    # inter_cfg.json wire format
    entry = wrappers.create_internal((2960, 1664), Kmat(synth.FISHEYE_TRUE_K), np.array(synth.FISHEYE_TRUE_D).reshape(4, 1))
    assert len(entry) == 15 and entry[:2] == [2960, 1664] and entry[4] == synth.FISHEYE_TRUE_K[2]
    size, K, D = wrappers.parse_internal(entry)
    assert size == (2960, 1664) and np.array_equal(K, Kmat(synth.FISHEYE_TRUE_K)) and D.shape == (4,)
    import json
    txt = wrappers.dumps_internal({"left_calib": entry})
    assert json.loads(txt)["left_calib"] == entry and txt.startswith('{\n    "left_calib": [\n        2960, \n')


ALL = [case_project_points, case_reproj_errors, case_undistort_points, case_solve_pnp, case_rectify_maps,
       case_wrappers]


def case_nonplanar_blocks():
    """Object points with Z != 0 (the kernels' general path; the reference's boards
    are planar so its callers never reach it): every evaluation form against the oracle."""
    import torch
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    rng = np.random.default_rng(5)
    for name, model, mid in (("C1", "pinhole", 0), ("C3", "fisheye", 1)):
        P = synth.config(name, n_views=70)
        obj = P.obj.numpy().copy()
        obj[:, 2] = rng.uniform(-30, 30, obj.shape[0]).astype(np.float32)
        vp = P.view_ptr
        objs = [obj[vp[k]:vp[k + 1]].reshape(-1, 1, 3) for k in range(P.n_views)]
        imgs = P.as_lists()[1]
        views = O.Views(objs, imgs)
        g, poses = P.intr_guess(), P.poses_true.numpy()
        E = Engine(mid, vp, obj, P.img, dev, lib=lib)
        assert E.tiles.planar == 0
        E.set_intr(0, g)
        E.set_poses(0, poses)
        blk = O.normal_blocks(model, views, g, poses)
        NI, nu = E.NI, E.NI * (E.NI + 1) // 2
        i6 = np.triu_indices(6)

        def rel(a, b):
            return np.abs(np.asarray(a) - b).max() / np.abs(b).max()

        for fused in (False, True):
            E.normal_eq(0, 0, 1e-3, store_blocks=True, fused=fused)
            G = E.glob[0].cpu().numpy()
            U = np.zeros((NI, NI))
            U[np.triu_indices(NI)] = G[:nu]
            U = U + U.T - np.diag(np.diag(U))
            assert rel(U, blk["U"]) < 1e-11 and rel(G[nu:nu + NI], blk["b"]) < 1e-10
            assert abs(G[-1] / blk["cost"] - 1) < 1e-12
            assert rel(E.slots_to_views(E.V.T), blk["V"][:, i6[0], i6[1]]) < 1e-11
            assert rel(E.slots_to_views(E.W.T).reshape(-1, NI, 6), blk["W"]) < 1e-11
            assert rel(E.slots_to_views(E.g.T), blk["g"]) < 1e-10
        # projectPoints takes any Z
        pp = cc.projectPoints if model == "pinhole" else cc.fisheye.projectPoints
        nd = 5 if model == "pinhole" else 4
        got, _ = pp(objs[3].astype(np.float64), poses[3, :3], poses[3, 3:], Kmat(g[:4]), g[4:4 + nd])
        ref = (M.project_pinhole if model == "pinhole" else M.project_fisheye)(
            objs[3].reshape(-1, 3).astype(np.float64), poses[3, :3], poses[3, 3:], g[:4], g[4:4 + nd])
        assert np.abs(got.reshape(-1, 2) - ref).max() < 1e-9
    # the solver API itself rejects non-planar targets (no planar-free pose initialiser)
    try:
        cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1)
    except cc.error:
        pass
    else:
        raise AssertionError("non-planar target accepted")


ALL.append(case_nonplanar_blocks)


def case_remap():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (104, 185, 3), dtype=np.uint8)
    K = Kmat((100.0, 101.0, 92.0, 51.0))
    D = np.array(synth.FISHEYE_TRUE_D)
    newK = K.copy()
    newK[0, 0] *= 0.5
    newK[1, 1] *= 0.5
    for size in ((185, 104), (370, 208)):                      # the reference doubles the fisheye output size
        newK[0, 2], newK[1, 2] = size[0] / 2, size[1] / 2
        for mtype in (cc.CV_32FC1, cc.CV_16SC2):
            m1, m2 = cc.fisheye.initUndistortRectifyMap(K, D, np.eye(3), newK, size, mtype)
            out = cc.remap(img, m1, m2, cc.INTER_LINEAR)
            ref = M.remap_bilinear_u8(img, m1, m2)
            assert out.shape == (size[1], size[0], 3) and out.dtype == np.uint8
            assert np.array_equal(out, ref)                    # byte work: bit-exact
            assert (ref == 0).all(axis=2).any() and (ref != 0).any()    # border and interior both exercised
    gray = img[..., 0].copy()
    m1, m2 = cc.initUndistortRectifyMap(K, np.array(synth.PINHOLE_TRUE_D), None, K, (185, 104), cc.CV_32FC1)
    assert np.array_equal(cc.remap(gray, m1, m2, cc.INTER_LINEAR, borderValue=7), M.remap_bilinear_u8(gray, m1, m2, 7))
    # identity map reproduces the image
    jj, ii = np.meshgrid(np.arange(185, dtype=np.float32), np.arange(104, dtype=np.float32))
    assert np.array_equal(cc.remap(img, jj, ii, cc.INTER_LINEAR), img)


ALL.append(case_remap)


def case_on_axis_point():
    """cv2 switches to cdist = 1 for points within 1e-8 of the optical axis; the kernels
    evaluate them by the general formula at r = 1e-8 (camcal_math.h).  The projection and
    the normal-equation blocks of a view that contains such a point match the oracle."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    K4, D = synth.FISHEYE_TRUE_K, np.array(synth.FISHEYE_TRUE_D)
    X = np.array([[0.0, 0.0, 0.0], [1e-7, 0.0, 0.0], [25.0, 0.0, 0.0], [0.0, 25.0, 0.0], [25.0, 25.0, 0.0],
                  [50.0, 25.0, 0.0]])
    pose = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 500.0])          # first point exactly on the axis
    got, _ = cc.fisheye.projectPoints(X.reshape(-1, 1, 3), pose[:3], pose[3:], Kmat(K4), D)
    ref = M.project_fisheye(X, pose[:3], pose[3:], K4, D)
    assert np.all(np.isfinite(got)) and np.abs(got.reshape(-1, 2) - ref).max() < 1e-9
    assert np.allclose(got[0, 0], [K4[2], K4[3]], atol=1e-12)
    obj = X.astype(np.float32)
    img = (ref + 0.1).astype(np.float32)
    E = Engine(_lib.FISHEYE, np.array([0, 6]), obj, img, dev, lib=lib)
    intr = np.concatenate([K4, D])
    E.set_intr(0, intr)
    E.set_poses(0, pose[None])
    E.normal_eq(0, 0, 0.0, store_blocks=True, fused=True)
    blk = O.normal_blocks("fisheye", O.Views([obj.reshape(-1, 1, 3)], [img.reshape(-1, 1, 2)]), intr, pose[None])
    G = E.glob[0].cpu().numpy()
    U = np.zeros((8, 8))
    U[np.triu_indices(8)] = G[:36]
    U = U + U.T - np.diag(np.diag(U))
    assert np.all(np.isfinite(G))
    assert np.abs(U - blk["U"]).max() < 1e-11 * np.abs(blk["U"]).max()
    assert np.abs(G[36:44] - blk["b"]).max() < 1e-10 * np.abs(blk["b"]).max()
    i6 = np.triu_indices(6)
    V = E.slots_to_views(E.V.T)[0]
    assert np.abs(V - blk["V"][0][i6]).max() < 1e-11 * np.abs(blk["V"]).max()


ALL.append(case_on_axis_point)


def case_session_growing_lists():
    """cc.CalibrationSession (the growing-list caller of intrinsicCalib.py:538-564: one more view per call, 100
    iterations first, 10-iteration refinements after) against fresh one-shot calls on the same lists: the views
    already packed stay on the device, old poses warm-start the refinement, and every call returns what a fresh
    call returns to 1e-9; lists that are not an extension of the previous ones start the session over."""
    for model, cfg, fl in (("fisheye", "C2", FE), ("pinhole", "C1", 1)):
        P = synth.config(cfg, n_views=70)                      # crosses a tile boundary and the first capacity
        objs, imgs = P.as_lists()
        ses = cc.CalibrationSession(model)
        one = cc.fisheye.calibrate if model == "fisheye" else cc.calibrateCamera
        inc = ses.calibrate if model == "fisheye" else ses.calibrateCamera
        K, D = P.K_guess.copy(), P.D_guess.copy()
        Kf, Df = P.K_guess.copy(), P.D_guess.copy()
        # production-sized lists take the one-call path (a fresh solve is cheaper than the bookkeeping): same bits
        a = inc(objs[:12], imgs[:12], P.image_size, K.copy(), D.copy(), flags=fl, criteria=(3, 100, 1e-6))
        b = one(objs[:12], imgs[:12], P.image_size, Kf.copy(), Df.copy(), flags=fl, criteria=(3, 100, 1e-6))
        assert ses._E is None and a[0] == b[0] and np.array_equal(a[1], b[1])
        ses.ONE_SHOT_BELOW = 0                                  # from here on: the incremental machinery itself
        for step, n in enumerate((6, 7, 8, 30, 64, 65, 70)):
            crit = (3, 100 if step == 0 else 10, 1e-6)
            a = inc(objs[:n], imgs[:n], P.image_size, K, D, flags=fl, criteria=crit)
            b = one(objs[:n], imgs[:n], P.image_size, Kf, Df, flags=fl, criteria=crit)
            assert a[1] is K and ses._E.n_views == n
            assert abs(a[0] - b[0]) < 1e-9, (model, n, a[0], b[0])
            assert np.allclose(a[1], b[1], rtol=1e-9) and np.allclose(a[2], b[2], rtol=1e-6, atol=1e-10)
            assert np.allclose(np.hstack(a[3]), np.hstack(b[3]), atol=1e-7)
            assert np.allclose(np.hstack(a[4]), np.hstack(b[4]), atol=1e-5)
        # another list (here: the views in another order) is not an extension: the session starts over
        E_before = ses._E
        a = inc(objs[10:20], imgs[10:20], P.image_size, K, D, flags=fl, criteria=(3, 10, 1e-6))
        b = one(objs[10:20], imgs[10:20], P.image_size, Kf, Df, flags=fl, criteria=(3, 10, 1e-6))
        assert ses._E is not E_before and ses._E.n_views == 10 and abs(a[0] - b[0]) < 1e-9
        with pytest.raises(cc.error):
            inc(objs[:3], imgs[:2], P.image_size, K, D, flags=fl, criteria=(3, 10, 1e-6))


ALL.append(case_session_growing_lists)


def case_condition_guard():
    """cv2's ComputeExtrinsicRefine leaves a view alone once cond(J) > 1e6 (fisheye.cpp,
    thresh_cond of cv::fisheye::calibrate).  Views far from the camera trip it (the
    translation along the axis is barely observable); the refinement must stop for exactly
    those views, with the pose it had and the step count of the oracle, and go on for the
    rest -- under every lane mapping (the group kernels share the guard)."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    P = synth.config("C2", n_views=96, seed=5)
    K4, D = synth.FISHEYE_TRUE_K, np.array(synth.FISHEYE_TRUE_D)
    vp = P.view_ptr.cpu().numpy() if hasattr(P.view_ptr, "cpu") else np.asarray(P.view_ptr)
    obj = P.obj.cpu().numpy().copy() if hasattr(P.obj, "cpu") else np.array(P.obj)
    img = P.img.cpu().numpy().copy() if hasattr(P.img, "cpu") else np.array(P.img)
    start = P.poses_true.cpu().numpy().copy() if hasattr(P.poses_true, "cpu") else np.array(P.poses_true)
    rng = np.random.default_rng(11)
    far = np.arange(0, 96, 7)                       # 14 views moved far away, one of them only moderately
    for k, v in enumerate(far):
        z = 3.0e4 if k == 0 else 10.0 ** rng.uniform(5.3, 6.5)
        pose = np.array([0.1, -0.15, 0.2, 30.0, -20.0, z])
        a, b = vp[v], vp[v + 1]
        uv = M.project_fisheye(obj[a:b].astype(np.float64), pose[:3], pose[3:], K4, D)
        img[a:b] = (uv + rng.normal(0.0, 0.05, uv.shape)).astype(np.float32)
        start[v] = pose + np.array([1e-3, -1e-3, 2e-3, 1.0, -1.0, 1e-3 * z])
    g = np.concatenate([K4, D])
    objs = [obj[vp[i]:vp[i + 1]].reshape(-1, 1, 3) for i in range(96)]
    imgs = [img[vp[i]:vp[i + 1]].reshape(-1, 1, 2) for i in range(96)]
    views = O.Views(objs, imgs)
    want, iters = O.refine_poses_fisheye_batched(views, start, g)
    tripped = np.array([O.extrinsic_refine_fisheye(objs[v].reshape(-1, 3).astype(np.float64),
                                                   imgs[v].reshape(-1, 2).astype(np.float64), start[v], g)[2]
                        for v in range(96)])
    assert 8 <= int(tripped.sum()) < 96 and not tripped[far[0]]      # the guard matters, and not for every far view
    for lpv in (1, 16, 64):
        E = Engine(_lib.FISHEYE, vp, obj, img, dev, lib=lib, lanes_per_view=lpv)
        E.set_intr(0, g)
        E.set_poses(0, start)
        E.pose_refine(0, 0, 20, 1e-10)
        got, it = E.get_poses(0), E.slots_to_views(E.iters)
        assert np.array_equal(it[tripped], iters[tripped])           # stopped where cv2 stops
        stopped0 = tripped & (iters == 0)
        assert stopped0.any() and np.array_equal(got[stopped0], start[stopped0])   # untouched poses
        assert np.allclose(got, want, rtol=1e-8, atol=1e-7)
        assert (it == iters).mean() > 0.95


ALL.append(case_condition_guard)


def case_extreme_raggedness():
    """Views of 4 corners (cv2's minimum) beside views of a 60 x 40 = 2400-corner board in one set: the
    tiles then differ by a factor 600 in length, a group of lanes walks a view in 38 / 150 trips or has
    60 of 64 lanes idle.  Whole solve against the oracle, every lane mapping."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    big = synth.make_problem("fisheye", 40, synth.board_points(60, 40, 6.0), (2960, 1664), synth.FISHEYE_TRUE_K,
                             synth.FISHEYE_TRUE_D, synth.FISHEYE_GUESS_K, synth.FISHEYE_GUESS_D, seed=21, noise=0.2,
                             ragged=(4, 2400))
    n = np.diff(big.view_ptr)
    n[::5] = 4                                          # every fifth view keeps its first four kept corners
    obj, img = big.obj.numpy(), big.img.numpy()
    objs, imgs = [], []
    for v in range(40):
        a = big.view_ptr[v]
        o, i = obj[a:a + n[v]], img[a:a + n[v]]
        if n[v] == 4:                                   # four corners in general position (not collinear)
            full_o, full_i = obj[a:big.view_ptr[v + 1]], img[a:big.view_ptr[v + 1]]
            sm, df = full_o[:, 0] + full_o[:, 1], full_o[:, 0] - full_o[:, 1]
            pick = np.array([sm.argmin(), sm.argmax(), df.argmin(), df.argmax()])   # the extreme corners of the board
            assert len(set(pick.tolist())) == 4
            o, i = full_o[pick], full_i[pick]
        objs.append(o.reshape(-1, 1, 3))
        imgs.append(i.reshape(-1, 1, 2))
    assert min(len(o) for o in objs) == 4 and max(len(o) for o in objs) > 1500
    crit = (3, 200, 1e-8)
    ref = O.fisheye_calibrate(objs, imgs, big.image_size, big.K_guess, big.D_guess, flags=FE, criteria=crit,
                              solver="schur")
    for lpv in (1, 16, 64):
        Engine.LANES_PER_VIEW = lpv
        try:
            info = {}
            r = cc.fisheye.calibrate(objs, imgs, big.image_size, big.K_guess.copy(), big.D_guess.copy(), flags=FE,
                                     criteria=crit, _info=info)
        finally:
            Engine.LANES_PER_VIEW = 0
        assert info["iters"] == ref[5]["iters"] and abs(r[0] - ref[0]) < 1e-9
        assert np.allclose(r[1], ref[1], rtol=1e-8) and np.allclose(r[2].ravel(), ref[2].ravel(), rtol=1e-6, atol=1e-10)
        assert np.allclose(np.hstack(r[4]).T, ref[4], rtol=1e-6, atol=1e-5)


ALL.append(case_extreme_raggedness)


def case_pose_response_and_predicted_warm_start():
    """The response stage (camcal_normal_eq_fused_response) and the move it feeds (camcal_pose_predict):
    (1) same record as the plain single-pass evaluation, Y = V^-1 W^T against the oracle's blocks;
    (2) the predicted pose is pose + Y s for the intrinsic step taken (the pose part of the joint
    Gauss-Newton step at g = 0); a view whose last refinement took no step, or ran into the cap, keeps its pose;
    (3) a whole solve with the prediction returns what the solve without it returns (the warm-start path the
    oracle restates step for step), in fewer refinement steps -- every lane mapping."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    P = synth.config("C3", n_views=150)
    objs, imgs = P.as_lists()
    views = O.Views(objs, imgs)
    g = P.intr_guess()
    for lpv in (1, 16, 64):
        E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib, lanes_per_view=lpv)
        E.set_intr(0, g)
        E.pose_init(0, 0)
        E.pose_refine(0, 0, 20, 1e-10)
        poses = E.get_poses(0)
        E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
        plain = E.rec.cpu().numpy().copy()
        E.normal_eq_response(0, 0)
        rec = E.rec.cpu().numpy().copy()
        assert np.abs(rec - plain).max() <= 1e-12 * np.abs(plain).max()
        blk = O.normal_blocks("fisheye", views, g, poses)
        want = np.linalg.solve(blk["V"], blk["W"].transpose(0, 2, 1))
        Y = E.slots_to_views(E.Y.T).reshape(-1, 8, 6).transpose(0, 2, 1)          # (N, 6, NI)
        sc = np.abs(want).max(axis=(0, 1))
        assert (np.abs(Y - want).max(axis=(0, 1)) < 1e-8 * sc).all()
        mask = np.ones(8, dtype=bool)
        a2 = 0.4
        E.solve_reduced(0, mask, 0.0, -a2, 0, 1)
        dK = E.dK[:8].cpu().numpy()
        it = E.iters.clone()
        frozen = [int(E.layout["slot_view"][3]), int(E.layout["slot_view"][70])]   # pretend: no step / at the cap
        E.iters[3] = 0
        E.iters[70] = 20
        E.pose_predict(0, -a2)
        E.iters.copy_(it)
        got = E.get_poses(0)
        d = np.einsum("bij,j->bi", want, a2 * dK)
        keep = ~np.isin(np.arange(P.n_views), frozen)
        assert np.array_equal(got[frozen], poses[frozen])
        assert np.abs(got[keep] - (poses + d)[keep]).max() < 1e-9
        del E
    res = []
    for pred in (False, True):
        for lpv in ((0,) if not pred else (0, 1, 16, 64)):
            Engine.PREDICT_POSES, Engine.LANES_PER_VIEW = pred, lpv
            try:
                info = {}
                r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE,
                                         criteria=(3, 1000, 1e-6), _info=info)
            finally:
                Engine.PREDICT_POSES, Engine.LANES_PER_VIEW = True, 0
            res.append((r, info["iters"], float(info["engine"].iters.double().sum().item())))
    (a, ia, sa) = res[0]
    for (b, ib, sb) in res[1:]:
        assert ia == ib and abs(a[0] - b[0]) < 1e-11
        assert np.allclose(a[1], b[1], rtol=1e-11) and np.allclose(a[2], b[2], rtol=1e-8, atol=1e-13)
        assert np.allclose(np.hstack(a[3]), np.hstack(b[3]), atol=1e-9)
        assert np.allclose(np.hstack(a[4]), np.hstack(b[4]), atol=1e-7)
        assert sb < sa                                               # the last re-estimation alone needs fewer steps


ALL.append(case_pose_response_and_predicted_warm_start)


def case_two_launch_iteration_of_small_sets():
    """camcal_refine_normal_eq_partial + camcal_reduce_solve (what camcal_fisheye_gn runs on small sets: two
    launches per iteration) against the five separate stages: the same record, intrinsic step, poses and step
    counts, 16 and 64 lanes per view.  Same device functions, but inlined into different kernels (different
    multiply-add contraction): equal to 1e-12, not to the bit (the shared-call build, camcal_kernels.hip
    CAMCAL_SHARED_CALL, is bit-identical).  One lane per view (large sets): the re-estimation and the evaluation
    stay launches of their own, only the second stage is shared; the pinhole model is refused."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    P = synth.config("C3", n_views=150)
    g = P.intr_guess()
    mask = np.ones(8, dtype=bool)

    def close(x, y, tol=1e-12):
        return np.abs(x - y).max() <= tol * max(np.abs(x).max(), 1e-300)
    for lpv in (16, 64, 1):            # one lane per view: the large-set form (separate launches, same second stage)
        A = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib, lanes_per_view=lpv)
        B = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib, lanes_per_view=lpv)
        for E in (A, B):
            E.set_intr(0, g)
            E.pose_init(0, 0)
            E.pose_refine(0, 0, 20, 1e-10)
        A.normal_eq_response(0, 0)                       # stages: evaluation | solve | predict + refine | evaluation
        B.refine_normal_eq_partial(0, 0, max_iter=0, first=True)
        for k in range(3):
            a2 = 1.0 - 0.6 ** (k + 1)
            A.solve_reduced(0, mask, 0.0, -a2, 0, 0)
            B.reduce_solve(mask, 0.0, -a2, 0, 0)
            assert close(A.rec.cpu().numpy()[:89], B.rec.cpu().numpy()[:89]), (lpv, k)
            assert close(A.dK.cpu().numpy()[:8], B.dK.cpu().numpy()[:8], 1e-9), (lpv, k)       # a solve amplifies by cond
            assert close(A.stats.cpu().numpy()[:5], B.stats.cpu().numpy()[:5], 1e-9), (lpv, k)
            assert close(A.get_intr(0), B.get_intr(0)), (lpv, k)
            A._call("camcal_pose_predict_refine", A.model, A._tp(), A.intr[0].data_ptr(), A.Y.data_ptr(), A.dK.data_ptr(),
                    -a2, 20, A.pose[0].data_ptr(), 20, 1e-10, A.iters.data_ptr(), A._stream())
            A.normal_eq_response(0, 0)
            B.refine_normal_eq_partial(0, 0, max_iter=20, predict=True, scale=-a2)
            pa, pb = A.get_poses(0), B.get_poses(0)
            assert np.abs(pa - pb).max() < 1e-9 * max(1.0, np.abs(pa).max()), (lpv, k)
            ia, ib = A.iters.cpu().numpy(), B.iters.cpu().numpy()
            assert (ia != ib).mean() <= 0.02 and np.abs(ia - ib).max() <= 1      # a view at the stop rule's edge may take one step more
            ya, yb = A.Y.cpu().numpy(), B.Y.cpu().numpy()
            assert np.abs(ya - yb).max() <= 1e-7 * np.abs(ya).max()
        B.reduce_solve(mask, 0.0, 0.0, 0, 0, solve=False)
        assert close(A.rec.cpu().numpy()[:89], B.rec.cpu().numpy()[:89], 1e-11)
        del A, B
    Pp = synth.config("C1R", n_views=30)
    E = Engine(_lib.PINHOLE, Pp.view_ptr, Pp.obj, Pp.img, dev, lib=lib)
    try:
        E.refine_normal_eq_partial(0, 0, max_iter=0, first=True)
        raise AssertionError("the pinhole model must be refused (its loop is CvLevMarq, not this one)")
    except _lib.CamcalStatus as exc:
        assert exc.code == _lib.E_UNSUPPORTED


ALL.append(case_two_launch_iteration_of_small_sets)


def case_folded_pinhole_lm_step_of_small_sets():
    """camcal_cross_schur_solve / camcal_schur_solve / camcal_backsub_intr_normal (what camcal_pinhole_lm runs on small
    sets: four launches per LM step) against the eight separate stage launches: Schur record, intrinsic step, trial
    poses, trial record and blocks, step norms -- equal to the order of the sums; 16 and 64 lanes per view; one lane
    per view and the fisheye model are refused (the stages run then)."""
    from cameracalibration_amd import api, _lib
    from cameracalibration_amd.engine import Engine
    lib, dev = api._resolve()
    P = synth.config("C1R", n_views=150)
    K = P.K_guess
    g0 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]] + list(np.asarray(P.D_guess).ravel()[:5]))
    mask = np.ones(9, dtype=bool)

    def close(x, y, tol=1e-11):
        return np.abs(x - y).max() <= tol * max(np.abs(x).max(), 1e-300)
    for lpv in (16, 64):
        A = Engine(_lib.PINHOLE, P.view_ptr, P.obj, P.img, dev, lib=lib, lanes_per_view=lpv)
        B = Engine(_lib.PINHOLE, P.view_ptr, P.obj, P.img, dev, lib=lib, lanes_per_view=lpv)
        for E in (A, B):
            E.set_intr(0, g0)
            E.pose_init(0, 0)
            E.pose_refine(0, 0, 20, 1e-10)
            E.intr_normal(0, 0, 0, blocks=0)
        B.small_set_begin()
        for lam, fresh in ((1e-3, True), (1e-2, False)):
            if fresh:
                A.cross_schur(0, 0, lam)
                B.cross_schur_solve(0, 0, lam, mask, -1.0, 1)
            else:
                A.schur_complement(lam)
                B.schur_solve(lam, mask, -1.0, 0, 1)
            A.solve_reduced(0, mask, lam, -1.0, 0, 1)
            assert close(A.schur.cpu().numpy(), B.schur.cpu().numpy()), (lpv, lam)
            assert close(A.dK.cpu().numpy()[:9], B.dK.cpu().numpy()[:9], 1e-8), (lpv, lam)
            assert close(A.stats.cpu().numpy()[:5], B.stats.cpu().numpy()[:5], 1e-8)
            assert close(A.get_intr(1), B.get_intr(1))
            A.backsub(lam, 0, 1)
            A.intr_normal(1, 1, 1, blocks=1)
            B.backsub_intr_normal(lam, 0, 1, 1)
            pa, pb = A.get_poses(1), B.get_poses(1)
            assert np.abs(pa - pb).max() < 1e-8 * max(1.0, np.abs(pa).max()), (lpv, lam)
            assert close(A.norms.cpu().numpy()[:2], B.norms.cpu().numpy()[:2], 1e-8)
            assert close(A.glob[1].cpu().numpy(), B.glob[1].cpu().numpy(), 1e-8)
            assert close(A.Vs[1].cpu().numpy(), B.Vs[1].cpu().numpy(), 1e-8) and close(A.gs[1].cpu().numpy(), B.gs[1].cpu().numpy(), 1e-6)
        del A, B
    for model, lpv, Pq in ((_lib.PINHOLE, 1, P), (_lib.FISHEYE, 64, synth.config("C3", n_views=30))):
        E = Engine(model, Pq.view_ptr, Pq.obj, Pq.img, dev, lib=lib, lanes_per_view=lpv)
        try:
            E.small_set_begin()
            raise AssertionError("must be refused")
        except _lib.CamcalStatus as exc:
            assert exc.code == _lib.E_UNSUPPORTED


ALL.append(case_folded_pinhole_lm_step_of_small_sets)


def case_pinhole_lm_with_rejected_steps():
    """Poor starting intrinsics make CvLevMarq reject steps (the damping sequence of the iteration log goes up as
    well as down).  The one-call path -- which runs the first half of the next step ahead of the verdict and has to
    drop it after a rejection (camcal_driver.h: PinholeLM) -- returns what the stage-by-stage loop returns."""
    P = synth.config("C1R", n_views=40, seed=3)
    objs, imgs = P.as_lists()
    for scale, dd in ((1.6, 0.8), (2.5, 1.5)):
        K = P.K_guess.copy()
        K[0, 0] *= scale
        K[1, 1] *= scale
        D = np.asarray(P.D_guess, dtype=np.float64).copy()
        D = D + dd * np.array([1, -1, 0.01, 0.01, 0.5]).reshape(D.shape)
        a = cc.calibrateCamera(objs, imgs, P.image_size, K.copy(), D.copy(), flags=1, criteria=(3, 60, 1e-9))
        info = dict(log=True)
        b = cc.calibrateCamera(objs, imgs, P.image_size, K.copy(), D.copy(), flags=1, criteria=(3, 60, 1e-9), _info=info)
        lam = [e["lam_lg10"] for e in info["log"]]
        assert any(y >= x for x, y in zip(lam, lam[1:])), lam            # at least one rejection on the way
        assert abs(a[0] - b[0]) < 1e-9 and abs(a[0] - 0.2709) < 1e-3
        assert np.allclose(a[1], b[1], rtol=1e-8, atol=0) and np.allclose(a[2], b[2], rtol=1e-6, atol=1e-10)
        assert np.allclose(np.hstack(a[3]), np.hstack(b[3]), atol=1e-8)


ALL.append(case_pinhole_lm_with_rejected_steps)


def case_randomised_production_sized_calls_vs_oracle():
    """24 random production-sized calls (4 ... 40 views, full and ragged boards, random FIX_* flag sets, iteration caps
    1 ... 1000, both models) through the plain drop-in call -- the one-call path with its folded loops -- against the
    cv2-faithful oracle (dense CvLevMarq / Schur-eliminated fisheye loop): rms to 1e-9 px, K to 1e-8, D to 1e-6
    relative and the poses for runs that may converge (caps >= 100); 1e-7 / 1e-6 / 1e-4 for runs cut short."""
    rng = np.random.default_rng(2024)
    for case in range(24):
        fisheye = case % 2 == 0
        nv = int(rng.choice([4, 6, 9, 15, 25, 40]))
        name = ("C3" if rng.integers(2) else "C2") if fisheye else ("C1R" if rng.integers(2) else "C1")
        P = synth.config(name, n_views=nv, seed=int(rng.integers(1 << 30)))
        objs, imgs = P.as_lists()
        crit = (3, int(rng.choice([1, 3, 10, 100, 1000])), 1e-6)
        if fisheye:
            fl = FE
            for bit in (16, 32, 64, 128, 512, 2048):             # FIX_K1..K4, FIX_PRINCIPAL_POINT, FIX_FOCAL_LENGTH
                if rng.integers(6) == 0:
                    fl |= bit
            r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit)
            o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=fl, criteria=crit, solver="schur")
            assert np.allclose(np.hstack(r[3]).T, o[3], atol=1e-7) and np.allclose(np.hstack(r[4]).T, o[4], atol=1e-5), (case, fl)
        else:
            fl = 1
            for bit in (4, 8, 16, 32, 64, 128):                  # FIX_PRINCIPAL_POINT, ZERO_TANGENT, FIX_FOCAL, FIX_K1..K3
                if rng.integers(6) == 0:
                    fl |= bit
            r = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit)
            o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=fl, criteria=crit, solver="dense")
        # a run stopped by a small iteration cap is not at a fixed point: what the two implementations' starting poses
        # differ by (each refined to cv2's 1e-10 rule, not to the bit) has not been contracted away yet
        loose = crit[1] < 100
        assert abs(r[0] - o[0]) < (1e-7 if loose else 1e-9), (case, name, nv, fl, crit, r[0], o[0])
        assert np.allclose(r[1], o[1], rtol=1e-6 if loose else 1e-8, atol=0), (case, name, nv, fl, crit)
        assert np.allclose(np.asarray(r[2]).ravel(), np.asarray(o[2]).ravel(), rtol=1e-4 if loose else 1e-6, atol=1e-9), (case, name, nv, fl, crit)


ALL.append(case_randomised_production_sized_calls_vs_oracle)


def case_one_call_entry_and_calibrate_many():
    """camcal_calibrate (what the plain drop-in call runs) against the stage-by-stage Engine path (chosen by
    passing _info), bit for bit; camcal_calibrate_many on the reference's four-camera loop (cali_in:
    controller_internal_calibration.py:375-399 -- two fisheye and two pinhole cameras, <= 25 ragged views each)
    against the four calls made one after another, bit for bit, K / D written in place; a degenerate view in one
    problem fails that problem and names the view."""
    FEc = FE
    crit = (3, 1000, 1e-6)
    calls, single, engine, guesses = [], [], [], []
    for k, (name, model, nv, seed) in enumerate((("C3", "fisheye", 25, 31), ("C3", "fisheye", 21, 32),
                                                 ("C1R", "pinhole", 25, 33), ("C1", "pinhole", 20, 34))):
        P = synth.config(name, n_views=nv, seed=seed)
        objs, imgs = P.as_lists()
        fn = cc.fisheye.calibrate if model == "fisheye" else cc.calibrateCamera
        fl = FEc if model == "fisheye" else cc.CALIB_USE_INTRINSIC_GUESS
        single.append(fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit))
        info = {}
        engine.append(fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit, _info=info))
        calls.append(dict(model=model, objectPoints=objs, imagePoints=imgs, imageSize=P.image_size,
                          cameraMatrix=P.K_guess.copy(), distCoeffs=P.D_guess.copy(), flags=fl, criteria=crit))
        guesses.append((P.K_guess.copy(), P.D_guess.copy()))
    many = cc.calibrate_many(calls)
    for a, b, e, c in zip(many, single, engine, calls):
        for x in (b, e):
            assert a[0] == x[0] and np.array_equal(a[1], x[1]) and np.array_equal(a[2], x[2])
            assert np.array_equal(np.hstack(a[3]), np.hstack(x[3])) and np.array_equal(np.hstack(a[4]), np.hstack(x[4]))
        assert a[1] is c["cameraMatrix"] and a[2] is c["distCoeffs"] and a[2].shape == c["distCoeffs"].shape
    # more problems than the library has streams (they share them round robin): nine at once
    nine = [dict(c, cameraMatrix=P0.copy(), distCoeffs=D0.copy())
            for c, P0, D0 in ((calls[j % 4], guesses[j % 4][0], guesses[j % 4][1]) for j in range(9))]
    for j, a in enumerate(cc.calibrate_many(nine)):
        b = single[j % 4]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(np.hstack(a[3]), np.hstack(b[3]))
    # the 4-coefficient pinhole call through the one-call path: a new 5-element array comes back
    P = synth.config("C1", n_views=12)
    objs, imgs = P.as_lists()
    r = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess[:4].copy(), flags=1, criteria=crit)
    info = {}
    e = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess[:4].copy(), flags=1, criteria=crit, _info=info)
    assert r[2].shape == (5, 1) and np.array_equal(r[2], e[2]) and r[0] == e[0]
    # a degenerate view in one problem
    o5 = calls[1]["objectPoints"][5].copy()
    o5[:, 0, 1] = 0.0
    i5 = calls[1]["imagePoints"][5].copy()
    i5[:, 0, 1] = i5[0, 0, 1]
    i5[:, 0, 0] = i5[0, 0, 0] + o5[:, 0, 0]
    calls[1]["objectPoints"][5], calls[1]["imagePoints"][5] = o5, i5
    with pytest.raises(cc.error) as ei:
        cc.calibrate_many(calls)
    assert "[5]" in str(ei.value)


ALL.append(case_one_call_entry_and_calibrate_many)
