"""Host logic of the drop-in (no GPU): layout/packing, cv2-compatible argument
handling, and the solver drivers run against the CPU lane emulator (the same
per-lane code the HIP kernels execute) and checked with the oracle."""
import numpy as np
import pytest
import torch

import cameracalibration_amd as cc
from cameracalibration_amd import _lib, api, synth
from cameracalibration_amd.engine import Engine, make_layout
from cameracalibration_amd.distributed import shard_ranges
from oracle import cvcalib as O

CPU = torch.device("cpu")
CRIT = (cc.TERM_CRITERIA_EPS + cc.TERM_CRITERIA_COUNT, 1000, 1e-6)
FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS


def test_constants_match_cv2():
    assert cc.CALIB_USE_INTRINSIC_GUESS == 1 and cc.TERM_CRITERIA_COUNT == 1 and cc.TERM_CRITERIA_EPS == 2
    f = cc.fisheye
    assert (f.CALIB_USE_INTRINSIC_GUESS, f.CALIB_RECOMPUTE_EXTRINSIC, f.CALIB_CHECK_COND, f.CALIB_FIX_SKEW) == (1, 2, 4, 8)
    assert (f.CALIB_FIX_K1, f.CALIB_FIX_K2, f.CALIB_FIX_K3, f.CALIB_FIX_K4) == (16, 32, 64, 128)
    assert f.CALIB_FIX_PRINCIPAL_POINT == 512 and f.CALIB_FIX_FOCAL_LENGTH == 2048
    assert cc.CALIB_FIX_K3 == 128 and cc.CALIB_RATIONAL_MODEL == 16384 and cc.CALIB_FIX_TANGENT_DIST == 2097152


def test_layout_ragged():
    n = np.array([13, 63, 40, 13, 57] * 30)
    lay = make_layout(n)
    assert lay["n_tiles"] == 3 and lay["n_slots"] == 192
    assert (np.sort(lay["slot_view"][:150]) == np.arange(150)).all() and (lay["slot_view"][150:] == -1).all()
    assert (np.diff(lay["lane_n"][:150]) <= 0).all()            # sorted, longest first
    assert (lay["tile_nmax"] == [63, 40, 13]).all()
    assert lay["total_floats"] == 5 * 64 * (63 + 40 + 13)


def test_pack_tiles_places_every_point(emu):
    P = synth.config("C3", n_views=100)
    E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, CPU, lib=emu)
    pts = E.t_pts.numpy()
    lay = E.layout
    obj, img = P.obj.numpy(), P.img.numpy()
    for slot in (0, 17, 63, 64, 99):
        v = lay["slot_view"][slot]
        tile, lane = divmod(slot, 64)
        for k in (0, lay["lane_n"][slot] - 1):
            base = lay["tile_off"][tile] + k * 5 * 64 + lane
            got = pts[base + 64 * np.arange(5)]
            want = np.concatenate([obj[P.view_ptr[v] + k], img[P.view_ptr[v] + k]])
            assert (got == want).all()


def test_shard_ranges_balance_and_cover():
    n = np.random.default_rng(0).integers(13, 64, 1000)
    for w in (1, 2, 4, 8):
        r = shard_ranges(n, w)
        assert r[0][0] == 0 and r[-1][1] == 1000 and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        loads = [n[a:b].sum() for a, b in r]
        assert max(loads) - min(loads) <= 2 * 63


def _emu_kw(emu):
    return dict()                          # the emu fixture routes the package to the emulator


def test_calibrate_camera_signature_and_inplace(emu):
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    K, D = P.K_guess.copy(), P.D_guess.copy()
    out = cc.calibrateCamera(objs, imgs, P.image_size, K, D, flags=cc.CALIB_USE_INTRINSIC_GUESS, criteria=CRIT,
                             **_emu_kw(emu))
    rms, Ko, Do, rv, tv = out
    assert isinstance(rms, float) and Ko is K and Do is D          # written in place and returned
    assert Do.shape == (5, 1) and Ko.shape == (3, 3) and Ko[0, 1] == 0 and Ko[2, 2] == 1
    assert isinstance(rv, tuple) and len(rv) == 20 and rv[0].shape == (3, 1) and tv[0].dtype == np.float64
    assert not np.allclose(K, P.K_guess)
    o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=CRIT, solver="dense")
    assert abs(rms - o[0]) < 1e-9
    assert np.allclose(K, o[1], rtol=1e-8) and np.allclose(D, o[2], rtol=1e-6, atol=1e-10)


def test_fisheye_matches_cv2_faithful_oracle(emu):
    P = synth.config("C2", n_views=30)
    objs, imgs = P.as_lists()
    K, D = P.K_guess.copy(), P.D_guess.copy()
    info = {}
    rms, Ko, Do, rv, tv = cc.fisheye.calibrate(objs, imgs, P.image_size, K, D, flags=FE, criteria=CRIT,
                                               _info=info, **_emu_kw(emu))
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=CRIT, solver="dense")
    assert info["iters"] == o[5]["iters"]
    assert Do.shape == (4, 1) and Ko is K
    assert abs(rms - o[0]) < 1e-9
    assert np.allclose(Ko, o[1], rtol=1e-9) and np.allclose(Do, o[2], rtol=1e-6, atol=1e-11)
    assert np.allclose(np.hstack(rv).T, o[3], atol=1e-8) and np.allclose(np.hstack(tv).T, o[4], atol=1e-6)


def test_ragged_views_fisheye(emu):
    P = synth.config("C3", n_views=150)
    objs, imgs = P.as_lists()
    rms, K, D, rv, tv = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(),
                                             flags=FE, criteria=CRIT, **_emu_kw(emu))
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=CRIT, solver="schur")
    assert abs(rms - o[0]) < 1e-9 and np.allclose(K, o[1], rtol=1e-9) and np.allclose(D, o[2], rtol=1e-6, atol=1e-11)
    assert np.allclose(np.hstack(rv).T, o[3], atol=1e-8)


def test_refine_call_with_10_iterations(emu):
    """intrinsicCalib.py:131-134 / :223-226: warm-started 10-iteration calls."""
    P = synth.config("C2", n_views=20)
    objs, imgs = P.as_lists()
    crit = (3, 10, 1e-6)
    r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=crit,
                             **_emu_kw(emu))
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=crit, solver="schur")
    assert np.allclose(r[1], o[1], rtol=1e-10)


def test_fix_flags(emu):
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    fl = cc.CALIB_USE_INTRINSIC_GUESS | cc.CALIB_FIX_PRINCIPAL_POINT | cc.CALIB_FIX_K3
    K, D = P.K_guess.copy(), P.D_guess.copy()
    rms, Ko, Do, _, _ = cc.calibrateCamera(objs, imgs, P.image_size, K, D, flags=fl, criteria=CRIT, **_emu_kw(emu))
    assert Ko[0, 2] == P.K_guess[0, 2] and Ko[1, 2] == P.K_guess[1, 2] and Do[4, 0] == P.D_guess[4, 0]
    o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=fl, criteria=CRIT, solver="dense")
    assert np.allclose(Ko, o[1], rtol=1e-8) and abs(rms - o[0]) < 1e-9
    P = synth.config("C2", n_views=20)
    objs, imgs = P.as_lists()
    fl = FE | cc.fisheye.CALIB_FIX_K4 | cc.fisheye.CALIB_FIX_FOCAL_LENGTH
    rms, Ko, Do, _, _ = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl,
                                             criteria=CRIT, **_emu_kw(emu))
    assert Do[3, 0] == 0.0 and Ko[0, 0] == P.K_guess[0, 0]
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=fl, criteria=CRIT, solver="schur")
    assert np.allclose(Ko, o[1], rtol=1e-9) and np.allclose(Do, o[2], atol=1e-11)


def test_library_sequenced_loops_equal_the_python_loops(emu):
    """Single rank: the solver loops run inside the library (camcal_fisheye_gn / camcal_pinhole_lm);
    asking for the iteration log (or more than one rank) keeps them in solver.py.  Same stage
    calls in the same order, so the results are identical."""
    for cfg, nv in (("C1", 20), ("C2", 40), ("C3", 90)):
        P = synth.config(cfg, n_views=nv)
        objs, imgs = P.as_lists()
        res = []
        for info in ({}, {"log": True}):
            if P.model == "pinhole":
                r = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                                       criteria=CRIT, _info=info, **_emu_kw(emu))
            else:
                r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE,
                                         criteria=CRIT, _info=info, **_emu_kw(emu))
            res.append((r, info))
        (a, ia), (b, ib) = res
        assert ia["log"] is None and len(ib["log"]) == ib["iters"] == ia["iters"]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(np.hstack(a[3]), np.hstack(b[3])) and np.array_equal(np.hstack(a[4]), np.hstack(b[4]))


def test_view_lists_in_any_array_form():
    """Lists of ndarrays (the reference: float32 (n,1,3)/(n,1,2), camera_calib.py:70-75) take one
    concatenation; anything else (nested lists, float64, (n,3)) is converted per view -- same result."""
    from cameracalibration_amd.api import _gather_views
    P = synth.config("C3", n_views=40)
    objs, imgs = P.as_lists()
    vp, o, i = _gather_views(objs, imgs)
    assert o.dtype == np.float32 and o.shape == (vp[-1], 3) and i.shape == (vp[-1], 2)
    assert np.array_equal(np.diff(vp), [len(x) for x in objs])
    forms = [([x.reshape(-1, 3).astype(np.float64) for x in objs], [x.reshape(-1, 2) for x in imgs]),
             ([x.tolist() for x in objs], [x.tolist() for x in imgs]),
             (tuple(objs), tuple(imgs))]
    for fo, fi in forms:
        vp2, o2, i2 = _gather_views(fo, fi)
        assert np.array_equal(vp, vp2) and np.array_equal(o, o2) and np.array_equal(i, i2)
    # every view of one size, stacked into one array each: no per-view work at all
    Q = synth.config("C2", n_views=12)
    qo, qi = Q.as_lists()
    ref = _gather_views(qo, qi)
    for so, si in ((np.stack(qo), np.stack(qi)), (np.stack(qo).reshape(12, -1, 3).astype(np.float64), np.stack(qi).reshape(12, -1, 2))):
        got = _gather_views(so, si)
        assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(ref, got))
    with pytest.raises(cc.error):
        _gather_views(np.stack(qo)[:, :3], np.stack(qi)[:, :3])
    with pytest.raises(cc.error):
        _gather_views(np.stack(qo), np.stack(qi)[:, :-1])
    for fo, fi in ((objs, imgs), ([x.tolist() for x in objs], [x.tolist() for x in imgs])):
        short = list(fi)
        short[3] = short[3][:-1]
        with pytest.raises(cc.error):
            _gather_views(fo, short)
        with pytest.raises(cc.error):
            _gather_views([x[:3] for x in fo], [x[:3] for x in fi])      # fewer than 4 points per view


def test_stacked_views_through_the_drop_in_calls(emu):
    """(N,n,1,3) / (N,n,1,2) arrays instead of lists: the same calibration, bit for bit."""
    for name, call, flags in (("C1", cc.calibrateCamera, cc.CALIB_USE_INTRINSIC_GUESS), ("C2", cc.fisheye.calibrate, FE)):
        P = synth.config(name, n_views=16)
        objs, imgs = P.as_lists()
        a = call(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=flags, criteria=CRIT)
        b = call(np.stack(objs), np.stack(imgs), P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=flags, criteria=CRIT)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert len(b[3]) == 16 and np.array_equal(np.hstack(a[3]), np.hstack(b[3])) and np.array_equal(np.hstack(a[4]), np.hstack(b[4]))


def test_errors_like_cv2(emu):
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    kw = dict(flags=1, criteria=CRIT, **_emu_kw(emu))
    with pytest.raises(cc.error):
        cc.calibrateCamera(objs, imgs[:-1], P.image_size, P.K_guess.copy(), P.D_guess.copy(), **kw)
    with pytest.raises(cc.error):
        cc.calibrateCamera([], [], P.image_size, P.K_guess.copy(), P.D_guess.copy(), **kw)
    bad = [o.copy() for o in objs]
    bad[0] = bad[0][:3]
    with pytest.raises(cc.error):
        cc.calibrateCamera(bad, [imgs[0][:3]] + imgs[1:], P.image_size, P.K_guess.copy(), P.D_guess.copy(), **kw)
    K = P.K_guess.copy(); K[0, 0] = -1
    with pytest.raises(cc.error):
        cc.calibrateCamera(objs, imgs, P.image_size, K, P.D_guess.copy(), **kw)
    K = P.K_guess.copy(); K[0, 2] = 5000
    with pytest.raises(cc.error):
        cc.calibrateCamera(objs, imgs, P.image_size, K, P.D_guess.copy(), **kw)
    K = P.K_guess.copy(); K[0, 1] = 0.5
    with pytest.raises(cc.error):
        cc.calibrateCamera(objs, imgs, P.image_size, K, P.D_guess.copy(), **kw)
    with pytest.raises(cc.error):
        cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=0, criteria=CRIT,
                           **_emu_kw(emu))
    with pytest.raises(cc.error):
        cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), np.zeros((5, 1)), flags=FE, criteria=CRIT,
                             **_emu_kw(emu))
    with pytest.raises(cc.error):
        cc.fisheye.calibrate(objs, imgs, P.image_size, None, None, flags=FE, criteria=CRIT, **_emu_kw(emu))


def test_no_cpu_fallback_in_product():
    """Without a ROCm device the drop-in refuses to run instead of falling back."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    with pytest.raises(_lib.CamcalError):
        cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=CRIT)
    with pytest.raises(_lib.CamcalError):
        Engine(0, P.view_ptr, P.obj, P.img, CPU)            # HIP library + cpu tensor
    with pytest.raises(_lib.CamcalError):
        _lib.CamcalLib("/nonexistent/libcamcal.so")
    # and the package text holds no route to the emulator or the oracle (the tests replace api._resolve from outside)
    import glob
    import os
    pkg = os.path.dirname(os.path.abspath(cc.__file__))
    for f in glob.glob(os.path.join(pkg, "*.py")):
        text = open(f).read()
        assert "hostemu" not in text and "import oracle" not in text and "from oracle" not in text, f
        assert "_set_test_backend" not in text and "_test_backend" not in text, f


def test_pose_init_lands_in_basin(emu):
    """Homography initialisation + refinement reaches the oracle's per-view optimum."""
    for name, mid, model in (("C1", 0, "pinhole"), ("C2", 1, "fisheye")):
        P = synth.config(name, n_views=24)
        objs, imgs = P.as_lists()
        views = O.Views(objs, imgs)
        E = Engine(mid, P.view_ptr, P.obj, P.img, CPU, lib=emu)
        g = P.intr_guess()
        E.set_intr(0, g)
        E.pose_init(0, 0)
        assert int(E.status.abs().sum()) == 0
        p0 = E.get_poses(0)
        if model == "fisheye":
            want0 = np.stack([O.init_extrinsics_fisheye(o, i, g) for o, i in zip(views.obj, views.img)])
            assert np.allclose(p0, want0, rtol=1e-6, atol=1e-6)
            E.pose_refine(0, 0, 20, 1e-10)
            want, _ = O.refine_poses_fisheye_batched(views, want0, g)
            assert np.allclose(E.get_poses(0), want, rtol=1e-9, atol=1e-8)
        else:
            E.pose_refine(0, 0, 20, 1e-10)
            want = np.stack([O.find_extrinsics_pinhole(o, i, g, max_iter=200) for o, i in zip(views.obj, views.img)])
            assert np.allclose(E.get_poses(0), want, rtol=1e-5, atol=1e-4)


def test_four_coefficient_guess_is_padded_like_cv2(emu):
    """cv2.calibrateCamera pads a 4-element distCoeffs to its 5-coefficient model, estimates k3 unless
    CALIB_FIX_K3 and returns a NEW 5-element array in the input's orientation."""
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    for shape, want in (((4, 1), (5, 1)), ((1, 4), (1, 5)), ((4,), (5, 1))):
        D4 = P.D_guess.ravel()[:4].reshape(shape).copy()
        keep = D4.copy()
        r = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), D4, flags=1, criteria=CRIT)
        assert r[2].shape == want and r[2] is not D4 and np.array_equal(D4, keep)
        assert abs(r[2].ravel()[4]) > 1e-3                        # k3 was estimated
    o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess.ravel()[:4].reshape(4, 1), flags=1,
                           criteria=CRIT, solver="dense")
    assert o[2].shape == (5, 1) and abs(r[0] - o[0]) < 1e-9 and np.allclose(r[2].ravel(), o[2].ravel(), rtol=1e-6, atol=1e-9)
    rf = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.ravel()[:4].copy(),
                            flags=1 | cc.CALIB_FIX_K3, criteria=CRIT)
    assert rf[2].shape == (5, 1) and rf[2][4, 0] == 0.0


def test_degenerate_view_raises_and_names_it(emu):
    """One view with collinear corners: the pose start has no answer (cv2 returns garbage / non-finite values
    for the caller's checkRange); the drop-in raises `error` naming the view."""
    for cfg, fn, fl in (("C2", cc.fisheye.calibrate, FE), ("C1", cc.calibrateCamera, 1)):
        P = synth.config(cfg, n_views=12)
        objs, imgs = P.as_lists()
        o5 = objs[5].copy()
        o5[:, 0, 1] = 0.0                                         # every corner on one board line ...
        i5 = imgs[5].copy()
        i5[:, 0, 1] = i5[0, 0, 1]                                 # ... and on one image line
        i5[:, 0, 0] = i5[0, 0, 0] + o5[:, 0, 0]
        objs[5], imgs[5] = o5, i5
        with pytest.raises(cc.error) as ei:
            fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=CRIT)
        assert "5" in str(ei.value)


def cap_hitting_case(fn_kwargs):
    """A set in which some views run ComputeExtrinsicRefine to its 20-step cap from the homography start
    (noisy, ragged, strongly tilted boards).  cv2 restarts every pose from InitExtrinsics at each outer
    iteration and returns whatever the capped refinement reached; this build warm-starts the same refinement
    from the previous pose.  Checked against the oracle's cv2-faithful re-initialising path:
    intrinsics and rms to round-off, poses as ROTATIONS (a view on the |r| = pi boundary may come back in the
    other representation of the same rotation) -- and the set really contains views at the cap."""
    from cameracalibration_amd.synth import _rot_from_rvec
    P = synth.config("C3", n_views=250, seed=3, noise=0.5)
    objs, imgs = P.as_lists()
    info = {}
    r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=CRIT,
                             _info=info, **fn_kwargs)
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=CRIT, solver="schur",
                            reinit_poses=True)
    E = info["engine"]
    E.set_intr(0, P.intr_guess())
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    assert int((E.slots_to_views(E.iters) >= 20).sum()) >= 2          # views at the cap exist
    assert info["iters"] == o[5]["iters"] and abs(r[0] - o[0]) < 1e-10
    assert np.allclose(r[1], o[1], rtol=1e-10) and np.allclose(r[2].ravel(), o[2].ravel(), rtol=1e-8, atol=1e-12)
    rv, tv = np.hstack(r[3]).T, np.hstack(r[4]).T
    Ra = _rot_from_rvec(torch.from_numpy(rv)).numpy()
    Rb = _rot_from_rvec(torch.from_numpy(o[3])).numpy()
    assert np.abs(Ra - Rb).max() < 1e-8 and np.abs(tv - o[4]).max() < 1e-6
    assert (np.abs(rv - o[3]).max(axis=1) < 1e-8).mean() > 0.98      # and as vectors, bar the boundary


def test_cap_hitting_views_match_the_reinitialising_oracle(emu):
    cap_hitting_case(dict())


def test_static_entry_rejection_codes_match_the_hip_build(emu):
    """camcal_normal_eq_fused_static: the limit counts partial records (tiles x lanes per view of the mapping
    the set gets), and tiles / model are validated first -- the emulator shares the rule with the HIP build
    (camcal_lane.h: static_partials), so the CPU tier pins the entry point's contract."""
    import ctypes as C
    from cameracalibration_amd import _lib
    libs = [emu, _lib.load()]            # argument checks run before any device work: the HIP build answers without a GPU
    buf = (C.c_double * 8)()
    ptr = C.addressof(buf)

    def tiles(nt, lpv, planar=1, slots=None):
        return _lib.Tiles(ptr, ptr, ptr, ptr, nt, 64 * nt if slots is None else slots, planar, lpv)

    for lib in libs:
        call = lambda model, t: lib.camcal_normal_eq_fused_static(model, C.byref(t), ptr, ptr, 0.0, ptr, ptr, None)
        assert call(7, tiles(4, 0)) == -2                # bad model
        assert call(1, tiles(4, 5)) == -1                # bad lanes_per_view
        assert call(1, tiles(4, 0, slots=100)) == -1     # n_slots != 64 n_tiles
        assert call(1, tiles(16385, 1)) == -3            # one lane per view: 16385 partial records
        assert call(1, tiles(16385, 0, planar=0)) == -3
        assert call(1, tiles(20000, 64)) == -3           # too many tiles for 64 lanes per view: one lane per view, 20000 records
    # (a grouped set never exceeds the limit: 32 tiles x 64 lanes = 2048, 448 tiles x 16 lanes = 7168 partial records)


_PLANAR_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import cameracalibration_amd as cc
from cameracalibration_amd import synth, _lib
from tests.conftest import use_emulator, EMU_PATH
use_emulator(_lib.load(EMU_PATH))
out = []
for name, fn, fl in (("C2", cc.fisheye.calibrate, 8 | 2 | 1), ("C1", cc.calibrateCamera, 1)):
    P = synth.config(name, n_views=12)
    o, i = P.as_lists()
    r = fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 30, 1e-6))
    out.append(np.concatenate([[r[0]], r[1].ravel(), np.asarray(r[2]).ravel(), np.hstack(r[3]).ravel()]))
    off = [x.copy() for x in o]
    off[3][5, 0, 2] = 1.0
    try:
        fn(off, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 30, 1e-6))
        raise SystemExit("a non-planar target was accepted")
    except cc.error as e:
        assert "planar" in str(e).lower(), str(e)
np.save(sys.argv[1], np.concatenate(out))
"""


def test_planarity_verdict_from_the_device_side_test(tmp_path):
    """camcal_calibrate with the target's planarity tested behind the upload (the path of sets of a million
    points and more, forced here by CAMCAL_PLANAR_ON_DEVICE_MIN=0) on the emulator: same results as with the host
    test, and a board with a corner off its plane is refused just the same."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("host", {}), ("device", {"CAMCAL_PLANAR_ON_DEVICE_MIN": "0"}),
                     ("early", {"CAMCAL_PIN_LIMIT": "0", "CAMCAL_EARLY_UPLOAD_MIN": "0"})):      # the large calls' upload thread
        f = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", _PLANAR_SCRIPT % ROOT, f], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    assert np.array_equal(res["host"], res["device"]) and np.array_equal(res["host"], res["early"])
