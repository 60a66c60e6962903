"""Parity tests proper: the HIP library (through the C ABI / the drop-in API)
against the oracle on the same seeded inputs, against the committed golden
vectors, and -- at BASELINE.json's full sizes -- through size-independent
properties.  Bar: fp64 work, so agreement to round-off of the differently
ordered sums: blocks 1e-11 relative; solver outputs rms 1e-9 px, K 1e-8
relative, D 1e-6 relative (north star: 1e-6 px / 1e-6 relative)."""
import glob
import os

import numpy as np
import pytest
import torch

import cameracalibration_amd as cc
from cameracalibration_amd import _lib, synth
from cameracalibration_amd.engine import Engine
from oracle import cvcalib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
CRIT = (3, 1000, 1e-6)
FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tier needs a ROCm device"
    lib = _lib.load()
    assert not lib.is_emulator and lib.camcal_version() > 0      # the HIP build, not the emulator
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 16, 64], ids=["lane-per-view", "16-lanes-per-view", "64-lanes-per-view"])
def mapping(request, dev, monkeypatch):
    """Run a small-set test under each mapping of views to lanes (include/camcal.h:
    camcal_tiles.lanes_per_view; 0 = chosen by size is the default)."""
    monkeypatch.setattr(Engine, "LANES_PER_VIEW", request.param)
    yield request.param


def sym(tri, n):
    U = np.zeros((n, n))
    U[np.triu_indices(n)] = tri
    return U + U.T - np.diag(np.diag(U))


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("name,model,mid,nv", [("C1", "pinhole", 0, 20), ("C2", "fisheye", 1, 500),
                                               ("C3", "fisheye", 1, 700), ("C3", "fisheye", 1, 1),
                                               ("C1R", "pinhole", 0, 700)])
def test_normal_equation_stages_match_oracle(dev, name, model, mid, nv, mapping):
    P = synth.config(name, n_views=nv)
    objs, imgs = P.as_lists()
    views = O.Views(objs, imgs)
    g, poses = P.intr_guess(), P.poses_true.numpy()
    E = Engine(mid, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, g)
    E.set_poses(0, poses)
    lam = 1e-3
    E.normal_eq(0, 0, lam)
    blk = O.normal_blocks(model, views, g, poses)
    NI, nu = E.NI, E.NI * (E.NI + 1) // 2
    G = E.glob[0].cpu().numpy()
    assert rel(sym(G[:nu], NI), blk["U"]) < 1e-11 and rel(G[nu:nu + NI], blk["b"]) < 1e-10
    assert abs(G[-1] / blk["cost"] - 1) < 1e-12
    i6 = np.triu_indices(6)
    assert rel(E.slots_to_views(E.V.T), blk["V"][:, i6[0], i6[1]]) < 1e-11
    assert rel(E.slots_to_views(E.W.T).reshape(-1, NI, 6), blk["W"]) < 1e-11
    assert rel(E.slots_to_views(E.g.T), blk["g"]) < 1e-10
    # Schur complement, reduced solve and back-substitution against the oracle's elimination
    mask = np.ones(NI, dtype=bool)
    dK, di = O._schur_step(blk, lam, mask, damp=True)
    S = E.schur.cpu().numpy()
    Vd = blk["V"].copy()
    d6 = np.arange(6)
    Vd[:, d6, d6] *= 1 + lam
    Y = np.linalg.solve(Vd, np.concatenate([blk["W"].transpose(0, 2, 1), blk["g"][..., None]], axis=2))
    C = np.einsum("bik,bkj->ij", blk["W"], Y[..., :-1])
    d = np.einsum("bik,bk->i", blk["W"], Y[..., -1])
    assert rel(sym(S[:nu], NI), C) < 1e-10 and rel(S[nu:], d) < 1e-10
    if nv >= 20:
        E.solve_reduced(0, mask, lam, -1.0, 0, 1)
        E.backsub(lam, 0, 1)
        # the reduced system is ill-conditioned in k3/k4; compare the step in the metric of the problem
        got = E.dK[:NI].cpu().numpy()
        sc = np.sqrt(np.diag(blk["U"]))
        assert np.abs((got - dK) * sc).max() < 1e-6 * np.abs(dK * sc).max()
        assert np.allclose(E.get_poses(1), poses - di, rtol=1e-7, atol=1e-7)
        nr = E.norms.cpu().numpy()
        assert abs(nr[1] / (poses ** 2).sum() - 1) < 1e-12


@pytest.mark.parametrize("name,nv", [("C2", 500), ("C3", 700)])
def test_benchmarked_single_pass_record_matches_oracle(dev, name, nv, mapping):
    """The instantiation bench.py times -- camcal_normal_eq_fused without block stores at
    lambda = 0 (k_normal_fused<fisheye, no stores, planar>; cv2's ComputeJacobians + the
    elimination inside solve(JJ2, ex3), camera_calib.py:74-77) -- record by record against
    the oracle's blocks and their undamped Schur complement."""
    P = synth.config(name, n_views=nv)
    objs, imgs = P.as_lists()
    views = O.Views(objs, imgs)
    g, poses = P.intr_guess(), P.poses_true.numpy()
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, g)
    E.set_poses(0, poses)
    E.W.fill_(float("nan"))                      # the no-store form must not touch the block buffers
    E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
    assert bool(torch.isnan(E.W).all().item())
    blk = O.normal_blocks("fisheye", views, g, poses)
    NI, nu = 8, 36
    R = E.rec.cpu().numpy()
    G, S = R[:E.glob_len], R[E.glob_len:]
    assert rel(sym(G[:nu], NI), blk["U"]) < 1e-11 and rel(G[nu:nu + NI], blk["b"]) < 1e-10
    assert abs(G[-1] / blk["cost"] - 1) < 1e-12
    Y = np.linalg.solve(blk["V"], np.concatenate([blk["W"].transpose(0, 2, 1), blk["g"][..., None]], axis=2))
    C = np.einsum("bik,bkj->ij", blk["W"], Y[..., :-1])
    d = np.einsum("bik,bk->i", blk["W"], Y[..., -1])
    assert rel(sym(S[:nu], NI), C) < 1e-10 and rel(S[nu:], d) < 1e-10


def test_cost_only_and_per_view_cost(dev, mapping):
    P = synth.config("C3", n_views=300)
    objs, imgs = P.as_lists()
    views = O.Views(objs, imgs)
    g, poses = P.intr_guess(), P.poses_true.numpy()
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, g)
    E.set_poses(0, poses)
    E.intr_normal(0, 0, 1, with_normal=False, view_cost=True)
    assert abs(E.glob[1][-1].item() / O.reproj_cost("fisheye", views, g, poses) - 1) < 1e-12
    vc = E.slots_to_views(E.view_cost)
    want = O.per_view_reproj_err("fisheye", views, g, poses)          # ||e||/n
    assert np.allclose(np.sqrt(vc) / views.n, want, rtol=1e-11)


def test_pose_init_and_refine_match_oracle(dev, mapping):
    P = synth.config("C2", n_views=200)
    objs, imgs = P.as_lists()
    views = O.Views(objs, imgs)
    g = P.intr_guess()
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, g)
    E.pose_init(0, 0)
    assert int(E.status.abs().sum().item()) == 0
    want0 = np.stack([O.init_extrinsics_fisheye(o, i, g) for o, i in zip(views.obj[:40], views.img[:40])])
    assert np.allclose(E.get_poses(0)[:40], want0, rtol=1e-6, atol=1e-6)
    p0 = E.get_poses(0)
    E.pose_refine(0, 0, 20, 1e-10)
    want, iters = O.refine_poses_fisheye_batched(views, p0, g)
    assert np.allclose(E.get_poses(0), want, rtol=1e-9, atol=1e-8)
    assert (E.slots_to_views(E.iters) == iters).mean() > 0.95           # same stopping rule


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_golden_vectors_through_drop_in_api(dev, path, mapping):
    from tests.test_golden import check, load
    z, objs, imgs, crit = load(path)
    size = tuple(int(v) for v in z["image_size"])
    info = {}
    if str(z["model"]) == "pinhole":
        r = cc.calibrateCamera(objs, imgs, size, z["K_guess"].copy(), z["D_guess"].copy(), flags=int(z["flags"]),
                               criteria=crit, _info=info)
    else:
        r = cc.fisheye.calibrate(objs, imgs, size, z["K_guess"].copy(), z["D_guess"].copy(), flags=int(z["flags"]),
                                 criteria=crit, _info=info)
    check(z, r, info["iters"])


def test_config1_pinhole_vs_oracle(dev, mapping):
    P = synth.config("C1")
    objs, imgs = P.as_lists()
    K, D = P.K_guess.copy(), P.D_guess.copy()
    r = cc.calibrateCamera(objs, imgs, P.image_size, K, D, flags=cc.CALIB_USE_INTRINSIC_GUESS, criteria=CRIT)
    o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=CRIT, solver="dense")
    assert r[1] is K and abs(r[0] - o[0]) < 1e-9
    assert np.allclose(K, o[1], rtol=1e-8) and np.allclose(D, o[2], rtol=1e-6, atol=1e-10)


def test_config2_fisheye_500_views_vs_oracle(dev, mapping):
    P = synth.config("C2")
    objs, imgs = P.as_lists()
    info = {}
    r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=CRIT,
                             _info=info)
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=CRIT, solver="schur")
    assert info["iters"] == o[5]["iters"] and abs(r[0] - o[0]) < 1e-9
    assert np.allclose(r[1], o[1], rtol=1e-9) and np.allclose(r[2], o[2], rtol=1e-6, atol=1e-11)
    assert np.allclose(np.hstack(r[3]).T, o[3], atol=1e-8) and np.allclose(np.hstack(r[4]).T, o[4], atol=1e-6)
    # and the answer is the truth to within the noise
    assert np.allclose(r[1][[0, 1, 0, 1], [0, 1, 2, 2]], P.K_true, rtol=2e-3)


def test_config3_ragged_subsample_vs_oracle(dev, mapping):
    P = synth.config("C3", n_views=2000)
    objs, imgs = P.as_lists()
    r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=CRIT)
    o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=CRIT, solver="schur")
    assert abs(r[0] - o[0]) < 1e-9 and np.allclose(r[1], o[1], rtol=1e-9) and np.allclose(r[2], o[2], rtol=1e-6, atol=1e-11)


_PINHOLE_RAGGED = {}


def _pinhole_ragged_oracle():
    """2000 ragged pinhole views through the oracle's Schur-eliminated CvLevMarq, once per session."""
    if not _PINHOLE_RAGGED:
        P = synth.config("C1R", n_views=2000)
        objs, imgs = P.as_lists()
        o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=CRIT, solver="schur")
        _PINHOLE_RAGGED.update(P=P, objs=objs, imgs=imgs, o=o)
    return _PINHOLE_RAGGED


def test_config1_ragged_pinhole_2000_views_vs_oracle(dev, mapping):
    """The pinhole twin of the C3 sub-sample test: cv2.calibrateCamera as camera_calib.py:153-156 calls it,
    on 2000 ragged views (32 tiles, every lane mapping) against the oracle's CvLevMarq on the eliminated
    system -- same accepted steps, same answer."""
    c = _pinhole_ragged_oracle()
    P, o = c["P"], c["o"]
    info = {}
    K, D = P.K_guess.copy(), P.D_guess.copy()
    r = cc.calibrateCamera(c["objs"], c["imgs"], P.image_size, K, D, flags=cc.CALIB_USE_INTRINSIC_GUESS, criteria=CRIT,
                           _info=info)
    assert info["iters"] == o[5]["iters"] and abs(r[0] - o[0]) < 1e-9
    assert np.allclose(K, o[1], rtol=1e-8) and np.allclose(D, o[2], rtol=1e-6, atol=1e-10)
    assert np.allclose(np.hstack(r[3]).T, o[3], atol=1e-7) and np.allclose(np.hstack(r[4]).T, o[4], atol=1e-5)
    assert np.allclose(K[[0, 1, 0, 1], [0, 1, 2, 2]], P.K_true, rtol=2e-3)


def test_library_sequenced_loops_equal_the_python_loops_on_gpu(dev):
    """camcal_fisheye_gn / camcal_pinhole_lm (single rank, default) against the same loops
    sequenced by solver.py (chosen by asking for the iteration log): identical stage calls."""
    for cfg, nv in (("C1", 20), ("C2", 200)):
        P = synth.config(cfg, n_views=nv)
        objs, imgs = P.as_lists()
        res = []
        for info in ({}, {"log": True}):
            fn, fl = (cc.calibrateCamera, 1) if P.model == "pinhole" else (cc.fisheye.calibrate, FE)
            r = fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=CRIT, _info=info)
            res.append((r, info))
        (a, ia), (b, ib) = res
        assert ia["log"] is None and len(ib["log"]) == ib["iters"] == ia["iters"]
        if P.model == "pinhole":          # the same stage launches in both: the same bits
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(np.hstack(a[3]), np.hstack(b[3]))
        else:
            # small fisheye sets: the library folds an iteration's five launches into two (same device functions
            # inlined into other kernels: other multiply-add contraction), the Python loop keeps the five
            assert abs(a[0] - b[0]) <= 1e-12 * b[0] and np.allclose(a[1], b[1], rtol=1e-11, atol=0)
            assert np.allclose(a[2], b[2], rtol=1e-8, atol=1e-14)
            assert np.allclose(np.hstack(a[3]), np.hstack(b[3]), rtol=0, atol=1e-9)


@pytest.mark.parametrize("nv", [1, 63, 65, 2048, 2049, 28672, 28673])
@pytest.mark.parametrize("mid", [1, 0], ids=["fisheye", "pinhole"])
def test_automatic_mapping_at_its_size_boundaries(dev, nv, mid):
    """The mapping chosen by size (64 lanes per view up to 32 tiles, 16 up to 448, one above)
    against one lane per view, on ragged sets either side of every boundary, both models: the
    single-pass record, the stored blocks, a pose re-estimation and -- the stages the pinhole
    LM loop is made of (camera_calib.py:153-156) -- evaluation A + B, the stored-block Schur
    complement and the back-substitution."""
    P = synth.config("C3" if mid == 1 else "C1R", n_views=nv)
    out = []
    for mode in (1, 0):
        E = Engine(mid, P.view_ptr, P.obj, P.img, dev, lanes_per_view=mode)
        E.set_intr(0, P.intr_guess())
        E.set_poses(0, P.poses_true.numpy())
        E.normal_eq(0, 0, 1e-3, store_blocks=True, fused=True)
        rec, W = E.rec.cpu().numpy().copy(), E.slots_to_views(E.W.T).copy()
        E.normal_eq(0, 0, 1e-3)                                        # evaluation A + B
        rec2 = E.rec.cpu().numpy().copy()
        E.schur_complement(1e-2)
        sch = E.schur.cpu().numpy().copy()
        if nv >= 63:
            E.solve_reduced(0, np.ones(E.NI, dtype=bool), 1e-2, -1.0, 0, 1)
            E.backsub(1e-2, 0, 1)
            back = (E.get_poses(1), E.norms.cpu().numpy().copy())
        else:
            back = (np.zeros(1), np.zeros(1))
        E.pose_refine(0, 0, 20, 1e-10)
        out.append((rec, W, E.get_poses(0), E.slots_to_views(E.iters).copy(), rec2, sch, back))
    (r1, w1, p1, i1, q1, s1, b1), (r2, w2, p2, i2, q2, s2, b2) = out
    assert np.abs(r1 - r2).max() <= 1e-11 * np.abs(r1).max()
    assert np.abs(w1 - w2).max() <= 1e-11 * np.abs(w1).max()
    assert np.abs(q1 - q2).max() <= 1e-11 * np.abs(q1).max() and np.abs(q1 - r1).max() <= 1e-10 * np.abs(r1).max()
    assert np.abs(s1 - s2).max() <= 1e-10 * np.abs(s1).max()
    assert np.allclose(b1[0], b2[0], rtol=1e-7, atol=1e-7) and np.allclose(b1[1], b2[1], rtol=1e-7)
    assert np.allclose(p1, p2, rtol=1e-9, atol=1e-9) and (i1 == i2).mean() > 0.98


def test_noise_free_ground_truth_on_gpu(dev):
    P = synth.config("C2", n_views=300, noise=0.0)
    objs, imgs = P.as_lists()
    r = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE,
                             criteria=(3, 200, 1e-12))
    assert r[0] < 2e-4                                   # float32 pixel rounding floor
    assert np.allclose(r[1][[0, 1, 0, 1], [0, 1, 2, 2]], P.K_true, rtol=1e-6)
    assert np.allclose(r[2].ravel(), P.D_true, atol=1e-5)


def _engine_for(P, dev):
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, P.intr_guess())
    S, N = E.S, P.n_views
    pose = torch.zeros(6, S, dtype=torch.float64, device=dev)
    pose[5] = 1.0
    sv = torch.from_numpy(E.layout["slot_view"].astype(np.int64)).to(dev)
    valid = sv >= 0
    pose[:, valid] = P.poses_true.to(dev).T[:, sv[valid]]
    E.pose[0].copy_(pose)
    return E


@pytest.mark.parametrize("fused", [False, True], ids=["two-pass", "single-pass-no-stores"])
def test_full_size_properties_config3_and_5(dev, fused):
    """fused=True is the exact call bench.py times (camcal_normal_eq_fused, no block stores,
    lambda = 0) at the exact size it times it.  Size-independent properties at BASELINE sizes (the oracle cannot run
    these in seconds): (1) additivity -- the records of the full set equal the
    sum of the records of two halves; (2) view-order invariance; (3) the C
    oracle agrees on a 4000-view sample of the same set; (4) at the true
    parameters cost = sigma^2 * 2 * n_obs within sampling error."""
    for name, nv in (("C3", 100_000), ("C5", 1_000_000)):
        P = synth.config(name, n_views=nv, device=dev)
        E = _engine_for(P, dev)
        E.set_intr(0, P.intr_true())
        ne = dict(store_blocks=False, fused=True) if fused else {}
        E.normal_eq(0, 0, 0.0, **ne)
        full = E.rec.cpu().numpy().copy()
        sigma2 = full[E.glob_len - 1] / (2 * P.n_obs)
        assert abs(sigma2 - 0.04) < 0.04 * 0.01 + 1e-5            # noise 0.2 px (+ float32 rounding)
        vp = P.view_ptr
        h = nv // 2
        parts = []
        for lo, hi in ((0, h), (h, nv)):
            Q = synth.Problem(P.model, P.image_size, P.K_true, P.D_true, P.K_guess, P.D_guess,
                              vp[lo:hi + 1] - vp[lo], P.obj[vp[lo]:vp[hi]], P.img[vp[lo]:vp[hi]],
                              P.poses_true[lo:hi])
            Eh = _engine_for(Q, dev)
            Eh.set_intr(0, P.intr_true())
            Eh.normal_eq(0, 0, 0.0, **ne)
            parts.append(Eh.rec.cpu().numpy().copy())
            del Eh, Q
        assert np.allclose(parts[0] + parts[1], full, rtol=1e-9, atol=1e-6 * np.abs(full).max())
        # C oracle on a sample
        from oracle import cport
        ns = 4000
        r = cport.normal_eq(1, (vp[:ns + 1]).copy(), P.obj[:vp[ns]].cpu().numpy(), P.img[:vp[ns]].cpu().numpy(),
                            P.intr_true(), P.poses_true[:ns].cpu().numpy(), nthreads=8, want_blocks=False)
        Q = synth.Problem(P.model, P.image_size, P.K_true, P.D_true, P.K_guess, P.D_guess, vp[:ns + 1].copy(),
                          P.obj[:vp[ns]], P.img[:vp[ns]], P.poses_true[:ns])
        Es = _engine_for(Q, dev)
        Es.set_intr(0, P.intr_true())
        Es.normal_eq(0, 0, 0.0, **ne)
        G = Es.rec.cpu().numpy()
        nu = 36
        assert rel(sym(G[:nu], 8), r["U"]) < 1e-11 and rel(G[nu:nu + 8], r["b"]) < 1e-9
        assert rel(sym(G[45:45 + nu], 8), r["C"]) < 1e-10 and rel(G[45 + nu:], r["d"]) < 1e-9
        del E, Es, P, Q
        torch.cuda.empty_cache()


def test_full_size_properties_pinhole_stress_set(dev):
    """The pinhole evaluation at the 1M-view size bench.py reports it at (C5P: evaluation A + B, the form the
    LM loop uses, and the single pass): additivity over two halves, the C oracle on a 4000-view sample,
    the noise level at the true parameters -- the pinhole twin of the test above."""
    from oracle import cport
    nv = 1_000_000
    P = synth.config("C5P", n_views=nv, device=dev)
    vp = P.view_ptr

    def engine(Q):
        E = Engine(0, Q.view_ptr, Q.obj, Q.img, dev)
        E.set_intr(0, P.intr_true())
        sv = torch.from_numpy(E.layout["slot_view"].astype(np.int64)).to(dev)
        pose = torch.zeros(6, E.S, dtype=torch.float64, device=dev)
        pose[5] = 1.0
        valid = sv >= 0
        pose[:, valid] = Q.poses_true.to(dev).T[:, sv[valid]]
        E.pose[0].copy_(pose)
        return E

    def sub(lo, hi):
        return synth.Problem(P.model, P.image_size, P.K_true, P.D_true, P.K_guess, P.D_guess, vp[lo:hi + 1] - vp[lo],
                             P.obj[vp[lo]:vp[hi]], P.img[vp[lo]:vp[hi]], P.poses_true[lo:hi])

    for ne in ({}, dict(store_blocks=False, fused=True)):
        E = engine(P)
        E.normal_eq(0, 0, 0.0, **ne)
        full = E.rec.cpu().numpy().copy()
        sigma2 = full[E.glob_len - 1] / (2 * P.n_obs)
        assert abs(sigma2 - 0.04) < 0.04 * 0.01 + 1e-5
        del E
        parts = []
        for lo, hi in ((0, nv // 2), (nv // 2, nv)):
            Eh = engine(sub(lo, hi))
            Eh.normal_eq(0, 0, 0.0, **ne)
            parts.append(Eh.rec.cpu().numpy().copy())
            del Eh
        assert np.allclose(parts[0] + parts[1], full, rtol=1e-9, atol=1e-6 * np.abs(full).max())
        ns = 4000
        r = cport.normal_eq(0, vp[:ns + 1].copy(), P.obj[:vp[ns]].cpu().numpy(), P.img[:vp[ns]].cpu().numpy(),
                            P.intr_true(), P.poses_true[:ns].cpu().numpy(), nthreads=8, want_blocks=False)
        Es = engine(sub(0, ns))
        Es.normal_eq(0, 0, 0.0, **ne)
        G = Es.rec.cpu().numpy()
        nu = 45
        assert rel(sym(G[:nu], 9), r["U"]) < 1e-11 and rel(G[nu:nu + 9], r["b"]) < 1e-9
        assert rel(sym(G[55:55 + nu], 9), r["C"]) < 1e-10 and rel(G[55 + nu:], r["d"]) < 1e-9
        del Es
        torch.cuda.empty_cache()


def test_whole_stress_solves_fisheye_and_pinhole(dev):
    """The two 1M-view solves bench.py reports (solve_wall_clock.C5_fisheye_1M_views, kernels.C5p_pinhole_1M_views_lm),
    asserted: rms at the dof-corrected noise level, intrinsics at the truth to the sampling error of 88M
    observations, and the iteration count of the oracle's run on a 2000-view sample of the same generator (the
    stopping rule looks at the intrinsic step, which does not depend on the number of views once they are many)."""
    from cameracalibration_amd import solver
    for name, mid, nd in (("C5", 1, 4), ("C5P", 0, 5)):
        Ps = synth.config(name, n_views=2000)
        os_, is_ = Ps.as_lists()
        if mid == 1:
            o = O.fisheye_calibrate(os_, is_, Ps.image_size, Ps.K_guess, Ps.D_guess, flags=FE, criteria=CRIT, solver="schur")
        else:
            o = O.calibrate_camera(os_, is_, Ps.image_size, Ps.K_guess, Ps.D_guess, flags=1, criteria=CRIT, solver="schur")
        P = synth.config(name, n_views=1_000_000, device=dev)
        E = Engine(mid, P.view_ptr, P.obj, P.img, dev)
        E.set_intr(0, P.intr_guess())
        E.pose_init(0, 0)
        assert int(E.status.abs().sum().item()) == 0
        E.pose_refine(0, 0, 20, 1e-10)
        if mid == 1:
            cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), CRIT)
        else:
            cost, it = solver.lm_pinhole(E, np.ones(9, dtype=bool), 1000, 1e-6)
        intr = E.get_intr(0)
        rms = np.sqrt(cost / P.n_obs)
        want = 0.2 * np.sqrt((2 * P.n_obs - 6 * P.n_views - E.NI) / P.n_obs)
        assert abs(rms - want) < 2e-4, (name, rms, want)
        assert it == o[5]["iters"], (name, it, o[5]["iters"])
        assert np.allclose(intr[:4], P.K_true, rtol=2e-5), (name, intr[:4])
        assert np.allclose(intr[4:4 + nd], P.D_true, atol=2e-4 if mid == 1 else 1e-3), (name, intr[4:])
        # every pose at its per-view optimum: one more refinement pass moves nothing (fisheye: cv2's rule)
        if mid == 1:
            before = E.pose[0].clone()
            E.pose_refine(0, 0, 20, 1e-10)
            movedrel = (E.pose[0] - before).norm(dim=0) / before.norm(dim=0)
            # (a handful of views sit at the 20-step cap and keep creeping; the rest take the one confirming step)
            assert float((movedrel > 1e-9).double().mean().item()) < 1e-4 and float(movedrel.max().item()) < 1e-6
            assert float(E.iters.double().mean().item()) < 1.2
        del E, P
        torch.cuda.empty_cache()


def test_large_solve_recovers_truth(dev):
    """Config 3 at full size through the whole solver: 100k ragged views."""
    P = synth.config("C3", n_views=100_000, device=dev)
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    from cameracalibration_amd import solver
    E.set_intr(0, P.intr_guess())
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), CRIT)
    intr = E.get_intr(0)
    rms = np.sqrt(cost / P.n_obs)
    # least squares absorbs 6 dof per view + 8 global: E[cost] = sigma^2 (2 n_obs - 6 N - 8)
    want = 0.2 * np.sqrt((2 * P.n_obs - 6 * P.n_views - 8) / P.n_obs)
    assert abs(rms - want) < 1e-3
    assert np.allclose(intr[:4], P.K_true, rtol=1e-4) and np.allclose(intr[4:], P.D_true, atol=2e-4)


def test_multi_rank_loop_through_the_library_communicators(dev):
    """The in-library loops with a communicator (camcal_fisheye_gn / camcal_pinhole_lm, comm != NULL) on the
    GPU: a one-rank nccl process group gives (a) the library's own RCCL communicator (ncclAllReduce on the
    compute stream) and (b) the torch.distributed callback; with one rank both sums are the identity, so the
    results must equal the plain single-rank solve bit for bit.  (More ranks than one need more GPUs than
    this tier has; the N > 1 arithmetic is covered by the gloo world-2 tests.)"""
    import torch.distributed as dist
    from cameracalibration_amd.solver import Comm
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for cfg, nv in (("C2", 300), ("C1", 20)):
            P = synth.config(cfg, n_views=nv)
            objs, imgs = P.as_lists()
            fn, fl = (cc.calibrateCamera, 1) if P.model == "pinhole" else (cc.fisheye.calibrate, FE)
            ref = fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=CRIT)
            for backend in ("rccl", "torch"):
                comm = Comm(enabled=True, backend=backend)
                r = fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=CRIT, _comm=comm)
                assert comm.kind == backend
                if P.model == "pinhole":       # same stage launches with and without a communicator: same bits
                    assert r[0] == ref[0] and np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2])
                    assert np.array_equal(np.hstack(r[3]), np.hstack(ref[3]))
                else:                          # the single-rank fisheye loop of a small set uses the two-launch iteration
                    assert abs(r[0] - ref[0]) <= 1e-12 * ref[0] and np.allclose(r[1], ref[1], rtol=1e-11, atol=0)
                    assert np.allclose(r[2], ref[2], rtol=1e-8, atol=1e-14)
                    assert np.allclose(np.hstack(r[3]), np.hstack(ref[3]), rtol=0, atol=1e-9)
        # the collective itself: in place, on the stream, value-preserving for one rank
        lib = _lib.load()
        E = Engine(1, P.view_ptr, P.obj, P.img, dev) if P.model == "fisheye" else Engine(0, P.view_ptr, P.obj, P.img, dev)
        comm = Comm(enabled=True, backend="rccl")
        h = comm.handle(E)
        E.rec.copy_(torch.arange(E.rec.numel(), dtype=torch.float64, device=dev))
        lib.check(lib.camcal_comm_allreduce(h, E.rec.data_ptr(), E.rec.numel(), E._stream()), "allreduce")
        assert torch.equal(E.rec.cpu(), torch.arange(E.rec.numel(), dtype=torch.float64))
    finally:
        from cameracalibration_amd import solver
        solver.destroy_rccl_comms()
        dist.destroy_process_group()


def test_cap_hitting_views_match_the_reinitialising_oracle_gpu(dev):
    from tests.test_hostlogic import cap_hitting_case
    cap_hitting_case({})


def test_device_side_layout_equals_the_host_layout(dev):
    """Large sets build their slot / tile tables on the device (engine.py); same tables as make_layout /
    camcal_layout, bit for bit, on a ragged set -- and the packed points and a record built from them agree."""
    from cameracalibration_amd.engine import make_layout
    P = synth.config("C3", n_views=20000, device=dev)
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    assert E._layout is None                                   # the device path ran
    ref = make_layout(np.diff(np.asarray(P.view_ptr)))
    lay = E.layout
    for k in ("slot_view", "lane_n", "tile_nmax", "tile_off"):
        assert np.array_equal(lay[k], ref[k]), k
    assert (lay["total_floats"], lay["n_tiles"], lay["n_slots"]) == (ref["total_floats"], ref["n_tiles"], ref["n_slots"])
    old = Engine.DEVICE_LAYOUT_MIN_VIEWS
    Engine.DEVICE_LAYOUT_MIN_VIEWS = 10 ** 9
    try:
        F = Engine(1, P.view_ptr, P.obj, P.img, dev)
    finally:
        Engine.DEVICE_LAYOUT_MIN_VIEWS = old
    assert F._layout is not None and torch.equal(E.t_pts, F.t_pts) and E.n_obs == F.n_obs
    assert E.tiles.planar == 1 and F.tiles.planar == 1
    # what the device-side checks report: offsets that do not fit the point arrays, a board off its plane
    with pytest.raises(ValueError):
        Engine(1, P.view_ptr, P.obj[:-1], P.img[:-1], dev)
    bad = torch.as_tensor(P.view_ptr).clone()
    bad[5] = bad[7]                                            # view 5 gets a negative count
    with pytest.raises(ValueError):
        Engine(1, bad, P.obj, P.img, dev)
    off = P.obj.clone()
    off[-1, 2] = 0.5
    assert Engine(1, P.view_ptr, off, P.img, dev).tiles.planar == 0

