"""Randomised mid-size sets (1500 ... 40 000 views, both models, full and ragged boards): the one-call entry from host
arrays (host layout, 64 / 16 lanes per view or one, folded or separate stages by size, threaded upload where the
arrays are large) against the stage-by-stage engine solve forced to ONE lane per view (the large-set kernels, device
layout from 8192 views on) -- different kernels, different layout code, different loops; the results must agree to
the parity bar.  Usage: tools/gpu_fuzz_mid.py [seed=1] [cases=8]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import api, solver, synth, _lib
from cameracalibration_amd.engine import Engine

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cc.set_device("cuda:0")
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
crit = (3, 1000, 1e-6)
worst = dict(rms=0.0, K=0.0, D=0.0, pose=0.0)
bad = 0
for case in range(cases):
    fisheye = bool(rng.integers(2))
    ragged = bool(rng.integers(2))
    nv = int(np.exp(rng.uniform(np.log(1500), np.log(40000))))
    name = ("C3" if ragged else "C2") if fisheye else ("C1R" if ragged else "C1")
    P = synth.config(name, n_views=nv, seed=int(rng.integers(1 << 30)))
    vp = np.ascontiguousarray(np.asarray(P.view_ptr), dtype=np.int64)
    obj = np.ascontiguousarray(np.asarray(P.obj), dtype=np.float32)
    img = np.ascontiguousarray(np.asarray(P.img), dtype=np.float32)
    K9 = np.ascontiguousarray(P.K_guess, dtype=np.float64).ravel().copy()
    D5 = np.zeros(5)
    nd = 4 if fisheye else 5
    D5[:nd] = np.asarray(P.D_guess).ravel()[:nd]
    model = _lib.FISHEYE if fisheye else _lib.PINHOLE
    flags = (8 | 2 | 1) if fisheye else 1
    rms, K9, D5, rt, its = api._one_call(model, vp, obj, img, P.image_size, K9, D5, nd, flags, crit)
    E = Engine(model, vp, obj, img, dev, lanes_per_view=1)
    E.set_intr(0, P.intr_guess())
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    if fisheye:
        cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), crit)
    else:
        cost, it = solver.lm_pinhole(E, np.ones(9, dtype=bool), 1000, 1e-6)
    intr = E.get_intr(0)
    rms_e = float(np.sqrt(cost / vp[-1]))
    d = dict(rms=abs(rms - rms_e) / rms_e,
             K=float(np.max(np.abs(np.array([K9[0], K9[4], K9[2], K9[5]]) / intr[:4] - 1))),
             D=float(np.max(np.abs(D5[:nd] - intr[4:4 + nd]))),
             pose=float(np.max(np.abs(rt - E.get_poses(0)) / (1.0 + np.abs(rt)))))
    ok = its == it and d["rms"] < 1e-9 and d["K"] < 1e-8 and d["D"] < 1e-7 and d["pose"] < 1e-6
    bad += 0 if ok else 1
    for k in worst:
        worst[k] = max(worst[k], d[k])
    print("case %2d %-3s %6d views, %8d points: iterations %d / %d, rms %.6f  diffs %s%s" % (
        case, name, nv, int(vp[-1]), its, it, rms, {k: "%.1e" % v for k, v in d.items()}, "" if ok else "   <-- MISMATCH"), flush=True)
    del E
print("cases %d, mismatches %d, worst differences: %s" % (cases, bad, worst))
sys.exit(1 if bad else 0)
