"""Wall clock of production-sized calls through the drop-in API: C1, C2 (one-call path vs the Engine path),
and the reference's four-camera loop (4 x <= 25 ragged views) one call after another vs calibrate_many."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth

FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)


def med(fn, n=15):
    fn(); fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


for name, model in (("C1", "pinhole"), ("C2", "fisheye")):
    P = synth.config(name)
    o, i = P.as_lists()
    fn = cc.fisheye.calibrate if model == "fisheye" else cc.calibrateCamera
    fl = FE if model == "fisheye" else 1
    a = med(lambda: fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit))
    b = med(lambda: fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit, _info={}))
    print("%s one-call median %.3f min %.3f ms | engine path median %.3f min %.3f ms" % (name, a[0], a[1], b[0], b[1]), flush=True)
    so, si = np.stack(o), np.stack(i)                   # the views stacked in one array each: nothing to gather
    c = med(lambda: fn(so, si, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit))
    print("%s one-call, stacked arrays median %.3f min %.3f ms" % (name, c[0], c[1]), flush=True)

calls = []
for name, model, nv, seed in (("C3", "fisheye", 25, 31), ("C3", "fisheye", 25, 32), ("C1R", "pinhole", 25, 33), ("C1R", "pinhole", 25, 34)):
    P = synth.config(name, n_views=nv, seed=seed)
    o, i = P.as_lists()
    calls.append(dict(model=model, objectPoints=o, imagePoints=i, imageSize=P.image_size, cameraMatrix=P.K_guess,
                      distCoeffs=P.D_guess, flags=FE if model == "fisheye" else 1, criteria=crit))


def fresh():
    return [dict(c, cameraMatrix=c["cameraMatrix"].copy(), distCoeffs=c["distCoeffs"].copy()) for c in calls]


def seq():
    for c in fresh():
        fn = cc.fisheye.calibrate if c["model"] == "fisheye" else cc.calibrateCamera
        fn(c["objectPoints"], c["imagePoints"], c["imageSize"], c["cameraMatrix"], c["distCoeffs"], flags=c["flags"], criteria=c["criteria"])


a = med(seq)
b = med(lambda: cc.calibrate_many(fresh()))
print("production 4 cams (2 fisheye + 2 pinhole, 25 ragged views each): one after another median %.3f min %.3f ms | calibrate_many median %.3f min %.3f ms" % (a[0], a[1], b[0], b[1]), flush=True)
