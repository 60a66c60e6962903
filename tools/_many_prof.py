import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth
FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)
calls = []
for name, model, nv, seed in (("C3", "fisheye", 25, 31), ("C3", "fisheye", 25, 32), ("C1R", "pinhole", 25, 33), ("C1R", "pinhole", 25, 34)):
    P = synth.config(name, n_views=nv, seed=seed)
    o, i = P.as_lists()
    calls.append(dict(model=model, objectPoints=o, imagePoints=i, imageSize=P.image_size, cameraMatrix=P.K_guess,
                      distCoeffs=P.D_guess, flags=FE if model == "fisheye" else 1, criteria=crit))
def fresh():
    return [dict(c, cameraMatrix=c["cameraMatrix"].copy(), distCoeffs=c["distCoeffs"].copy()) for c in calls]
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cc.calibrate_many(fresh())
    torch.cuda.synchronize(); print((time.perf_counter() - t0) * 1e3)
