import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth
FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)
for name, kw, model in (("C2", {}, 1), ("C3", dict(n_views=25, seed=31), 1), ("C1", {}, 0), ("C1R", dict(n_views=25, seed=33), 0)):
    P = synth.config(name, **kw)
    o, i = P.as_lists()
    fn = cc.fisheye.calibrate if model else cc.calibrateCamera
    for _ in range(6):
        fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE if model else 1, criteria=crit)
    torch.cuda.synchronize()
