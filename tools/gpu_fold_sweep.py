"""Drop-in call wall clock over set sizes, both models (development tool): run once with CAMCAL_FOLD_MAX_WAVES=0
(separate stages everywhere) and once without (folded stages up to the limit) to see where folding pays."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth
crit = (3, 1000, 1e-6)


def med(fn, n=9):
    t_end = time.perf_counter() + 0.05
    fn()
    while time.perf_counter() < t_end:
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


out = []
for name, model, fl in (("C1R", "pinhole", 1), ("C3", "fisheye", 8 | 2 | 1)):
    for nv in (25, 500, 1000, 1500, 3000, 6000):
        P = synth.config(name, n_views=nv, seed=3)
        o, i = P.as_lists()
        fn = cc.fisheye.calibrate if model == "fisheye" else cc.calibrateCamera
        out.append("%s %d: %.3f" % (model, nv, med(lambda: fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit))))
print("limit=%s | " % os.environ.get("CAMCAL_FOLD_MAX_WAVES", "default") + " | ".join(out), flush=True)
