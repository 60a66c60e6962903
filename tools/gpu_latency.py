#!/usr/bin/env python3
"""End-to-end latency of the drop-in calls at the reference's production sizes
(development tool): C1 pinhole 20x54, C2 fisheye 500x88, a 25-view fisheye set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cameracalibration_amd as cc
from cameracalibration_amd import synth

FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
crit = (3, 1000, 1e-6)


def run(name, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        info = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: median {np.median(ts)*1e3:.2f} ms, min {min(ts)*1e3:.2f} ms, iters {info.get('iters')}", flush=True)


for cfg, nv in (("C1", 20), ("C2", 25), ("C2", 500)):
    P = synth.config(cfg, n_views=nv)
    objs, imgs = P.as_lists()

    def call():
        info = {}
        if P.model == "pinhole":
            cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=crit, _info=info)
        else:
            cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=crit, _info=info)
        return info
    run(f"{cfg} {nv} views", call)

if "--profile" in sys.argv:
    import cProfile, pstats
    P = synth.config("C1", n_views=20)
    objs, imgs = P.as_lists()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=crit)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
