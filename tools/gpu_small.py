#!/usr/bin/env python3
"""Production-sized solves (development tool): wall clock and results with one lane per
view and with the grouped mapping chosen for small sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cameracalibration_amd as cc
from cameracalibration_amd import synth, _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, -1]
FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
from cameracalibration_amd.engine import Engine
lib = _lib.load()
cases = (("C1", 20), ("C2", 25), ("C2", 500), ("C2", 2000), ("C2", 8000))
if len(sys.argv) > 3:
    cases = tuple((c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[3].split(","))
for cfg, nv in cases:
    P = synth.config(cfg, n_views=nv)
    objs, imgs = P.as_lists()
    ref = None
    for mode in modes:
        Engine.LANES_PER_VIEW = max(mode, 0)       # -1 / 0: chosen by size
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = {}
            if P.model == "pinhole":
                out = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                                         criteria=(3, 1000, 1e-6), _info=info)
            else:
                out = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE,
                                           criteria=(3, 1000, 1e-6), _info=info)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        rms, K, D = out[0], out[1], np.asarray(out[2]).ravel()
        msg = ""
        if ref is None:
            ref = (rms, K, D)
        else:
            msg = (f" | vs first mode: rms {abs(rms - ref[0]):.2e} K {np.abs(K - ref[1]).max() / ref[1].max():.2e}"
                   f" D {np.abs(D - ref[2]).max() / np.abs(ref[2]).max():.2e}")
        print(f"{cfg} {nv} views lanes/view {mode:3d}: median {np.median(ts[1:])*1e3:.2f} ms, iters {info.get('iters')}, rms {rms:.9f}{msg}", flush=True)
Engine.LANES_PER_VIEW = 0
if os.environ.get("CAMCAL_CPROFILE"):
    import cProfile, pstats
    P = synth.config("C2", n_views=500)
    objs, imgs = P.as_lists()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=(3, 1000, 1e-6))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
