"""Randomised production-sized calls (development tool): the one-call path (folded loops) against the stage-by-stage
Python loop (asked for with an iteration log) over random view counts, ragged corner counts, flag sets and iteration
caps, both models.  Prints the worst relative differences; exits non-zero on a mismatch beyond the parity bar."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = dict(rms=0.0, K=0.0, rv=0.0)
bad = 0
for case in range(n_cases):
    fisheye = bool(rng.integers(2))
    nv = int(rng.choice([4, 5, 7, 16, 25, 40, 63, 64, 65, 130, 300]))
    name = ("C3" if rng.integers(2) else "C2") if fisheye else ("C1R" if rng.integers(2) else "C1")
    P = synth.config(name, n_views=nv, seed=int(rng.integers(1 << 30)))
    o, i = P.as_lists()
    if fisheye:
        fl = 8 | 1 | (2 if rng.integers(4) else 0)
        for bit in (16, 32, 64, 128, 512, 2048):
            if rng.integers(8) == 0:
                fl |= bit
        fn = cc.fisheye.calibrate
    else:
        fl = 1
        for bit in (4, 8, 16, 32, 64, 128):
            if rng.integers(8) == 0:
                fl |= bit
        fn = cc.calibrateCamera
    crit = (3, int(rng.choice([1, 2, 5, 10, 100, 1000])), float(rng.choice([1e-6, 1e-9, 1e-3])))
    try:
        a = fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit)
        info = dict(log=True)
        b = fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit, _info=info)
    except cc.error as e:
        print("case %d %s nv %d flags %d crit %s: error %s" % (case, name, nv, fl, crit, e))
        continue
    d_rms = abs(a[0] - b[0]) / max(b[0], 1e-300)
    d_K = float(np.max(np.abs(a[1] - b[1]) / np.maximum(np.abs(b[1]), 1.0)))
    d_rv = float(np.max(np.abs(np.hstack(a[3]) - np.hstack(b[3]))))
    worst["rms"], worst["K"], worst["rv"] = max(worst["rms"], d_rms), max(worst["K"], d_K), max(worst["rv"], d_rv)
    if d_rms > 1e-8 or d_K > 1e-8 or d_rv > 1e-6:
        bad += 1
        print("MISMATCH case %d %s nv %d flags %d crit %s: rms %.3e K %.3e rvec %.3e (iters %s)" % (case, name, nv, fl, crit, d_rms, d_K, d_rv, info["iters"]))
print("cases %d, mismatches %d, worst relative differences: %s" % (n_cases, bad, worst))
sys.exit(1 if bad else 0)
