"""Small fisheye sets: the library's loop (two launches per iteration: camcal_reduce_solve +
camcal_refine_normal_eq_partial) against the stage-by-stage loop (solver.gn_fisheye with a log: five
launches per iteration).  Same K, D, poses bit for bit; wall clock of the one-call path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth

FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)


def med(fn, n=15):
    t_end = time.perf_counter() + 0.05
    fn()
    while time.perf_counter() < t_end:
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


cases = [("C2", dict()), ("C3", dict(n_views=25, seed=31)), ("C3", dict(n_views=100, seed=5)), ("C2", dict(n_views=1000)),
         ("C2", dict(n_views=20)), ("C3", dict(n_views=3000, seed=2))]
for name, kw in cases:
    P = synth.config(name, **kw)
    o, i = P.as_lists()

    def call(**extra):
        return cc.fisheye.calibrate(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=FE, criteria=crit, **extra)
    a = call()
    info = dict(log=True)
    b = call(_info=info)
    t = med(call)
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and \
        all(np.array_equal(x, y) for x, y in zip(a[3], b[3])) and all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
    print("%s %s: one-call median %.3f min %.3f ms | rms %.15f vs staged %.15f (rel diff %.1e) | K, D, poses bit-identical: %s, iterations %s"
          % (name, kw, t[0], t[1], a[0], b[0], abs(a[0] - b[0]) / b[0], same, info["iters"]), flush=True)
