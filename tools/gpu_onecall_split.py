"""Where a production-sized one-call solve spends its wall clock: list gathering, problem set-up (workspace),
the library call (camcal_calibrate), result formatting."""
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth, api, _lib

FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)
lib = _lib.load()
dev = torch.device("cuda:0")
for name, model, mid in (("C1", "pinhole", 0), ("C2", "fisheye", 1)):
    P = synth.config(name)
    o, i = P.as_lists()
    acc = np.zeros(4)
    n = 30
    for rep in range(n + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mid == 0:
            D, intr0, mask, maxc, eps = api._prep_pinhole(P.image_size, P.K_guess, P.D_guess, 1, crit)
            D5 = intr0[4:9].copy(); nd = 5; fl = 1
        else:
            intr0, mask = api._prep_fisheye(P.K_guess, P.D_guess, FE)
            D5 = P.D_guess.ravel().copy(); nd = 4; fl = FE
        vp, obj, img = api._gather_views(o, i)
        t1 = time.perf_counter()
        keep = []
        K9 = api._k9(intr0)
        p, ws, rt, status = api._problem(lib, dev, mid, vp, obj, img, P.image_size, K9, D5, nd, fl, crit, keep)
        t2 = time.perf_counter()
        rms, its = C.c_double(0.0), C.c_int32(0)
        rc = lib.camcal_calibrate(p.model, p.n_views, p.view_ptr, p.obj, p.img, p.width, p.height, p.K, p.D, p.n_dist,
                                  p.rt, p.view_status, p.flags, p.crit_type, p.max_iter, p.eps, p.workspace,
                                  p.workspace_bytes, None, api._stream_of(dev), C.byref(rms), C.byref(its))
        t3 = time.perf_counter()
        api._give_workspace(dev, ws)
        rv, tv = api._poses_to_tuples(rt)
        t4 = time.perf_counter()
        assert rc == 0
        if rep >= 3:
            acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3]) * 1e3
    acc /= n
    print("%s: gather %.3f  problem %.3f  camcal_calibrate %.3f (%d iterations)  results %.3f  = %.3f ms" % (
        name, acc[0], acc[1], acc[2], its.value, acc[3], acc.sum()), flush=True)
