"""A/B of library builds on production-sized one-call solves (development tool).
usage: gpu_small_ab.py lib1.so lib2.so ...   (alternating, three rounds; median wall clock per call)"""
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, synth, api
import cameracalibration_amd as cc
FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)
libs = [_lib.CamcalLib(p) if hasattr(_lib, "CamcalLib") else _lib.load(p) for p in sys.argv[1:]]
dev = torch.device("cuda:0")
cases = [("C2", {}, "fisheye"), ("C3", dict(n_views=25, seed=31), "fisheye"), ("C1", {}, "pinhole")]
sets = []
for name, kw, model in cases:
    P = synth.config(name, **kw)
    sets.append((name, model, P) + P.as_lists())


def med(fn, n=21):
    t_end = time.perf_counter() + 0.05
    fn()
    while time.perf_counter() < t_end:
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for rnd in range(3):
    for path, lib in zip(sys.argv[1:], libs):
        api._resolve = lambda lib=lib: (lib, dev)
        out = []
        for name, model, P, o, i in sets:
            fn = cc.fisheye.calibrate if model == "fisheye" else cc.calibrateCamera
            fl = FE if model == "fisheye" else 1
            out.append("%s %.3f" % (name, med(lambda: fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit))))
        print("%-26s %s" % (os.path.basename(path), "  ".join(out)), flush=True)
