#!/usr/bin/env python3
"""Latency of ONE Gauss-Newton pose step of one view under the group mappings (development tool):
a set of a few views refined with tol = -1 (never met) for N steps; device time / N.
Also: the fused evaluation and the reduced solve alone, on the same set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth, _lib
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
lib = _lib.load(os.environ["CAMCAL_LIB"]) if os.environ.get("CAMCAL_LIB") else None


def dev_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for name, model, nv in (("C2", 1, 4), ("C2", 1, 500), ("C1", 0, 4), ("C3", 1, 25)):
    P = synth.config(name, n_views=nv)
    for lpv in (64, 16):
        E = Engine(model, P.view_ptr, P.obj.to(dev), P.img.to(dev), dev, lib=lib, lanes_per_view=lpv)
        E.set_intr(0, P.intr_guess())
        E.pose_init(0, 0)
        E.pose_refine(0, 0, 20, 1e-10)
        saved = E.pose[0].clone()
        base = dev_ms(lambda: (E.pose[0].copy_(saved), E.pose_refine(0, 0, 1, -1.0)))
        for n in (21, 101):
            t = dev_ms(lambda: (E.pose[0].copy_(saved), E.pose_refine(0, 0, n, -1.0)))
            print("%s model %d views %d lanes/view %d corners %d: %3d steps %.1f us -> %.2f us per step" %
                  (name, model, nv, lpv, int(np.diff(P.view_ptr).max()), n, t * 1e3, (t - base) * 1e3 / (n - 1)), flush=True)
        E.pose[0].copy_(saved)
