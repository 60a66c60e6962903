#!/usr/bin/env python3
"""Where the wall clock of a C2-sized drop-in call goes, stage by stage (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cameracalibration_amd as cc
from cameracalibration_amd import synth, _lib
from cameracalibration_amd.api import _gather_views, _poses_to_tuples
from cameracalibration_amd.engine import Engine
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 500
P = synth.config("C2", n_views=nv)
objs, imgs = P.as_lists()
dev = torch.device("cuda:0")
acc = {}
def tick(name, t0):
    torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return time.perf_counter()
for rep in range(8):
    torch.cuda.synchronize()
    t = time.perf_counter()
    vp, o, i = _gather_views(objs, imgs); t = tick("gather lists", t)
    E = Engine(1, vp, o, i, dev); t = tick("Engine (uploads, pack, buffers)", t)
    E.set_intr(0, P.intr_guess()); t = tick("set_intr", t)
    E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10); t = tick("pose init + refine", t)
    cost, its = E.solve_fisheye(np.ones(8, bool), (3, 1000, 1e-6), True); t = tick("solve_fisheye (library loop)", t)
    intr = E.get_intr(0); poses = E.get_poses(0); t = tick("read results", t)
    rv, tv = _poses_to_tuples(poses); t = tick("result tuples", t)
tot = 0
for k, v in acc.items():
    m = float(np.median(v[2:])); tot += m
    print(f"{k:36s} {m:7.3f} ms")
print(f"{'sum (each stage synchronised)':36s} {tot:7.3f} ms, iterations {its}")
