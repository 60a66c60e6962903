#!/usr/bin/env python3
"""Does the joint Gauss-Newton pose step shorten the per-view re-estimation? (development tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
P = synth.config("C5", n_views=nv, device=dev)
E = Engine(1, P.view_ptr, P.obj, P.img, dev)
E.set_intr(0, P.intr_guess())
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
mask = np.ones(8, dtype=bool)
tot = 0.0
for k in range(7):
    a = 1 - 0.6 ** (k + 1)
    t_eval = timed(lambda: E.normal_eq(0, 0, 0.0, store_blocks=warm, fused=True))
    E.solve_reduced(0, mask, 0.0, -a, 0, 0)
    t_bs = 0.0
    if warm:
        E.dK.mul_(a)
        t_bs = timed(lambda: E.backsub(0.0, 0, 0))
    ms = timed(lambda: E.pose_refine(0, 0, 20, 1e-10))
    it = E.iters.cpu().numpy()
    tot += t_eval + t_bs + ms
    print("iter %d eval %.2f backsub %.2f refine %.2f ms, iters mean %.2f, per-tile max %.2f, intr %s" % (k, t_eval, t_bs, ms, it.mean(), it.reshape(-1, 64).max(axis=1).mean(), E.get_intr(0)[:2]))
print("total", tot)
