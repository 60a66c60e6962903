#!/usr/bin/env python3
"""Cost of k capped Gauss-Newton pose steps from the initialised poses (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
P = synth.config("C5", n_views=1000000, device=dev)
E = Engine(1, P.view_ptr, P.obj, P.img, dev)
E.set_intr(0, P.intr_guess())
E.pose_init(0, 0)
start = E.pose[0].clone()
for k in (1, 2, 3, 4, 5, 6, 8, 12, 20):
    E.pose[0].copy_(start)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E.pose_refine(0, 0, k, 1e-10)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    it = E.iters.cpu().numpy()
    print("max_iter %2d: %.2f ms, mean steps %.2f" % (k, ms, it.mean()))
