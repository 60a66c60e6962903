#!/usr/bin/env python3
"""Lane mappings on device-resident sets (development tool): whole fisheye solve (engine set-up, pose
initialisation, loop) for each lanes-per-view value (0 = by size, 1, 16, 64).  Usage: tools/gpu_lanes.py C3:100000,C2:20000 1,0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth, solver, _lib
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
cases = [(c.split(":")[0], int(c.split(":")[1])) for c in (sys.argv[1] if len(sys.argv) > 1 else "C3:100000").split(",")]
modes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,0").split(",")]
for name, nv in cases:
    P = synth.config(name, n_views=nv, device=dev)
    for lpv in modes:
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lanes_per_view=lpv)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            E.set_intr(0, P.intr_guess()); E.pose_init(0, 0)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            E.pose_refine(0, 0, 20, 1e-10)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
            torch.cuda.synchronize(); t4 = time.perf_counter()
            ts.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0)])
            intr = E.get_intr(0)
            del E
        m = np.median(np.array(ts[1:]), axis=0)
        print(f"{name} {nv} lanes/view {lpv:2d}: engine {m[0]:.2f} init {m[1]:.2f} refine0 {m[2]:.2f} loop({it}) {m[3]:.2f} total {m[4]:.2f} ms  fx {intr[0]:.6f} cost {cost:.6f}", flush=True)
