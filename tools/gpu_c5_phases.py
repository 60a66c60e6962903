#!/usr/bin/env python3
"""Where a large fisheye solve spends its time (development tool): engine set-up (layout, packing,
buffers), pose initialisation, first refinement, the damped Gauss-Newton loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth, solver, _lib
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else (1_000_000 if name == "C5" else 100_000)
P = synth.config(name, n_views=nv, device=dev)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev)
    sync(); t1 = time.perf_counter()
    E.set_intr(0, P.intr_guess()); E.pose_init(0, 0)
    sync(); t2 = time.perf_counter()
    E.pose_refine(0, 0, 20, 1e-10)
    sync(); t3 = time.perf_counter()
    cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
    sync(); t4 = time.perf_counter()
    intr = E.get_intr(0)
    t5 = time.perf_counter()
    print(f"{name} {nv}: engine {1e3*(t1-t0):.2f}  pose_init {1e3*(t2-t1):.2f}  refine0 {1e3*(t3-t2):.2f}  loop({it}) {1e3*(t4-t3):.2f}  read {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms", flush=True)
    del E
