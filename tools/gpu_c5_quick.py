"""C5 / C2 solve timing with and without the predicted warm start (Engine.PREDICT_POSES), one process."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import _lib, synth, solver
from cameracalibration_amd.engine import Engine

dev = torch.device("cuda:0")
FE = 8 | 2 | 1
P = synth.config("C5", n_views=1_000_000, device=dev)
for pred in (False, True, False, True):
    Engine.PREDICT_POSES = pred
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, P.intr_guess()); E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("C5 predict=%s setup %.2f ms loop %.2f ms total %.2f ms it %d rms %.9f fx %.9f" % (
        pred, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, it, np.sqrt(cost / P.n_obs), E.get_intr(0)[0]), flush=True)
    del E
del P
torch.cuda.empty_cache()
for name, nv in (("C2", 500), ("C3", 100_000)):
    P = synth.config(name, n_views=nv, device=dev if nv > 1000 else "cpu")
    for pred in (False, True, False, True):
        Engine.PREDICT_POSES = pred
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            E = Engine(1, P.view_ptr, P.obj, P.img, dev)
            E.set_intr(0, P.intr_guess()); E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
            cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%s predict=%s median %.3f ms it %d rms %.9f" % (name, pred, float(np.median(ts)), it, np.sqrt(cost / P.n_obs)), flush=True)
