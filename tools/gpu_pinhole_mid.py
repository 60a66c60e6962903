"""Pinhole LM loop time (library loop, points on the device) over set sizes (development tool); run with
CAMCAL_FOLD_MAX_WAVES=0 and without to compare the separate and the folded stages."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
out = []
for nv in (1000, 3000, 6000, 12000, 20000, 28000):
    P = synth.config("C1R", n_views=nv, seed=3, device=dev)
    K = P.K_guess
    g0 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]] + list(np.asarray(P.D_guess).ravel()[:5]))
    ts = []
    for rep in range(5):
        E = Engine(0, P.view_ptr, P.obj, P.img, dev)
        E.set_intr(0, g0); E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        err, it = E.solve_pinhole(np.ones(9, dtype=bool), 1000, 1e-6)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        del E
    out.append("%d: %.3f ms (%d it)" % (nv, float(np.median(ts[1:])), it))
print("limit=%s | " % os.environ.get("CAMCAL_FOLD_MAX_WAVES", "default") + " | ".join(out), flush=True)
