"""Phase times of a mid-size fisheye solve (default: C3, 100 000 ragged views): engine set-up, pose initialisation,
first refinement, loop -- each synchronised."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import synth, solver
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
from cameracalibration_amd import _lib
lib = _lib.load(os.environ["CAMCAL_LIB"]) if os.environ.get("CAMCAL_LIB") else None
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
P = synth.config(name, n_views=nv, device=dev)
def sync(): torch.cuda.synchronize()
for rep in range(4):
    sync(); t0 = time.perf_counter()
    E = Engine(1, P.view_ptr, P.obj, P.img, dev, lib=lib); sync(); t1 = time.perf_counter()
    E.set_intr(0, P.intr_guess()); E.pose_init(0, 0); sync(); t2 = time.perf_counter()
    E.pose_refine(0, 0, 20, 1e-10); sync(); t3 = time.perf_counter()
    cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6)); sync(); t4 = time.perf_counter()
    print("%s %d: engine %.3f  init %.3f  first refine %.3f  loop %.3f (%d it)  total %.3f ms" % (
        name, nv, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, it, (t4 - t0) * 1e3), flush=True)
    del E
