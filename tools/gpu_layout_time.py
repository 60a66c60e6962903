"""Engine set-up (layout + packing + buffers) of device-resident sets: wall clock per construction."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
for name, nv in (("C2", 8192), ("C2", 20000), ("C3", 100000), ("C5", 1000000)):
    P = synth.config(name, n_views=nv, device=dev)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E = Engine(1, P.view_ptr, P.obj, P.img, dev); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del E
    print("%s %8d views: engine set-up %s ms" % (name, nv, " ".join("%.3f" % t for t in ts)), flush=True)
