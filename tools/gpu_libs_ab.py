"""A/B of library builds (compiler flags, code variants) on the C5 set: the single-pass evaluation, evaluation A / B
and the first pose refinement.  usage: gpu_libs_ab.py lib1.so lib2.so ...  (each is loaded in turn, twice)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, synth
from cameracalibration_amd.engine import Engine
dev = torch.device("cuda:0")
P = synth.config("C5", n_views=1_000_000, device=dev)
libs = sys.argv[1:]
for rnd in range(2):
    for path in libs:
        lib = _lib.load(path)
        E = Engine(1, P.view_ptr, P.obj, P.img, dev, lib=lib)
        E.set_intr(0, P.intr_guess()); E.pose_init(0, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); E.pose_refine(0, 0, 20, 1e-10); torch.cuda.synchronize()
        ref = (time.perf_counter() - t0) * 1e3
        for st in (4, 0, 5): E.time_stage(st, 30)
        f, a, b = E.time_stage(4, 150), E.time_stage(0, 60), E.time_stage(5, 60)
        print("%-28s fused %.4f  A %.4f  B %.4f  refine0 %.3f ms" % (os.path.basename(path), f, a, b, ref), flush=True)
        del E
