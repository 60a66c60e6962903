"""The one-call entry (camcal_calibrate) on LARGE sets handed over as host arrays: wall clock including the layout on
the host, the upload over PCIe, packing, the solve and the download of the poses -- the PCIe-inclusive figure
beside the device-resident solve times of bench.py."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth, api

cc.set_device("cuda:0")
FE = 8 | 2 | 1
crit = (3, 1000, 1e-6)
for name, nv in (("C2", 20000), ("C3", 100000), ("C5", 1000000)):
    P = synth.config(name, n_views=nv)
    vp = np.ascontiguousarray(np.asarray(P.view_ptr), dtype=np.int64)
    obj = np.ascontiguousarray(np.asarray(P.obj), dtype=np.float32)
    img = np.ascontiguousarray(np.asarray(P.img), dtype=np.float32)
    mb = (obj.nbytes + img.nbytes + vp.nbytes) / 1e6
    ts = []
    for rep in range(4):
        K9 = np.ascontiguousarray(P.K_guess, dtype=np.float64).ravel().copy()
        D5 = np.zeros(5); D5[:4] = np.asarray(P.D_guess).ravel()[:4]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rms, K9, D5, rt, its = api._one_call(1, vp, obj, img, P.image_size, K9, D5, 4, FE, crit)
        ts.append((time.perf_counter() - t0) * 1e3)
    n_obs = int(vp[-1])
    print("%s %8d views %6.0f MB in: %s ms  (rms %.6f, %d iterations; best %.1f ms = %.2e obs/s whole solve, %.1f GB/s if it were all upload)"
          % (name, nv, mb, " ".join("%.1f" % t for t in ts), rms, its, min(ts), n_obs / (min(ts) * 1e-3), mb / 1e3 / (min(ts) * 1e-3)), flush=True)
