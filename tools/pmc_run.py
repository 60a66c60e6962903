#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: one calibration read of a known byte
count with the point-load access shape, then a few launches of the hot
kernels on the C5 set.  Usage (one counter group per run):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o x -- python3 tools/pmc_run.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cameracalibration_amd import _lib, synth
from cameracalibration_amd.engine import Engine
dev = torch.device('cuda:0')
lib = _lib.load()
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P = synth.config('C5', n_views=nv, device=dev)
E = Engine(1, P.view_ptr, P.obj, P.img, dev)
E.set_intr(0, P.intr_guess())
E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
del P
buf = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
res = C.c_double(0)
torch.cuda.synchronize()
lib.camcal_peak_probe(2, buf.data_ptr(), buf.numel(), 1, None, C.byref(res))   # 2 launches x 1 GiB, 4-byte loads
lib.camcal_peak_probe(1, buf.data_ptr(), buf.numel(), 1, None, C.byref(res))   # 2 launches x 1 GiB, 16-byte loads
for _ in range(3):
    E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
for _ in range(3):
    E.normal_eq_response(0, 0)              # the form the fisheye loop runs since round 3: + V^-1 W^T per view (384 B)
for _ in range(3):
    E.normal_eq(0, 0, 0.0, store_blocks=True, fused=False)
# pose re-estimation: from the homography initialisation (the first refinement of a solve)
E.pose_init(0, 0)
E.pose_refine(0, 0, 20, 1e-10)
print('refine steps', float(E.iters.double().sum().item()))
torch.cuda.synchronize()
print('done')
