#!/usr/bin/env python3
"""Sustained FP64 FMA rate (clock under load) -- development tool."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cameracalibration_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
buf = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
res = C.c_double(0)
for reps in (1, 5, 20, 100, 400):
    lib.camcal_peak_probe(0, buf.data_ptr(), buf.numel(), reps, None, C.byref(res))
    print('fma reps', reps, 'TFLOP/s', res.value, flush=True)
