#!/usr/bin/env python3
"""A/B of the single-pass evaluation (development tool): prints the fused
stage time and the record it produced, for comparing kernel variants selected
by environment variables."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device('cuda:0')
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P = synth.config('C5', n_views=nv, device=dev)
from cameracalibration_amd import _lib
lib = _lib.load(os.environ['CAMCAL_LIB']) if os.environ.get('CAMCAL_LIB') else None
E = Engine(1, P.view_ptr, P.obj, P.img, dev, lib=lib)
E.set_intr(0, P.intr_guess())
E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
torch.cuda.synchronize()
E.time_stage(4, 3)
for rnd in range(3):
    ms = E.time_stage(4, 20)
    msA, msB = E.time_stage(0, 10), E.time_stage(5, 10)
    print(f'  evalA {msA:.4f} evalB {msB:.4f}')
    print(f'variant LDS={os.environ.get("CAMCAL_FUSED_LDS","0")} fused: {ms:.4f} ms -> {P.n_obs/ms*1e3:.3e} obs/s', flush=True)
E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
r = E.rec.cpu().numpy()
print('rec checksum', repr(float(r.sum())), repr(float(abs(r).sum())))
