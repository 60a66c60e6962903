#!/usr/bin/env python3
"""Stage timings on the C5 workload (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cameracalibration_amd import synth
from cameracalibration_amd.engine import Engine
dev = torch.device('cuda:0')
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cfg = sys.argv[2] if len(sys.argv) > 2 else 'C5'
P = synth.config(cfg, n_views=nv, device=dev)
E = Engine(1, P.view_ptr, P.obj, P.img, dev)
E.set_intr(0, P.intr_guess())
E.pose_init(0, 0); E.pose_refine(0, 0, 20, 1e-10)
torch.cuda.synchronize()
for rnd in range(3):
    for st, nm in ((0, 'evalA'), (5, 'evalB_cross_schur'), (3, 'normal_eq'), (4, 'fused_normal_eq')):
        E.time_stage(st, 2)
        ms = E.time_stage(st, 20)
        print(f'{cfg} {nv} {nm}: {ms:.4f} ms -> {P.n_obs/ms*1e3:.3e} obs/s', flush=True)
