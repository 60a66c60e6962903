#!/usr/bin/env python3
"""First-contact GPU check: stage parity against the oracle on small problems
and stage timings on a large fisheye set.  Development tool, not a test."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cameracalibration_amd import synth, api, _lib
from cameracalibration_amd.engine import Engine
from oracle import cvcalib as O

dev = torch.device('cuda:0')
print('device', torch.cuda.get_device_name(0), flush=True)

def sym(tri, n):
    U = np.zeros((n, n)); iu = np.triu_indices(n); U[iu] = tri; return U + U.T - np.diag(np.diag(U))

for name, model, mid, nv in (('C1', 'pinhole', 0, 20), ('C2', 'fisheye', 1, 130), ('C3', 'fisheye', 1, 300)):
    P = synth.config(name, n_views=nv)
    objs, imgs = P.as_lists(); views = O.Views(objs, imgs)
    g = P.intr_guess(); poses = P.poses_true.numpy()
    E = Engine(mid, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, g); E.set_poses(0, poses)
    E.intr_normal(0, 0, 0, True); E.view_blocks(0, 0)
    blk = O.normal_blocks(model, views, g, poses)
    NI = E.NI; nu = NI*(NI+1)//2
    G = E.glob[0].cpu().numpy()
    print(name, 'U', np.abs(sym(G[:nu], NI)-blk['U']).max()/np.abs(blk['U']).max(),
          'b', np.abs(G[nu:nu+NI]-blk['b']).max()/np.abs(blk['b']).max(), 'cost', G[-1]/blk['cost']-1)
    V = E.slots_to_views(E.V.T); W = E.slots_to_views(E.W.T); gg = E.slots_to_views(E.g.T)
    i6 = np.triu_indices(6)
    print(name, 'V', np.abs(V-blk['V'][:, i6[0], i6[1]]).max()/np.abs(blk['V']).max(),
          'W', np.abs(W.reshape(-1, NI, 6)-blk['W']).max()/np.abs(blk['W']).max(),
          'g', np.abs(gg-blk['g']).max()/np.abs(blk['g']).max(), flush=True)

crit = (3, 1000, 1e-6)
P = synth.config('C1'); objs, imgs = P.as_lists()
t0 = time.time()
r = api.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=crit)
t1 = time.time()
o = O.calibrate_camera(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=crit, solver='schur')
print('C1 gpu', t1-t0, 's rms', r[0], 'vs oracle', o[0], 'K', np.abs(r[1]-o[1]).max(), 'D', np.abs(r[2]-o[2]).max(), flush=True)
P = synth.config('C2'); objs, imgs = P.as_lists()
info = {}
t0 = time.time()
r = api.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=11, criteria=crit, _info=info)
t1 = time.time()
o = O.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess, P.D_guess, flags=11, criteria=crit, solver='schur')
print('C2 gpu', t1-t0, 's iters', info['iters'], 'rms', r[0], 'vs oracle', o[0], 'K', np.abs(r[1]-o[1]).max(), 'D', np.abs(r[2]-o[2]).max(), flush=True)
t0 = time.time()
r = api.fisheye_calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=11, criteria=crit)
print('C2 gpu second call', time.time()-t0, flush=True)

for nv in (65536, 1000000):
    t0 = time.time()
    P = synth.config('C5', n_views=nv, device=dev)
    torch.cuda.synchronize(); print('gen', nv, time.time()-t0, 's', flush=True)
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, P.intr_guess())
    E.pose[0].copy_(torch.zeros_like(E.pose[0]))
    E.pose[0][:, :nv] = P.poses_true.T
    E.pose[0][5, nv:] = 1.0
    for st, nm in ((0, 'intr_normal'), (1, 'view_blocks'), (2, 'schur')):
        E.time_stage(st, 2)
        ms = E.time_stage(st, 10)
        print(f'  {nm}: {ms:.3f} ms  -> {P.n_obs/ms*1e3:.3e} obs/s', flush=True)
    t0 = time.time(); E.pose_init(0, 1); torch.cuda.synchronize(); print('  pose_init', time.time()-t0)
    t0 = time.time(); E.pose_refine(0, 1, 20, 1e-10); torch.cuda.synchronize(); print('  pose_refine', time.time()-t0, 'mean iters', E.iters.float().mean().item())
    d = (E.pose[1][:, :nv] - E.pose[0][:, :nv]).abs().max(dim=1).values
    print('  refined-vs-truth pose maxabs', d.cpu().numpy())
    del E, P
