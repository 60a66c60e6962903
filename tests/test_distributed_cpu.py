"""N > 1 path on CPU: world_size 2 over gloo, views sharded by observation
count, only the small global records all-reduced.  The per-lane compute is the
CPU lane emulator (tests/hostemu); the sharding, the collectives and the
drivers are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.conftest import use_emulator  # noqa: E402
EMU = os.path.join(ROOT, "tests", "hostemu", "libcamcal_hostemu.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cameracalibration_amd as cc
    from cameracalibration_amd import _lib, synth
    from cameracalibration_amd.distributed import shard_lists
    from cameracalibration_amd.solver import Comm
    emu = _lib.load(EMU)
    from cameracalibration_amd import api
    use_emulator(emu)      # the workers run on the CPU lane emulator
    crit = (3, 1000, 1e-6)
    if model == "fisheye":
        P = synth.config("C3", n_views=90)
        objs, imgs = P.as_lists()
        o, i, (lo, hi) = shard_lists(objs, imgs, rank, world)
        fl = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
        r = cc.fisheye.calibrate(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=crit, _comm=Comm())
    else:
        P = synth.config("C1")
        objs, imgs = P.as_lists()
        o, i, (lo, hi) = shard_lists(objs, imgs, rank, world)
        r = cc.calibrateCamera(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=crit, _comm=Comm())
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), rms=r[0], K=r[1], D=r[2], rv=np.hstack(r[3]).T,
             tv=np.hstack(r[4]).T, lo=lo, hi=hi)
    dist.destroy_process_group()


@pytest.mark.parametrize("model", ["fisheye", "pinhole"])
def test_two_rank_solve_equals_single_rank(model, emu, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), model, str(tmp_path)), nprocs=world, join=True)
    import cameracalibration_amd as cc
    from cameracalibration_amd import synth
    crit = (3, 1000, 1e-6)
    if model == "fisheye":
        P = synth.config("C3", n_views=90)
        objs, imgs = P.as_lists()
        fl = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
        ref = cc.fisheye.calibrate(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl,
                                   criteria=crit)
    else:
        P = synth.config("C1")
        objs, imgs = P.as_lists()
        ref = cc.calibrateCamera(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1,
                                 criteria=crit)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert parts[0]["hi"] == parts[1]["lo"] and parts[0]["lo"] == 0 and parts[1]["hi"] == len(objs)
    for p in parts:                                   # every rank holds the same global answer
        assert abs(float(p["rms"]) - ref[0]) < 1e-10
        assert np.allclose(p["K"], ref[1], rtol=1e-10) and np.allclose(p["D"], ref[2], rtol=1e-8, atol=1e-13)
    rv = np.concatenate([p["rv"] for p in parts])
    tv = np.concatenate([p["tv"] for p in parts])
    assert np.allclose(rv, np.hstack(ref[3]).T, atol=1e-9) and np.allclose(tv, np.hstack(ref[4]).T, atol=1e-7)


def _worker_degenerate(rank, world, port, out_dir):
    """Rank 1 holds a view with collinear corners; both ranks must raise (a rank-local raise
    leaves the other rank alone in the next collective)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cameracalibration_amd as cc
    from cameracalibration_amd import _lib, synth, api
    from cameracalibration_amd.distributed import shard_lists
    from cameracalibration_amd.solver import Comm
    use_emulator(_lib.load(EMU))
    msgs = []
    for cfg, fn, fl in (("C2", cc.fisheye.calibrate, 8 | 2 | 1), ("C1", cc.calibrateCamera, 1)):
        P = synth.config(cfg, n_views=16)
        objs, imgs = P.as_lists()
        o5 = objs[13].copy(); o5[:, 0, 1] = 0.0
        i5 = imgs[13].copy(); i5[:, 0, 1] = i5[0, 0, 1]; i5[:, 0, 0] = i5[0, 0, 0] + o5[:, 0, 0]
        objs[13], imgs[13] = o5, i5                                 # view 13 lives on rank 1
        o, i, (lo, hi) = shard_lists(objs, imgs, rank, world)
        assert (lo <= 13 < hi) == (rank == 1)
        try:
            fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 30, 1e-6), _comm=Comm())
            msgs.append("no error")
        except cc.error as exc:
            msgs.append(str(exc))
    # the ranks are still in step: one more collective completes
    t = torch.ones(1)
    dist.all_reduce(t)
    assert t.item() == world
    with open(os.path.join(out_dir, f"d{rank}.txt"), "w") as f:
        f.write("\n".join(msgs))
    dist.destroy_process_group()


def test_degenerate_view_on_one_rank_raises_on_every_rank(emu, tmp_path):
    mp.spawn(_worker_degenerate, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msgs = open(tmp_path / f"d{r}.txt").read().split("\n")
        assert len(msgs) == 2 and all(m != "no error" for m in msgs), (r, msgs)


def _worker_one_call(rank, world, port, out_dir):
    """camcal_calibrate (the one-call C entry) with a communicator on this rank's shard of the views: same K, D,
    rms as the single-rank call; and a shard that is rejected on one rank only (a view with three corners)
    fails the call on BOTH ranks instead of leaving the other in the loop's first all-reduce."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cameracalibration_amd import _lib, synth
    from cameracalibration_amd.distributed import shard_ranges
    lib = _lib.load(EMU)
    res = {}
    for name, mid, nd, flags in (("C3", 1, 4, 8 | 2 | 1), ("C1", 0, 5, 1)):
        P = synth.config(name, n_views=40)
        n = np.diff(P.view_ptr)
        lo, hi = shard_ranges(n, world)[rank]
        vp = (P.view_ptr[lo:hi + 1] - P.view_ptr[lo]).astype(np.int64)
        obj = np.ascontiguousarray(P.obj.numpy()[P.view_ptr[lo]:P.view_ptr[hi]])
        img = np.ascontiguousarray(P.img.numpy()[P.view_ptr[lo]:P.view_ptr[hi]])
        for bad in (False, True):
            vpb = vp.copy()
            if bad and rank == 1:
                vpb[-1] = vpb[-2] + 3                                 # the last view of rank 1 keeps three corners
            nbytes = lib.camcal_workspace_len(mid, len(vpb) - 1, vpb.ctypes.data)
            ws = torch.empty(nbytes // 8 + 64, dtype=torch.float64)
            base = (ws.data_ptr() + 255) & ~255

            def cb(ctx, buf, cnt, stream, ws=ws):
                off = (int(buf) - ws.data_ptr()) // 8
                dist.all_reduce(ws[off:off + cnt])
                return 0
            fn = _lib.ALLREDUCE_FN(cb)
            comm = _lib.CommStruct(fn, None)
            K = np.array([P.K_guess[0, 0], 0, P.K_guess[0, 2], 0, P.K_guess[1, 1], P.K_guess[1, 2], 0, 0, 1.0])
            D = np.zeros(5)
            D[:nd] = P.D_guess.ravel()
            rt = np.zeros((len(vpb) - 1, 6))
            rms, its = C.c_double(0), C.c_int32(0)
            rc = lib.camcal_calibrate(mid, len(vpb) - 1, vpb.ctypes.data, obj.ctypes.data, img.ctypes.data, P.image_size[0],
                                      P.image_size[1], K.ctypes.data, D.ctypes.data, nd, rt.ctypes.data, None, flags, 3, 1000,
                                      1e-6, base, ws.numel() * 8 - (base - ws.data_ptr()), C.byref(comm), None,
                                      C.byref(rms), C.byref(its))
            res["%s_%s" % (name, "bad" if bad else "ok")] = np.array([rc, rms.value, its.value] + K.tolist() + D.tolist())
            if not bad:
                res["%s_rt" % name] = rt
    t = torch.ones(1)
    dist.all_reduce(t)                       # still in step
    assert t.item() == world
    np.savez(os.path.join(out_dir, f"o{rank}.npz"), **res)
    dist.destroy_process_group()


def test_one_call_entry_with_a_communicator(emu, tmp_path):
    mp.spawn(_worker_one_call, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    import cameracalibration_amd as cc
    from cameracalibration_amd import synth
    parts = [np.load(tmp_path / f"o{r}.npz") for r in range(2)]
    for name, fn, fl in (("C3", cc.fisheye.calibrate, 8 | 2 | 1), ("C1", cc.calibrateCamera, 1)):
        P = synth.config(name, n_views=40)
        objs, imgs = P.as_lists()
        ref = fn(objs, imgs, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 1000, 1e-6))
        for p in parts:
            r = p[name + "_ok"]
            assert r[0] == 0 and abs(r[1] - ref[0]) < 1e-10
            assert np.allclose(r[3:12].reshape(3, 3), ref[1], rtol=1e-10)
            assert np.allclose(r[12:12 + ref[2].size], ref[2].ravel(), rtol=1e-8, atol=1e-13)
        rt = np.concatenate([p[name + "_rt"] for p in parts])
        assert np.allclose(rt[:, :3], np.hstack(ref[3]).T, atol=1e-9) and np.allclose(rt[:, 3:], np.hstack(ref[4]).T, atol=1e-7)
        assert parts[1][name + "_bad"][0] == -1 and parts[0][name + "_bad"][0] == -7      # BADARG where it is, PEER elsewhere
