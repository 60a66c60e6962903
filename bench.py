#!/usr/bin/env python3
"""Benchmark of the calibration hot path on MI355X.

One "step" = one full normal-equation evaluation (projection + residual +
analytic Jacobian + per-view blocks + Schur complement onto the intrinsics,
plus -- on N > 1 GPUs -- the all-reduce of the small global record) over the
rank's resident views.  Workload at every N: BASELINE.json config 5, the
stress set of Kannala-Brandt fisheye views x 88 corners, ``--views`` views
PER GPU (weak scaling, default 1,000,000 = the config's size on one GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from cameracalibration_amd import _lib, synth  # noqa: E402
from cameracalibration_amd.engine import Engine  # noqa: E402
from cameracalibration_amd.solver import _all_reduce  # noqa: E402  (device tensors on a gloo group go through the host)

# Algorithmic work model (BASELINE.md section 3 / SURVEY.md 8d; re-counted in DESIGN.md)
FLOP_PER_OBS = {"fisheye": 605.0, "pinhole": 732.0}
FLOP_PER_VIEW = 1200.0
BYTES_PER_OBS = 20.0      # X Y Z u v float32 (the planar fast path reads 16: Z rows are skipped)
BYTES_PER_VIEW = 696.0
# what the timed instantiation (planar, no block stores) moves: X Y u v, the pose, and one partial
# record [U|b|cost|C|d] of 81 doubles per tile of 64 views written and read once by the second stage
MOVED_BYTES_PER_OBS = 16.0
MOVED_BYTES_PER_VIEW = 48.0 + 2.0 * 8.0 * 81.0 / 64.0
PEAK_FP64_TFLOPS = 78.6     # 256 CU x 4 SIMD x 16 FMA lanes x 2 flop x 2.4 GHz (vector FP64)
PEAK_HBM_GBS = 8000.0


def kernel_source_sha():
    """Hash of the kernel sources: a committed PMC profile only describes the kernels it was taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("camcal_kernels.hip", "camcal_lane.h", "camcal_math.h"):
        h.update(open(os.path.join(HERE, "cameracalibration_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(n_obs):
    """HBM traffic and executed flops of the step's kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, collected by tools/pmc_run.py; PMC cannot be read from inside this
    process), scaled by observation count.  Values are None when the file is absent and flagged stale
    when the kernel sources changed since the passes were taken."""
    out = {"traffic": None, "executed_flop_per_obs": None, "stale": None, "source": None, "counters": None}
    try:
        tj = json.load(open(os.path.join(HERE, "profiles", "pmc_traffic.json")))
        k = tj["k_normal_fused<fisheye,no block stores,planar>"]
        out["traffic"] = k["hbm_bytes"] * n_obs / tj["workload_obs"]
        out["executed_flop_per_obs"] = k.get("executed_flop_per_obs")
        out["counters"] = {c: k[c] for c in ("valu_busy_of_wave_cycles", "lane_utilisation",
                                             "fp64_share_of_valu_instructions") if c in k} or None
        out["stale"] = tj.get("kernel_source_sha") != kernel_source_sha()
        out["source"] = tj.get("source")
    except Exception:
        pass
    return out


def cpu_baseline(seconds_target=12.0):
    """Times the C oracle (oracle/normal_eq.c, a plain restatement of cv2's
    per-view accumulation) on a bounded sample of the same workload."""
    try:
        from oracle import cport
        nthreads = host_threads()
        n_views = 20000
        P = synth.config("C5", n_views=n_views, device="cpu")
        intr = P.intr_guess()
        poses = P.poses_true.numpy()
        obj, img = P.obj.numpy(), P.img.numpy()
        cport.normal_eq(1, P.view_ptr, obj, img, intr, poses, nthreads=nthreads, want_blocks=False)
        reps, t0 = 0, time.perf_counter()
        while True:
            cport.normal_eq(1, P.view_ptr, obj, img, intr, poses, nthreads=nthreads, want_blocks=False)
            reps += 1
            el = time.perf_counter() - t0
            if el > seconds_target or reps >= 100000:
                break
        return {"value": P.n_obs * reps / el, "unit": "corner-obs/s", "cores": nthreads, "kind": "port",
                "host_cpu_count": os.cpu_count(),
                "sample": f"{reps} evaluations of {n_views} fisheye views x 88 corners (oracle/normal_eq.c, "
                          f"OpenMP {nthreads} threads = the box's CPU share of {os.cpu_count()} host CPUs, {el:.1f} s)"}
    except Exception as exc:  # the baseline is reported, never required
        return {"value": None, "unit": "corner-obs/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}


def host_threads():
    """Threads the CPU legs use: the box's CPU share (OMP_NUM_THREADS is set to it there), else every core."""
    try:
        return max(1, int(os.environ.get("OMP_NUM_THREADS", "") or (os.cpu_count() or 1)))
    except ValueError:
        return os.cpu_count() or 1


def cpu_port_solves():
    """the repo's own Schur-based CPU restatement -- the solver's per-view C++ code looped
    over the views with OpenMP on all host cores (oracle/libcamcal_cpu.so, built from tests/hostemu) -- timed on
    every config: C1 / C2 in full, C3 / C5 on a 20 000-view sample of the same generator (the form is linear in
    the views), C4 on a 200-pose rig.  A stated baseline, never the target."""
    from cameracalibration_amd import rig, solver
    out = {"threads": host_threads(), "host_cpu_count": os.cpu_count(),
           "what": "oracle/libcamcal_cpu.so: the solver's per-view C++ code behind the same C ABI, OpenMP over tiles"}
    cpu = _lib.load(os.path.join(HERE, "oracle", "libcamcal_cpu.so"))
    dev = torch.device("cpu")

    def timed(fn, reps=2):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), r

    for key, name, nv, mid in (("C1_pinhole_20x54", "C1", 20, 0), ("C2_fisheye_500x88", "C2", 500, 1),
                               ("C3_fisheye_100k_ragged_views", "C3", 20000, 1), ("C5_fisheye_1M_views", "C5", 20000, 1)):
        P = synth.config(name, n_views=nv)

        def solve():
            E = Engine(mid, P.view_ptr, P.obj, P.img, dev, lib=cpu)
            E.set_intr(0, P.intr_guess())
            E.pose_init(0, 0)
            E.pose_refine(0, 0, 20, 1e-10)
            if mid == 1:
                return solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
            return solver.lm_pinhole(E, np.ones(9, dtype=bool), 1000, 1e-6)
        ms, (cost, it) = timed(solve)
        out[key] = {"views": nv, "ms": ms, "iterations": it, "rms_px": float(np.sqrt(cost / P.n_obs)),
                    "ms_per_1000_views": ms * 1000.0 / nv}
    R = synth.config("C4", n_views=200)
    cams = [dict(pose_idx=R.pose_idx[c], view_ptr=R.view_ptr[c], obj=R.obj[c], img=R.img[c]) for c in range(R.n_cams)]
    Ks, Ds = R.guess_matrices()
    from cameracalibration_amd import api
    orig = api._resolve
    api._resolve = lambda: (cpu, dev)        # this leg only: the rig front end resolves its library through api._resolve
    try:
        def rig_solve():
            info = {}
            r = rig.calibrate_rig(None, None, R.image_size, Ks, Ds, criteria=(3, 100, 1e-7), cams=cams, n_poses=R.n_poses,
                                  _info=info)
            return r[0], info["iters"]
        ms, (rms, it) = timed(rig_solve)
    finally:
        api._resolve = orig
    out["C4_rig_4x10000_poses"] = {"poses": 200, "ms": ms, "iterations": it, "rms_px": float(rms),
                                   "ms_per_1000_poses": ms * 1000.0 / 200}
    return out


def big_solve(name, n_views, dev, lib):
    """Whole fisheye solve (engine set-up with the layout and the packing, pose initialisation, damped
    Gauss-Newton to the reference's criteria) of a device-resident synthetic set: second of two runs (the
    first pays one-time costs -- loading torch's sort kernels, the allocator's first blocks -- and is
    reported as first_run_ms)."""
    from cameracalibration_amd import solver
    P = synth.config(name, n_views=n_views, device=dev)
    n_obs = P.n_obs
    runs = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib)
        E.set_intr(0, P.intr_guess())
        E.pose_init(0, 0)
        E.pose_refine(0, 0, 20, 1e-10)
        cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6))
        intr = E.get_intr(0)
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) * 1e3)
        del E
    res = {"ms": runs[1], "first_run_ms": runs[0], "iterations": it, "observations": n_obs,
           "rms_px": float(np.sqrt(cost / n_obs)),
           "fx_rel_err": float(abs(intr[0] / P.K_true[0] - 1)), "note": "points already on the device; "
           "includes the layout, packing, pose initialisation and every host round trip"}
    try:
        res["from_host_arrays"] = host_array_solve(P)
    except Exception as exc:              # reported, never required
        res["from_host_arrays"] = {"error": repr(exc)}
    return res


def host_array_solve(P):
    """The same set through the one-call entry (camcal_calibrate) from HOST arrays to host results: layout,
    upload over PCIe, packing, solve, poses back in view order -- the PCIe-inclusive figure (never `value`)."""
    from cameracalibration_amd import api
    vp = np.ascontiguousarray(torch.as_tensor(P.view_ptr).cpu().numpy(), dtype=np.int64)
    obj = np.ascontiguousarray(torch.as_tensor(P.obj).cpu().numpy(), dtype=np.float32)
    img = np.ascontiguousarray(torch.as_tensor(P.img).cpu().numpy(), dtype=np.float32)
    ts = []
    for _ in range(3):
        K9 = np.ascontiguousarray(P.K_guess, dtype=np.float64).ravel().copy()
        D5 = np.zeros(5)
        D5[:4] = np.asarray(P.D_guess).ravel()[:4]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rms, K9, D5, rt, its = api._one_call(_lib.FISHEYE, vp, obj, img, P.image_size, K9, D5, 4, 8 | 2 | 1, (3, 1000, 1e-6))
        ts.append((time.perf_counter() - t0) * 1e3)
    mb = (vp.nbytes + obj.nbytes + img.nbytes) / 1e6
    return {"ms": float(min(ts[1:])), "first_run_ms": ts[0], "iterations": int(its), "rms_px": float(rms), "megabytes_in": mb,
            "obs_per_s": float(vp[-1]) / (min(ts[1:]) * 1e-3),
            "note": "camcal_calibrate: host arrays in, host results out (PCIe-inclusive; upload through page-locked staging "
                    "filled by several threads, planarity tested on the device)"}


def solve_times(dev, with_cpu=True):
    """Wall-clock of whole solves through the drop-in API (SURVEY.md 8d asks for it
    next to the evaluation rate): median of 3 after one warm-up, host lists in,
    results out, so H2D, launches and every host round trip are inside.  with_cpu:
    the same C1 / C2 calls through the numpy oracle (one host thread) beside them."""
    import cameracalibration_amd as cc
    from cameracalibration_amd import rig
    out = {}
    fe = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
    crit = (3, 1000, 1e-6)

    def med(fn, reps=9, warm_s=0.05):
        # steady state: the CPU legs between these measurements leave the device idle for seconds, and a
        # millisecond call right after that is timed at idle clocks; so repeat the call for warm_s first
        t_end = time.perf_counter() + warm_s
        fn()
        while time.perf_counter() < t_end:
            fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, info

    try:
        P = synth.config("C1")
        o, i = P.as_lists()

        def c1(info=None):          # the plain drop-in call: the one-call path (camcal_calibrate)
            cc.calibrateCamera(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=1, criteria=crit, _info=info)
            return info
        ms, _ = med(c1)
        out["C1_pinhole_20x54"] = {"ms": ms, "iterations": c1({})["iters"]}
        if with_cpu:
            from oracle import cvcalib as O
            t0 = time.perf_counter()
            O.calibrate_camera(o, i, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=crit, solver="dense")
            out["C1_pinhole_20x54"]["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
            out["C1_pinhole_20x54"]["cpu_oracle"] = "oracle/cvcalib.py dense CvLevMarq (cv2's formulation), numpy, 1 thread"
        P2 = synth.config("C2")
        o2, i2 = P2.as_lists()

        def c2(info=None):
            cc.fisheye.calibrate(o2, i2, P2.image_size, P2.K_guess.copy(), P2.D_guess.copy(), flags=fe,
                                 criteria=crit, _info=info)
            return info
        ms, _ = med(c2)
        out["C2_fisheye_500x88"] = {"ms": ms, "iterations": c2({})["iters"]}
        if with_cpu:
            t0 = time.perf_counter()
            O.fisheye_calibrate(o2, i2, P2.image_size, P2.K_guess, P2.D_guess, flags=fe, criteria=crit, solver="schur")
            out["C2_fisheye_500x88"]["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
            out["C2_fisheye_500x88"]["cpu_oracle"] = ("oracle/cvcalib.py Schur-eliminated Gauss-Newton, numpy, 1 thread (cv2's dense "
                                                      "formulation of the same loop needs a 3008 x 3008 J^T J per iteration)")
        # the reference's real loop: cali_in solves its four cameras one after another
        # (controller/controller_internal_calibration.py:375-399; two fisheye + two pinhole, <= 25 ragged views each)
        calls = []
        for nm, model, seed in (("C3", "fisheye", 31), ("C3", "fisheye", 32), ("C1R", "pinhole", 33), ("C1R", "pinhole", 34)):
            Pc = synth.config(nm, n_views=25, seed=seed)
            oc, ic = Pc.as_lists()
            calls.append(dict(model=model, objectPoints=oc, imagePoints=ic, imageSize=Pc.image_size, cameraMatrix=Pc.K_guess,
                              distCoeffs=Pc.D_guess, flags=fe if model == "fisheye" else 1, criteria=crit))

        def fresh():
            return [dict(c, cameraMatrix=c["cameraMatrix"].copy(), distCoeffs=c["distCoeffs"].copy()) for c in calls]

        def one_by_one():
            for c in fresh():
                fn = cc.fisheye.calibrate if c["model"] == "fisheye" else cc.calibrateCamera
                fn(c["objectPoints"], c["imagePoints"], c["imageSize"], c["cameraMatrix"], c["distCoeffs"], flags=c["flags"],
                   criteria=c["criteria"])
        ms_seq, _ = med(one_by_one)
        ms_many, _ = med(lambda: cc.calibrate_many(fresh()))
        out["production_4cams"] = {"ms": ms_many, "one_after_another_ms": ms_seq,
                                   "what": "2 fisheye + 2 pinhole cameras, 25 ragged views (13..63 corners) each: "
                                           "calibrate_many (camcal_calibrate_many: one host thread and stream per camera) "
                                           "against the four drop-in calls in sequence; host lists in, results out"}
        R = synth.config("C4", device=dev)
        cams = [dict(pose_idx=R.pose_idx[c], view_ptr=R.view_ptr[c], obj=R.obj[c], img=R.img[c])
                for c in range(R.n_cams)]
        Ks, Ds = R.guess_matrices()

        def c4():
            info = {}
            rig.calibrate_rig(None, None, R.image_size, Ks, Ds, criteria=(3, 100, 1e-7), cams=cams,
                              n_poses=R.n_poses, _info=info)
            return info
        ms, info = med(c4)
        out["C4_rig_4x10000_poses"] = {"ms": ms, "iterations": info["iters"], "observations": R.n_obs,
                                        "note": "points already on the device; LM loop inside the library (camcal_rig_lm), "
                                                "stops on eps = 1e-7, not on the iteration cap"}
        del R, cams
        torch.cuda.empty_cache()
        lib = _lib.load()
        out["C3_fisheye_100k_ragged_views"] = big_solve("C3", 100_000, dev, lib)
        torch.cuda.empty_cache()
        out["C5_fisheye_1M_views"] = big_solve("C5", 1_000_000, dev, lib)
        if with_cpu:
            # the CPU restatement beside every config -- on a bounded sample where the full
            # set would take minutes (the Schur form is linear in the views; cv2's dense form is cubic)
            from oracle import cvcalib as O
            for key, name, nv in (("C3_fisheye_100k_ragged_views", "C3", 2000), ("C5_fisheye_1M_views", "C5", 1000)):
                Ps = synth.config(name, n_views=nv)
                os_, is_ = Ps.as_lists()
                t0 = time.perf_counter()
                r = O.fisheye_calibrate(os_, is_, Ps.image_size, Ps.K_guess, Ps.D_guess, flags=fe, criteria=crit, solver="schur")
                dt = (time.perf_counter() - t0) * 1e3
                out[key]["cpu_oracle_sample"] = {"views": nv, "ms": dt, "iterations": int(r[5]["iters"]),
                                                 "ms_per_1000_views": dt * 1000.0 / nv,
                                                 "what": "oracle/cvcalib.py Schur-eliminated Gauss-Newton, numpy, 1 thread, on a "
                                                         "%d-view sample of the same generator (linear in the views)" % nv}
    except Exception as exc:      # reported, never required
        out["error"] = str(exc)
    if with_cpu:
        try:
            out["cpu_port"] = cpu_port_solves()
        except Exception as exc:
            out["cpu_port"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return out


def sharded_solves(dev, lib, rank, world, views_total):
    """Whole solves with the set sharded over the ranks (N > 1, or one rank under --force-comm): the
    'LM solve time at 1/2/4/8 GPUs' half of BASELINE.json's metric.  C5: views_total fisheye views in
    contiguous ranges, the library's loop with its RCCL communicator (one message per iteration);
    C4: the 4-camera rig with its 10 000 board poses split over the ranks.  Wall clock = max over the
    ranks between two barriers, second of two runs; every rank generates its own shard on the device."""
    from cameracalibration_amd import rig, solver
    from cameracalibration_amd.distributed import shard_ranges, shard_rig_cams
    out = {}

    def wall(fn):
        ms = []
        for _ in range(2):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) * 1e3], dtype=torch.float64, device=dev)
            if world > 1:
                _all_reduce(t, None, dist.ReduceOp.MAX)
            ms.append(float(t.item()))
        return ms, res

    def all_ok(ok):
        """Collective: did the section succeed on EVERY rank?  (A rank that failed before or after its
        collectives must not leave the others to start the next section alone; a rank that dies INSIDE a
        collective is what the watchdog in main() is for.)"""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            _all_reduce(t, None, dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    ok = True
    try:
        lo, hi = shard_ranges(np.full(views_total, 88, dtype=np.int64), world)[rank]
        P = synth.config("C5", n_views=hi - lo, device=dev, seed=500 + rank)
        comm = solver.Comm(enabled=True)

        def c5():
            E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib)
            E.set_intr(0, P.intr_guess())
            E.pose_init(0, 0)
            E.pose_refine(0, 0, 20, 1e-10)
            cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), (3, 1000, 1e-6), comm=comm)
            return cost, it, E.get_intr(0)
        progress("sharded C5: %d views on this rank" % (hi - lo))
        ms, (cost, it, intr) = wall(c5)
        progress("sharded C5: %.2f ms" % ms[1])
        n_total = comm.total(P.n_obs, dev)
        out["C5_fisheye_sharded"] = {"ms": ms[1], "first_run_ms": ms[0], "views_total": views_total, "views_on_rank0": hi - lo,
                                     "iterations": it, "rms_px": float(np.sqrt(cost / n_total)),
                                     "fx_rel_err": float(abs(intr[0] / P.K_true[0] - 1)), "communicator": comm.kind,
                                     "note": "engine set-up, pose initialisation and the damped Gauss-Newton loop on every "
                                             "rank, points already on the device; max over the ranks"}
        del P
        torch.cuda.empty_cache()
    except Exception as exc:      # reported, never required
        ok = False
        out["C5_error"] = "%s: %s" % (type(exc).__name__, exc)
    if not all_ok(ok):
        out.setdefault("C5_error", "failed on another rank")
        return out
    try:
        R = synth.config("C4", device=dev)            # the same rig on every rank (seeded), then this rank's poses
        cams = [dict(pose_idx=R.pose_idx[c], view_ptr=R.view_ptr[c], obj=R.obj[c], img=R.img[c]) for c in range(R.n_cams)]
        mine, n_mine, (plo, phi) = shard_rig_cams(cams, R.n_poses, rank, world)
        Ks, Ds = R.guess_matrices()
        rcomm = solver.Comm(enabled=True)

        def c4():
            info = {}
            r = rig.calibrate_rig(None, None, R.image_size, Ks, Ds, criteria=(3, 100, 1e-7), cams=mine, n_poses=n_mine,
                                  _comm=rcomm, _info=info)
            return r[0], info["iters"]
        progress("sharded C4: %d poses on this rank" % n_mine)
        ms, (rms, it) = wall(c4)
        progress("sharded C4: %.2f ms" % ms[1])
        out["C4_rig_sharded"] = {"ms": ms[1], "first_run_ms": ms[0], "poses_total": R.n_poses, "poses_on_rank0": n_mine,
                                 "iterations": it, "rms_px": float(rms), "communicator": rcomm.kind,
                                 "note": "board poses split over the ranks, [C|d] and the records summed through the "
                                         "communicator inside camcal_rig_lm; max over the ranks"}
    except Exception as exc:      # reported, never required
        ok = False
        out["C4_error"] = "%s: %s" % (type(exc).__name__, exc)
    if not all_ok(ok):
        out.setdefault("C4_error", "failed on another rank")
    return out


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr per stage (rank, seconds since start): a run that takes minutes shows where it is."""
    sys.stderr.write("[bench rank %s +%.1fs] %s\n" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, msg))
    sys.stderr.flush()


def timed_steps(step, steps, world, dev):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; returns (wall seconds, max
    over ranks; HIP-event milliseconds per step on this rank's stream)."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # the library launches on torch's current stream
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    region_ms = ev0.elapsed_time(ev1) / steps
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tmax = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        _all_reduce(tmax, None, dist.ReduceOp.MAX)
    return float(tmax.item()), region_ms


def make_engine(lib, n_views, dev, seed):
    """C5 views resident on this rank, in the state of the solver at its first evaluation: poses
    initialised and refined per view at the guessed intrinsics."""
    P = synth.config("C5", n_views=n_views, device=dev, seed=seed)
    E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib)
    E.set_intr(0, P.intr_guess())
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    n_obs, nv = P.n_obs, P.n_views
    del P
    torch.cuda.synchronize()
    return E, n_obs, nv


def make_step(E, lib, handle):
    import ctypes as C
    n_rec = E.glob_len + E.schur_len

    def step():
        E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
        if handle is not None:         # [U|b|cost|C|d]: one ncclAllReduce on the compute stream
            with torch.cuda.device(E.device):
                lib.check(lib.camcal_comm_allreduce(handle, E.rec.data_ptr(), n_rec, E._stream()), "allreduce")
    return step


def kernel_extras(dev, lib):
    """Bench lines of the other heavy kernels of a solve: the per-view pose
    re-estimation at C5 size and the pinhole evaluation at the same scale."""
    out = {}
    try:
        P = synth.config("C5", n_views=1_000_000, device=dev)
        E = Engine(_lib.FISHEYE, P.view_ptr, P.obj, P.img, dev, lib=lib)
        E.set_intr(0, P.intr_guess())
        n_obs = P.n_obs
        del P
        E.pose_init(0, 0)
        start = E.pose[0].clone()
        ts = []
        for _ in range(3):
            E.pose[0].copy_(start)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            E.pose_refine(0, 0, 20, 1e-10)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(ts))
        steps = float(E.iters.double().sum().item()) / 1_000_000
        flop = 340.0                                   # projection + pose Jacobian + V, g per observation and step
        tf = flop * n_obs * steps / (ms * 1e-3) * 1e-12
        out["pose_refine_first"] = {"ms": ms, "mean_gn_steps_per_view": steps, "obs_steps_per_s": n_obs * steps / (ms * 1e-3),
                                    "flop_per_obs_step_model": flop, "achieved_tflops": tf, "frac_fp64_peak": tf / PEAK_FP64_TFLOPS,
                                    "note": "k_pose_refine from the homography initialisation, 1M views x 88 corners; views need "
                                            "2..20 steps each (cv2's rule per view)"}
        del E, start
        torch.cuda.empty_cache()
        P = synth.config("C5P", n_views=1_000_000, device=dev)
        E = Engine(_lib.PINHOLE, P.view_ptr, P.obj, P.img, dev, lib=lib)
        E.set_intr(0, P.intr_guess())
        n_obs, nv = P.n_obs, P.n_views
        E.pose_init(0, 0)
        E.pose_refine(0, 0, 20, 1e-10)
        for st in (0, 5, 4):
            E.time_stage(st, 20)
        a, b, f = E.time_stage(0, 100), E.time_stage(5, 100), E.time_stage(4, 100)
        flops = FLOP_PER_OBS["pinhole"] * n_obs + FLOP_PER_VIEW * nv
        out["pinhole_evaluation_1M_views"] = {
            "evaluation_A_ms": a, "evaluation_B_ms": b, "single_pass_ms": f,
            "obs_per_s_two_pass": n_obs / ((a + b) * 1e-3), "obs_per_s_single_pass": n_obs / (f * 1e-3),
            "frac_fp64_peak_two_pass": flops / ((a + b) * 1e-3) * 1e-12 / PEAK_FP64_TFLOPS,
            "frac_fp64_peak_single_pass": flops / (f * 1e-3) * 1e-12 / PEAK_FP64_TFLOPS,
            "work_model": "732 flop/obs + 1200 flop/view (BASELINE.md section 3); the LM driver uses A + B",
        }
        # whole LM solve of the same set (CvLevMarq, criteria of camera_calib.py:153-156)
        from cameracalibration_amd import solver
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        err, it = solver.lm_pinhole(E, np.ones(9, dtype=bool), 1000, 1e-6)
        torch.cuda.synchronize()
        intr = E.get_intr(0)
        out["C5p_pinhole_1M_views_lm"] = {"ms": (time.perf_counter() - t0) * 1e3, "iterations": it,
                                         "rms_px": float(np.sqrt(err / n_obs)),
                                         "fx_rel_err": float(abs(intr[0] / P.K_true[0] - 1)),
                                         "note": "LM loop only (poses already initialised), points on the device"}
    except Exception as exc:
        out["error"] = repr(exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults long enough to get past the device's clock ramp: the first ~25 launches after the set-up
    # kernels run up to 17 % slower than the steady state (profiles/README.md, r01_i)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--views", type=int, default=1_000_000, help="views per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="weak: --views per GPU (the headline value); strong: --views in total, sharded; both: "
                         "the line is the weak run and carries the strong run as 'strong_scaling' (N > 1 only)")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal of the N > 1 code path on one GPU: a one-rank nccl group, the step includes the "
                         "library's ncclAllReduce, the strong-scaling run is made too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solves", action="store_true", help="skip the whole-solve wall-clock extras")
    args = ap.parse_args()

    # ONE JSON line on stdout: everything else a library may print there (RCCL's version banner at
    # communicator creation, ...) goes to stderr; the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # CAMCAL_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
    # (the ranks share devices; the records travel through torch.distributed over gloo, not RCCL)
    backend = os.environ.get("CAMCAL_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    multi = world > 1 or args.force_comm
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = _lib.load()
    from cameracalibration_amd import solver
    from cameracalibration_amd.distributed import shard_ranges
    comm = solver.Comm(enabled=multi)
    headline = "strong" if (args.scaling == "strong" and multi) else "weak"
    runs = {}
    # clock ramp: the device needs ~25 launches of this kernel to reach its steady
    # clock, so a short --warmup is topped up to 30 untimed steps in all (reported as "prewarm_steps")
    prewarm = max(0, 30 - args.warmup)
    modes = [headline] + (["strong"] if (args.scaling == "both" and multi) else [])
    for mode in modes:
        if mode == "weak":
            nv_rank = args.views
        else:                          # BASELINE config 5 as written: --views in total, sharded by observation count
            lo, hi = shard_ranges(np.full(args.views, 88, dtype=np.int64), world)[rank]
            nv_rank = hi - lo
        progress("%s scaling: building %d views" % (mode, nv_rank))
        E, n_obs, n_views = make_engine(lib, nv_rank, dev, seed=5 + rank)
        handle = comm.handle(E)
        progress("communicator: %s" % comm.kind)
        step = make_step(E, lib, handle)
        for _ in range(prewarm + args.warmup):
            step()
        progress("warm-up done, timing %d steps" % args.steps)
        el, region_ms = timed_steps(step, args.steps, world, dev)
        progress("timed: %.3f ms/step" % (el / args.steps * 1e3))
        tot = torch.tensor([float(n_obs)], dtype=torch.float64, device=dev)
        if world > 1:
            _all_reduce(tot, None)
        # what every rank held and how long its kernels took: the line must let a reader check the sharding
        mine = torch.tensor([float(n_views), float(n_obs), float(region_ms)], dtype=torch.float64, device=dev)
        if world > 1:
            allr = [torch.zeros_like(mine) for _ in range(world)]
            if backend == "nccl":
                dist.all_gather(allr, mine)
            else:
                hl = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(hl, mine.cpu())
                allr = hl
            per_rank = [dict(rank=r, views=int(t[0].item()), obs=int(t[1].item()), kernel_region_ms=float(t[2].item()))
                        for r, t in enumerate(allr)]
        else:
            per_rank = [dict(rank=0, views=n_views, obs=n_obs, kernel_region_ms=float(region_ms))]
        # ranks the collective of the step really spanned, as RCCL reports it for the library's communicator
        if handle is None:
            nranks = 1
        elif comm.kind == "rccl":
            nranks = int(lib.camcal_comm_size(handle))
        else:
            nranks = None                 # torch.distributed callback: not an RCCL communicator of the library
        if comm.kind == "rccl" and nranks != world:
            raise SystemExit("bench: the RCCL communicator spans %s ranks, --gpus / WORLD_SIZE say %d" % (nranks, world))
        runs[mode] = dict(E=E, n_obs=n_obs, n_views=n_views, el=el, region_ms=region_ms, total_obs=float(tot.item()),
                          per_rank=per_rank, rccl_nranks=nranks, comm_kind=comm.kind,
                          comm={"rccl": "library RCCL communicator: one ncclAllReduce of 81 doubles per step on the "
                                        "compute stream", "torch": "torch.distributed all_reduce (callback)",
                                None: None}[comm.kind])
        if mode != headline:
            del runs[mode]["E"]
            del E
            torch.cuda.empty_cache()
    R = runs[headline]
    E, n_obs, n_views, el, region_ms = R["E"], R["n_obs"], R["n_views"], R["el"], R["region_ms"]

    # per-kernel durations (HIP events on the launch stream), rank 0
    stage_ms = {}
    if rank == 0:
        for st, nm in ((4, "k_normal_fused+finalize (the step)"), (0, "evaluation_A"), (5, "evaluation_B")):
            E.time_stage(st, 3)
            stage_ms[nm] = E.time_stage(st, max(args.steps, 10))
    if world > 1:
        dist.barrier()
    progress("stage timings done")
    out = None
    if rank == 0:
        # the kernel (+ its two tiny reduction launches) over the timed region itself when nothing else
        # runs in it; with an all-reduce in the step, the same launches timed back to back afterwards
        pass_ms = region_ms if world == 1 else stage_ms["k_normal_fused+finalize (the step)"]
        stage_ms["timed_region_per_step"] = region_ms
        flops = FLOP_PER_OBS["fisheye"] * n_obs + FLOP_PER_VIEW * n_views
        bytes_model = BYTES_PER_OBS * n_obs + BYTES_PER_VIEW * n_views
        bytes_moved = MOVED_BYTES_PER_OBS * n_obs + MOVED_BYTES_PER_VIEW * n_views
        ach_tf = flops / (pass_ms * 1e-3) * 1e-12
        ach_gb = bytes_moved / (pass_ms * 1e-3) * 1e-9
        # measured peaks on this device (BASELINE.md section 3 asks for both)
        import ctypes as C
        res = C.c_double(0.0)
        probe = {}
        buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        for kind, key in ((0, "fp64_fma_tflops"), (1, "hbm_read_gbs")):
            rc = lib.camcal_peak_probe(kind, buf.data_ptr(), buf.numel(), 5,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(res))
            probe[key] = float(res.value) if rc == 0 else None
        del buf
        pmc = pmc_profile(n_obs)
        traffic = pmc["traffic"]
        ex_flop = pmc["executed_flop_per_obs"]
        ex_tf = (ex_flop * n_obs / (pass_ms * 1e-3) * 1e-12) if ex_flop else None
        per = "per GPU" if headline == "weak" else f"in total, sharded over {world} GPUs"
        out = {
            "metric": "corner-obs/s for residual+Jacobian+Schur (one normal-equation evaluation)",
            "value": R["total_obs"] * args.steps / el,
            "unit": "corner-obs/s",
            "n_gpus": world, "rccl_nranks": R["rccl_nranks"], "per_rank": R["per_rank"],
            "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": headline, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C5 stress: {args.views} Kannala-Brandt fisheye views x 88 corners {per} "
                                   f"({n_obs} corner observations on rank 0), 8 intrinsics + 6 per view",
                       "views_per_gpu": n_views, "obs_per_gpu": n_obs, "parallelism": f"views sharded x{world}",
                       "collective": R["comm"]},
            "roofline": {
                "bound": "fp64-valu", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": traffic,
                "kernels_ms": stage_ms, "pass_ms": pass_ms,
                "work_model": "605 flop/obs + 1200 flop/view (BASELINE.md section 3) over the single-pass "
                              "kernel k_normal_fused (+ its two tiny reduction launches)",
                "measured_peak_fp64_tflops": probe.get("fp64_fma_tflops"),
                "frac_of_measured_peak": (ach_tf / probe["fp64_fma_tflops"]) if probe.get("fp64_fma_tflops") else None,
                # what the FP64 pipe really executes (PMC: FMA x2 + add/mul; less than the model because of the
                # Hankel / shared-row structure of the fisheye blocks): the busy fraction of the pipe
                "executed_flop_per_obs": ex_flop, "achieved_executed": ex_tf,
                "frac_executed": (ex_tf / PEAK_FP64_TFLOPS) if ex_tf else None,
                "traffic_source": pmc["source"], "traffic_stale": pmc["stale"],
                # why frac_executed stops where it does (committed PMC passes): the one resident wavefront keeps the
                # VALU busy for valu_busy of its cycles with every lane active; fp64_share of the issue slots are
                # FP64 instructions (the rest: v_accvgpr moves of the 112 accumulators that do not fit 128 doubles
                # of architectural VGPRs, address / loop code); DESIGN.md section 4 has the table
                "counters": pmc["counters"],
            },
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gb, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": ach_gb / PEAK_HBM_GBS, "traffic": traffic, "traffic_stale": pmc["stale"],
                             "achieved_from_traffic": (traffic / (pass_ms * 1e-3) * 1e-9) if traffic else None,
                             "bytes_moved": bytes_moved, "bytes_model": bytes_model,
                             "measured_peak_read_gbs": probe.get("hbm_read_gbs"),
                             "work_model": "moved by this instantiation (planar, no block stores): 16 B/obs + 48 B/view "
                                           "pose + one 81-double partial record per 64 views written and re-read; "
                                           "bytes_model is the frozen 20 B/obs + 696 B/view of SURVEY.md 8d "
                                           "(3-D points, blocks stored), which this launch does not move"},
        }
        # a line whose records did not travel over the library's RCCL communicator between GPUs of their own
        # (gloo rehearsal with ranks sharing a device, records staged through the host) is a rehearsal of the
        # control flow, never a measurement: it carries no value
        rehearsal = multi and (backend != "nccl" or R["comm_kind"] != "rccl")
        if rehearsal:
            out["rehearsal"] = True
            out["rehearsal_value"] = out["value"]
            out["value"] = None
        if "strong" in runs and headline != "strong":
            S = runs["strong"]
            out["strong_scaling"] = {"value": None if rehearsal else S["total_obs"] * args.steps / S["el"], "unit": "corner-obs/s",
                                     "ms_per_step": S["el"] / args.steps * 1e3, "views_total": args.views,
                                     "views_on_rank0": S["n_views"], "kernel_region_ms_rank0": S["region_ms"],
                                     "per_rank": S["per_rank"], "rccl_nranks": S["rccl_nranks"],
                                     "note": "BASELINE config 5 as written: the 1M-view set sharded over the ranks "
                                             "(shard_ranges), same step, same collective"}
    # whole solves with the set sharded over the ranks: every rank takes part.  They are extras of the line:
    # the headline above is complete before they start, and a deadline makes sure it is written even if a
    # rank dies inside a collective and leaves the others waiting -- then every rank ends with
    # status 0 after rank 0 has written the line with the failure noted in it.
    sharded = None
    if multi and not args.no_solves:
        import threading
        deadline = float(os.environ.get("CAMCAL_BENCH_WATCHDOG", "600"))

        def expired():
            if rank == 0:
                out["solve_wall_clock_sharded"] = {"error": "no result within %.0f s: a rank failed inside a "
                                                            "collective or the solves hung" % deadline}
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            progress("sharded solves: deadline of %.0f s passed, giving up" % deadline)
            os._exit(0)
        timer = threading.Timer(deadline + (0.0 if rank == 0 else 10.0), expired)
        timer.daemon = True
        timer.start()
        try:
            sharded = sharded_solves(dev, lib, rank, world, args.views)
        finally:
            timer.cancel()
    progress("sharded solves done" if sharded is not None else "no sharded solves")
    if rank == 0:
        if sharded is not None:
            out["solve_wall_clock_sharded"] = sharded
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_solves:
            del E, R, runs
            torch.cuda.empty_cache()
            out["solve_wall_clock"] = solve_times(dev, with_cpu=not args.no_cpu_baseline)
            torch.cuda.empty_cache()
            out["kernels"] = kernel_extras(dev, lib)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        solver.destroy_rccl_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
