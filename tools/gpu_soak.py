"""Soak run of the production-sized drop-in calls: the same handful of problems over and over for a given time,
every result compared bit for bit with the first one of its problem (the loops are deterministic: fixed reduction
order, the atomic counter only elects the workgroup that finishes a sum).  A lost notification, a stale landing
buffer or a race between the speculative launches and the host's verdict would show as a mismatch or an error.
Usage: tools/gpu_soak.py [seconds=120] [threads=1]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cc.set_device("cuda:0")
FE = cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC | cc.CALIB_USE_INTRINSIC_GUESS
CRIT = (3, 100, 1e-6)


def problems():
    out = []
    for name, nv, ragged in (("C1", 20, False), ("C2", 500, False), ("C2", 25, True), ("C1", 25, True), ("C2", 137, True),
                             ("C1", 300, False), ("C2", 2500, False)):
        P = synth.config(name, n_views=nv, seed=nv)
        objs, imgs = P.as_lists()
        if ragged:
            rng = np.random.default_rng(nv)
            keep = [np.sort(rng.choice(len(o), rng.integers(13, len(o) + 1), replace=False)) for o in objs]
            objs = [o[k] for o, k in zip(objs, keep)]
            imgs = [i[k] for i, k in zip(imgs, keep)]
        fisheye = P.model == "fisheye"
        out.append((fisheye, objs, imgs, P.image_size, P.K_guess, P.D_guess))
    return out


def call(p):
    fisheye, objs, imgs, size, K, D = p
    if fisheye:
        r = cc.fisheye.calibrate(objs, imgs, size, K.copy(), D.copy(), flags=FE, criteria=CRIT)
    else:
        r = cc.calibrateCamera(objs, imgs, size, K.copy(), D.copy(), flags=cc.CALIB_USE_INTRINSIC_GUESS, criteria=CRIT)
    return np.concatenate([[r[0]], r[1].ravel(), r[2].ravel(), np.hstack(r[3]).ravel(), np.hstack(r[4]).ravel()])


probs = problems()
first = [call(p) for p in probs]
stats = dict(calls=0, mismatches=0, errors=0)
lock = threading.Lock()
t_end = time.perf_counter() + seconds


def worker(tid):
    rng = np.random.default_rng(tid)
    while time.perf_counter() < t_end:
        i = int(rng.integers(len(probs)))
        try:
            ok = np.array_equal(call(probs[i]), first[i])
            err = 0
        except Exception as e:             # noqa: BLE001 -- counted and shown
            ok, err = True, 1
            print("error in problem %d: %r" % (i, e), flush=True)
        with lock:
            stats["calls"] += 1
            stats["mismatches"] += 0 if ok else 1
            stats["errors"] += err


ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
[t.start() for t in ths]
while any(t.is_alive() for t in ths):
    time.sleep(15)
    with lock:
        print("%.0f s left: %s" % (max(t_end - time.perf_counter(), 0), stats), flush=True)
[t.join() for t in ths]
print("soak done (%d thread(s), %.0f s): %s" % (n_threads, seconds, stats), flush=True)
sys.exit(1 if stats["mismatches"] or stats["errors"] else 0)
