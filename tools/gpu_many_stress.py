"""calibrate_many under load (development tool): random batches of 1 ... 9 production-sized problems, repeated,
each result compared bit for bit with the same problem solved alone; also from several Python threads at once."""
import os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from cameracalibration_amd import synth

rng = np.random.default_rng(5)
pool = []
for k in range(12):
    fisheye = k % 2 == 0
    name = ("C3" if k % 4 == 0 else "C2") if fisheye else ("C1R" if k % 4 == 1 else "C1")
    P = synth.config(name, n_views=int(rng.choice([5, 12, 25, 40, 70])), seed=100 + k)
    o, i = P.as_lists()
    call = dict(model="fisheye" if fisheye else "pinhole", objectPoints=o, imagePoints=i, imageSize=P.image_size,
                cameraMatrix=P.K_guess, distCoeffs=P.D_guess, flags=(8 | 2 | 1) if fisheye else 1, criteria=(3, 1000, 1e-6))
    fn = cc.fisheye.calibrate if fisheye else cc.calibrateCamera
    ref = fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=call["flags"], criteria=call["criteria"])
    pool.append((call, ref))


def fresh(c):
    return dict(c, cameraMatrix=c["cameraMatrix"].copy(), distCoeffs=c["distCoeffs"].copy())


def same(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and \
        np.array_equal(np.hstack(a[3]), np.hstack(b[3])) and np.array_equal(np.hstack(a[4]), np.hstack(b[4]))


bad = [0]


def worker(seed, rounds):
    r = np.random.default_rng(seed)
    for _ in range(rounds):
        idx = r.integers(0, len(pool), size=int(r.integers(1, 10)))
        out = cc.calibrate_many([fresh(pool[j][0]) for j in idx])
        for j, a in zip(idx, out):
            if not same(a, pool[j][1]):
                bad[0] += 1


worker(1, 150)
print("single thread: 150 batches, mismatches", bad[0], flush=True)
ts = [threading.Thread(target=worker, args=(10 + k, 60)) for k in range(4)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print("four Python threads x 60 batches, mismatches", bad[0], flush=True)
sys.exit(1 if bad[0] else 0)
