#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: seeded synthetic inputs (float32, exactly as
handed to the solver) and the outputs of the cv2-faithful dense oracle
(oracle/cvcalib.py, solver="dense").

PARITY UNPINNED: cv2 is not installable here, and the reference holds no
fixtures for this path, so these vectors pin the build against its own
restatement of OpenCV's algorithm (plus the ground truth they were generated
from), not against OpenCV's binaries.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cameracalibration_amd import synth  # noqa: E402
from oracle import cvcalib as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FE = O.FISHEYE_FIX_SKEW | O.FISHEYE_RECOMPUTE_EXTRINSIC | O.FISHEYE_USE_INTRINSIC_GUESS


def save(name, P, flags, criteria, res):
    np.savez_compressed(
        os.path.join(HERE, name), model=P.model, image_size=np.array(P.image_size), view_ptr=P.view_ptr,
        obj=P.obj.numpy(), img=P.img.numpy(), K_guess=P.K_guess, D_guess=P.D_guess,
        K_true=np.array(P.K_true), D_true=np.array(P.D_true), poses_true=P.poses_true.numpy(),
        flags=flags, criteria=np.array(criteria, dtype=np.float64),
        rms=res[0], K=res[1], D=res[2], rvecs=res[3], tvecs=res[4], iters=res[5]["iters"])
    print(name, "rms", res[0], "iters", res[5]["iters"])


def main():
    crit = (3, 1000, 1e-6)
    P = synth.config("C1")
    o, i = P.as_lists()
    save("c1_pinhole_20x54.npz", P, 1, crit,
         O.calibrate_camera(o, i, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=crit, solver="dense"))
    P = synth.config("C2", n_views=96)
    o, i = P.as_lists()
    save("c2_fisheye_96x88.npz", P, FE, crit,
         O.fisheye_calibrate(o, i, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=crit, solver="dense"))
    P = synth.config("C3", n_views=80)
    o, i = P.as_lists()
    save("c3_ragged_80.npz", P, FE, crit,
         O.fisheye_calibrate(o, i, P.image_size, P.K_guess, P.D_guess, flags=FE, criteria=crit, solver="dense"))
    crit10 = (3, 10, 1e-6)
    P = synth.config("C1", seed=11)
    o, i = P.as_lists()
    save("c1_pinhole_refine10.npz", P, 1, crit10,
         O.calibrate_camera(o, i, P.image_size, P.K_guess, P.D_guess, flags=1, criteria=crit10, solver="dense"))
    make_rig()


def make_rig():
    """Config C4 at oracle size: 3 fisheye cameras x 12 board poses, dense
    joint CvLevMarq (oracle/rig.py).  eps 1e-7: further iterations only move at
    rounding level, where accept / reject decisions are not reproducible."""
    from oracle import rig as OR
    R = synth.make_rig(12, 3, board=synth.board_points(6, 5, 40.0), seed=21)
    objs, imgs = R.as_lists()
    rms, intr, ext, poses, info = OR.rig_calibrate("fisheye", objs, imgs, R.intr_guess, max_iter=40, eps=1e-7)
    os.makedirs(os.path.join(HERE, "rig"), exist_ok=True)
    np.savez_compressed(
        os.path.join(HERE, "rig", "rig_fisheye_3x12.npz"), image_size=np.array(R.image_size), n_poses=R.n_poses,
        **{f"pose_idx{c}": R.pose_idx[c] for c in range(3)}, **{f"view_ptr{c}": R.view_ptr[c] for c in range(3)},
        **{f"obj{c}": R.obj[c].numpy() for c in range(3)}, **{f"img{c}": R.img[c].numpy() for c in range(3)},
        intr_guess=R.intr_guess, intr_true=R.intr_true, ext_true=R.ext_true, poses_true=R.poses_true,
        criteria=np.array([3, 40, 1e-7]), rms=rms, intr=intr, ext=ext, poses=poses, iters=info["iters"])
    print("rig_fisheye_3x12.npz rms", rms, "iters", info["iters"])


if __name__ == "__main__":
    main()
