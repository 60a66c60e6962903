#!/usr/bin/env python3
"""HIP-event timings of the point-level kernels through the C ABI (development tool;
DESIGN.md section 10 quotes its output)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cameracalibration_amd import _lib, synth
from cameracalibration_amd.engine import Engine

dev = torch.device("cuda:0")
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def line(name, ms, nbytes, units, unit_name):
    print(f"{name:44s} {ms:8.4f} ms  {nbytes / ms * 1e-6:8.1f} GB/s  {units / ms * 1e-6:9.2f} G{unit_name}/s", flush=True)


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)


fish = np.zeros(16)
fish[:8] = [1066.58, 1066.31, 1478.65, 791.19, -0.02334, 0.007086, -0.005574, 9.94e-4]
pin = np.zeros(16); pin[:9] = [1365.77, 1367.35, 959.51, 556.16, -0.4096, 0.2351, -1.7e-4, -9.5e-4, -0.0869]
t_fish, t_pin = dv(fish), dv(pin)

# rectify maps + remap
for model, name, intr, (w, h) in ((1, "fisheye 2960x1664", t_fish, (2960, 1664)), (0, "pinhole 1920x1080", t_pin, (1920, 1080))):
    K = np.array([[intr[0].item(), 0, intr[2].item()], [0, intr[1].item(), intr[3].item()], [0, 0, 1]])
    iR = dv(np.linalg.inv(K).reshape(-1))
    m1f = torch.empty(h, w, dtype=torch.float32, device=dev); m2f = torch.empty_like(m1f)
    m1s = torch.empty(h, w, 2, dtype=torch.int16, device=dev); m2s = torch.empty(h, w, dtype=torch.int16, device=dev)
    px = w * h
    ms = timed(lambda: lib.check(lib.camcal_undistort_rectify_map(model, w, h, intr.data_ptr(), iR.data_ptr(), 0, m1f.data_ptr(), m2f.data_ptr(), st), "map"))
    line(f"rectify map {name} CV_32FC1", ms, px * 8, px, "px")
    ms = timed(lambda: lib.check(lib.camcal_undistort_rectify_map(model, w, h, intr.data_ptr(), iR.data_ptr(), 1, m1s.data_ptr(), m2s.data_ptr(), st), "map"))
    line(f"rectify map {name} CV_16SC2", ms, px * 6, px, "px")
    for cn in (1, 3):
        src = torch.randint(0, 256, (h, w, cn), dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        for mt, a, b, mb in ((0, m1f, m2f, 8), (1, m1s, m2s, 6)):
            ms = timed(lambda: lib.check(lib.camcal_remap_u8(src.data_ptr(), w, h, w * cn, cn, a.data_ptr(), b.data_ptr(), mt, w, h, dst.data_ptr(), w * cn, 0, st), "remap"))
            line(f"remap {name} x{cn} {'float' if mt == 0 else 'fixed'} maps", ms, px * (mb + 2 * cn), px, "px")

# projection / undistortion of 16M points
n = 1 << 24
g = torch.Generator(device="cpu").manual_seed(1)
xyz = torch.rand(n, 3, generator=g, dtype=torch.float64) * torch.tensor([400.0, 300.0, 0.0]) - torch.tensor([200.0, 150.0, 0.0])
xyz = xyz.to(dev)
pose = dv([0.1, -0.2, 0.05, 10.0, -5.0, 900.0])
uv = torch.empty(n, 2, dtype=torch.float64, device=dev)
out = torch.empty_like(uv)
for model, name, intr in ((0, "pinhole", t_pin), (1, "fisheye", t_fish)):
    ms = timed(lambda: lib.check(lib.camcal_project_points(model, n, xyz.data_ptr(), pose.data_ptr(), intr.data_ptr(), uv.data_ptr(), st), "proj"))
    line(f"projectPoints {name} 16.8M points", ms, n * 40, n, "pt")
    ms = timed(lambda: lib.check(lib.camcal_undistort_points(model, n, uv.data_ptr(), intr.data_ptr(), None, out.data_ptr(), st), "und"))
    line(f"undistortPoints {name} 16.8M points", ms, n * 32, n, "pt")

# per-view reprojection error of C5
P = synth.config("C5", device=dev)
E = Engine(1, P.view_ptr, P.obj, P.img, dev)
E.set_intr(0, P.intr_guess())
E.pose_init(0, 0)
torch.cuda.synchronize()
out2 = torch.zeros(2, dtype=torch.float64, device=dev)
ms = timed(lambda: lib.check(lib.camcal_reproj_err(1, E._tp(), E.intr[0].data_ptr(), E.pose[0].data_ptr(), 1, E.view_cost.data_ptr(),
                                                   out2.data_ptr(), E.scratch.data_ptr(), st), "reproj"), reps=10)
line("reprojection error C5 (1M views x 88)", ms, P.n_obs * 16 + 56 * P.n_views, P.n_obs, "obs")
ms = timed(lambda: E.pose_init(0, 0), reps=5)
line("pose initialisation C5", ms, P.n_obs * 16, P.n_obs, "obs")
