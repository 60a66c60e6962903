"""The C ABI without Python in the loop: examples/cabi_fisheye (built by `make
examples`, links libcamcal.so, uses only include/camcal.h and the HIP runtime)
calibrates a synthetic fisheye camera and checks the recovered intrinsics."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_host_program_calibrates_through_the_abi():
    exe = os.path.join(ROOT, "examples", "cabi_fisheye")
    assert os.path.exists(exe), "run `make examples` (the driver's build() does)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")


@pytest.mark.gpu
def test_c_host_program_calibrates_a_rig_through_the_abi():
    """The multi-camera rig (BASELINE config 4) from include/camcal.h alone: camcal_rig_init + camcal_rig_lm."""
    exe = os.path.join(ROOT, "examples", "cabi_rig")
    assert os.path.exists(exe), "run `make examples` (the driver's build() does)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")


_FALLBACK_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import cameracalibration_amd as cc
from cameracalibration_amd import synth
out = []
for name, fn, fl in (("C2", cc.fisheye.calibrate, 8 | 2 | 1), ("C1", cc.calibrateCamera, 1)):
    P = synth.config(name, n_views=60 if name == "C2" else 20)
    o, i = P.as_lists()
    r = fn(o, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 1000, 1e-6))
    out.append(np.concatenate([[r[0]], r[1].ravel(), np.asarray(r[2]).ravel(), np.hstack(r[3]).ravel(), np.hstack(r[4]).ravel()]))
np.save(sys.argv[1], np.concatenate(out))
"""


@pytest.mark.gpu
def test_one_call_entry_on_its_fallback_transfer_paths(tmp_path):
    """camcal_calibrate without pinned staging (CAMCAL_PIN_LIMIT=0: the copies go from / to the caller's arrays, the
    path of sets above 8 MB) and without the kernel-written notify buffer (CAMCAL_NO_NOTIFY: copy + event read-backs):
    the same results to the bit as the default paths -- only the transfers differ."""
    import numpy as np
    res = {}
    for tag, env in (("default", {}), ("direct", {"CAMCAL_PIN_LIMIT": "0"}), ("copies", {"CAMCAL_NO_NOTIFY": "1"}),
                     ("both", {"CAMCAL_PIN_LIMIT": "0", "CAMCAL_NO_NOTIFY": "1"})):
        f = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([os.sys.executable, "-c", _FALLBACK_SCRIPT % ROOT, f], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    for tag in ("direct", "copies", "both"):
        assert np.array_equal(res[tag], res["default"]), tag


@pytest.mark.gpu
def test_randomised_small_calls_one_call_path_equals_the_stage_loop():
    """tools/gpu_fuzz_small.py: 80 random production-sized calls (view counts 4 ... 300, ragged corner counts, random FIX_*
    flags and iteration caps, both models) through the one-call path (folded loops, run-ahead half step, notify buffer)
    against the stage-by-stage Python loop: rms and K to 1e-8 relative, rotation vectors to 1e-6."""
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz_small.py"), "11", "80"],
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0


@pytest.mark.gpu
def test_device_layout_equals_the_host_layout():
    """camcal_layout_device (sort, prefix sums, planarity and offset checks on the device) writes the tables
    camcal_layout / engine.make_layout write, element for element, ties included."""
    import torch
    from cameracalibration_amd import _lib
    from tests.test_abi import check_device_layout
    lib = _lib.load()
    assert not lib.is_emulator
    dev = torch.device("cuda:0")
    to_dev = lambda a: torch.from_numpy(a).to(dev)
    to_host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    check_device_layout(lib, to_dev, to_host, (0, 1, 63, 64, 65, 1000, 8191, 8192, 100000, 1000001))
