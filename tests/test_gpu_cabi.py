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
    off = [x.copy() for x in o]
    off[3][5, 0, 2] = 1.0                       # one corner off the board's plane: refused, on the host or on the device
    try:
        fn(off, i, P.image_size, P.K_guess.copy(), P.D_guess.copy(), flags=fl, criteria=(3, 1000, 1e-6))
        raise SystemExit("a non-planar target was accepted")
    except cc.error as e:
        assert "planar" in str(e).lower(), str(e)
np.save(sys.argv[1], np.concatenate(out))
"""


@pytest.mark.gpu
def test_one_call_entry_on_its_fallback_transfer_paths(tmp_path):
    """camcal_calibrate without pinned staging (CAMCAL_PIN_LIMIT=0: the copies go from / to the caller's arrays, the
    path of sets above 8 MB) and without the kernel-written notify buffer (CAMCAL_NO_NOTIFY: copy + event read-backs):
    the same results to the bit as the default paths -- only the transfers differ.  Likewise with the planarity test
    of the target moved onto the device (CAMCAL_PLANAR_ON_DEVICE_MIN=0: what sets of a million points and more do);
    and with the point arrays uploaded the way large ones are (several threads through page-locked staging; forced here by
    a 1 KB chunk; and started before the host lays the views out, CAMCAL_EARLY_UPLOAD_MIN=0); every variant refuses a board with one corner off its plane."""
    import numpy as np
    res = {}
    for tag, env in (("default", {}), ("direct", {"CAMCAL_PIN_LIMIT": "0"}), ("copies", {"CAMCAL_NO_NOTIFY": "1"}),
                     ("both", {"CAMCAL_PIN_LIMIT": "0", "CAMCAL_NO_NOTIFY": "1"}),
                     ("planarity on the device", {"CAMCAL_PLANAR_ON_DEVICE_MIN": "0"}),
                     ("threaded upload", {"CAMCAL_PIN_LIMIT": "0", "CAMCAL_BULK_CHUNK": "1024"}),
                     ("early threaded upload", {"CAMCAL_PIN_LIMIT": "0", "CAMCAL_BULK_CHUNK": "1024", "CAMCAL_EARLY_UPLOAD_MIN": "0"})):
        f = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([os.sys.executable, "-c", _FALLBACK_SCRIPT % ROOT, f], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    for tag in ("direct", "copies", "both", "planarity on the device", "threaded upload", "early threaded upload"):
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
def test_randomised_mid_size_calls_grouped_kernels_equal_the_large_set_kernels():
    """tools/gpu_fuzz_mid.py: 16 random sets of 1500 ... 40 000 views (both models, full and ragged boards) through the
    one-call entry from host arrays (host layout; 64 / 16 lanes per view or one, folded or separate stages, by size)
    against the engine solve forced to one lane per view (the large-set kernels; device layout from 8192 views on):
    equal iteration counts, rms to 1e-9, K to 1e-8, poses to 1e-6 (measured: round-off, 1e-13)."""
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz_mid.py"), "7", "16"],
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


@pytest.mark.gpu
def test_one_call_entry_on_a_large_set_of_host_arrays():
    """camcal_calibrate on 80 000 ragged fisheye views handed over as host arrays -- the paths only large sets
    take: planarity tested on the device behind the upload, point arrays uploaded by several threads through
    page-locked staging, results brought into the caller's view order by several threads -- against the
    stage-by-stage engine solve of the same set (device layout, same library loop)."""
    import numpy as np
    import torch
    import cameracalibration_amd as cc
    from cameracalibration_amd import api, solver, synth
    from cameracalibration_amd.engine import Engine
    cc.set_device("cuda:0")
    dev = torch.device("cuda:0")
    P = synth.config("C3", n_views=80000)
    vp = np.ascontiguousarray(np.asarray(P.view_ptr), dtype=np.int64)
    obj = np.ascontiguousarray(np.asarray(P.obj), dtype=np.float32)
    img = np.ascontiguousarray(np.asarray(P.img), dtype=np.float32)
    assert obj.nbytes > 4 * (8 << 20) and vp[-1] > (1 << 20)          # large enough for the threaded upload and the device test
    crit = (3, 1000, 1e-6)
    K9 = np.ascontiguousarray(P.K_guess, dtype=np.float64).ravel().copy()
    D5 = np.zeros(5)
    D5[:4] = np.asarray(P.D_guess).ravel()[:4]
    rms, K9, D5, rt, its = api._one_call(1, vp, obj, img, P.image_size, K9, D5, 4, 8 | 2 | 1, crit)
    E = Engine(1, vp, obj, img, dev)
    E.set_intr(0, P.intr_guess())
    E.pose_init(0, 0)
    E.pose_refine(0, 0, 20, 1e-10)
    cost, it = solver.gn_fisheye(E, np.ones(8, dtype=bool), crit)
    intr = E.get_intr(0)
    assert its == it and abs(rms - np.sqrt(cost / vp[-1])) < 1e-12 * rms
    assert np.allclose([K9[0], K9[4], K9[2], K9[5]], intr[:4], rtol=1e-12, atol=0) and np.allclose(D5[:4], intr[4:8], rtol=1e-10, atol=1e-14)
    assert np.allclose(rt, E.get_poses(0), rtol=1e-9, atol=1e-12)
    off = obj.copy()
    off[-1, 2] = 0.25                                                 # the last point of the last view leaves the plane
    with pytest.raises(cc.error):
        try:
            api._one_call(1, vp, off, img, P.image_size, K9.copy(), D5.copy(), 4, 8 | 2 | 1, crit)
        except Exception as e:      # noqa: BLE001 -- _one_call raises the package's own error types
            assert "planar" in str(e).lower(), str(e)
            raise
