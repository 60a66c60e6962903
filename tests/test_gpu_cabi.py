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
