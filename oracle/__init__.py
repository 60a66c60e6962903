"""CPU oracle for the calibration hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``cameracalibration_amd/`` may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / reported baseline, never as the product.

PARITY UNPINNED: the arithmetic being restated lives in the third-party
dependency ``opencv-contrib-python==4.10.0.84`` (reference
``requirements.txt:11``), which is neither vendored under ``/root/reference``
nor installable here (``import cv2`` -> ModuleNotFoundError, no network).  The
reference holds no golden vectors or tests for this path.
The restatement therefore follows OpenCV 4.10's *published* algorithm
(``calibration.cpp``, ``compat_ptsetreg.cpp``, ``fisheye.cpp``) as recalled in
SURVEY.md section 8(c)-appendix, anchored on the reference call sites

* ``server/aruco/camera_calib.py:74-77``   cv2.fisheye.calibrate (GUI path)
* ``server/aruco/camera_calib.py:153-156`` cv2.calibrateCamera   (GUI path)
* ``server/internal/IntrinsicCalibration/intrinsicCalib.py:102-105,131-134``
* ``server/internal/IntrinsicCalibration/intrinsicCalib.py:194-197,223-226``

and pinned by build-created evidence: synthetic ground truth, analytic-vs-
central-difference Jacobians and ``scipy.optimize.least_squares`` on the same
residual (tests/test_oracle_*.py; fixtures under tests/golden/).
"""
