"""cameracalibration_amd -- MI355X-native replacement for the calibration
solver calls of huangqi2002/CameraCalibration.

Use it where the reference uses cv2 for calibration::

    import cameracalibration_amd as cc
    rms, K, D, rvecs, tvecs = cc.calibrateCamera(obj, img, size, K, D,
        flags=cc.CALIB_USE_INTRINSIC_GUESS,
        criteria=(cc.TERM_CRITERIA_EPS + cc.TERM_CRITERIA_COUNT, 1000, 1e-6))
    rms, K, D, rvecs, tvecs = cc.fisheye.calibrate(obj, img, size, K, D,
        flags=cc.fisheye.CALIB_FIX_SKEW | cc.fisheye.CALIB_RECOMPUTE_EXTRINSIC
              | cc.CALIB_USE_INTRINSIC_GUESS, criteria=...)

``cc.CalibrationSession`` is the same pair of calls for a caller whose view lists
grow from call to call (the capture loops of intrinsicCalib.py:538-564): it keeps the
packed views and the poses on the device between calls.

``cc.calibrate_rig`` (rig.py) is the joint multi-camera solve of BASELINE.json
config 4, which the reference does not have.

(reference call sites: server/aruco/camera_calib.py:74-77,153-156 and
server/internal/IntrinsicCalibration/intrinsicCalib.py:102-105,131-134,194-197,223-226).
"""
from .api import (  # noqa: F401
    calibrateCamera, calibrate_many, error, set_device, CalibrationSession,
    TERM_CRITERIA_COUNT, TERM_CRITERIA_MAX_ITER, TERM_CRITERIA_EPS,
    CALIB_USE_INTRINSIC_GUESS, CALIB_FIX_ASPECT_RATIO, CALIB_FIX_PRINCIPAL_POINT,
    CALIB_ZERO_TANGENT_DIST, CALIB_FIX_FOCAL_LENGTH, CALIB_FIX_K1, CALIB_FIX_K2, CALIB_FIX_K3,
    CALIB_FIX_K4, CALIB_FIX_K5, CALIB_FIX_K6, CALIB_RATIONAL_MODEL, CALIB_THIN_PRISM_MODEL,
    CALIB_FIX_S1_S2_S3_S4, CALIB_TILTED_MODEL, CALIB_FIX_TAUX_TAUY, CALIB_FIX_TANGENT_DIST,
)
from .points import (  # noqa: F401
    projectPoints, norm, checkRange, undistortPoints, solvePnP, solvePnP_batch, initUndistortRectifyMap,
    Rodrigues, reproj_errors, remap, INTER_LINEAR, BORDER_CONSTANT, NORM_L2, CV_16SC2, CV_32FC1, SOLVEPNP_ITERATIVE,
)
from . import fisheye  # noqa: F401
from .rig import calibrate_rig  # noqa: F401

__all__ = ["calibrateCamera", "fisheye", "calibrate_rig", "calibrate_many", "CalibrationSession", "error", "set_device"]
