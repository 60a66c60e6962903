"""``cv2.fisheye`` namespace stand-in: ``calibrate`` and its flag constants
(reference: server/aruco/camera_calib.py:74-77), ``projectPoints`` (:97),
``undistortPoints`` (:241) and ``initUndistortRectifyMap``
(server/video/video_server.py:473)."""
from .api import fisheye_calibrate as calibrate  # noqa: F401
from .points import fisheye_projectPoints as projectPoints  # noqa: F401
from .points import fisheye_undistortPoints as undistortPoints  # noqa: F401
from .points import fisheye_initUndistortRectifyMap as initUndistortRectifyMap  # noqa: F401
from .api import (  # noqa: F401
    FISHEYE_CALIB_USE_INTRINSIC_GUESS as CALIB_USE_INTRINSIC_GUESS,
    FISHEYE_CALIB_RECOMPUTE_EXTRINSIC as CALIB_RECOMPUTE_EXTRINSIC,
    FISHEYE_CALIB_CHECK_COND as CALIB_CHECK_COND,
    FISHEYE_CALIB_FIX_SKEW as CALIB_FIX_SKEW,
    FISHEYE_CALIB_FIX_K1 as CALIB_FIX_K1,
    FISHEYE_CALIB_FIX_K2 as CALIB_FIX_K2,
    FISHEYE_CALIB_FIX_K3 as CALIB_FIX_K3,
    FISHEYE_CALIB_FIX_K4 as CALIB_FIX_K4,
    FISHEYE_CALIB_FIX_INTRINSIC as CALIB_FIX_INTRINSIC,
    FISHEYE_CALIB_FIX_PRINCIPAL_POINT as CALIB_FIX_PRINCIPAL_POINT,
    FISHEYE_CALIB_ZERO_DISPARITY as CALIB_ZERO_DISPARITY,
    FISHEYE_CALIB_FIX_FOCAL_LENGTH as CALIB_FIX_FOCAL_LENGTH,
)
