"""Host-side mirrors of the reference's two calibration wrappers, running on the
device library:

* :class:`Camera_Cali`, :class:`Calib_Data`, :func:`split_points`
  <- server/aruco/camera_calib.py:9-18,21-41,44-310
* :class:`Fisheye`, :class:`Normal`, :class:`CalibData`
  <- server/internal/IntrinsicCalibration/intrinsicCalib.py:59-70,73-161,164-253

Same names, argument meaning, return objects and gates, including the quirks
the callers depend on:

* the fisheye paths hand the solver their ``init_camera_mat`` / ``init_dist_coeff``
  arrays themselves (camera_calib.py:73, intrinsicCalib.py:99-100), the solver
  writes its result into them, so the +-15 % gate compares the result with
  itself and the "initial" values drift from call to call; the pinhole path of
  ``Camera_Cali`` copies (:145), that of ``Normal`` does not (:191-192);
* the hold-out error of ``calib_in_fisheye`` is evaluated with ``self.init_*``
  (:98-99), i.e. the aliased result;
* the acceptance number is ``||proj - corners||_2 / n`` per view (not an RMS),
  averaged over views.

The per-view loops of the reference (projectPoints + norm per view) are one
kernel launch here (:func:`points.reproj_errors`); results are identical to
calling :func:`points.projectPoints` and :func:`points.norm` view by view.
Settings the reference reads from its global config (``m_global``) or argparse
are constructor arguments.
"""
from __future__ import annotations

import copy

import numpy as np

from . import api, points
from . import fisheye as _fe

_CRIT = api.TERM_CRITERIA_EPS + api.TERM_CRITERIA_COUNT
_FE_FLAGS = _fe.CALIB_FIX_SKEW | _fe.CALIB_RECOMPUTE_EXTRINSIC | api.CALIB_USE_INTRINSIC_GUESS

FISHEYE_INIT_K = ((1062.267560, 0.0, 1501.057278), (0.0, 1062.212564, 867.887492), (0.0, 0.0, 1.0))
FISHEYE_INIT_D = ((-0.018123981144576858,), (-0.0026101267970621823,), (0.0006215782316954739,),
                  (-0.00030905667520437564,))
NORMAL_INIT_K = ((1345.159861, 0.0, 973.776987), (0.0, 1344.626518, 541.407299), (0.0, 0.0, 1.0))
NORMAL_INIT_D = ((-0.404256259818931,), (0.21922420928359876,), (-0.00016181802429342013,),
                 (-7.126641144546051e-05,), (-0.06924570764420157,))


def _solve(kind, obj, img, frame_size, K, D, max_iter, session=None):
    """The reference's solver call; ``session``: an api.CalibrationSession of a growing-list caller."""
    if kind == "fisheye":
        fn = session.calibrate if session is not None else _fe.calibrate
        return fn(obj, img, frame_size, K, D, flags=_FE_FLAGS, criteria=(_CRIT, max_iter, 1e-6))
    fn = session.calibrateCamera if session is not None else api.calibrateCamera
    return fn(obj, img, frame_size, K, D, flags=api.CALIB_USE_INTRINSIC_GUESS, criteria=(_CRIT, max_iter, 1e-6))


def _within_gates(K, D, K_nominal, D_nominal, rel):
    """+-rel on every camera-matrix entry against the nominal, |D| <= 1."""
    assert K.shape == K_nominal.shape, "matrix shapes differ"
    assert D.shape == D_nominal.shape, "matrix shapes differ"
    return bool(np.all(np.abs(K - K_nominal) <= rel * np.abs(K_nominal)) and np.all(np.abs(D) <= 1.0))


# ---------------------------------------------------------------------------
# server/aruco/camera_calib.py
class Calib_Data:
    def __init__(self):
        self.type = None
        self.camera_mat = None
        self.dist_coeff = None
        self.rvecs = None
        self.tvecs = None
        self.reproj_err = None
        self.ok = False
        self.check = True


def split_points(obj_point_list, img_point_list, test_size=0.2, random_state=None):
    """Per-view hold-out split: numpy's global generator is seeded once (if
    ``random_state`` is given), then every view shuffles ``arange(n)`` and gives
    its first ``int(n * test_size)`` points to the test set."""
    if random_state is not None:
        np.random.seed(random_state)
    parts = ([], [], [], [])
    for o, i in zip(obj_point_list, img_point_list):
        order = np.arange(len(o))
        np.random.shuffle(order)
        cut = int(len(o) * test_size)
        test, train = order[:cut], order[cut:]
        for dst, src in zip(parts, (o[train], o[test], i[train], i[test])):
            dst.append(src)
    return parts


class Camera_Cali:
    def __init__(self, similar_threshold=0.15, board_w=9, board_h=6, board_size=50.0):
        self.similar_threshold = similar_threshold
        self.board_w, self.board_h, self.board_size = board_w, board_h, board_size
        self.init_camera_mat = np.array(FISHEYE_INIT_K)
        self.init_dist_coeff = np.array(FISHEYE_INIT_D)
        self.init_camera_mat_normal = np.array(NORMAL_INIT_K)
        self.init_dist_coeff_normal = np.array(NORMAL_INIT_D)

    def calib_in(self, obj_point_list, img_point_list, frame_size, camera_type):
        if camera_type == "fisheye":
            return self.calib_in_fisheye(obj_point_list, img_point_list, frame_size)
        return self.calib_in_normal(obj_point_list, img_point_list, frame_size)

    def _calib_in(self, kind, obj_point_list, img_point_list, frame_size, K0, D0, K_eval, D_eval, K_nom, D_nom):
        data = Calib_Data()
        if len(obj_point_list) <= 3:
            return data
        obj_train, obj_test, img_train, img_test = split_points(obj_point_list, img_point_list, test_size=0.1,
                                                                random_state=42)
        data.ok, data.camera_mat, data.dist_coeff, data.rvecs, data.tvecs = _solve(
            kind, obj_train, img_train, frame_size, K0, D0, 1000)
        if data.ok:
            K, D = K_eval(data), D_eval(data)
            _, data.reproj_err = points.reproj_errors(kind, obj_test, img_test, data.rvecs, data.tvecs, K, D)
        if data.ok and points.checkRange(data.camera_mat) and points.checkRange(data.dist_coeff):
            if not _within_gates(data.camera_mat, data.dist_coeff, K_nom(), D_nom(), self.similar_threshold):
                data.ok = False
        return data

    def calib_in_fisheye(self, obj_point_list, img_point_list, frame_size):
        # the nominal arrays go to the solver uncopied and are evaluated / gated after it wrote into them
        return self._calib_in("fisheye", obj_point_list, img_point_list, frame_size,
                              self.init_camera_mat, self.init_dist_coeff,
                              lambda d: self.init_camera_mat, lambda d: self.init_dist_coeff,
                              lambda: self.init_camera_mat, lambda: self.init_dist_coeff)

    def calib_in_normal(self, obj_point_list, img_point_list, frame_size):
        return self._calib_in("pinhole", obj_point_list, img_point_list, frame_size,
                              self.init_camera_mat_normal.copy(), self.init_dist_coeff_normal.copy(),
                              lambda d: d.camera_mat, lambda d: d.dist_coeff,
                              lambda: self.init_camera_mat_normal, lambda: self.init_dist_coeff_normal)

    # -- extrinsic (one solvePnP per camera on a shared board) -----------------
    def calib_ex(self, objectPoints, imgPoints, point_id_list, mtx, dist, check_mode=False, camera_type="normal",
                 rotate=0):
        end_point = np.array([(self.board_w + 1) * self.board_size, (self.board_h + 1) * self.board_size, 0])
        if rotate == 1:      # 90 degrees clockwise (in place, like the reference)
            objectPoints[:, :, 0:2] = np.flip(objectPoints[:, :, 0:2], 2)
            objectPoints[:, :, 1] = end_point[1] - objectPoints[:, :, 1]
        elif rotate == 2:    # 180
            objectPoints = end_point - objectPoints
        elif rotate == 3:    # 270
            objectPoints[:, :, 0:2] = np.flip(objectPoints[:, :, 0:2], 2)
            objectPoints[:, :, 0] = end_point[0] - objectPoints[:, :, 0]
        if camera_type == "fisheye":
            return self.calib_ex_fisheye(imgPoints, objectPoints, point_id_list, mtx, dist, (2960, 1664), check_mode)
        return self.calib_ex_normal(imgPoints, objectPoints, point_id_list, mtx, dist, (1920, 1080), check_mode)

    def _calib_ex(self, undist, focal_scale, imgPoints, objectPoints, point_id_list, mtx, dist, check_mode):
        point_dict, point_dict_perspec = {}, {}
        flat = copy.deepcopy(imgPoints)
        undist(flat, mtx, dist, flat, P=mtx)                     # pixels of an ideal pinhole with the same K
        ret, rvecs, tvecs = points.solvePnP(objectPoints, flat, mtx, np.zeros_like(dist))
        persp = copy.deepcopy(imgPoints)
        mtx_p = mtx.copy()
        mtx_p[0, 0] *= focal_scale
        mtx_p[1, 1] *= focal_scale
        undist(persp, mtx, dist, persp, P=mtx_p)
        if check_mode:
            Rinv = np.linalg.inv(points.Rodrigues(rvecs)[0])
            t1 = (Rinv @ tvecs).ravel()
            levelled = copy.deepcopy(imgPoints)
            undist(levelled, mtx, dist, levelled, Rinv)
            for j in range(levelled.shape[0]):
                pt = levelled[j][0]
                pt[0] = (pt[0] - 1 * t1[0] / 1000) * (t1[2] / 1000) + 0.1
                pt[1] = (pt[1] - 1 * t1[1] / 1000) * (t1[2] / 1000) + 0.1
                point_dict[f"{point_id_list[j]}"] = pt
                point_dict_perspec[f"{point_id_list[j]}"] = persp[j][0]
        return ret, rvecs, tvecs, point_dict, point_dict_perspec

    def calib_ex_fisheye(self, imgPoints, objectPoints, point_id_list, mtx, dist, img_shape, check_mode=False):
        return self._calib_ex(_fe.undistortPoints, 0.6, imgPoints, objectPoints, point_id_list, mtx, dist, check_mode)

    def calib_ex_normal(self, imgPoints, objectPoints, point_id_list, mtx, dist, img_shape, check_mode=False):
        return self._calib_ex(points.undistortPoints, 0.8, imgPoints, objectPoints, point_id_list, mtx, dist,
                              check_mode)


# ---------------------------------------------------------------------------
# server/internal/IntrinsicCalibration/intrinsicCalib.py
class CalibData:
    def __init__(self):
        self.type = None
        self.camera_mat = None
        self.dist_coeff = None
        self.rvecs = None
        self.tvecs = None
        self.map1 = None
        self.map2 = None
        self.reproj_err = None
        self.ok = False
        self.check = True


class _Incremental:
    """Shared body of Fisheye / Normal: first call 100 iterations from the
    nominal values plus the gates, later calls 10 iterations from the previous
    result; after each call the per-view errors over all points and the
    undistortion maps."""
    _kind = None
    _type = None
    _maps = None

    def __init__(self, frame_width=1280, frame_height=1024, focal_scale=0.5, size_scale=1):
        self.data = CalibData()
        self.inited = False
        self.board = []
        self.frame_width, self.frame_height = frame_width, frame_height
        self.focal_scale, self.size_scale = focal_scale, size_scale
        # update() is called after every accepted frame with the same lists, one view longer
        # (intrinsicCalib.py:538-564): the session keeps the views already packed on the device
        self._session = api.CalibrationSession(self._kind)

    def update(self, corners, frame_size):
        if not self.inited:
            self._update_init(self.board, corners, frame_size)
            self.inited = True
        else:
            self._update_refine(self.board, corners, frame_size)
        self._calc_reproj_err(corners)
        self._get_undistort_maps()

    def _update_init(self, board, corners, frame_size):
        data = self.data
        data.type = self._type
        data.camera_mat = self.init_camera_mat           # not copied: the solver writes into the nominal arrays
        data.dist_coeff = self.init_dist_coeff
        data.ok, data.camera_mat, data.dist_coeff, data.rvecs, data.tvecs = _solve(
            self._kind, board, corners, frame_size, data.camera_mat, data.dist_coeff, 100, self._session)
        if not _within_gates(data.camera_mat, data.dist_coeff, self.init_camera_mat, self.init_dist_coeff, 0.15):
            data.check = False
        data.ok = data.ok and data.check and points.checkRange(data.camera_mat) and points.checkRange(data.dist_coeff)

    def _update_refine(self, board, corners, frame_size):
        data = self.data
        data.ok, data.camera_mat, data.dist_coeff, data.rvecs, data.tvecs = _solve(
            self._kind, board, corners, frame_size, data.camera_mat, data.dist_coeff, 10, self._session)
        data.ok = data.ok and points.checkRange(data.camera_mat) and points.checkRange(data.dist_coeff)

    def _calc_reproj_err(self, corners):
        if not self.inited:
            return
        data = self.data
        n = len(corners)
        err, _ = points.reproj_errors(self._kind, self.board[:n], corners, data.rvecs[:n], data.tvecs[:n],
                                      data.camera_mat, data.dist_coeff)
        data.reproj_err = [float(e) for e in err]

    def _get_camera_mat_dst(self, camera_mat):
        dst = camera_mat.copy()
        dst[0][0] *= self.focal_scale
        dst[1][1] *= self.focal_scale
        dst[0][2] = self.frame_width / 2 * self.size_scale
        dst[1][2] = self.frame_height / 2 * self.size_scale
        return dst

    def _get_undistort_maps(self):
        data = self.data
        dst = self._get_camera_mat_dst(data.camera_mat)
        size = (int(self.frame_width * self.size_scale), int(self.frame_height * self.size_scale))
        data.map1, data.map2 = type(self)._maps(data.camera_mat, data.dist_coeff, np.eye(3, 3), dst, size,
                                                points.CV_16SC2)


class Fisheye(_Incremental):
    _kind, _type = "fisheye", "FISHEYE"
    _maps = staticmethod(_fe.initUndistortRectifyMap)

    def __init__(self, **kw):
        super().__init__(**kw)
        self.init_camera_mat = np.array(FISHEYE_INIT_K)
        self.init_dist_coeff = np.array(FISHEYE_INIT_D)


class Normal(_Incremental):
    _kind, _type = "pinhole", "NORMAL"
    _maps = staticmethod(points.initUndistortRectifyMap)

    def __init__(self, **kw):
        super().__init__(**kw)
        self.init_camera_mat = np.array(NORMAL_INIT_K)
        self.init_dist_coeff = np.array(NORMAL_INIT_D)


# ---------------------------------------------------------------------------
# inter_cfg.json wire format (controller/controller_internal_calibration.py:404-416,
# read back at :547-549 and server/video/video_server.py:455-461; sample
# configs/fg/inter_cfg.json): per camera "<name>_calib": [w, h, K row-major (9), D (4 | 5)].
def create_internal(img_size, mtx, distortion):
    out = [int(v) for v in img_size]
    out.extend(float(v) for v in np.asarray(mtx, dtype=np.float64).reshape(-1))
    out.extend(float(v) for v in np.asarray(distortion, dtype=np.float64).reshape(-1))
    return out


def parse_internal(entry):
    """-> (img_size (w, h), K (3,3), D (n,))."""
    if len(entry) not in (15, 16):
        raise ValueError("an inter_cfg entry holds 2 + 9 + 4|5 numbers")
    return (int(entry[0]), int(entry[1])), np.array(entry[2:11], dtype=np.float64).reshape(3, 3), \
        np.array(entry[11:], dtype=np.float64)


def dumps_internal(calib_parameter):
    import json
    return json.dumps(calib_parameter, indent=4, separators=(', ', ': '), ensure_ascii=False)
