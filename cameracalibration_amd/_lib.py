"""ctypes binding of libcamcal.so (include/camcal.h).

The product path has exactly one implementation: the HIP library built from
``csrc/camcal_kernels.hip``.  :func:`load` raises if it is missing -- there is
no CPU fallback.  Tests may pass an explicit ``path`` (the CPU lane emulator
of the test tree) to exercise the host logic without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "csrc", "libcamcal.so")

PINHOLE, FISHEYE = 0, 1
LAYOUT_NOT_PLANAR, LAYOUT_BAD_COUNT, LAYOUT_BAD_TOTAL = 1, 2, 4     # camcal_layout_device: head[2]


class Tiles(C.Structure):
    """struct camcal_tiles (include/camcal.h)."""
    _fields_ = [("pts", C.c_void_p), ("tile_off", C.c_void_p), ("tile_nmax", C.c_void_p),
                ("lane_n", C.c_void_p), ("n_tiles", C.c_int32), ("n_slots", C.c_int64),
                ("planar", C.c_int32), ("lanes_per_view", C.c_int32)]


_P, _I, _D = C.c_void_p, C.c_int, C.c_double
_TP = C.POINTER(Tiles)

# int (*allreduce_sum)(void* ctx, double* buf, int64_t n, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class RigCam(C.Structure):
    """struct camcal_rig_cam (include/camcal.h)."""
    _fields_ = [("tiles", Tiles), ("slot_pose", C.c_void_p), ("pose_slot", C.c_void_p), ("pose", C.c_void_p),
                ("V", C.c_void_p), ("W", C.c_void_p), ("g", C.c_void_p)]


RIG_MAX_CAMS = 4


class Rig(C.Structure):
    """struct camcal_rig (include/camcal.h)."""
    _fields_ = [("model", C.c_int32), ("n_cams", C.c_int32), ("P", C.c_int64), ("cam", RigCam * RIG_MAX_CAMS),
                ("intr", C.c_void_p * 2), ("ext", C.c_void_p * 2), ("rig_pose", C.c_void_p * 2),
                ("Vp", C.c_void_p * 2), ("gp", C.c_void_p * 2), ("Wp", C.c_void_p * 2), ("rec", C.c_void_p * 2),
                ("Yt", C.c_void_p), ("z", C.c_void_p), ("dG", C.c_void_p), ("scratch", C.c_void_p),
                ("status", C.c_void_p)]


class Problem(C.Structure):
    """struct camcal_problem (include/camcal.h)."""
    _fields_ = [("model", C.c_int32), ("flags", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("crit_type", C.c_int32), ("max_iter", C.c_int32), ("eps", C.c_double), ("n_views", C.c_int64),
                ("view_ptr", C.c_void_p), ("obj", C.c_void_p), ("img", C.c_void_p), ("K", C.c_void_p), ("D", C.c_void_p),
                ("n_dist", C.c_int32), ("lanes_per_view", C.c_int32), ("rt", C.c_void_p), ("view_status", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("rms", C.c_double),
                ("iterations", C.c_int32), ("status", C.c_int32)]


class CommStruct(C.Structure):
    """struct camcal_comm (include/camcal.h)."""
    _fields_ = [("allreduce_sum", ALLREDUCE_FN), ("ctx", C.c_void_p)]


_SIGNATURES = {
    "camcal_version": (C.c_int, []),
    "camcal_n_intr": (C.c_int, [_I]),
    "camcal_glob_len": (C.c_int, [_I]),
    "camcal_schur_len": (C.c_int, [_I]),
    "camcal_scratch_len": (C.c_int64, [_I, _I]),
    "camcal_layout": (C.c_int, [C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "camcal_layout_device_temp_len": (C.c_int64, [C.c_int64]),
    "camcal_points_off_plane": (C.c_int, [C.c_int64, _P, _P, _P]),
    "camcal_layout_device": (C.c_int, [C.c_int64, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "camcal_pack_tiles": (C.c_int, [_P, _P, _P, _P, _TP, _P, _P]),
    "camcal_pose_init": (C.c_int, [_I, _TP, _P, _P, _P, _P]),
    "camcal_pose_refine": (C.c_int, [_I, _TP, _P, _P, _I, _D, _P, _P]),
    "camcal_intr_normal": (C.c_int, [_I, _TP, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "camcal_cross_schur": (C.c_int, [_I, _TP, _P, _P, _P, _P, _D, _P, _P, _P, _P]),
    "camcal_view_blocks": (C.c_int, [_I, _TP, _P, _P, _P, _P, _P, _P]),
    "camcal_schur": (C.c_int, [_I, _TP, _P, _P, _P, _D, _P, _P, _P]),
    "camcal_solve_reduced": (C.c_int, [_I, _P, _P, C.c_char_p, _D, _D, _P, _P, _P, _P, _P]),
    "camcal_backsub": (C.c_int, [_I, _TP, _P, _P, _P, _D, _P, _P, _P, _P, _P, _P]),
    "camcal_normal_eq": (C.c_int, [_I, _TP, _P, _P, _D, _P, _P, _P, _P, _P, _P, _P]),
    "camcal_normal_eq_fused": (C.c_int, [_I, _TP, _P, _P, _D, _P, _P, _P, _P, _P, _P, _P]),
    "camcal_normal_eq_fused_static": (C.c_int, [_I, _TP, _P, _P, _D, _P, _P, _P]),
    "camcal_rccl_unique_id": (C.c_int, [_P]),
    "camcal_comm_rccl_create": (C.c_int, [_P, _I, _I, C.POINTER(_P)]),
    "camcal_comm_destroy": (C.c_int, [_P]),
    "camcal_refine_normal_eq_partial": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_double, _P, _P, C.c_int, _P, C.c_double,
                                                  C.c_int, C.c_int, _P, _P]),
    "camcal_reduce_solve": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, _P, C.c_double, C.c_double, _P, _P, _P, _P, _P, C.c_double, _P]),
    "camcal_small_set_begin": (C.c_int, [C.c_int, _P, _P, _P]),
    "camcal_cross_schur_solve": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_double, _P, _P, _P, _P, C.c_double, _P, _P, _P, _P, _P]),
    "camcal_schur_solve": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_double, _P, _P, _P, C.c_double, _P, _P, _P, _P, _P, _P]),
    "camcal_backsub_intr_normal": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                             C.c_double, _P, _P]),
    "camcal_comm_size": (C.c_int, [_P]),
    "camcal_comm_allreduce": (C.c_int, [_P, _P, C.c_int64, _P]),
    "camcal_normal_eq_fused_response": (C.c_int, [_I, _TP, _P, _P, _P, _P, _P, _P, _P]),
    "camcal_pose_predict": (C.c_int, [_I, _TP, _P, _P, _D, _P, _I, _P, _P]),
    "camcal_pose_predict_refine": (C.c_int, [_I, _TP, _P, _P, _P, _D, _I, _P, _I, _D, _P, _P]),
    "camcal_fisheye_gn": (C.c_int, [_TP, _P, _P, C.c_char_p, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P, _P, _P,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "camcal_pinhole_lm": (C.c_int, [_TP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_char_p, _I, _D, _P, _P,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "camcal_workspace_len": (C.c_int64, [_I, C.c_int64, _P]),
    "camcal_calibrate": (C.c_int, [_I, C.c_int64, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _D, _P, C.c_int64,
                                   _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "camcal_calibrate_many": (C.c_int, [C.POINTER(Problem), _I, _P]),
    "camcal_time_stage": (C.c_int, [_I, _I, _I, _TP, _P, _P, _D, _P, _P, _P, _P, _P, _P, _P,
                                    C.POINTER(C.c_float)]),
    "camcal_rig_rec_len": (C.c_int64, [_I, _I]),
    "camcal_rig_scratch_len": (C.c_int64, [_I, _I, _P, C.c_int64]),
    "camcal_rig_evaluate": (C.c_int, [_P, _I, _P]),
    "camcal_rig_residuals": (C.c_int, [_P, _I, _P]),
    "camcal_rig_residuals_notify": (C.c_int, [_P, _I, _P, C.c_double, _P]),
    "camcal_rig_chain": (C.c_int, [_P, _I, _P]),
    "camcal_rig_eliminate": (C.c_int, [_P, _I, _D, _P]),
    "camcal_rig_solve": (C.c_int, [_P, _I, C.c_char_p, _D, _I, _P]),
    "camcal_rig_backsub": (C.c_int, [_P, _I, _D, _I, _P]),
    "camcal_rig_init": (C.c_int, [_P, _P, _P]),
    "camcal_rig_lm": (C.c_int, [_P, C.c_char_p, _I, _D, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32)]),
    "camcal_project_points": (C.c_int, [_I, C.c_int64, _P, _P, _P, _P, _P]),
    "camcal_reproj_err": (C.c_int, [_I, _TP, _P, _P, _I, _P, _P, _P, _P]),
    "camcal_undistort_points": (C.c_int, [_I, C.c_int64, _P, _P, _P, _P, _P]),
    "camcal_undistort_rectify_map": (C.c_int, [_I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "camcal_remap_u8": (C.c_int, [_P, _I, _I, C.c_int64, _I, _P, _P, _I, _I, _I, _P, C.c_int64, _I, _P]),
    "camcal_peak_probe": (C.c_int, [_I, _P, C.c_int64, _I, _P, C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


E_NUMERIC, E_COMM, E_DISJOINT, E_PEER, E_UNSUPPORTED = -4, -5, -6, -7, -8


class CamcalError(RuntimeError):
    pass


class CamcalNumericError(CamcalError):
    """CAMCAL_E_NUMERIC: a solver loop met a singular system or a non-finite cost."""


class CamcalStatus(CamcalError):
    """A call returned a CAMCAL_E_* / hipError_t status; ``code`` holds it."""

    def __init__(self, msg, code):
        super().__init__(msg)
        self.code = int(code)


class CamcalLib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise CamcalError(
                f"{path} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        self.is_emulator = self.camcal_version() < 0

    def check(self, rc: int, what: str):
        if rc == E_NUMERIC:
            raise CamcalNumericError(f"{what}: singular system or non-finite cost (degenerate views?)")
        if rc != 0:
            raise CamcalStatus(f"{what} failed with status {rc}"
                               + (" (hipError_t)" if rc > 0 else " (CAMCAL_E_*)"), rc)


_cached = None


def load(path: str | None = None) -> CamcalLib:
    """Open the HIP library (cached).  ``path`` overrides for tests only."""
    global _cached
    if path is not None:
        return CamcalLib(path)
    if _cached is None:
        _cached = CamcalLib(DEFAULT_PATH)
    return _cached
