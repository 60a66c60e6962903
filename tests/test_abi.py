"""The C-ABI library loads without a GPU and exports every symbol that
include/camcal.h declares; the ctypes table covers the same set."""
import ctypes
import os
import re

from cameracalibration_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "camcal.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(camcal_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_by_hip_library():
    path = _lib.DEFAULT_PATH
    assert os.path.exists(path), "libcamcal.so must be built in-tree (make lib)"
    dll = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(dll, n), n


def test_ctypes_table_matches_header():
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared()


def test_query_entry_points_need_no_gpu():
    lib = _lib.load()
    assert lib.camcal_version() > 0 and not lib.is_emulator
    assert lib.camcal_n_intr(0) == 9 and lib.camcal_n_intr(1) == 8 and lib.camcal_n_intr(7) < 0
    assert lib.camcal_glob_len(0) == 55 and lib.camcal_glob_len(1) == 45
    assert lib.camcal_schur_len(0) == 54 and lib.camcal_schur_len(1) == 44
    assert lib.camcal_scratch_len(1, 100) >= 64 * 64 and lib.camcal_scratch_len(1, 20000) >= 89 * 20000


def test_tiles_struct_layout_matches_the_header():
    """camcal_tiles as ctypes sees it: four pointers, n_tiles, n_slots, planar, lanes_per_view (56 bytes)."""
    T = _lib.Tiles
    assert ctypes.sizeof(T) == 56
    assert (T.n_tiles.offset, T.n_slots.offset, T.planar.offset, T.lanes_per_view.offset) == (32, 40, 48, 52)


def test_emulator_exports_the_same_abi(emu):
    for n in _declared():
        assert hasattr(emu.dll, n), n
    assert emu.is_emulator


def test_host_layout_matches_python_layout():
    """camcal_layout (host side of the C ABI, what a C caller uses) against
    engine.make_layout (what the Python drop-in uses), both libraries."""
    import ctypes as C
    import numpy as np
    from cameracalibration_amd import _lib
    from cameracalibration_amd.engine import make_layout
    rng = np.random.default_rng(0)
    libs = [_lib.load(os.path.join(ROOT, "tests", "hostemu", "libcamcal_hostemu.so"))]
    if os.path.exists(_lib.DEFAULT_PATH):
        libs.append(_lib.load(_lib.DEFAULT_PATH))
    for lib in libs:
        for n_views in (0, 1, 63, 64, 65, 1000):
            n = rng.integers(4, 89, n_views).astype(np.int32)
            ref = make_layout(n)
            nt, ns, tot = C.c_int32(), C.c_int64(), C.c_int64()
            ptr = n.ctypes.data if n_views else None
            assert lib.camcal_layout(n_views, ptr, C.byref(nt), C.byref(ns), C.byref(tot), None, None, None, None) == 0
            assert (nt.value, ns.value, tot.value) == (ref["n_tiles"], ref["n_slots"], ref["total_floats"])
            sv = np.empty(ns.value, np.int32); ln = np.empty(ns.value, np.int32)
            nm = np.empty(nt.value, np.int32); to = np.empty(nt.value, np.int64)
            assert lib.camcal_layout(n_views, ptr, C.byref(nt), C.byref(ns), C.byref(tot), sv.ctypes.data,
                                     ln.ctypes.data, nm.ctypes.data, to.ctypes.data) == 0
            assert np.array_equal(sv, ref["slot_view"]) and np.array_equal(ln, ref["lane_n"])
            assert np.array_equal(nm, ref["tile_nmax"]) and np.array_equal(to, ref["tile_off"])
