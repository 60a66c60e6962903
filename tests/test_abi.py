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
        for n_views in (0, 1, 63, 64, 65, 1000, 1001):
            n = rng.integers(4, 89, n_views).astype(np.int32)
            if n_views == 1001:
                n[17] = 5_000_000                      # past the counting sort's table: the comparison sort
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


def _device_layout(lib, view_ptr, obj, to_dev, to_host):
    """camcal_layout_device through ctypes; to_dev / to_host move arrays (identity on the emulator)."""
    import ctypes as C
    import numpy as np
    n_views = len(view_ptr) - 1
    T = max((n_views + 63) // 64, 1)
    S = 64 * T
    d_vp, d_obj = to_dev(np.ascontiguousarray(view_ptr, np.int64)), to_dev(np.ascontiguousarray(obj, np.float32))
    sv, ln = to_dev(np.full(S, 7, np.int32)), to_dev(np.full(S, 7, np.int32))
    nm, off, head = to_dev(np.full(T, 7, np.int32)), to_dev(np.full(T, 7, np.int64)), to_dev(np.full(4, 7, np.int64))
    tmp_len = lib.camcal_layout_device_temp_len(n_views)
    assert tmp_len >= 0
    tmp = to_dev(np.zeros(max(tmp_len, 1), np.uint8))
    ptr = lambda a: a.ctypes.data if isinstance(a, np.ndarray) else a.data_ptr()
    rc = lib.camcal_layout_device(n_views, ptr(d_vp), obj.shape[0], ptr(d_obj) if obj.size else None, ptr(sv), ptr(ln),
                                  ptr(nm), ptr(off), ptr(head), ptr(tmp), tmp_len, None)
    assert rc == 0, rc
    return [to_host(a) for a in (sv, ln, nm, off, head)]


def check_device_layout(lib, to_dev, to_host, sizes):
    """Shared by the emulator test below and the GPU tier (tests/test_gpu_cabi.py)."""
    import numpy as np
    from cameracalibration_amd.engine import make_layout
    rng = np.random.default_rng(5)
    for n_views in sizes:
        for hi in (6, 89, 2400):                                   # few distinct counts (many ties) ... ragged
            n = rng.integers(4, hi, n_views).astype(np.int64) if n_views else np.zeros(0, np.int64)
            if n_views > 2000 and hi == 2400:
                n = np.minimum(n, 300)
            if n_views > 200000:
                n = np.minimum(n, 24)                              # keeps the point arrays of the test small
            vp = np.concatenate([[0], np.cumsum(n)])
            obj = np.zeros((int(vp[-1]), 3), np.float32)
            obj[:, :2] = 1.5
            ref = make_layout(n.astype(np.int32))
            sv, ln, nm, off, head = _device_layout(lib, vp, obj, to_dev, to_host)
            assert np.array_equal(sv, ref["slot_view"]) and np.array_equal(ln, ref["lane_n"]), (n_views, hi)
            assert np.array_equal(nm, ref["tile_nmax"]) and np.array_equal(off, ref["tile_off"]), (n_views, hi)
            assert list(head) == [int(vp[-1]), ref["total_floats"], 0, int(n.max()) if n_views else 0]
            if n_views:
                obj[int(vp[-1]) - 1, 2] = 1e-30                    # one point off the plane, the last one
                assert _device_layout(lib, vp, obj, to_dev, to_host)[4][2] == _lib.LAYOUT_NOT_PLANAR
                obj[int(vp[-1]) - 1, 2] = 0
                assert _device_layout(lib, vp, obj[:-1], to_dev, to_host)[4][2] & _lib.LAYOUT_BAD_TOTAL
                bad = vp.copy()
                bad[n_views // 2] += 10 ** 6 if n_views > 1 else 0   # an offset out of order: a negative count
                if n_views > 2:
                    assert _device_layout(lib, bad, obj, to_dev, to_host)[4][2] & _lib.LAYOUT_BAD_COUNT


def test_device_layout_entry_on_the_emulator(emu):
    check_device_layout(emu, lambda a: a, lambda a: a, (0, 1, 63, 64, 65, 1000))
