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


def test_emulator_exports_the_same_abi(emu):
    for n in _declared():
        assert hasattr(emu.dll, n), n
    assert emu.is_emulator
