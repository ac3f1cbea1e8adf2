"""CPU: the C-ABI library loads and exports every symbol include/hp_hip.h declares (no compute calls)."""
import os
import re

from hyperpose_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"libhp_hip.so does not export: {missing}"


def test_python_symbol_list_matches_header():
    assert sorted(_lib.SYMBOLS) == _declared()


def test_struct_layout_matches_reference_human_t():
    # include/hyperpose/utility/human.hpp:14-31: 18 x {bool,f32,f32,f32} + f32 = 292 bytes
    import ctypes as C
    assert C.sizeof(_lib.BodyPart) == 16
    assert C.sizeof(_lib.Human) == 292
    assert _lib.HUMAN_DTYPE.itemsize == 292


def test_no_device_reports_error_not_crash():
    L = _lib.lib()
    n = L.hp_device_count()
    if n <= 0:
        rc = L.hp_init(0)
        assert rc == _lib.HP_ERR_NO_DEVICE
        assert len(L.hp_last_error()) > 0
