"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/orbslam3_hip.h declares (no compute calls without a GPU), and fails loudly without a device."""
import ctypes as C
import re
from pathlib import Path

import pytest

from orb_slam3_study_kr_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "orbslam3_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osh_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_mirror_agree():
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/orbslam3_hip.h but not exported"
    assert b"gfx950" in lib.osh_version()


def test_struct_sizes_match_header_layout():
    # 4 ints + 8 pointers + 3 doubles + int (+pad) + 4 pointers (stop_flag, kb8, cam2, trl)
    assert C.sizeof(capi.LbaProblem) == 16 + 8 * 8 + 24 + 8 + 32
    assert C.sizeof(capi.LbaResult) == 4 * 8 + 16 + 128 * 8 * 2 + 128 * 4 + 8
    assert C.sizeof(capi.OrbBatch) == 16 + 6 * 8


def test_no_device_fails_loudly_not_silently():
    lib = capi.load_library()
    if lib.osh_device_count() > 0:
        pytest.skip("a GPU is visible; this test covers the no-GPU container")
    ctx = C.c_void_p()
    rc = lib.osh_lba_create(0, C.byref(ctx))
    assert rc == capi.OSH_ERR_NO_DEVICE and not ctx
    assert "device" in capi.last_error(lib).lower()
    rc = lib.osh_orb_create(0, C.byref(ctx))
    assert rc == capi.OSH_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    for py in (ROOT / "orb_slam3_study_kr_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py
    for src in (ROOT / "orb_slam3_study_kr_amd" / "csrc").rglob("*"):
        if src.suffix in (".hip", ".cpp", ".cc", ".h"):
            assert "oracle" not in src.read_text().lower().replace("oracle/", "ORACLEDIR"), src
