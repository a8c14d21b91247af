"""CPU-only checks of the drop-in boundary: libsan_hip.so builds for gfx950,
loads, and exports every symbol include/san_hip.h declares.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from spatialalignmentnetwork_amd import _lib
    return _lib


def test_header_parses_and_lib_exports_every_symbol(built):
    protos = built.parse_header()
    text = open(os.path.join(ROOT, "include", "san_hip.h")).read()
    declared = set(re.findall(r"\b(san_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 25
    dll = ctypes.CDLL(built.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name


def test_error_path_without_gpu(built):
    """Argument validation happens before any HIP call, so it is testable here."""
    lib = built.lib()
    with pytest.raises(RuntimeError, match="null"):
        lib.call("san_rss", None, None, 1, 1, 16, 0, None)
    assert "null" in lib.last_error()
    # MFMA layout [groups][cin + 4 pad rows][taps][4][quads padded to 4]: 18 -> one group of 20 (quads 5 -> 8)
    assert lib.query("san_conv_packed_floats", 18, 3, 3) == 1 * (3 + 4) * 9 * 4 * 8
    assert lib.query("san_conv_packed_floats", 36, 18, 3) == 2 * 18 * 9 * 32 + 4 * 9 * 32   # 36 -> 2 groups of 20
    assert lib.query("san_conv_packed_floats", 64, 64, 1) == 4 * 64 * 1 * 16 + 4 * 1 * 16
    # transposed 2x2 conv == 1x1 conv to 4*cout virtual channels: 72 -> 4 groups of 20
    assert lib.query("san_conv_packed_floats", 18, 36, 2) == 4 * 36 * 32 + 4 * 32
    assert lib.query("san_fft_workspace_bytes", 8, 320, 320) == 8 * 320 * 320 * 8


def test_missing_library_fails_loudly(built, tmp_path):
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        built.SanLibrary(str(tmp_path / "libsan_hip.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spatialalignmentnetwork_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f
