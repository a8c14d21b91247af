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


def test_replay_table_is_generated_from_the_header(built):
    """csrc/san_replay_table.inc (the dispatcher cases of san_replay_run) is what build.py derives from include/san_hip.h, and
    the function ids Python puts on a tape are the header's order."""
    from spatialalignmentnetwork_amd import build
    assert open(build.REPLAY_TABLE).read() == build.replay_table_text()
    lib = built.lib()
    assert list(lib.func_index) == list(built.parse_header())
    text = open(build.REPLAY_TABLE).read()
    for name in ("san_rss", "san_adamw_step_hyper", "san_conv_stream_set_tuning"):
        assert f"case {lib.func_index[name]}: rc = {name}(" in text
    assert "san_replay_run(" not in text and "san_last_error_string(" not in text


def test_replay_tape_walks_calls_and_reports_the_failing_entry(built):
    """A tape of C-ABI calls is walked in order by ONE foreign call; the first failing entry stops it and is named; entries flagged
    as weight packing are skipped on request; a value that is not an error code is ignored when flagged so.  (Argument
    validation happens before any HIP call: testable without a GPU.)"""
    lib = built.lib()
    ok = built.tape_call_words(lib._san_version, (), built.TAPE_IGNORE_RC)      # returns the version number, not an rc
    bad = built.tape_call_words(lib._san_rss, (None, None, 1, 1, 16, 0, None))
    assert ok is not None and bad is not None and bad[0] >> 24 == 7
    built.Tape(ok + ok, {}).run()
    with pytest.raises(RuntimeError, match=r"san_rss failed \(argument error -1\).*null"):
        built.Tape(ok + bad + ok, {len(ok): "san_rss"}).run()
    skipped = built.tape_call_words(lib._san_rss, (None, None, 1, 1, 16, 0, None), built.TAPE_PACK)
    built.Tape(ok + skipped, {}).run(skip_packs=True)
    with pytest.raises(RuntimeError, match="argument error"):
        built.Tape(ok + skipped, {}).run(skip_packs=False)
    # floats / doubles travel as bit patterns, negative ints sign-extended
    w = built.tape_call_words(lib._san_adamw_step, (None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, -3, 1.0, None))
    assert w is not None and w[6] == 0x3A83126F and w[11] == 0xFFFFFFFFFFFFFFFD
    with pytest.raises(RuntimeError, match="san_adamw_step"):
        built.Tape(w, {0: "san_adamw_step"}).run()
    # a truncated tape is refused, not read past
    with pytest.raises(RuntimeError, match="past the end"):
        built.Tape(bad[:3], {0: "truncated"}).run()
    assert built.tape_call_words(lib._san_last_error_string, ()) is None


def _device_disassembly(lib_path, tmp_path):
    """gfx950 disassembly of every code object in the library's .hip_fatbin section (one clang offload bundle per
    translation unit, concatenated by the linker)."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = str(tmp_path / "fat.bin")
    subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib_path], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, data)]
    assert starts, "no offload bundle in the library"
    out = []
    for k, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(data)])):
        bundle, co = str(tmp_path / f"b{k}.bin"), str(tmp_path / f"b{k}.hsaco")
        open(bundle, "wb").write(data[lo:hi])
        ids = subprocess.run([f"{llvm}/clang-offload-bundler", "--list", "--type=o", f"--input={bundle}"], check=True,
                             capture_output=True, text=True).stdout.split()
        target = [t for t in ids if "gfx950" in t]
        assert len(target) == 1, ids
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={target[0]}", f"--input={bundle}",
                        f"--output={co}"], check=True)
        out.append(subprocess.run([f"{llvm}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout)
    return out


def test_library_has_no_packed_fp32_instruction(built, tmp_path):
    """Round 4 found `v_pk_mul_f32 ... op_sel:[0,1]` reading the wrong register of its source pair while another stream's MFMA
    kernel shared the compute unit.  The library is built without the packed-fp32 target feature (build.py, NO_PK32): no kernel
    of any stream can contain such an instruction, whatever shares the chip with it.  Checked on the shipped binary."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    texts = _device_disassembly(built.LIB_PATH, tmp_path)
    assert len(texts) >= 10                      # one per .hip source
    pk = [line.strip() for t in texts for line in t.splitlines() if re.search(r"\bv_pk_(mul|add|fma|mov)_(f32|b32)\b", line)]
    assert not pk, f"{len(pk)} packed-fp32 instructions in libsan_hip.so, e.g. {pk[:3]}"
    assert sum(t.count("v_mfma_") for t in texts) > 10000        # (the disassembly is the real one)


def test_rccl_binding_loads_the_process_library_and_reports_errors(built):
    """csrc/san_rccl.cpp binds RCCL at run time by path (no link dependency): the library torch ships resolves all six entry points
    and reports its version; a bogus path is an error with a message, not a crash; a communicator handle that does not exist is
    refused before any RCCL call (no GPU needed for any of this)."""
    import glob
    import torch
    lib = built.lib()
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
    if not cands:
        pytest.skip("torch ships no librccl here")
    with pytest.raises(RuntimeError, match="cannot open"):
        lib.call("san_rccl_load", b"/nonexistent/librccl.so", None)
    ver = ctypes.c_int(0)
    lib.call("san_rccl_load", cands[0].encode(), ctypes.byref(ver))
    assert ver.value >= 20000, ver.value                     # NCCL-style version code (2.26.6 -> 22606)
    with pytest.raises(RuntimeError, match="no such communicator"):
        lib.call("san_rccl_allreduce_sum_f32", 12345, ctypes.c_void_p(64), 4, None)


def test_native_rccl_is_not_attempted_without_an_nccl_group(built, monkeypatch):
    """dist.native_rccl(): only on the nccl backend with a CUDA device; anything else answers None without touching RCCL."""
    from spatialalignmentnetwork_amd import dist as sdist
    monkeypatch.setitem(sdist.NATIVE, "tried", False)
    monkeypatch.setitem(sdist.NATIVE, "handle", None)
    assert sdist.native_rccl(None, "cpu") is None
    assert sdist.NATIVE["tried"] is True and sdist.NATIVE["handle"] is None
    monkeypatch.setitem(sdist.NATIVE, "tried", False)
    monkeypatch.setenv("SAN_NATIVE_RCCL", "0")
    assert sdist.native_rccl(object(), "cuda:0") is None and sdist.NATIVE["why"] == "SAN_NATIVE_RCCL=0"
