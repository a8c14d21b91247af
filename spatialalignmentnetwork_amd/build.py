"""Build libsan_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m spatialalignmentnetwork_amd.build [--force]

The shared object lands next to this file so that it travels with the source
snapshot to the GPU box; nothing is installed into site-packages.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.environ.get("SAN_BUILD_OBJ", os.path.join(HERE, "build"))           # A/B builds: objects and library elsewhere
LIB = os.environ.get("SAN_BUILD_LIB", os.path.join(HERE, "libsan_hip.so"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# NO packed-fp32 instructions anywhere in the device code (round 5).  `v_pk_mul_f32 ... op_sel:[0,1]` was measured to take the
# wrong register of its source pair for the last 16 lanes of a wave while another stream's MFMA kernel shared the compute unit
# (scratch/attempts/r4_sens_overlap_notes.md); round 4 dropped the feature kernel by kernel with a function
# attribute, which breaks inlining of helpers compiled with the default features.  Dropping the target feature for the whole
# device compilation has no such problem: every kernel that can ever share a compute unit with another stream's kernels is
# covered by construction (tests/test_abi.py disassembles the library and counts), and the step got 0.6 % FASTER (the guide
# prices a packed fp32 instruction beside MFMAs above the two scalar ones it replaces).  The host pass does not know the
# feature and says so once per file: that line is filtered below.
NO_PK32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"] + NO_PK32 + os.environ.get("SAN_EXTRA_HIPCC_FLAGS", "").split()      # tuning builds (-DSAN_B16_RING=3 ...): use force


REPLAY_TABLE = os.path.join(CSRC, "san_replay_table.inc")


def replay_table_text() -> str:
    """The dispatcher cases of csrc/san_replay.cpp, one per int-returning prototype of include/san_hip.h, in header order (the
    order ``_lib.parse_header`` sees: a function's id on a replay tape is its index there).  An argument travels as one 64-bit
    word: pointers and integers by value, float / double as their bit patterns."""
    import re
    text = open(os.path.join(os.path.dirname(HERE), "include", "san_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = ["// GENERATED from include/san_hip.h by spatialalignmentnetwork_amd/build.py (replay_table_text) -- do not edit;",
           "// tests/test_abi.py checks that it matches the header."]
    idx = 0
    for m in re.finditer(r"(const\s+char\s*\*|int|size_t)\s+(san_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        if ret == "int" and name != "san_replay_run":
            parts = []
            if args and args != "void":
                for i, a in enumerate(args.split(",")):
                    t = re.sub(r"\b\w+$", "", a.strip()).strip()
                    if "*" in t:
                        parts.append(f"({t})(uintptr_t)a[{i}]")
                    elif t == "float":
                        parts.append(f"bits_f(a[{i}])")
                    elif t == "double":
                        parts.append(f"bits_d(a[{i}])")
                    elif t == "size_t":
                        parts.append(f"(size_t)a[{i}]")
                    else:
                        parts.append(f"(int)(int64_t)a[{i}]")
            out.append(f"case {idx}: rc = {name}({', '.join(parts)}); break;")
        idx += 1
    return "\n".join(out) + "\n"


def _write_replay_table() -> None:
    text = replay_table_text()
    if not os.path.exists(REPLAY_TABLE) or open(REPLAY_TABLE).read() != text:
        with open(REPLAY_TABLE, "w") as f:
            f.write(text)


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "san_hip.h"))
    return hdrs


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(p) for p in [src] + _deps())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    err = "\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
    if err.strip():
        sys.stderr.write(err + "\n")
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    _write_replay_table()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
