"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/mst_hip.h declares (no kernel is launched here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from musicstyletransfer_amd.csrc import build
    build.build(verbose=False)
    from musicstyletransfer_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mst_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mst_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mst_hip.h but not exported"


def test_binding_table_matches_header(lib):
    from musicstyletransfer_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_string(lib):
    assert lib.mst_version() >= 100
    assert isinstance(lib.mst_last_error(), bytes)


def test_struct_sizes_match_c(lib):
    # sizes computed by the C compiler for the same declarations
    import ctypes, subprocess, tempfile
    from musicstyletransfer_amd import _lib
    src = '#include <stdio.h>\n#include "mst_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(mst_gemm_args), sizeof(mst_wgrad_args));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(_lib.GemmArgs) == a
    assert ctypes.sizeof(_lib.WgradArgs) == b


def test_invalid_arguments_fail_loudly(lib):
    # argument validation happens before any HIP call, so it is testable without a GPU
    import ctypes
    from musicstyletransfer_amd import _lib
    g = _lib.GemmArgs()
    g.M, g.N, g.K = 4, 4, 3  # K not a multiple of 8
    rc = lib.mst_gemm_nt(ctypes.byref(g), None)
    assert rc == -1
    assert b"multiples of 8" in lib.mst_last_error()
    with pytest.raises(_lib.MstError):
        _lib.call("mst_layernorm_fwd", 0, 4, 6, None, 8, None, None, 1e-5, None, 8, None, None, 1, None)
