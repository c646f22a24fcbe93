"""The C-ABI boundary: libaa_mi355.so (cross-compiled for gfx950) loads on a machine without a GPU and exports every
entry point include/aa_mi355.h declares; the ctypes mirror binds them all.  No compute calls here."""
import ctypes
import os
import re

import torch  # noqa: F401  (loads the HIP runtime the library links against)

from animate_anything_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aa_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("aa_conv_gemm", "aa_conv_gemm_workspace", "aa_conv_gemm_tile_info", "aa_groupnorm", "aa_layernorm",
              "aa_attention", "aa_softmax_rows", "aa_cfg_dpm_step", "aa_version", "aa_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    path = build.build()
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib.SYMBOLS) == set(declared_symbols())
    bound = _lib.bind(path)
    assert bound.aa_version() >= 1
    info = (ctypes.c_int32 * 7)()
    n = 0
    while bound.aa_conv_gemm_tile_info(n, info) == 0:
        assert info[0] % 32 == 0 and info[1] % 32 == 0 and info[4] in (32, 64)
        n += 1
    assert n >= 20 and bound.aa_conv_gemm_tile_info(-1, info) == -1


def test_missing_library_fails_loudly(tmp_path):
    try:
        _lib.bind(str(tmp_path / "libaa_mi355.so"))
    except RuntimeError as e:
        assert "no fallback backend" in str(e)
    else:
        raise AssertionError("binding a missing library must raise")
