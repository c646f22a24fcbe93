"""The C-ABI boundary: libaa_mi355.so (cross-compiled for gfx950) loads on a machine without a GPU and exports every
entry point include/aa_mi355.h declares; the ctypes mirror binds them all.  No compute calls here."""
import ctypes
import os
import re

import pytest
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


def _device_code_objects(tmp_path):
    """The gfx950 code objects inside libaa_mi355.so (animate_anything_amd/build.py::device_code_objects)."""
    from animate_anything_amd import build
    out = build.device_code_objects(build.build(), str(tmp_path))
    assert out, "no gfx950 code object in the library"
    return out


def _device_code_object(tmp_path):
    """(the unit that holds the first tile group: scripts that only need some contraction kernels)"""
    return _device_code_objects(tmp_path)[0]


def test_x_kernels_keep_hipcc_out_of_the_accumulators():
    """The hand-scheduled contraction kernels (csrc/kernels/conv_gemm_x.h) name their accumulators a[0:255] literally: hipcc must
    neither spill (scratch) nor touch accumulation registers itself.  The audit of the built code object lives in the build
    (animate_anything_amd/build.py::audit_x_kernels - a library that fails it is rejected at build time); here it runs on the library
    the tests use, and the build record next to it names the compiler."""
    import json
    import shutil
    if not shutil.which(os.path.join(build.LLVM_TOOLS, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    lib = build.build()
    assert build.audit_x_kernels(lib) == []
    rec = lib[:-3] + ".buildinfo.json"
    if os.path.exists(rec):                                 # (written by the build that compiled the library)
        info = json.load(open(rec))
        assert info["audit"] == "ok" and any("clang" in line or "HIP" in line for line in info["hipcc"])


def test_the_audit_notices_a_foreign_accumulator_write(tmp_path, monkeypatch):
    """The audit itself: a disassembly in which hipcc parked one value in an accumulation register (one extra v_accvgpr_write) or
    gave an MFMA an accumulation-register destination that is no literal block is reported."""
    import shutil
    import subprocess
    if not shutil.which(os.path.join(build.LLVM_TOOLS, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    real = subprocess.run

    def tampered(cmd, **kw):
        r = real(cmd, **kw)
        if "-d" in cmd and "conv_gemm_x_kernel" in r.stdout:
            lines = r.stdout.split("\n")
            k = next(i for i, ln in enumerate(lines) if "v_mfma_f32_32x32x16" in ln and " a[" in ln)
            lines.insert(k, "\tv_accvgpr_write_b32 a3, v7")
            lines.insert(k, "\tv_mfma_f32_32x32x16_f16 a[8:23], v[2:5], v[6:9], a[8:23]")
            r = subprocess.CompletedProcess(r.args, r.returncode, "\n".join(lines), r.stderr)
        return r
    monkeypatch.setattr(subprocess, "run", tampered)
    problems = build.audit_x_kernels(build.build())
    assert any("v_accvgpr_write" in p for p in problems) and any("not one of the source's statements" in p for p in problems)
