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


def _device_code_objects(tmp_path):
    """The gfx950 code objects inside libaa_mi355.so: one clang offload bundle per translation unit (csrc/aa_api.hip and the
    groups of csrc/aa_tiles.hip)."""
    import struct
    from animate_anything_amd import build
    data = open(build.build(), "rb").read()
    out, start = [], 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", start)
        if i < 0:
            break
        start = i + 24
        n = struct.unpack_from("<Q", data, i + 24)[0]
        if not 0 < n < 16:
            continue
        off = i + 32
        for _ in range(n):
            o, s_, ln = struct.unpack_from("<QQQ", data, off)
            off += 24
            name = data[off:off + ln].decode(errors="replace")
            off += ln
            if "gfx950" in name and s_ > 0:
                p = tmp_path / f"dev{len(out)}.co"
                p.write_bytes(data[i + o:i + o + s_])
                out.append(str(p))
    assert out, "no gfx950 code object in the library"
    return out


def _device_code_object(tmp_path):
    """(the unit that holds the first tile group: scripts that only need some contraction kernels)"""
    return _device_code_objects(tmp_path)[0]


def test_x_kernels_keep_hipcc_out_of_the_accumulators(tmp_path):
    """The hand-scheduled contraction kernels (csrc/kernels/conv_gemm_x.h) name their accumulators a[0:255] literally:
    hipcc must neither spill (scratch) nor touch accumulation registers itself.  Audit of the built code object
    (cdna guide 5.7 item 4): per kernel no private segment, no VGPR spills, and exactly the v_accvgpr traffic the source
    writes - 16 initialising writes per literal block and site, 16 reads per block for each read-out site (epilogue(s), split-K)."""
    import re
    import shutil
    import subprocess
    tools = "/opt/rocm/lib/llvm/bin"
    if not shutil.which(os.path.join(tools, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    cos = _device_code_objects(tmp_path)
    notes = "".join(subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout for co in cos)
    meta = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
        meta[m.group(2)] = tuple(int(m.group(k)) for k in (1, 3, 4, 5))
    xk = {k: v for k, v in meta.items() if "conv_gemm_x_kernel" in k}
    assert len(xk) >= 16                                    # 8 tiles x {fp16, bf16}
    dis = "".join(subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout for co in cos)
    bodies = {}
    cur = None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(line)
    for name, (agpr, scratch, vgpr, spills) in xk.items():
        assert scratch == 0 and spills == 0, (name, scratch, spills)
        m = re.search(r"x_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        wm, wn, per_cu = int(m.group(3)), int(m.group(4)), int(m.group(10))
        waves_per_simd = wm * wn * per_cu // 4             # 1: the whole 512-register file per lane; 2: half of it
        assert vgpr <= 512 // waves_per_simd, (name, vgpr)
        body = "\n".join(bodies[name])
        literal_blocks = agpr // 16
        assert literal_blocks in (4, 8, 12, 15, 16), (name, agpr)
        # the accumulators are written by the source only: started from the bias (or a folded LayerNorm's terms) at the K loop prologue / empty K range
        writes = len(re.findall(r"v_accvgpr_write", body))        # (r04: the BK = 64 prologue has two sites (one or more K steps), tiles that can start
        assert writes % (16 * literal_blocks) == 0 and 2 <= writes // (16 * literal_blocks) <= 8, (name, writes)      #  from a folded LayerNorm two forms per site)
        # read-out sites (split-K partials, the general epilogue, its branch-free forms) read every block exactly once each
        reads = len(re.findall(r"v_accvgpr_read", body))
        assert 2 * 16 * literal_blocks <= reads <= 32 * 16 * literal_blocks, (name, reads)     # (r04: + LayerNorm-fold / row-statistics forms, a second epilogue instance behind an in-kernel K-split finish; hipcc may clone part of a form)
        assert "scratch_" not in body, name
        # a matrix-core product the COMPILER places must not land in accumulation registers: hipcc does not know the literal blocks are
        # live there (r04: the row-statistics epilogue used the MFMA builtin, its results reused literal blocks and some (tile, epilogue
        # form) pairs overwrote blocks that had not been read out yet - the statistics are packed dot products now).  Every MFMA with
        # an a[...] destination is one of the source's literal statements: destination = third source, aligned to a block, inside the
        # literal range.
        mf = re.findall(r"v_mfma_f32_32x32x16_\w+\s+([av]\[\d+:\d+\]),\s*\S+,\s*\S+,\s*([av]\[\d+:\d+\]|\S+)", body)
        a_dst = [(d.rstrip(","), c) for d, c in mf if d.startswith("a")]
        for d, c in a_dst:
            lo = int(re.match(r"a\[(\d+):", d).group(1))
            assert d == c and lo % 16 == 0 and lo < 16 * literal_blocks, (name, d, c)
        # ... and nothing else names an accumulation register at all (gfx950 loads could target them directly)
        for line in bodies[name]:
            if re.search(r"[\s,]a(\[\d+:\d+\]|\d+)\b", line) and not re.search(r"^\s*(v_mfma_|v_accvgpr_)", line.strip()):
                raise AssertionError((name, "accumulation register outside the source's statements", line.strip()))

