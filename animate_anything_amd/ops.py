"""Torch-tensor front end of the C ABI (include/aa_mi355.h).

Tensors only carry device memory here: every function checks its operands, fills the C descriptor
and enqueues the HIP kernels on torch's current stream.  Activations are channels-last token
matrices `[tokens, C]` (see DESIGN.md "Data layout").  There is no fallback: operands that are not
on the GPU raise, as does a missing library.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import (AA_ACT_GELU, AA_ACT_NONE, AA_ACT_QUICK_GELU, AA_ACT_SILU, AA_BF16, AA_F16, AA_F32, AaAttention, AaAttnOperand,
                   AaBlend, AaConvGemm, AaFFFused, AaLinearRows, AaDpmStep, AaDpmStepTok, AaEulerStepTok, AaGroupNorm, AaPackFrames, AaPackLatents, AaSeqSelfAttn)

_DT = {torch.float16: AA_F16, torch.bfloat16: AA_BF16, torch.float32: AA_F32}

# bench.py sets this to a list to record every contraction launch of one step (descriptor + the
# tensors that keep its pointers alive) so the dominant kernel can be re-timed in isolation.
TRACE = None

# Tile-shape autotuning of aa_conv_gemm: the first time a contraction signature is seen on the GPU
# (outside hipGraph capture) every tile shape of the library's table that fits is timed on the real
# operands and the fastest index is remembered for that signature (descriptor field `tile`).
AUTOTUNE = True
LAST_STAMPS = None
GN_PLANS = None            # a list: groupnorm() appends (groups per workgroup, pieces per thread, parts, grid) or None per call
MAX_OPERAND_BYTES = 1 << 31      # the contraction kernel's 32-bit operand offsets
FORCE_TILE = -1       # tests / sweeps: >= 0 puts this tile-table index into AaConvGemm.tile of every conv_gemm call (strict: ineligible = error)
# experiments: AaAttention._pad.  bit 0 (1): s_setprio 1 around the matrix clusters of the head_dim-64 kernel; bit 2 (4): short key sequences take the
# general kernel instead of attention_shortkv_kernel (A/B); bit 3 (8): long sequences take the two-query-blocks-per-wave kernel (round 6 experiment: slower, profiles/r06g_*).  (The eager row maximum of rounds 2-4 is a COMPILE-time variant: build.build(variant=
# ("eager", ["-DAA_ATTN_EAGER_MAX=1"])) loaded through AA_LIBRARY - there is no run-time bit for it.)  Unknown bits are rejected.
ATTN_FLAGS = int(os.environ.get("AA_ATTN_FLAGS", "0"))
if ATTN_FLAGS & ~13:
    raise RuntimeError(f"AA_ATTN_FLAGS={ATTN_FLAGS}: only bits 0 (value 1), 2 (value 4) and 3 (value 8) exist")
# A folded LayerNorm's row statistics reach the consumer as aa_ln_finalize's per-row coefficients (one 4.7 us launch per consumer, 99 per
# step) or raw (ABI 106 `ln_parts`: the consumer finalises 2 / 10 partial sums per row itself with independent loads).  Measured (r04h/i):
# the raw form removes 80 launches (-0.37 ms) and costs the K = 320 ... 1280 consumers 4-7 % (+0.7 ms: square root, reciprocal and the
# sums sit in every tile's prologue) - the launch stays the default.
LN_FINALIZE_LAUNCH = os.environ.get("AA_LN_RAW", "0") != "1"
TILE_PICKER = None    # tests / scripts/debug: callable(signature key, [(tile, K splits), ...]) -> the pair a conv_gemm call runs with (GPU or emulator)
PRODUCER_COEF = os.environ.get("AA_PRODUCER_COEF", "1") == "1"   # ABI 107: a producer whose tile spans the row writes the LayerNorm coefficients itself (no aa_ln_finalize launch)
LN_RAW_ANY_PARTS = False     # tests: let the consumer finalise any number of partial sums per row (a chain of dependent loads: +12-23 %)
K_SPLITS = 0          # tests: explicit K split count for calls that are not autotuned (0 = library decides)
# K-split launches of the hand-scheduled tiles can finish inside the kernel (AaConvGemm.tickets, ABI 106).  Measured on the step (r04g):
# not faster than partials + reduce launch (the autotuner moved the 8x8-level shapes to the compiled tile + reduce launch; step 62.5 ms
# with, 61.9 ms without) - the release / acquire fences around the ticket write back and invalidate the XCD's L2.  Off unless asked for.
USE_TICKETS = os.environ.get("AA_TICKETS", "0") == "1"
DEBUG_ABLATE = int(os.environ.get("AA_DEBUG_ABLATE", "0"))      # profiling only: forwarded to AaConvGemm.debug (16: row-major tile order, A/B of the grouped order)
class _TileTable:
    """The library's tile table (aa_conv_gemm_tile_info), read on first use: entries (rows, columns, K step, stages)."""

    def __init__(self):
        self._rows = None

    def _load(self):
        if self._rows is None:
            lib, rows, info = _lib.get(), [], (C.c_int32 * 7)()
            while lib.aa_conv_gemm_tile_info(len(rows), info) == 0:
                rows.append(tuple(info) + (int(lib.aa_conv_gemm_tile_flags(len(rows))),))
            self._rows = rows
        return self._rows

    def __iter__(self):
        return iter([(r[0], r[1], r[4], r[5]) for r in self._load()])

    def __len__(self):
        return len(self._load())

    def __getitem__(self, i):
        r = self._load()[i]
        return (r[0], r[1], r[4], r[5])

    def wave_cols(self, i):
        return self._load()[i][3]

    def per_cu(self, i):
        return self._load()[i][6]

    def flags(self, i):
        """aa_conv_gemm_tile_flags: bit 0 halo-slab kernel, bit 1 hand-scheduled stream, higher bits the DMA schedule."""
        return self._load()[i][7]


TILE_TABLE = _TileTable()
_tile_cache = {}


def _tile_table_id():
    """What a saved tile choice is valid for: the library version and the exact tile table it indexes."""
    lib = _lib.get()
    return {"aa_version": int(lib.aa_version()), "arch": "gfx950",
            "tiles": [list(TILE_TABLE._load()[i]) for i in range(len(TILE_TABLE))]}


def save_tile_cache(path):
    """Persist the autotuned (tile, K splits) choices (json) so a later process can skip the tuning launches."""
    import json
    with open(path, "w") as f:
        json.dump({"id": _tile_table_id(), "choices": [[list(k), list(v)] for k, v in sorted(_tile_cache.items())]}, f)


def load_tile_cache(path):
    """Load choices saved by `save_tile_cache`; a file written against another library version / tile table is ignored
    (returns False): its indices would select arbitrary tiles."""
    import json
    with open(path) as f:
        blob = json.load(f)
    if not isinstance(blob, dict) or blob.get("id") != _tile_table_id():
        return False
    for k, v in blob["choices"]:
        _tile_cache[tuple(k)] = tuple(v)
    return True


if os.environ.get("AA_DUMP_TILE_CACHE"):                  # debugging aid: what did this process's autotuner choose
    import atexit
    atexit.register(lambda: save_tile_cache(os.environ["AA_DUMP_TILE_CACHE"]))

DEFAULT_TILE_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_cache_gfx950.json")
_default_cache_tried = False


def _load_default_tile_cache():
    """The committed tile choices for the benchmarked shapes (written by `bench.py --tile-cache`): fresh boxes then skip
    the tuning launches - and their box-to-box jitter - for every signature the file knows."""
    global _default_cache_tried
    if not _default_cache_tried:
        _default_cache_tried = True
        if os.path.exists(DEFAULT_TILE_CACHE) and os.environ.get("AA_NO_TILE_CACHE", "0") != "1":
            try:
                load_tile_cache(DEFAULT_TILE_CACHE)
            except Exception:
                pass


TICKET_COUNTERS = 1 << 16
_tickets = {}


def _ticket_array(t: torch.Tensor):
    """The zero-initialised counter array of AaConvGemm.tickets for the stream `t`'s calls run on: one per (device, stream) -
    K-split launches in flight on different streams must not share counters; the kernels hand every counter back as zero.
    Launches recorded into a hipGraph use a separate array per device (a capture stream is short-lived; the replays of the
    graphs of one device run one at a time in this package).  Never allocated inside a capture: a call captured before any eager
    call on that device gets None (partials + reduce launch)."""
    if not t.is_cuda:
        if not _lib.host_pointers_ok():
            return None
        a = _tickets.get("host")
        if a is None:
            a = _tickets["host"] = torch.zeros(TICKET_COUNTERS, dtype=torch.int32)
        return a
    capturing = torch.cuda.is_current_stream_capturing()
    key = (str(t.device), "graph" if capturing else torch.cuda.current_stream(t.device).cuda_stream)
    a = _tickets.get(key)
    if a is None and not capturing:
        a = _tickets[key] = torch.zeros(TICKET_COUNTERS, dtype=torch.int32, device=t.device)
        gkey = (str(t.device), "graph")
        if gkey not in _tickets:
            _tickets[gkey] = torch.zeros(TICKET_COUNTERS, dtype=torch.int32, device=t.device)
    return a


def _tile_candidates(d, rows):
    """(tile, k_splits) pairs worth timing: every tile shape that divides the packed width with the library's own split
    decision (0), plus explicit K splits where they turn an under-filled single round of workgroups into (nearly)
    full ones - the 16x16 / 8x8 levels, where 170 tiles of 256x256 would leave a third of the CUs idle."""
    dma = (d.c0 + d.c1) % 64 == 0 and d.c0 % 64 == 0 and d.out_dtype == d.dtype
    if not dma:
        return []
    out = []
    lib = _lib.get()
    for i, (bm, bn, bk, _st) in enumerate(TILE_TABLE):
        if not lib.aa_conv_gemm_tile_ok(C.byref(d), i):                 # width, GEGLU pairing, slab-kernel geometry
            continue
        out.append((i, 0))
        if TILE_TABLE.flags(i) & 1:                                     # halo-slab kernel: no K split
            continue
        if d.geglu or bm < 128:
            continue
        tiles = -(-rows // bm) * (d.n_pad // bn)
        slots = 256 * TILE_TABLE.per_cu(i)
        nk = d.k_pad // bk
        if tiles >= slots:
            continue
        fill1 = tiles / slots
        for sp in (2, 3, 4, 5, 6, 8):
            if nk // sp < 8:
                break
            wgs = tiles * sp
            fill = wgs / (-(-wgs // slots) * slots)
            if wgs <= 3 * slots and fill >= 0.8 and fill > fill1 + 0.1:
                out.append((i, sp))
    return out


AUTOTUNE_EVENTS = 0      # signatures tuned by timing launches in this process (0 when the committed tile cache covers the workload)


def _autotune(lib, d, stream, key, rows, dev):
    global AUTOTUNE_EVENTS
    AUTOTUNE_EVENTS += 1
    cands = _tile_candidates(d, rows)
    if "stats" in key:                                   # the caller wants the row statistics: only ways of carrying the call out that emit them
        def emits(c, sp):
            d.tile, d.k_splits = c, sp
            need = lib.aa_conv_gemm_workspace(C.byref(d))
            keep = (d.workspace, d.workspace_bytes)
            d.workspace, d.workspace_bytes = (C.c_void_p(1), need) if need else (None, 0)      # (only its presence is looked at)
            n = lib.aa_conv_gemm_row_stats_parts(C.byref(d))
            d.workspace, d.workspace_bytes = keep
            return n > 0
        cands = [c for c in cands if emits(*c)] or cands
    best = (-1, 0)
    if len(cands) == 1:                                  # (one way left - e.g. the only tile that emits the row statistics: take it, do not
        best = cands[0]                                  #  fall back to the library's own choice, which may be another: ADVICE r04)
    if len(cands) > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ws_keep = (d.workspace, d.workspace_bytes)

        def timed(c, sp, reps, raw=False):
            d.tile, d.k_splits = c, sp
            need = lib.aa_conv_gemm_workspace(C.byref(d))
            ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=dev)
            d.workspace, d.workspace_bytes = C.c_void_p(ws.data_ptr()), need
            if lib.aa_conv_gemm(C.byref(d), stream) != 0:
                return None
            # one launch per measurement, synchronised in between: back-to-back launches of one kernel overlap
            # their tails, which hides exactly the last-round under-fill that a dependent chain of kernels pays
            ts = []
            for _ in range(reps):
                e0.record()
                lib.aa_conv_gemm(C.byref(d), stream)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            if raw:
                return ts
            ts.sort()
            return ts[len(ts) // 2] if reps > 5 else ts[1]

        # screening pass over every candidate, then a run-off between the four fastest: three interleaved passes of seven launches
        # each, median of the 21 (r04: with one 11-launch pass per finalist the choice flipped between runs on close calls - clock /
        # power drift during a pass favours whoever runs at the right moment - and a flipped 3x3 choice costs 0.1 ms of the step)
        first = sorted((t, c) for c in cands for t in [timed(c[0], c[1], 5)] if t is not None)
        finalists = [c for _, c in first[:4]]
        samples = {c: [] for c in finalists}
        for _ in range(3):
            for c in finalists:
                ts = timed(c[0], c[1], 7, raw=True)
                if ts is not None:
                    samples[c] += ts
        final = sorted((sorted(v)[len(v) // 2], c) for c, v in samples.items() if v)
        if final:
            best = final[0][1]
        d.workspace, d.workspace_bytes = ws_keep
    _tile_cache[key] = best
    return best


def _autotune_groupnorm(lib, d, ws, nbytes, stream, key):
    """(1, 0) if the statistics + apply pair beats the single kernel for this signature, else (0, 0)."""
    global AUTOTUNE_EVENTS
    info = (C.c_int32 * 4)()
    best = (0, 0)
    if lib.aa_groupnorm_plan(C.byref(d), info):
        AUTOTUNE_EVENTS += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(two):
            lib.aa_set_groupnorm_two_pass(two)
            try:
                ts = []
                for i in range(9):
                    e0.record()
                    lib.aa_groupnorm(C.byref(d), C.c_void_p(ws.data_ptr()), nbytes, stream)
                    e1.record()
                    e1.synchronize()
                    if i >= 2:
                        ts.append(e0.elapsed_time(e1))
            finally:
                lib.aa_set_groupnorm_two_pass(0)
            ts.sort()
            return ts[len(ts) // 2]

        t1, t2 = timed(0), timed(1)
        best = (1, 0) if t2 < t1 else (0, 0)
    _tile_cache[key] = best
    return best


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd ops run on the GPU only (tensor is on %s)" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("animate_anything_amd ops need contiguous tensors")


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _run(fn, *args):
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("libaa_mi355: " + _lib.get().aa_last_error().decode())


def _round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------- weights
@dataclass
class PackedWeight:
    """Weights laid out for aa_conv_gemm: [n_pad, k_pad], k ordered (tap, channel)."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    n_out: int
    kh: int
    kw: int
    cin: int          # channels per tap as stored (after padding to a multiple of 8)
    geglu: int = 0    # 0, or the value/gate interleave granularity of a GEGLU projection
    k_order: int = 0  # 0: K = (tap, channel); 1: K = (64-channel chunk, tap, channel)
    ln_cols: Optional[torch.Tensor] = None   # LayerNorm folded into these weights: fp32 [2, n_pad] = colsum(W'), b + beta W^T
    ln_eps: float = 0.0

    @property
    def n_pad(self):
        return self.w.shape[0]

    @property
    def k_pad(self):
        return self.w.shape[1]


def pack_weight(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, geglu: bool = False, ln=None) -> PackedWeight:
    """Pack an nn.Linear [n,k], nn.Conv2d [n,c,kh,kw] or nn.Conv3d [n,c,kt,1,1] weight.

    GEGLU (diffusers `GEGLU.proj`, rows [0,d) = value, [d,2d) = gate) is re-ordered into alternating
    blocks of 32 value rows / 32 gate rows: a value block and its gate block are then neighbouring accumulator
    blocks of the same wavefront and the gating happens in registers.

    `ln` = (gamma, beta, eps) folds the LayerNorm in front of an nn.Linear into it (AaConvGemm.ln_stats):
    LN(x) W^T + b = rstd (x W'^T) - rstd mean colsum(W') + (b + beta W^T) with W' = W diag(gamma) rounded to the storage type;
    colsum is taken over the ROUNDED W', so the mean term cancels exactly what the matrix cores accumulate."""
    w = weight.detach()
    b_ln = None
    if ln is not None:
        gamma, beta, eps = ln
        assert w.dim() == 2, "the LayerNorm fold is for nn.Linear weights"
        wf = w.float()
        b_ln = wf @ beta.detach().float() + (0.0 if bias is None else bias.detach().float())      # fp32 [n]
        w = (wf * gamma.detach().float()[None, :]).to(w.dtype)
        bias = None
    if w.dim() == 2:
        n, kh, kw, cin = w.shape[0], 1, 1, w.shape[1]
        w4 = w.reshape(n, 1, 1, cin)
    elif w.dim() == 4:
        n, cin, kh, kw = w.shape
        w4 = w.permute(0, 2, 3, 1)
    elif w.dim() == 5:
        n, cin, kh, kw = w.shape[0], w.shape[1], w.shape[2], 1
        assert w.shape[3] == 1 and w.shape[4] == 1
        w4 = w.reshape(n, cin, kh).permute(0, 2, 1).reshape(n, kh, 1, cin)
    else:
        raise ValueError("unsupported weight rank %d" % w.dim())
    cpad = _round_up(cin, 8)
    if cpad != cin:
        w4 = torch.nn.functional.pad(w4, (0, cpad - cin))
    k = kh * kw * cpad
    k_order = 0
    if kh * kw > 1 and cpad % 64 == 0 and n % 8 == 0:      # (only shapes that take the LDS-DMA path)
        # multi-tap filter: chunk-major K so the 9 (3) taps of one 64-channel slab are consecutive K steps
        w4 = w4.reshape(n, kh * kw, cpad // 64, 64).permute(0, 2, 1, 3)
        k_order = 1
    w2 = w4.reshape(n, k)
    b = None if bias is None else bias.detach()
    if geglu:
        d = n // 2
        gran = 32                                          # one 32-column accumulator block of value, the next one of gate
        assert d % gran == 0, "GEGLU inner width must be a multiple of 32"
        val = torch.arange(d, device=w.device).reshape(d // gran, 1, gran)
        src = torch.cat([val, val + d], dim=1).reshape(-1)
        w2 = w2[src]
        b = None if b is None else b[src]
        b_ln = None if b_ln is None else b_ln[src]
    n_pad, k_pad = _round_up(n, 64), _round_up(k, 64)
    out = torch.zeros(n_pad, k_pad, dtype=w.dtype, device=w.device)
    out[:n, :k] = w2
    if b is not None and geglu and n_pad != n:
        b = torch.nn.functional.pad(b, (0, n_pad - n))
    ln_cols, ln_eps = None, 0.0
    if ln is not None:
        ln_cols = torch.zeros(2, n_pad, dtype=torch.float32, device=w.device)
        ln_cols[0] = out.float().sum(dim=1)
        ln_cols[1, :n] = b_ln
        ln_eps = float(ln[2])
    return PackedWeight(out.contiguous(), None if b is None else b.contiguous(), n, kh, kw, cpad, gran if geglu else 0, k_order,
                        ln_cols, ln_eps)


def pack_upsample2x_weights(weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """Nearest x2 upsample + 3x3 / pad 1 convolution (diffusers Upsample2D) == four 2x2 convolutions on the STORED grid, one per
    output parity class (a, b) = (Y & 1, X & 1): virtual rows Y-1, Y, Y+1 of output row Y = 2y + a read stored rows
    (y-1, y, y) for a = 0 and (y, y, y+1) for a = 1 - the taps that read the same stored pixel are summed here (fp32, one
    rounding).  Returns {(a, b): PackedWeight of the [N, C, 2, 2] filter}; class (a, b) runs with padding (1 - a, 1 - b) and
    out_map (2, 2, a, b).  4/9 of the multiply-adds of the 3x3 form; zero padding of the virtual grid == zero padding of the
    stored grid for every class (the summed taps of a class fall inside or outside together)."""
    n, c, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().float()
    groups = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}          # parity -> taps of the two stored rows / columns
    out = {}
    for a in (0, 1):
        for b in (0, 1):
            f = torch.zeros(n, c, 2, 2, dtype=torch.float32, device=w.device)
            for r, dys in enumerate(groups[a]):
                for q, dxs in enumerate(groups[b]):
                    for dy in dys:
                        for dx in dxs:
                            f[:, :, r, q] += w[:, :, dy, dx]
            out[(a, b)] = pack_weight(f.to(weight.dtype), bias)
    return out


@dataclass
class RowStats:
    """Partial (sum, sum of squares) per row of a token matrix, [rows, parts, 2] fp32, left by the contraction that wrote it
    (AaConvGemm.row_stats); `coef(channels, eps)` turns them (aa_ln_finalize, once) into the per-row (-mean, sqrt(var + eps),
    rstd, 0) that the contraction folding the LayerNorm of that matrix takes (AaConvGemm.ln_stats)."""
    data: Optional[torch.Tensor]      # None: the producer wrote the finished coefficients itself (AaConvGemm.row_coef, ABI 107)
    rows: int
    parts: int
    _coef: Optional[torch.Tensor] = None
    _coef_key: tuple = ()

    def coef(self, channels, eps):
        if self._coef is None or self._coef_key != (channels, eps):
            if self.data is None:
                raise RuntimeError("RowStats: finished coefficients for (channels, eps) = %r were asked for %r" % (self._coef_key, (channels, eps)))
            out = torch.empty(self.rows, 4, dtype=torch.float32, device=self.data.device)
            _run(_lib.get().aa_ln_finalize, _ptr(self.data), self.parts, _ptr(out), self.rows, channels, float(eps), _stream(self.data))
            self._coef, self._coef_key = out, (channels, eps)
        return self._coef

    def repeat(self, n):
        if self.data is None:
            return RowStats(None, self.rows * n, 0, torch.cat([self._coef] * n), self._coef_key)
        return RowStats(torch.cat([self.data] * n), self.rows * n, self.parts)


# ------------------------------------------------------------------------------------- contraction
@dataclass
class Geom:
    """Output grid and source mapping of one aa_conv_gemm call (all in pixels / tokens)."""
    n_img: int
    h_in: int
    w_in: int
    h_out: int
    w_out: int
    stride: int = 1
    pad_h: int = 0
    pad_w: int = 0
    h_virt: int = 0
    w_virt: int = 0

    @property
    def rows(self):
        return self.n_img * self.h_out * self.w_out


def linear_geom(rows: int) -> Geom:
    return Geom(1, 1, rows, 1, rows)


def conv3x3_geom(n, h, w, stride=1, pad=1, up_to=None) -> Geom:
    """3x3 conv over [n,h,w]; `up_to`=(H,W) inserts a nearest resize in front (Upsample2D);
    stride 2 / pad 1 is the UNet Downsample2D, stride 2 / pad 0 with a one-pixel bottom/right halo
    the VAE one (F.pad (0,1,0,1))."""
    hv, wv = (h, w) if up_to is None else up_to
    if stride == 1:
        ho, wo = hv, wv
    elif pad == 1:
        ho, wo = (hv + 2 - 3) // stride + 1, (wv + 2 - 3) // stride + 1
    else:
        ho, wo = (hv + 1 - 3) // stride + 1, (wv + 1 - 3) // stride + 1
    return Geom(n, h, w, ho, wo, stride, pad, pad, hv, wv)


def tconv_geom(clips, frames, hw) -> Geom:
    """Conv3d (3,1,1), padding (1,0,0): a 3x1 conv over a [clips, frames, hw] grid."""
    return Geom(clips, frames, hw, frames, hw, 1, 1, 0)


def conv_gemm(x0: torch.Tensor, pw: PackedWeight, g: Geom, x1: Optional[torch.Tensor] = None,
              rowvec: Optional[torch.Tensor] = None, rowvec_div: int = 1,
              residual: Optional[torch.Tensor] = None, act: int = AA_ACT_NONE, out_dtype=None,
              out_scale: float = 1.0, bias_per_row: bool = False, out: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = "packed", acc_scale: float = 1.0, out_map=None, ln_stats=None, row_stats: bool = False,
              coef_eps: Optional[float] = None):
    """`out_map` = (sy, sx, oy, ox): GEMM row (img, y, x) goes to pixel (y * sy + oy, x * sx + ox) of the [n_img, h_out * sy,
    w_out * sx] grid that `out` (required then) holds - AaConvGemm.out_sy .. out_ox.
    `ln_stats` = RowStats of x0's rows (from the call that produced x0): `pw` must carry a folded LayerNorm (pack_weight(ln=...)).
    `row_stats=True`: returns (out, RowStats or None) - the partial row statistics of `out` when this call can emit them; with
    `coef_eps` (the epsilon of the LayerNorm that will consume them) a call whose tile spans the output row writes the finished
    per-row coefficients itself (ABI 107: no aa_ln_finalize launch in front of the consumer)."""
    lib = _lib.get()
    osc = 1 if out_map is None else out_map[0] * out_map[1]
    if out_map is not None and (out is None or out.shape[0] != g.rows * osc):
        raise RuntimeError("conv_gemm: out_map needs `out` = the whole [n_img * h_out * sy * w_out * sx, N] grid")
    b = pw.bias if isinstance(bias, str) else bias
    _check(x0, x1, pw.w, b, residual, out)
    if rowvec is not None:                                 # may be a column slice of a wider matrix (row pitch = stride(0))
        if (not rowvec.is_cuda and not _lib.host_pointers_ok()) or rowvec.stride(-1) != 1:
            raise RuntimeError("conv_gemm: rowvec must be a GPU tensor with contiguous rows")
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    # The LDS-DMA kernel addresses every operand with 32-bit byte offsets (< 2 GiB).  Bigger activations (the VAE above
    # 512x512 x 16 frames) are cut along the image axis - images are independent rows of the implicit GEMM.
    n_cols_ = pw.n_out // 2 if pw.geglu else pw.n_out
    in_rows = g.n_img * g.h_in * g.w_in
    biggest = 2 * max(in_rows * max(c0, c1), g.rows * max(osc * (n_cols_ if out is None else out.stride(0)),
                                                          0 if residual is None else residual.stride(0)))
    if biggest >= MAX_OPERAND_BYTES and g.n_img > 1 and not bias_per_row:
        if out is None:
            out = torch.empty(g.rows, n_cols_, dtype=x0.dtype if out_dtype is None else out_dtype, device=x0.device)
        parts = min(g.n_img, -(-biggest // (MAX_OPERAND_BYTES // 2)))
        per = -(-g.n_img // parts)
        ri, ro = g.h_in * g.w_in, g.h_out * g.w_out
        for i0 in range(0, g.n_img, per):
            n = min(per, g.n_img - i0)
            gi = Geom(n, g.h_in, g.w_in, g.h_out, g.w_out, g.stride, g.pad_h, g.pad_w, g.h_virt, g.w_virt)
            rv = rowvec
            if rowvec is not None:                         # row-vector groups are whole images (or whole clips of them)
                assert (i0 * ro) % rowvec_div == 0 and ((i0 + n) * ro) % rowvec_div == 0 or i0 + n == g.n_img
                rv = rowvec[(i0 * ro) // rowvec_div:]
            ls = None
            if ln_stats is not None:                       # the statistics of a chunk's rows (the folded call is linear: ri == ro)
                if ln_stats.data is None:                  # finished coefficients from the producer (ABI 107): slice those
                    ls = RowStats(None, n * ri, 0, ln_stats._coef[i0 * ri:(i0 + n) * ri], ln_stats._coef_key)
                else:
                    ls = RowStats(ln_stats.data[i0 * ri:(i0 + n) * ri], n * ri, ln_stats.parts)
            conv_gemm(x0[i0 * ri:(i0 + n) * ri], pw, gi, None if x1 is None else x1[i0 * ri:(i0 + n) * ri], rv, rowvec_div,
                      None if residual is None else residual[i0 * ro:(i0 + n) * ro], act, out_dtype, out_scale, False,
                      out[i0 * ro * osc:(i0 + n) * ro * osc], bias, acc_scale, out_map, ls)
        return (out, None) if row_stats else out           # (chunks may run different tiles: no common statistics layout - the caller keeps its LayerNorm)
    if c0 + c1 != pw.cin:
        raise RuntimeError(f"conv_gemm: activation has {c0}+{c1} channels, weight expects {pw.cin}")
    n_cols = pw.n_out // 2 if pw.geglu else pw.n_out
    odt = x0.dtype if out_dtype is None else out_dtype
    if out is None:
        out = torch.empty(g.rows, n_cols, dtype=odt, device=x0.device)
    d = AaConvGemm()
    d.a0, d.a1, d.w, d.bias = _ptr(x0), _ptr(x1), _ptr(pw.w), _ptr(b)
    d.rowvec, d.residual, d.out = _ptr(rowvec), _ptr(residual), _ptr(out)
    d.c0, d.c1 = c0, c1
    d.n_img, d.h_in, d.w_in = g.n_img, g.h_in, g.w_in
    d.h_virt, d.w_virt = g.h_virt or g.h_in, g.w_virt or g.w_in
    d.h_out, d.w_out = g.h_out, g.w_out
    d.kh, d.kw, d.stride, d.pad_h, d.pad_w = pw.kh, pw.kw, g.stride, g.pad_h, g.pad_w
    d.n_out, d.n_pad, d.k_pad = pw.n_out, pw.n_pad, pw.k_pad
    d.rowvec_div = rowvec_div
    d.rowvec_ld = 0 if rowvec is None or rowvec.dim() < 2 else rowvec.stride(0)
    d.ldo = out.stride(0)
    d.ldr = 0 if residual is None else residual.stride(0)
    d.act, d.geglu, d.bias_per_row = act, int(pw.geglu), int(bias_per_row)
    d.dtype, d.out_dtype, d.out_scale = _DT[x0.dtype], _DT[odt], out_scale
    d.acc_scale = 0.0 if acc_scale == 1.0 else float(acc_scale)
    if out_map is not None:
        d.out_sy, d.out_sx, d.out_oy, d.out_ox = (int(v) for v in out_map)
    if ln_stats is not None:
        if pw.ln_cols is None or ln_stats.rows != g.rows:
            raise RuntimeError("conv_gemm: ln_stats needs weights packed with a folded LayerNorm and statistics of every row of x0")
        if ln_stats.data is None or LN_FINALIZE_LAUNCH or (ln_stats.parts not in (2, 10) and not LN_RAW_ANY_PARTS):      # aa_ln_finalize in between (r04h: finalising the pieces of a row one dependent load
                                                                                      # after the other costs the consumer 12-23 %: unrolled forms for 2 / 10 parts only)
            d.ln_stats, d.ln_cols = _ptr(ln_stats.coef(c0, pw.ln_eps)), _ptr(pw.ln_cols)
        else:                                             # the kernel finalises the producer's partial sums itself
            d.ln_stats, d.ln_cols = _ptr(ln_stats.data), _ptr(pw.ln_cols)
            d.ln_parts, d.ln_eps = ln_stats.parts, float(pw.ln_eps)
    elif pw.ln_cols is not None:
        raise RuntimeError("conv_gemm: these weights carry a folded LayerNorm: pass the row statistics of x0 (ln_stats)")
    if acc_scale == 0.0:
        raise ValueError("conv_gemm: acc_scale == 0 is not representable (0 means 1 in the C ABI)")
    d.k_order = pw.k_order
    d.debug = DEBUG_ABLATE
    tk = _ticket_array(x0) if USE_TICKETS else None
    if tk is not None:
        d.tickets, d.tickets_len = _ptr(tk), tk.numel()
    d.tile, d.k_splits = -1, K_SPLITS
    key = (d.dtype, g.n_img, g.h_in, g.w_in, d.h_virt, d.w_virt, g.h_out, g.w_out, g.stride, pw.kh, pw.kw, c0, c1,
           pw.n_out, d.geglu, residual is not None) + (("ln",) if ln_stats is not None else ()) + (("stats",) if row_stats else ())
    if TILE_PICKER is not None and FORCE_TILE < 0 and K_SPLITS == 0:      # tile fuzzing (tests, scripts/debug): any eligible pair must be right
        cands = _tile_candidates(d, g.rows)
        d.tile, d.k_splits = TILE_PICKER(key, cands) if cands else (-1, 0)
    elif AUTOTUNE and x0.is_cuda and FORCE_TILE < 0 and K_SPLITS == 0:    # (explicit tile / split requests of tests and sweeps are not re-tuned)
        _load_default_tile_cache()
        choice = _tile_cache.get(key)
        # the tuning launches write `out` repeatedly: only safe when no input of the call aliases it
        aliased = any(t is not None and t.untyped_storage().data_ptr() == out.untyped_storage().data_ptr()
                      for t in (x0, x1, residual, rowvec))
        if choice is None and not aliased and not torch.cuda.is_current_stream_capturing():
            choice = _autotune(lib, d, _stream(x0), key, g.rows, x0.device)
        d.tile, d.k_splits = (-1, 0) if choice is None else choice
    if FORCE_TILE >= 0:                                   # tests / sweeps: THIS tile or an error (aa_conv_gemm rejects a tile that
        d.tile = FORCE_TILE                               # cannot carry out the call; the thread-local override merely prefers)
    ws = None
    need = lib.aa_conv_gemm_workspace(C.byref(d))        # split-K scratch for few-tile / long-K calls
    if DEBUG_ABLATE & 8:                                  # phase probe: [workgroup][8] shader-clock stamps
        global LAST_STAMPS
        ws = LAST_STAMPS = torch.zeros(2, 1 << 16, 8, dtype=torch.int64, device=x0.device)
        d.workspace, d.workspace_bytes = _ptr(ws), ws.numel() * 8
    elif need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=x0.device)
        d.workspace, d.workspace_bytes = _ptr(ws), need
    stats = None
    if row_stats:                                         # after tile / workspace are settled: the plan decides whether it can emit them
        parts = lib.aa_conv_gemm_row_stats_parts(C.byref(d))
        if parts > 0 and coef_eps is not None and PRODUCER_COEF and lib.aa_conv_gemm_row_coef_ok(C.byref(d)):
            coef = torch.empty(g.rows, 4, dtype=torch.float32, device=x0.device)       # the tile spans the row: finished coefficients
            stats = RowStats(None, g.rows, 0, coef, (pw.n_out, float(coef_eps)))
            d.row_coef, d.row_coef_eps = _ptr(coef), float(coef_eps)
        elif parts > 0:
            stats = RowStats(torch.empty(g.rows, parts, 2, dtype=torch.float32, device=x0.device), g.rows, parts)
            d.row_stats, d.row_stats_parts = _ptr(stats.data), parts
    _run(lib.aa_conv_gemm, C.byref(d), _stream(x0))
    if TRACE is not None:
        TRACE.append((d, (x0, x1, pw, b, rowvec, residual, out, ws, stats, ln_stats)))
    return (out, stats) if row_stats else out


# ------------------------------------------------------------------------------------- norms
def groupnorm(x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_groups_img: int,
              tokens_per_group: int, num_groups: int = 32, eps: float = 1e-5, silu: bool = False,
              x1: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.get()
    _check(x0, x1, gamma, beta, out)
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    y = torch.empty(n_groups_img * tokens_per_group, c0 + c1, dtype=x0.dtype, device=x0.device) if out is None else out
    d = AaGroupNorm()
    d.x0, d.x1, d.gamma, d.beta, d.y = _ptr(x0), _ptr(x1), _ptr(gamma), _ptr(beta), _ptr(y)
    d.c0, d.c1 = c0, c1
    d.n_groups_img, d.tokens_per_group, d.num_groups = n_groups_img, tokens_per_group, num_groups
    d.silu, d.dtype, d.eps = int(silu), _DT[x0.dtype], eps
    nbytes = lib.aa_groupnorm_workspace(C.byref(d))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x0.device)
    # one kernel (x kept in registers between statistics and normalisation) or the statistics + apply pair: the library offers
    # the single kernel wherever the shape allows; which of the two is faster is measured once per signature and kept with
    # the tile choices (key (-1, ...): tile_cache_gfx950.json)
    two_pass = False
    if AUTOTUNE and x0.is_cuda:
        key = (-1, d.dtype, c0, c1, n_groups_img, tokens_per_group, num_groups, int(silu))
        _load_default_tile_cache()
        choice = _tile_cache.get(key)
        if choice is None and y.data_ptr() != x0.data_ptr() and not torch.cuda.is_current_stream_capturing():
            choice = _autotune_groupnorm(lib, d, ws, nbytes, _stream(x0), key)
        two_pass = bool(choice and choice[0])
    if two_pass:
        lib.aa_set_groupnorm_two_pass(1)
    try:
        if GN_PLANS is not None:                          # (tests / bench breakdown) how this call runs: None = two kernels
            info = (C.c_int32 * 4)()
            GN_PLANS.append(tuple(info) if lib.aa_groupnorm_plan(C.byref(d), info) else None)
        _run(lib.aa_groupnorm, C.byref(d), _ptr(ws), nbytes, _stream(x0))
    finally:
        if two_pass:
            lib.aa_set_groupnorm_two_pass(0)
    return y


def groupnorm_coef(x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_groups_img: int, tokens_per_group: int, num_groups: int = 32,
                   eps: float = 1e-5) -> torch.Tensor:
    """The statistics pass of `groupnorm` alone: fp32 [n_groups_img, 2, C] = per-channel (scale, shift) of every image group, for
    linear_rows(affine=...) - the normalised tensor is never written."""
    lib = _lib.get()
    _check(x0, gamma, beta)
    c0 = x0.shape[-1]
    d = AaGroupNorm()
    d.x0, d.x1, d.gamma, d.beta, d.y = _ptr(x0), None, _ptr(gamma), _ptr(beta), None
    d.c0, d.c1 = c0, 0
    d.n_groups_img, d.tokens_per_group, d.num_groups = n_groups_img, tokens_per_group, num_groups
    d.silu, d.dtype, d.eps = 0, _DT[x0.dtype], eps
    nbytes = lib.aa_groupnorm_workspace(C.byref(d))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x0.device)
    coef = torch.empty(n_groups_img, 2, c0, dtype=torch.float32, device=x0.device)
    _run(lib.aa_groupnorm_coef, C.byref(d), _ptr(ws), nbytes, _ptr(coef), _stream(x0))
    return coef


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    lib = _lib.get()
    _check(x, gamma, beta)
    y = torch.empty_like(x)
    _run(lib.aa_layernorm, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), x.shape[0], x.shape[1], eps,
         _DT[x.dtype], _stream(x))
    return y


# ------------------------------------------------------------------------------------- attention
def _operand(t: torch.Tensor, col0: int, outer_stride: int, inner_stride: int, pos_stride: int, outer_div: int = 1):
    o = AaAttnOperand()
    o.ptr, o.ld, o.col0 = _ptr(t), t.stride(0), col0
    o.outer_stride, o.inner_stride, o.pos_stride, o.outer_div = outer_stride, inner_stride, pos_stride, outer_div
    return o


def attention(q: torch.Tensor, q_col0: int, k: torch.Tensor, k_col0: int, v: torch.Tensor, v_col0: int,
              heads: int, n_outer: int, n_inner: int, q_len: int, kv_len: int,
              q_strides, kv_strides, kv_outer_div: int = 1, scale: Optional[float] = None, head_dim: int = 64,
              kv_seq_mod: int = 0, causal: bool = False) -> torch.Tensor:
    """softmax(q k^T * scale) v for every (outer, inner, head).  `*_strides` = (outer, inner, pos)
    row strides of the token matrices; output rows use the q addressing, columns [0, heads*64)."""
    lib = _lib.get()
    for t in (q, k, v):                                   # operands may be column slices of wider matrices (ld = stride(0))
        if (not t.is_cuda and not _lib.host_pointers_ok()) or t.stride(-1) != 1:
            raise RuntimeError("attention: operands must be GPU tensors with contiguous rows")
    out = torch.empty(q.shape[0], heads * head_dim, dtype=q.dtype, device=q.device)
    d = AaAttention()
    d.q = _operand(q, q_col0, *q_strides)
    d.k = _operand(k, k_col0, *kv_strides, outer_div=kv_outer_div)
    d.v = _operand(v, v_col0, *kv_strides, outer_div=kv_outer_div)
    d.o = _operand(out, 0, *q_strides)
    d.k.seq_mod = d.v.seq_mod = kv_seq_mod       # > 0: sequence number n reads K / V table entry n % kv_seq_mod
    d.causal = int(causal)                       # key position > query position masked (CLIP text encoder)
    d._pad = ATTN_FLAGS
    d.n_outer, d.n_inner, d.heads, d.head_dim = n_outer, n_inner, heads, head_dim
    d.q_len, d.kv_len, d.dtype = q_len, kv_len, _DT[q.dtype]
    d.scale = float(head_dim) ** -0.5 if scale is None else scale
    _run(lib.aa_attention, C.byref(d), _stream(q))
    return out


# ------------------------------------------------------------------------------------- short-sequence self-attention with the projections inside
SEQ_ATTN_DEBUG = 0     # profiling only: AaSeqSelfAttn.flags (timing ablations)


@dataclass
class SeqQKV:
    """Operands of aa_seq_self_attention: `w` [heads, 3, 64, C] (per head the to_k rows, the permuted to_v rows, the to_q rows; a
    LayerNorm's gamma folded in), `bias` fp32 [heads, 3, 64] = W beta of that LayerNorm (None without one), its epsilon."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    ln_eps: float


def pack_seq_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, ln=None) -> SeqQKV:
    """to_q / to_k / to_v [heads * 64, C] (no bias: diffusers Attention) -> the operands of aa_seq_self_attention.  The to_v rows of a
    head go in the order 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) per 32-row half (a lane's 16 output registers are then 16
    consecutive channels).  `ln` = (gamma, beta, eps) folds the LayerNorm in front: LN(x) W^T = x~ W'^T + W beta with x~ the
    normalised rows (the kernel's part), W' = W diag(gamma) rounded to the storage type, the bias in fp32 (cf. pack_weight(ln=...))."""
    n, c = wq.shape
    assert n % 64 == 0 and wk.shape == (n, c) and wv.shape == (n, c)
    i = torch.arange(32, device=wq.device)
    perm = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
    perm64 = torch.cat([perm, perm + 32])
    heads = n // 64

    def order(t):                                                     # [3, n, ...] (q, k, v) -> [heads, 3, 64, ...] (k, permuted v, q)
        q, k, v = (t[j].reshape((heads, 64) + tuple(t.shape[2:])) for j in range(3))
        return torch.stack([k, v[:, perm64], q], dim=1).contiguous()

    w3 = torch.stack([wq.detach(), wk.detach(), wv.detach()])
    bias, eps = None, 0.0
    if ln is not None:
        gamma, beta, eps = ln
        wf = w3.float()
        bias = order(wf @ beta.detach().float())                      # fp32 [heads, 3, 64]
        w3 = (wf * gamma.detach().float()[None, None, :]).to(wq.dtype)
    return SeqQKV(order(w3), bias, float(eps))


@dataclass
class SeqPre:
    """A linear layer run INSIDE aa_seq_self_attention in front of the normalisation (AaSeqSelfAttn.pre_w): `w` [C / 64, 64, C] = its rows
    in 64-row passes, each 32-row half in the order 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); `bias` fp32 [C] in the same order."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]


def pack_seq_pre(weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> SeqPre:
    """nn.Linear(C, C) (diffusers Transformer*Model.proj_in, Attention.to_out[0]) -> the `pre` operand of seq_self_attention."""
    n, c = weight.shape
    assert n == c and n % 64 == 0
    i = torch.arange(32, device=weight.device)
    perm = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
    rows = (torch.arange(0, n, 32, device=weight.device)[:, None] + perm[None, :]).reshape(-1)
    b = None if bias is None else bias.detach().float()[rows].contiguous()
    return SeqPre(weight.detach()[rows].reshape(n // 64, 64, c).contiguous(), b)


def _seq_self_attn_desc(x, pk, out, n_outer, n_inner, seq_len, strides, scale):
    d = AaSeqSelfAttn()
    d.x, d.w, d.w_bias, d.o = _ptr(x), _ptr(pk.w), _ptr(pk.bias), _ptr(out)
    d.outer_stride, d.inner_stride, d.pos_stride = strides
    last = (n_outer - 1) * strides[0] + (n_inner - 1) * strides[1] + (seq_len - 1) * strides[2]
    d.x_bytes, d.o_bytes = (last + 1) * x.stride(0) * 2, (last + 1) * out.stride(0) * 2
    d.n_outer, d.n_inner, d.seq_len, d.channels = n_outer, n_inner, seq_len, x.shape[1]
    d.ldx, d.ldo = x.stride(0), out.stride(0)
    d.normalize, d.ln_eps, d.scale, d.dtype = int(pk.ln_eps > 0.0), float(pk.ln_eps), float(scale), _DT[x.dtype]
    d.flags = SEQ_ATTN_DEBUG
    return d


def seq_self_attention_ok(channels: int, seq_len: int, rows: int, dtype) -> bool:
    """Does aa_seq_self_attention cover a [rows, channels] token matrix with sequences of `seq_len` positions?"""
    d = AaSeqSelfAttn()
    d.channels, d.seq_len, d.n_outer, d.n_inner, d.ldx, d.ldo = channels, seq_len, 1, 1, channels, channels
    d.x_bytes = d.o_bytes = rows * channels * 2
    d.dtype = _DT.get(dtype, -1)
    return bool(_lib.get().aa_seq_self_attention_ok(C.byref(d)))


def seq_self_attention(x: torch.Tensor, pk: SeqQKV, n_outer: int, n_inner: int, seq_len: int, strides, scale: Optional[float] = None,
                       pre: Optional[SeqPre] = None, residual: Optional[torch.Tensor] = None):
    """softmax(q k^T * scale) v with [q | k | v] = LayerNorm(x) W^T for every (outer, inner) sequence and head (head_dim 64), in one
    kernel; `pk` from pack_seq_qkv (with the LayerNorm folded in, or without one), `strides` = (outer, inner, pos) row strides of x
    (and of the result, [rows, C]).
    `pre` (pack_seq_pre): the same kernel first runs x' = x W_pre^T + b_pre (+ residual), writes it and attends over LayerNorm(x') -
    the projection in front of the block (proj_in) or the previous attention layer's to_out + residual.  Returns (o, x') then."""
    lib = _lib.get()
    _check(x, pk.w, pk.bias, residual)
    if pk.w.dtype != x.dtype or pk.w.shape[-1] != x.shape[1]:
        raise RuntimeError("seq_self_attention: weights were packed for another dtype / width")
    out = torch.empty(x.shape[0], x.shape[1], dtype=x.dtype, device=x.device)
    d = _seq_self_attn_desc(x, pk, out, n_outer, n_inner, seq_len, strides, 64.0 ** -0.5 if scale is None else scale)
    x_pre = None
    if pre is not None:
        _check(pre.w, pre.bias)
        if pre.w.dtype != x.dtype or pre.w.shape[-1] != x.shape[1]:
            raise RuntimeError("seq_self_attention: the projection in front was packed for another dtype / width")
        x_pre = torch.empty_like(out)
        d.pre_w, d.pre_bias, d.pre_residual, d.pre_out = _ptr(pre.w), _ptr(pre.bias), _ptr(residual), _ptr(x_pre)
        d.ld_res, d.ld_pre = 0 if residual is None else residual.stride(0), x_pre.stride(0)
    elif residual is not None:
        raise RuntimeError("seq_self_attention: `residual` belongs to the projection in front (`pre`)")
    _run(lib.aa_seq_self_attention, C.byref(d), _stream(x))
    return out if pre is None else (out, x_pre)


# ------------------------------------------------------------------------------------- FeedForward + proj_out in one kernel
FF_FUSED_DEBUG = 0     # profiling only: AaFFFused.flags (timing ablations)


FF_STAGE_BYTES, FF_STAGES = 41984, 65          # include/aa_mi355.h: AA_FF_STAGE_BYTES, AA_FF_STAGES


@dataclass
class FFFused:
    """Operand of aa_ff_fused (include/aa_mi355.h): `w` = the weight stream, [65, 41984 / 2] of the storage type - the LDS stage images of
    GEGLU.proj (LayerNorm folded in), Wp W2 and Wp in the order the kernel consumes them, biases as (hi, lo) rows; see pack_ff_fused."""
    w: torch.Tensor
    ln_eps: float
    normalize: bool


def _bias_hi_lo(b: Optional[torch.Tensor], n: int, dtype, device) -> torch.Tensor:
    """fp32 bias [n] (None: zeros) -> [n, 8] of the storage type: (hi, lo, 0 x 6) with hi = round(b), lo = round(b - hi) - the A operand of the
    bias k-slice of aa_ff_fused (the matrix pipe adds hi * 1 + lo * 1 in fp32: the bias to 2^-22 (fp16) / 2^-16 (bf16) relative)."""
    out = torch.zeros(n, 8, dtype=dtype, device=device)
    if b is not None:
        hi = b.to(dtype)
        out[:, 0], out[:, 1] = hi, (b - hi.float()).to(dtype)
    return out


def _ff_stage_image(block: torch.Tensor, bias8: Optional[torch.Tensor], k_chunks: bool) -> torch.Tensor:
    """One LDS stage image [41984 / 2]: `block` = [64, 320] (64 weight rows x all K: five K chunks of 64) or [320, 64] (all rows x 64 K: five
    row chunks of 64); every chunk [64 rows][8 slots][8] with slot s of row r holding source slot s ^ ((r >> 1) & 7); then `bias8` [64, 8]
    (zeros if None)."""
    dev = block.device
    chunks = block.reshape(64, 5, 64).permute(1, 0, 2) if k_chunks else block.reshape(5, 64, 64)       # [5][64 rows][64 K]
    r = torch.arange(64, device=dev)[:, None]
    src_slot = torch.arange(8, device=dev)[None, :] ^ ((r >> 1) & 7)                                    # [64, 8]
    sw = torch.gather(chunks.reshape(5, 64, 8, 8), 2, src_slot[None, :, :, None].expand(5, 64, 8, 8))
    b = torch.zeros(64, 8, dtype=block.dtype, device=dev) if bias8 is None else bias8
    return torch.cat([sw.reshape(-1), b.reshape(-1)])


def pack_ff_fused(w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor],
                  wp: Optional[torch.Tensor] = None, bp: Optional[torch.Tensor] = None, ln=None) -> FFFused:
    """diffusers FeedForward = GEGLU.proj (w1 [8 C, C]: value rows, then gate rows; b1) -> value * gelu(gate) -> Linear (w2 [C, 4 C], b2), `+ x`
    behind it, and the transformer wrapper's proj_out (wp [C, C], bp; None: identity) -> the weight stream of ff_fused:
        out = [h | x] [Wp W2 | Wp]^T + (Wp b2 + bp) + outer.
    The product Wp W2 is formed in fp32 and rounded once (layers._MergedTail checks that this is harmless for the weights at hand).
    `ln` = (gamma, beta, eps) folds the LayerNorm in front of GEGLU.proj as pack_weight(ln=...) does."""
    c = w1.shape[1]
    hid = w2.shape[1]
    assert c == 320 and w1.shape == (2 * hid, c) and w2.shape == (c, hid) and hid == 4 * c
    dt, dev = w1.dtype, w1.device
    w1f = w1.detach().float()
    b1f = None if b1 is None else b1.detach().float()
    eps, normalize = 0.0, False
    if ln is not None:
        gamma, beta, eps = ln
        extra = w1f @ beta.detach().float()
        b1f = extra if b1f is None else b1f + extra
        w1f = w1f * gamma.detach().float()[None, :]
        normalize = True
    # W1 chunk ch of 64 rows: value rows of hidden units 32 ch .. + 31, then their gate rows
    idx = torch.arange(hid, device=dev).reshape(hid // 32, 32)
    rows1 = torch.cat([idx, idx + hid], dim=1).reshape(-1)
    w1p = w1f[rows1].to(dt)
    b1p = _bias_hi_lo(None if b1f is None else b1f[rows1], 2 * hid, dt, dev)
    wpf = torch.eye(c, device=dev) if wp is None else wp.detach().float()
    wh = wpf @ w2.detach().float()                                                  # [C, 4 C]
    bm = None
    if b2 is not None or bp is not None:
        bm = (0.0 if bp is None else bp.detach().float()) + (0.0 if b2 is None else wpf @ b2.detach().float())
    i = torch.arange(32, device=dev)
    perm = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
    rows = (torch.arange(0, c, 32, device=dev)[:, None] + perm[None, :]).reshape(-1)
    # hidden axis: position 32 cc + 16 h + 8 t + e of every 64 holds unit 32 cc + (e & 3) + 4 h + 8 (e >> 2) + 16 t (the order in which a lane's
    # accumulator registers of value * gelu(gate) become the B operand of the second product)
    pz = torch.arange(64, device=dev)
    cc, hh, tt, ee = pz >> 5, (pz >> 4) & 1, (pz >> 3) & 1, pz & 7
    unit = 32 * cc + (ee & 3) + 4 * hh + 8 * (ee >> 2) + 16 * tt
    cols = (torch.arange(0, hid, 64, device=dev)[:, None] + unit[None, :]).reshape(-1)
    whp, wxp = wh[rows][:, cols].to(dt), wpf[rows].to(dt)
    bmp = _bias_hi_lo(None if bm is None else bm[rows], c, dt, dev)
    stage_x = lambda n: _ff_stage_image(wxp[64 * n:64 * n + 64], bmp[64 * n:64 * n + 64], True)
    stage_w1 = lambda ch: _ff_stage_image(w1p[64 * ch:64 * ch + 64], b1p[64 * ch:64 * ch + 64], True)
    stage_h = lambda g: _ff_stage_image(whp[:, 64 * g:64 * g + 64], None, False)
    npair = hid // 64
    seq = [stage_x(n) for n in range(c // 64)] + [stage_w1(0), stage_w1(1)]
    for g in range(1, npair):
        seq += [stage_w1(2 * g), stage_h(g - 1), stage_w1(2 * g + 1)]
    seq.append(stage_h(npair - 1))
    w = torch.stack(seq).contiguous()
    assert w.shape == (FF_STAGES, FF_STAGE_BYTES // 2)
    return FFFused(w, float(eps), normalize)


def _ff_fused_desc(x, pk, outer, out):
    d = AaFFFused()
    d.x, d.outer, d.out, d.w = _ptr(x), _ptr(outer), _ptr(out), _ptr(pk.w)
    d.rows, d.channels = x.shape[0], x.shape[1]
    d.ldx, d.ld_outer, d.ldo = x.stride(0), 0 if outer is None else outer.stride(0), out.stride(0)
    d.normalize, d.ln_eps, d.dtype, d.flags = int(pk.normalize), pk.ln_eps, _DT[x.dtype], FF_FUSED_DEBUG
    return d


def ff_fused_ok(channels: int, rows: int, dtype) -> bool:
    d = AaFFFused()
    d.rows, d.channels, d.ldx, d.ld_outer, d.ldo, d.dtype = rows, channels, channels, channels, channels, _DT.get(dtype, -1)
    return bool(_lib.get().aa_ff_fused_ok(C.byref(d)))


def ff_fused(x: torch.Tensor, pk: FFFused, outer: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[GEGLU(LayerNorm(x) W1^T + b1) | x] Wm^T + bm + outer in one kernel (pk from pack_ff_fused): the FeedForward of a transformer block, its
    residual, the wrapper's proj_out and the wrapper's residual; the [rows, 4 C] activation is never written."""
    lib = _lib.get()
    _check(x, outer, pk.w, out)
    if pk.w.dtype != x.dtype or x.shape[1] != 320:
        raise RuntimeError("ff_fused: weights were packed for another dtype / width")
    if out is None:
        out = torch.empty(x.shape[0], x.shape[1], dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != x.dtype or out.stride(1) != 1:
        raise RuntimeError("ff_fused: `out` must have x's shape and dtype")
    d = _ff_fused_desc(x, pk, outer, out)
    _run(lib.aa_ff_fused, C.byref(d), _stream(x))
    return out


# ------------------------------------------------------------------------------------- K = C projections with the rows in registers
LR_STAGE_BYTES = 21504                          # include/aa_mi355.h: AA_LR_STAGE_BYTES
LINEAR_ROWS_DEBUG = int(os.environ.get("AA_LINEAR_ROWS_DEBUG", "0"))      # AaLinearRows.flags (1: plan for a 2-CU chip, 2: no stage split of the last round)


@dataclass
class LinearRows:
    """Operand of aa_linear_rows: `w` = the weight stream [n_out / 32, 21504 / 2] of the storage type (see pack_linear_rows)."""
    w: torch.Tensor
    n_out: int
    ln_eps: float
    normalize: bool


def pack_linear_rows(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, ln=None) -> LinearRows:
    """nn.Linear(320, n_out) [+ the LayerNorm (gamma, beta, eps) in front of it] -> the weight stream of linear_rows: per 32 output channels the
    weight rows in the order 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) (a lane's accumulator registers are then 16 consecutive channels),
    five [32][64] K chunks with the kernel's XOR swizzle applied, the biases as (hi, lo) rows (_bias_hi_lo)."""
    n, c = weight.shape
    assert c == 320 and n % 32 == 0
    dt, dev = weight.dtype, weight.device
    wf = weight.detach().float()
    bf = None if bias is None else bias.detach().float()
    eps, normalize = 0.0, False
    if ln is not None:
        gamma, beta, eps = ln
        extra = wf @ beta.detach().float()
        bf = extra if bf is None else bf + extra
        wf = wf * gamma.detach().float()[None, :]
        normalize = True
    i = torch.arange(32, device=dev)
    perm = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
    rows = (torch.arange(0, n, 32, device=dev)[:, None] + perm[None, :]).reshape(-1)
    wp = wf[rows].to(dt).reshape(n // 32, 32, 5, 8, 8)                       # [stage][row][K chunk][slot][8]
    r = torch.arange(32, device=dev)[:, None]
    src_slot = torch.arange(8, device=dev)[None, :] ^ ((r >> 1) & 7)           # [32, 8]: slot s of row r holds source slot s ^ ((r >> 1) & 7)
    sw = torch.gather(wp.permute(0, 2, 1, 3, 4), 3, src_slot[None, None, :, :, None].expand(n // 32, 5, 32, 8, 8))     # [stage][chunk][row][slot][8]
    b8 = _bias_hi_lo(None if bf is None else bf[rows], n, dt, dev).reshape(n // 32, 32 * 8)
    pad = torch.zeros(n // 32, 256, dtype=dt, device=dev)
    w = torch.cat([sw.reshape(n // 32, -1), b8, pad], dim=1).contiguous()
    assert w.shape == (n // 32, LR_STAGE_BYTES // 2)
    return LinearRows(w, n, float(eps), normalize)


def linear_rows_ok(channels: int, n_out: int, rows: int, dtype, rows_per_group: int = 0) -> bool:
    """`rows_per_group` > 0: with a GroupNorm's coefficients in front (linear_rows(affine=...))."""
    d = AaLinearRows()
    d.rows, d.channels, d.n_out, d.ldx, d.ld_res, d.ldo, d.dtype = rows, channels, n_out, channels, n_out, n_out, _DT.get(dtype, -1)
    if rows_per_group:
        d.row_affine, d.rows_per_group = 16, rows_per_group          # (any non-null pointer: the check reads the geometry only)
    return bool(_lib.get().aa_linear_rows_ok(C.byref(d)))


def linear_rows(x: torch.Tensor, pk: LinearRows, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, affine=None) -> torch.Tensor:
    """[LayerNorm](x) W^T + b (+ residual) for 320-channel token rows, rows in registers (pk from pack_linear_rows).
    `affine` = (groupnorm_coef(...), rows per image group): the GroupNorm in front of this layer, applied to the rows in the registers."""
    lib = _lib.get()
    _check(x, residual, pk.w, out)
    if pk.w.dtype != x.dtype or x.shape[1] != 320:
        raise RuntimeError("linear_rows: weights were packed for another dtype / width")
    if out is None:
        out = torch.empty(x.shape[0], pk.n_out, dtype=x.dtype, device=x.device)
    d = AaLinearRows()
    d.x, d.residual, d.out, d.w = _ptr(x), _ptr(residual), _ptr(out), _ptr(pk.w)
    d.rows, d.channels, d.n_out = x.shape[0], x.shape[1], pk.n_out
    d.ldx, d.ld_res, d.ldo = x.stride(0), 0 if residual is None else residual.stride(0), out.stride(0)
    d.normalize, d.ln_eps, d.dtype, d.flags = int(pk.normalize), pk.ln_eps, _DT[x.dtype], LINEAR_ROWS_DEBUG
    if affine is not None:
        coef, per = affine
        if coef.dtype != torch.float32 or not coef.is_contiguous() or coef.shape[-1] != x.shape[1] or coef.shape[0] * per < x.shape[0] or coef.device != x.device:
            raise RuntimeError("linear_rows: `affine` must be groupnorm_coef's fp32 [groups, 2, channels] covering every row")
        d.row_affine, d.rows_per_group = _ptr(coef), per
    _run(lib.aa_linear_rows, C.byref(d), _stream(x))
    return out


def softmax_rows(x: torch.Tensor, dtype, cols: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row softmax of fp32 scores over the first `cols` columns; `out` may be wider (its padding is
    left untouched)."""
    lib = _lib.get()
    _check(x, out)
    assert x.dtype == torch.float32
    cols = x.shape[1] if cols is None else cols
    y = torch.empty(x.shape, dtype=dtype, device=x.device) if out is None else out
    _run(lib.aa_softmax_rows, _ptr(x), _ptr(y), x.shape[0], cols, x.stride(0), y.stride(0), _DT[y.dtype], _stream(x))
    return y


def cfg_dpm_step(eps_uncond, eps_text, latents, x0_prev, latents_lp, guidance, sigma_s, alpha_s, c_x, c_d0, c_d1):
    lib = _lib.get()
    _check(eps_uncond, eps_text, latents, x0_prev, latents_lp)
    d = AaDpmStep()
    d.eps_uncond, d.eps_text = _ptr(eps_uncond), _ptr(eps_text)
    d.latents, d.x0_prev, d.latents_lp = _ptr(latents), _ptr(x0_prev), _ptr(latents_lp)
    d.n = latents.numel()
    d.guidance, d.sigma_s, d.alpha_s, d.c_x, d.c_d0, d.c_d1 = guidance, sigma_s, alpha_s, c_x, c_d0, c_d1
    d.dtype = _DT[eps_uncond.dtype]
    _run(lib.aa_cfg_dpm_step, C.byref(d), _stream(latents))


# ------------------------------------------------------------------------------------- step glue
def timestep_embedding(t: torch.Tensor, dim: int, dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) of the fp32 DEVICE array `t` [n] -> [n, dim]."""
    lib = _lib.get()
    _check(t, out)
    assert t.dtype == torch.float32 and t.dim() == 1
    y = torch.empty(t.shape[0], dim, dtype=dtype, device=t.device) if out is None else out
    _run(lib.aa_timestep_embedding, _ptr(t), _ptr(y), t.shape[0], dim, _DT[y.dtype], _stream(t))
    return y


def pack_latents(sample: torch.Tensor, cond: torch.Tensor, mask: Optional[torch.Tensor], batch: int, dtype,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sample [Bs,C,T,h,w] (fp32 or `dtype`), cond [Bc,C,1,h,w], mask [Bm,1,1,h,w] or None -> tokens [batch*(T+1)*h*w, 8]."""
    lib = _lib.get()
    _check(sample, cond, mask, out)
    bs, c, frames, h, w = sample.shape
    y = torch.empty(batch * (frames + 1) * h * w, 8, dtype=dtype, device=sample.device) if out is None else out
    d = AaPackLatents()
    d.sample, d.cond, d.mask, d.out = _ptr(sample), _ptr(cond), _ptr(mask), _ptr(y)
    d.batch, d.sample_batch, d.cond_batch = batch, bs, cond.shape[0]
    d.mask_batch = 0 if mask is None else mask.shape[0]
    d.channels, d.frames, d.hw = c, frames, h * w
    d.dtype, d.sample_dtype = _DT[dtype], _DT[sample.dtype]
    assert cond.dtype == dtype and (mask is None or mask.dtype == dtype)
    _run(lib.aa_pack_latents, C.byref(d), _stream(sample))
    return y


def cfg_dpm_step_tokens(eps_tokens, latents, x0_prev, latents_lp, guidance, coeffs, next_t=None, next_t_value=0.0):
    """Fused CFG + DPM-Solver++ update reading the UNet's token-layout output [(2*)clips*(T+1)*hw, eps_ld]."""
    lib = _lib.get()
    _check(eps_tokens, latents, x0_prev, latents_lp, next_t)
    clips, c, frames, h, w = latents.shape
    d = AaDpmStepTok()
    d.eps_tokens, d.latents, d.x0_prev, d.latents_lp = _ptr(eps_tokens), _ptr(latents), _ptr(x0_prev), _ptr(latents_lp)
    d.next_t, d.next_t_count, d.next_t_value = _ptr(next_t), 0 if next_t is None else next_t.numel(), float(next_t_value)
    d.clips, d.channels, d.frames, d.hw = clips, c, frames, h * w
    d.eps_ld = eps_tokens.stride(0)
    d.guidance_on, d.guidance = int(guidance is not None), float(guidance or 0.0)
    d.sigma_s, d.alpha_s, d.c_x, d.c_d0, d.c_d1 = (coeffs[k] for k in ("sigma_s", "alpha_s", "c_x", "c_d0", "c_d1"))
    d.dtype = _DT[eps_tokens.dtype]
    assert latents.dtype == torch.float32 and x0_prev.dtype == torch.float32
    _run(lib.aa_cfg_dpm_step_tokens, C.byref(d), _stream(latents))


# ------------------------------------------------------------------------------------- Stable-Video-Diffusion glue
def blend(x: torch.Tensor, y: Optional[torch.Tensor] = None, a: float = 1.0, b: float = 1.0,
          rowvec: Optional[torch.Tensor] = None, rowvec_div: int = 1, rowvec_mod: int = 0, act: int = AA_ACT_NONE,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(a*x + b*y + rowvec[(row // rowvec_div) % rowvec_mod]) on [rows, C] token matrices (diffusers AlphaBlender etc.)."""
    lib = _lib.get()
    _check(x, y, out)
    if rowvec is not None and ((not rowvec.is_cuda and not _lib.host_pointers_ok()) or rowvec.stride(-1) != 1):
        raise RuntimeError("blend: rowvec must be a GPU tensor with contiguous rows")
    o = torch.empty_like(x) if out is None else out
    d = AaBlend()
    d.x, d.y, d.rowvec, d.out = _ptr(x), _ptr(y), _ptr(rowvec), _ptr(o)
    d.rows, d.channels = x.shape[0], x.shape[1]
    d.rowvec_div, d.rowvec_mod = rowvec_div, rowvec_mod
    d.rowvec_ld = 0 if rowvec is None or rowvec.dim() < 2 else rowvec.stride(0)
    d.a, d.b, d.act, d.dtype = float(a), float(b), act, _DT[x.dtype]
    _run(lib.aa_blend, C.byref(d), _stream(x))
    return o


def pack_frames(sources, batch: int, dtype, scale: Optional[torch.Tensor] = None, scaled_src: int = -1,
                out_channels: int = 16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cat(sources, dim=2) of [Bs, F, Cs, h, w] tensors (fp32 or `dtype`) -> tokens [batch*F*h*w, out_channels];
    source `scaled_src` is multiplied by the DEVICE scalar `scale` (EulerDiscreteScheduler.scale_model_input)."""
    lib = _lib.get()
    srcs = [s_ for s_ in sources if s_ is not None]
    _check(*srcs, scale, out)
    assert 1 <= len(srcs) <= 3
    _, frames, _, h, w = srcs[0].shape
    y = torch.empty(batch * frames * h * w, out_channels, dtype=dtype, device=srcs[0].device) if out is None else out
    d = AaPackFrames()
    k = 0
    for i, s_ in enumerate(sources):
        if s_ is None:
            continue
        assert s_.shape[1] == frames and s_.shape[3:] == (h, w) and s_.dtype in (torch.float32, dtype)
        d.src[k], d.src_channels[k], d.src_batch[k], d.src_f32[k] = _ptr(s_), s_.shape[2], s_.shape[0], int(s_.dtype == torch.float32)
        if i == scaled_src:
            d.scaled_src = k
        k += 1
    if scaled_src < 0 or scale is None:
        d.scaled_src = -1
    d.scale, d.out = _ptr(scale), _ptr(y)
    d.batch, d.frames, d.hw, d.out_channels, d.dtype = batch, frames, h * w, out_channels, _DT[dtype]
    _run(lib.aa_pack_frames, C.byref(d), _stream(srcs[0]))
    return y


def cfg_euler_step_tokens(v_tokens, latents, guidance, c_x, c_v, next_t=None, next_t_value=0.0, next_scale=None,
                          next_scale_value=1.0):
    """Per-frame guidance + Euler (v-prediction) update x' = c_x x + c_v v, reading the UNet's token-layout output;
    latents fp32 [clips, F, C, h, w] updated in place; `guidance` DEVICE fp32 [F] or None."""
    lib = _lib.get()
    _check(v_tokens, latents, guidance, next_t, next_scale)
    clips, frames, c, h, w = latents.shape
    assert latents.dtype == torch.float32 and (guidance is None or (guidance.dtype == torch.float32 and guidance.numel() == frames))
    d = AaEulerStepTok()
    d.v_tokens, d.latents, d.guidance = _ptr(v_tokens), _ptr(latents), _ptr(guidance)
    d.next_t, d.next_t_count, d.next_t_value = _ptr(next_t), 0 if next_t is None else next_t.numel(), float(next_t_value)
    d.next_scale, d.next_scale_value = _ptr(next_scale), float(next_scale_value)
    d.clips, d.channels, d.frames, d.hw, d.ld = clips, c, frames, h * w, v_tokens.stride(0)
    d.c_x, d.c_v, d.dtype = float(c_x), float(c_v), _DT[v_tokens.dtype]
    _run(lib.aa_cfg_euler_step_tokens, C.byref(d), _stream(latents))
