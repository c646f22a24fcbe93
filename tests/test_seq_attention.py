"""aa_seq_self_attention (LayerNorm -> to_q | to_k | to_v -> softmax(q k^T) v over short sequences in ONE kernel) against plain fp32
torch: F.layer_norm, F.linear, F.scaled_dot_product_attention - the operations diffusers' BasicTransformerBlock runs for the
temporal transformer (reference models/unet_3d_blocks.py:379,526,759; oracle/layers.py).  Two backends as in test_kernels.py:
the SIMT emulator (index arithmetic, operand layouts, the LDS exchange of K / V^T operand registers) and the MI355X (`-m gpu`)."""
import pytest
import torch
import torch.nn.functional as F

from animate_anything_amd import ops


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "emu":
        request.getfixturevalue("emu")
        yield "cpu"
    else:
        assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
        yield "cuda"


def reference(x, wq, wk, wv, gamma, beta, eps, clips, frames, hw):
    """x [clips * frames * hw, C] (token order clip, frame, pixel) -> attention over the frames of each pixel, fp32."""
    C = x.shape[1]
    heads = C // 64
    xn = x.float() if gamma is None else F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps).to(x.dtype).float()
    proj = lambda w: F.linear(xn, w.float()).to(x.dtype).float().reshape(clips, frames, hw, heads, 64).permute(0, 2, 3, 1, 4)   # [b, hw, h, T, d]
    o = F.scaled_dot_product_attention(proj(wq), proj(wk), proj(wv))
    return o.permute(0, 3, 1, 2, 4).reshape(-1, C)


def run_case(dev, C, clips, frames, hw, dtype=torch.float16, ln=True, seed=0, x_scale=1.0, x_mean=0.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (r(clips * frames * hw, C, sc=x_scale) + x_mean).to(dtype)
    wq, wk, wv = (r(C, C, sc=C ** -0.5).to(dtype) for _ in range(3))
    gamma, beta = ((1.0 + 0.3 * r(C)).to(dtype), (0.2 * r(C)).to(dtype)) if ln else (None, None)
    eps = 1e-5
    want = reference(x, wq, wk, wv, gamma, beta, eps, clips, frames, hw)
    to = lambda t: None if t is None else t.to(dev)
    pk = ops.pack_seq_qkv(to(wq), to(wk), to(wv), ln=None if gamma is None else (to(gamma), to(beta), eps))
    assert ops.seq_self_attention_ok(C, frames, x.shape[0], dtype)
    got = ops.seq_self_attention(to(x), pk, clips, hw, frames, (frames * hw, 1, hw))
    got, ref = got.float().cpu(), want
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    return err, scale


@pytest.mark.parametrize("C,clips,frames,hw", [(320, 2, 17, 15), (320, 1, 17, 37), (320, 2, 5, 9), (320, 1, 32, 3), (320, 3, 1, 50)])
def test_seq_self_attention(backend, C, clips, frames, hw):
    """17 frames x 15 pixels = exactly one tile; 37 pixels = two full tiles + a tail tile with idle waves; 5 / 32 / 1 frames:
    sequences that never / always / trivially straddle the 32-row blocks."""
    err, scale = run_case(backend, C, clips, frames, hw)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("C,frames,hw", [(512, 17, 16), (640, 17, 9)])
def test_seq_self_attention_other_widths(backend, C, frames, hw):
    """512 channels (transformer_in: stages of 256 K, two per pass) and 640 (four waves, one per SIMD)."""
    err, scale = run_case(backend, C, 1, frames, hw, seed=3)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


def test_seq_self_attention_without_layernorm_and_bf16(backend):
    err, scale = run_case(backend, 320, 1, 17, 20, ln=False, seed=5)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)
    err, scale = run_case(backend, 320, 1, 17, 20, dtype=torch.bfloat16, seed=6)
    assert err <= 6e-2 * max(1.0, scale), (err, scale)


def test_seq_self_attention_rows_with_large_mean(backend):
    """LayerNorm statistics on rows with |mean| >> std (the in-register two-pass form must not cancel)."""
    err, scale = run_case(backend, 320, 1, 17, 8, seed=7, x_scale=1.0, x_mean=50.0)
    assert err <= 2e-2 * max(1.0, scale), (err, scale)


def test_seq_self_attention_strided_output_matches_three_launch_form(backend):
    """Same inputs through the existing path: LayerNorm kernel, one Q|K|V contraction, aa_attention on strided rows."""
    dev, C, clips, frames, hw = backend, 320, 2, 17, 23
    g = torch.Generator().manual_seed(11)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().to(dev)
    x = r(clips * frames * hw, C)
    wq, wk, wv = r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5)
    gamma, beta = (1.0 + 0.3 * r(C).float()).half(), r(C, sc=0.2)
    fused = ops.seq_self_attention(x, ops.pack_seq_qkv(wq, wk, wv, ln=(gamma, beta, 1e-5)), clips, hw, frames, (frames * hw, 1, hw))
    qkv = ops.conv_gemm(ops.layernorm(x, gamma, beta, 1e-5), ops.pack_weight(torch.cat([wq, wk, wv])), ops.linear_geom(x.shape[0]))
    st = (frames * hw, 1, hw)
    three = ops.attention(qkv, 0, qkv, C, qkv, 2 * C, C // 64, clips, hw, frames, frames, st, st)
    err = (fused.float() - three.float()).abs().max().item()
    assert err <= 6e-3 * max(1.0, three.float().abs().max().item()), err


@pytest.mark.parametrize("C,clips,frames,hw,with_res", [(320, 1, 17, 20, True), (320, 2, 17, 9, False), (512, 1, 17, 10, True), (640, 1, 9, 11, True)])
def test_seq_self_attention_with_the_projection_in_front(backend, C, clips, frames, hw, with_res):
    """`pre`: x' = x W_pre^T + b_pre (+ residual) inside the kernel (proj_in in front of the first attention layer of a temporal transformer,
    the first layer's to_out + residual in front of the second): x' itself and the attention over LayerNorm(x') against fp32 torch."""
    dev, dtype = backend, torch.float16
    g = torch.Generator().manual_seed(21)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    rows = clips * frames * hw
    x, res = r(rows, C).to(dtype), (r(rows, C).to(dtype) if with_res else None)
    wp, bp = r(C, C, sc=C ** -0.5).to(dtype), r(C, sc=0.3).to(dtype)
    wq, wk, wv = (r(C, C, sc=C ** -0.5).to(dtype) for _ in range(3))
    gamma, beta = (1.0 + 0.3 * r(C)).to(dtype), (0.2 * r(C)).to(dtype)
    xp = F.linear(x.float(), wp.float(), bp.float()) + (0.0 if res is None else res.float())
    want_x = xp.to(dtype)
    want_o = reference(want_x, wq, wk, wv, gamma, beta, 1e-5, clips, frames, hw)
    to = lambda t: None if t is None else t.to(dev)
    pk = ops.pack_seq_qkv(to(wq), to(wk), to(wv), ln=(to(gamma), to(beta), 1e-5))
    o, x_pre = ops.seq_self_attention(to(x), pk, clips, hw, frames, (frames * hw, 1, hw), pre=ops.pack_seq_pre(to(wp), to(bp)), residual=to(res))
    ex = (x_pre.float().cpu() - xp).abs().max().item()
    assert ex <= 4e-3 * max(1.0, xp.abs().max().item()), ex                  # (one rounding to fp16)
    eo = (o.float().cpu() - want_o).abs().max().item()
    assert torch.isfinite(o.float()).all() and eo <= 1.5e-2 * max(1.0, want_o.abs().max().item()), eo
