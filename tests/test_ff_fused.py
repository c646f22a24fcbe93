"""aa_ff_fused (LayerNorm -> GEGLU.proj -> value * gelu(gate) -> Linear -> + x -> proj_out -> + outer in ONE kernel) against plain fp32 torch:
the operations diffusers' BasicTransformerBlock.norm3 / FeedForward and Transformer2DModel / TransformerTemporalModel.proj_out run
(reference models/unet_3d_blocks.py:287,446,681 and :379,526,759; oracle/layers.py).  Two backends as in test_kernels.py: the SIMT emulator
(index arithmetic, operand layouts, the weight ring) and the MI355X (`-m gpu`)."""
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


def reference(x, outer, w1, b1, w2, b2, wp, bp, gamma, beta, eps):
    """fp32 arithmetic on the stored operands, the layer order of diffusers (no merged weights)."""
    C = x.shape[1]
    xf = x.float()
    xn = xf if gamma is None else F.layer_norm(xf, (C,), gamma.float(), beta.float(), eps)
    y = F.linear(xn, w1.float(), None if b1 is None else b1.float())
    val, gate = y.chunk(2, dim=-1)
    h = val * F.gelu(gate)
    f = F.linear(h, w2.float(), None if b2 is None else b2.float()) + xf
    o = f if wp is None else F.linear(f, wp.float(), None if bp is None else bp.float())
    return o if outer is None else o + outer.float()


def run_case(dev, rows, dtype=torch.float16, ln=True, proj=True, with_outer=True, bias=True, seed=0, x_scale=1.0, x_mean=0.0, C=320):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (r(rows, C, sc=x_scale) + x_mean).to(dtype)
    outer = r(rows, C).to(dtype) if with_outer else None
    w1, w2 = r(8 * C, C, sc=C ** -0.5).to(dtype), r(C, 4 * C, sc=(4 * C) ** -0.5).to(dtype)
    b1, b2 = (r(8 * C, sc=0.2).to(dtype), r(C, sc=0.2).to(dtype)) if bias else (None, None)
    wp, bp = (r(C, C, sc=C ** -0.5).to(dtype), r(C, sc=0.2).to(dtype) if bias else None) if proj else (None, None)
    gamma, beta = ((1.0 + 0.3 * r(C)).to(dtype), (0.2 * r(C)).to(dtype)) if ln else (None, None)
    eps = 1e-5
    want = reference(x, outer, w1, b1, w2, b2, wp, bp, gamma, beta, eps)
    to = lambda t: None if t is None else t.to(dev)
    pk = ops.pack_ff_fused(to(w1), to(b1), to(w2), to(b2), to(wp), to(bp), ln=None if gamma is None else (to(gamma), to(beta), eps))
    assert ops.ff_fused_ok(C, rows, dtype)
    got = ops.ff_fused(to(x), pk, to(outer)).float().cpu()
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item(), want.abs().max().item()


@pytest.mark.parametrize("rows", [128, 300, 37])
def test_ff_fused(backend, rows):
    """One full tile; two tiles + a tail tile with an idle wave and a ragged one; a single ragged wave."""
    err, scale = run_case(backend, rows)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("ln,proj,with_outer,bias", [(False, True, True, True), (True, False, False, True), (True, True, True, False)])
def test_ff_fused_optional_operands(backend, ln, proj, with_outer, bias):
    """No LayerNorm in front (x as it is); no proj_out / outer residual (a bare FeedForward + residual); no biases anywhere."""
    err, scale = run_case(backend, 160, ln=ln, proj=proj, with_outer=with_outer, bias=bias, seed=3)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


def test_ff_fused_rows_far_from_zero_mean(backend):
    """Rows with mean 30 and unit spread: the statistics are taken in fp32 from the stored values, two passes (no E[x^2] - mean^2)."""
    err, scale = run_case(backend, 96, seed=5, x_mean=30.0)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


def test_ff_fused_bf16(backend):
    err, scale = run_case(backend, 200, dtype=torch.bfloat16, seed=7)
    assert err <= 6e-2 * max(1.0, scale), (err, scale)


def test_ff_fused_rejects_other_widths(backend):
    assert not ops.ff_fused_ok(640, 1024, torch.float16)
    assert not ops.ff_fused_ok(320, 1024, torch.float32)
