"""aa_linear_rows ([LayerNorm](x) W^T + b (+ residual) over 320-channel token rows held in registers) against plain fp32 torch: F.layer_norm,
F.linear - the K = C projections of the 320-channel transformers (diffusers proj_in / to_q / to_out / the fused Q|K|V; reference
models/unet_3d_blocks.py:287,446,681 and :379,526,759).  Two backends as in test_kernels.py: the SIMT emulator and the MI355X (`-m gpu`)."""
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


def run_case(dev, rows, n_out, dtype=torch.float16, ln=True, with_res=True, bias=True, seed=0, x_mean=0.0, C=320):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (r(rows, C) + x_mean).to(dtype)
    res = r(rows, n_out).to(dtype) if with_res else None
    w, b = r(n_out, C, sc=C ** -0.5).to(dtype), (r(n_out, sc=0.3).to(dtype) if bias else None)
    gamma, beta = ((1.0 + 0.3 * r(C)).to(dtype), (0.2 * r(C)).to(dtype)) if ln else (None, None)
    xn = x.float() if gamma is None else F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    want = F.linear(xn, w.float(), None if b is None else b.float()) + (0.0 if res is None else res.float())
    to = lambda t: None if t is None else t.to(dev)
    pk = ops.pack_linear_rows(to(w), to(b), ln=None if gamma is None else (to(gamma), to(beta), 1e-5))
    assert ops.linear_rows_ok(C, n_out, rows, dtype)
    got = ops.linear_rows(to(x), pk, to(res)).float().cpu()
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item(), want.abs().max().item()


@pytest.mark.parametrize("rows,n_out", [(128, 320), (300, 320), (37, 960), (130, 32), (64, 64)])
def test_linear_rows(backend, rows, n_out):
    """One full tile; two tiles + a ragged one; the fused Q|K|V width on a ragged wave; one and two stages (the ring's start-up)."""
    err, scale = run_case(backend, rows, n_out)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("ln,with_res,bias", [(False, True, True), (True, False, True), (False, False, False)])
def test_linear_rows_optional_operands(backend, ln, with_res, bias):
    err, scale = run_case(backend, 160, 320, ln=ln, with_res=with_res, bias=bias, seed=3)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)


def test_linear_rows_far_from_zero_mean_and_bf16(backend):
    err, scale = run_case(backend, 96, 320, seed=5, x_mean=30.0)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)
    err, scale = run_case(backend, 200, 320, dtype=torch.bfloat16, seed=7)
    assert err <= 6e-2 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("rows,n_out", [(680, 320), (200, 960), (520, 64)])
def test_linear_rows_last_round_split_over_stages(backend, rows, n_out, monkeypatch):
    """The launch planned for a 2-CU chip (4 resident tiles): 6 tiles = one full round + 2 tiles split into 2 x 5 stages; 2 tiles of 30 stages as
    2 x 2 workgroups; 5 tiles of 2 stages = 4 + 1 x 2 - and the same bits as the unsplit launch."""
    monkeypatch.setattr(ops, "LINEAR_ROWS_DEBUG", 1)
    err, scale = run_case(backend, rows, n_out, seed=11)
    assert err <= 1e-2 * max(1.0, scale), (err, scale)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows, 320, generator=g).half().to(backend)
    pk = ops.pack_linear_rows((torch.randn(n_out, 320, generator=g) * 0.05).half().to(backend), None, ln=None)
    a = ops.linear_rows(x, pk)
    monkeypatch.setattr(ops, "LINEAR_ROWS_DEBUG", 2)
    assert torch.equal(a, ops.linear_rows(x, pk))


@pytest.mark.parametrize("groups_img,per,x_mean,dtype", [(3, 64, 0.0, torch.float16), (2, 160, 50.0, torch.float16), (5, 32, 3.0, torch.bfloat16)])
def test_linear_rows_with_a_groupnorm_in_front(backend, groups_img, per, x_mean, dtype):
    """GroupNorm(32 groups) -> Linear as ops.groupnorm_coef + ops.linear_rows(affine=...): against F.group_norm + F.linear in fp32 (rows far from
    zero mean included: the statistics are pivot-centred, the rows are normalised in fp32 before they are rounded) and against the two-pass
    form (ops.groupnorm, then ops.linear_rows) - the normalised rows are rounded identically, so the outputs agree to the last bit or two."""
    C, n_out, rows = 320, 320, groups_img * per
    g = torch.Generator().manual_seed(21)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (r(rows, C) * (1.0 + r(1, C).abs()) + x_mean + r(groups_img, 1, 1).repeat(1, per, 1).reshape(rows, 1)).to(dtype)
    w, b = r(n_out, C, sc=C ** -0.5).to(dtype), r(n_out, sc=0.3).to(dtype)
    gamma, beta = (1.0 + 0.3 * r(C)).to(dtype), (0.2 * r(C)).to(dtype)
    xi = x.float().reshape(groups_img, per, C).permute(0, 2, 1)                                        # [image group, C, tokens]
    xn = F.group_norm(xi, 32, gamma.float(), beta.float(), 1e-6).permute(0, 2, 1).reshape(rows, C)
    want = F.linear(xn, w.float(), b.float())
    dev = backend
    pk = ops.pack_linear_rows(w.to(dev), b.to(dev))
    coef = ops.groupnorm_coef(x.to(dev), gamma.to(dev), beta.to(dev), groups_img, per, 32, 1e-6)
    assert ops.linear_rows_ok(C, n_out, rows, dtype, per) and not ops.linear_rows_ok(C, n_out, rows, dtype, per + 8)
    got = ops.linear_rows(x.to(dev), pk, affine=(coef, per)).float().cpu()
    two = ops.linear_rows(ops.groupnorm(x.to(dev), gamma.to(dev), beta.to(dev), groups_img, per, 32, 1e-6), pk).float().cpu()
    tol = 1e-2 if dtype == torch.float16 else 6e-2
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol * scale
    assert (got - two).abs().max().item() <= (2e-3 if dtype == torch.float16 else 2e-2) * scale


def test_linear_rows_rejects_other_shapes(backend):
    assert not ops.linear_rows_ok(640, 640, 1024, torch.float16)
    assert not ops.linear_rows_ok(320, 330, 1024, torch.float16)
