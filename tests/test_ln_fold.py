"""Folded LayerNorm at the module level: the product Transformer2DModel / TransformerTemporalModel (320 channels, the
64x64-level width) with every LayerNorm of the BasicTransformerBlock folded into the linear layer behind it
(layers.LN_FOLD_MIN_ROWS lowered so the small test sizes take the path the 139264-row level takes) against the oracle, and
against the same product modules running the stand-alone LayerNorm kernel.  Emulator on CPU, MI355X under -m gpu."""
import pytest
import torch

import oracle
from animate_anything_amd import layers as L
from util import rel_err, seeded_state


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    if request.param == "emu":
        request.getfixturevalue("emu")
        return "cpu"
    return "cuda"


def _tokens(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("temporal", [False, True])
def test_transformer_with_folded_layernorms(dev, temporal, monkeypatch):
    C, heads, clips, frames, h, w, text_len, text_dim = 320, 5, 2, 3, 4, 5, 7, 64
    torch.manual_seed(0)
    if temporal:
        ref = oracle.TransformerTemporalModel(heads, 64, C).eval()
        net = L.TransformerTemporalModel(heads, 64, C).eval()
    else:
        ref = oracle.Transformer2DModel(heads, 64, C, cross_attention_dim=text_dim).eval()
        net = L.Transformer2DModel(heads, 64, C, cross_attention_dim=text_dim).eval()
    state = seeded_state(ref)
    g = torch.Generator().manual_seed(3)
    for k in state:                                            # non-trivial LayerNorm / GroupNorm affine parameters
        if ".norm" in k or k.startswith("norm"):
            state[k] = state[k] + 0.3 * torch.randn(state[k].shape, generator=g)
    ref.load_state_dict(state)
    net.load_state_dict(state)
    net = net.half().to(dev)
    x = torch.randn(clips * frames, C, h, w, generator=g) * 1.5 + 0.7
    text = torch.randn(clips, text_len, text_dim, generator=g)
    grid = L.Grid(clips, frames, h, w)
    with torch.no_grad():
        if temporal:
            want = ref(x, num_frames=frames).sample
            run = lambda: net.tokens(_tokens(x).half().to(dev), grid)
        else:
            want = ref(x, text.repeat_interleave(frames, 0)).sample
            ttok = text.reshape(-1, text_dim).half().to(dev)
            run = lambda: net.tokens(_tokens(x).half().to(dev), grid, ttok, text_len)
        plain = run()                                          # stand-alone LayerNorm kernels (rows < LN_FOLD_MIN_ROWS)
        monkeypatch.setattr(L, "LN_FOLD_MIN_ROWS", 0)
        folded = run()
    want_tok = _tokens(want)
    assert rel_err(plain, want_tok) < 3e-2
    assert rel_err(folded, want_tok) < 3e-2
    assert not torch.equal(plain, folded)                      # a different arithmetic path really ran
