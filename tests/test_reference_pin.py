"""Pin the oracle to the reference's OWN graph code.

`/root/reference/models/unet_3d_condition_mask.py` and `models/unet_3d_blocks.py` are imported unmodified, from where they
lie, on top of tests/refstub/diffusers (leaf classes -> oracle.layers).  Same state dict -> the reference's
`UNet3DConditionModel.forward` must equal `oracle.UNet3DConditionModel.forward` to fp32 round-off.  That ties block order,
skip wiring, frame-0 concat / drop, mask repeat order, embedding arithmetic and the `upsample_size` path of the oracle to the
reference itself; the arithmetic inside the leaves stays a restatement of diffusers 0.24.0 (absent here).
Needs /root/reference: runs in the build container, skips on the GPU box.
"""
import os
import sys

import pytest
import torch

from util import SMALL_UNET, TINY_UNET, seeded_state, unet_inputs

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref_mod():
    here = os.path.dirname(os.path.abspath(__file__))
    saved = list(sys.path)
    saved_mods = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.") or k == "diffusers" or k.startswith("diffusers.") or k == "utils" or k.startswith("utils.")}
    for k in saved_mods:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(here, "refstub"))
    sys.path.insert(0, REF)
    try:
        import models.unet_3d_condition_mask as m          # the reference file itself
        import models.unet_3d_blocks as blocks
        assert m.__file__.startswith(REF) and blocks.__file__.startswith(REF)
        yield m
    finally:
        sys.path[:] = saved
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "diffusers" or k.startswith("diffusers.")]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


def _pair(ref_mod, cfg):
    import oracle
    torch.manual_seed(0)
    orc = oracle.UNet3DConditionModel(**cfg).eval()
    state = seeded_state(orc)
    orc.load_state_dict(state)
    ref = ref_mod.UNet3DConditionModel(**cfg).eval()
    # identical key sets and shapes: Appendix C's layout is the reference's, not ours
    assert sorted(ref.state_dict().keys()) == sorted(state.keys())
    ref.load_state_dict(state, strict=True)
    return ref, orc


def _call(model, inp, mask="given", **kw):
    m = inp["mask"] if mask == "given" else None
    with torch.no_grad():
        out = model(inp["sample"], inp["t"], inp["text"], inp["cond"], m, motion=inp["motion"], **kw)
    return out.sample if hasattr(out, "sample") else out[0]


@pytest.mark.parametrize("cfg,kw", [
    (TINY_UNET, dict(b=2, frames=3, h=8, w=8, text_dim=64)),
    (TINY_UNET, dict(b=1, frames=2, h=6, w=10, text_dim=64, text_len=5)),
    (SMALL_UNET, dict(b=2, frames=3, h=8, w=8, text_dim=128)),
    (SMALL_UNET, dict(b=2, frames=2, h=11, w=14, text_dim=128)),        # 11 -> 6 -> 3 -> 2: the `upsample_size` path (real eval sizes: 55 x 74)
    (SMALL_UNET, dict(b=1, frames=4, h=7, w=9, text_dim=128)),
])
def test_reference_forward_equals_oracle_forward(ref_mod, cfg, kw):
    ref, orc = _pair(ref_mod, cfg)
    inp = unet_inputs(**kw)
    a, b = _call(ref, inp), _call(orc, inp)
    assert a.shape == b.shape == inp["sample"].shape
    assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())


def test_reference_forward_without_mask_uses_conv_in(ref_mod):
    """`mask is None` takes `conv_in` instead of `conv_in2` (unet_3d_condition_mask.py:424-431)."""
    ref, orc = _pair(ref_mod, TINY_UNET)
    inp = unet_inputs(b=2, frames=3, h=8, w=8, text_dim=64)
    a, b = _call(ref, inp, mask=None), _call(orc, inp, mask=None)
    assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    assert (a - _call(ref, inp)).abs().max().item() > 1e-3          # and the mask path really is a different function


def test_reference_return_dict_and_timestep_forms(ref_mod):
    """Python-number / 0-d tensor / [B] tensor timesteps and `return_dict=False` (unet_3d_condition_mask.py:391-405,523-526)."""
    ref, orc = _pair(ref_mod, TINY_UNET)
    inp = unet_inputs(b=2, frames=2, h=8, w=8, text_dim=64)
    base = _call(orc, inp)
    for t in (inp["t"], torch.tensor(inp["t"]), torch.tensor([inp["t"]] * 2)):
        i2 = dict(inp, t=t)
        assert (_call(ref, i2) - base).abs().max().item() <= 1e-5 * max(1.0, base.abs().max().item())
        assert (_call(orc, i2) - base).abs().max().item() <= 1e-5 * max(1.0, base.abs().max().item())
    tup = ref(inp["sample"], inp["t"], inp["text"], inp["cond"], inp["mask"], motion=inp["motion"], return_dict=False)
    assert isinstance(tup, tuple) and torch.equal(tup[0], _call(ref, inp))


def test_named_modules_order_is_the_reference_order(ref_mod):
    """LoRA files address layers by traversal index (utils/lora.py): the module registration order of the oracle must be the
    reference's - down_blocks, up_blocks, mid_block (unet_3d_condition_mask.py:171-172,202), ADVICE r02."""
    ref, orc = _pair(ref_mod, TINY_UNET)
    leaf = lambda m: [n for n, mod in m.named_modules() if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d))]
    assert leaf(ref) == leaf(orc)
    first = lambda names, pre: next(i for i, n in enumerate(names) if n.startswith(pre))
    assert first(leaf(ref), "down_blocks") < first(leaf(ref), "up_blocks") < first(leaf(ref), "mid_block")
