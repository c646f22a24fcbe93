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


# --------------------------------------------------------------------------------------------- pipeline loop, helpers
class _StepOut:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _SchedulerAdapter:
    """diffusers calling convention (`step(...).prev_sample`, `scale_model_input`, `set_timesteps(n, device=)`) around the
    oracle's DPM-Solver++ restatement."""
    order = 1

    def __init__(self):
        import oracle
        self.inner = oracle.DPMSolverMultistepScheduler()

    def set_timesteps(self, n, device=None):
        self.inner.set_timesteps(n)
        self.timesteps = self.inner.timesteps

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x, **kw):
        return _StepOut(self.inner.step(eps, t, x))


def test_reference_pipeline_call_equals_oracle_pipeline(ref_mod):
    """The reference's own `LatentToVideoPipeline.__call__` (models/pipeline.py:14-214: guidance batching, condition-latent
    duplication, the permute / reshape around `scheduler.step`, callback cadence) run on the stub base class with oracle parts
    == `oracle.LatentToVideoPipeline` on the same inputs, with and without guidance."""
    import oracle
    import models.pipeline as P                       # the reference file (sys.path set up by the ref_mod fixture)
    assert P.__file__.startswith(REF)
    torch.manual_seed(0)
    unet = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    unet.load_state_dict(seeded_state(unet))
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g)
    frames, h, w, steps = 3, 8, 8, 4
    x0, noise, pos, neg = r(1, 4, 1, h, w) * 0.5, r(1, 4, frames, h, w), r(1, 7, 64), r(1, 7, 64)
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., 2:6, 2:6] = 1
    for guidance in (9.0, 1.0):
        osched = oracle.DPMSolverMultistepScheduler()
        osched.set_timesteps(steps)
        init = oracle.ddpm_add_noise(x0.repeat(1, 1, frames, 1, 1), noise, int(osched.timesteps[0]))
        seen_o, seen_r = [], []
        _, want = oracle.LatentToVideoPipeline(None, unet, osched)(
            latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0, mask=mask, motion=[4.0],
            num_inference_steps=steps, guidance_scale=guidance, return_dict=False, callback=lambda i, t, l: seen_o.append(l.clone()))
        ref_pipe = P.LatentToVideoPipeline(None, None, None, unet, _SchedulerAdapter())
        ref_pipe.decode_latents = lambda lat: torch.zeros(1, 3, lat.shape[2], 8, 8)        # (the VAE is not under test here)
        with torch.no_grad():
            _, got = ref_pipe(height=h * 8, width=w * 8, latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0, mask=mask,
                              motion=[4.0], num_inference_steps=steps, guidance_scale=guidance, return_dict=False, output_type="pt",
                              callback=lambda i, t, l: seen_r.append(l.clone()))
        assert len(seen_o) == len(seen_r) == steps
        for a, b in zip(seen_r, seen_o):
            assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
        assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


def test_append_dims_and_offset_encoder_are_the_references(ref_mod):
    """`_append_dims` (models/pipeline.py:216-221) against the product's; `LatentTransparencyOffsetEncoder`
    (models/layerdiffuse_VAE.py:17-41, pure torch) against the oracle restatement on one state dict."""
    import models.pipeline as P
    import models.layerdiffuse_VAE as LV
    from oracle import layerdiffuse as OL
    from animate_anything_amd.svd_pipeline import _append_dims
    x = torch.arange(6.0).reshape(2, 3)
    for nd in (2, 3, 5):
        assert torch.equal(P._append_dims(x, nd), _append_dims(x, nd))
    with pytest.raises(ValueError):
        _append_dims(x, 1)
    with pytest.raises(ValueError):
        P._append_dims(x, 1)
    torch.manual_seed(0)
    ref = LV.LatentTransparencyOffsetEncoder().eval()
    orc = OL.LatentTransparencyOffsetEncoder().eval()
    state = seeded_state(ref, rezero_std=0.05)                   # the zero-initialised last conv is re-drawn
    ref.load_state_dict(state)
    orc.load_state_dict(state)                                   # identical key layout
    img = torch.rand(2, 4, 24, 40, generator=torch.Generator().manual_seed(1)) * 2 - 1
    with torch.no_grad():
        a, b = ref(img), orc(img)
    assert a.shape == b.shape == (2, 4, 3, 5)
    assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())


def test_latent_helpers_are_the_references():
    """utils/common.py:12-20,32-48,296-300 (`tensor_to_vae_latent`, `DDPM_forward_timesteps`, `calculate_latent_motion_score`)
    executed from the reference file itself against the oracle's and the product's host-side restatements."""
    import oracle
    import refload
    from oracle import pipeline as OP
    from animate_anything_amd import pipeline as PP
    from util import TINY_VAE
    C = refload.load("utils/common.py", absent=("cv2", "torchvision", "torchvision.transforms", "imageio"))
    torch.manual_seed(0)
    vae = oracle.AutoencoderKL(**TINY_VAE).eval()
    vae.load_state_dict(seeded_state(vae))
    g = torch.Generator().manual_seed(2)
    frames = torch.rand(2, 3, 3, 16, 24, generator=g) * 2 - 1                    # [b, f, c, h, w]
    with torch.no_grad():
        a, b = C.tensor_to_vae_latent(frames, vae), OP.tensor_to_vae_latent(frames, vae)
    assert a.shape == b.shape == (2, 4, 3, 8, 12) and torch.allclose(a, b, atol=1e-6)      # TINY_VAE: one downsampling level
    lat = torch.randn(2, 4, 5, 6, 7, generator=g)
    want = C.calculate_latent_motion_score(lat)
    assert torch.allclose(OP.calculate_latent_motion_score(lat), want, atol=1e-6)
    assert torch.allclose(PP.calculate_latent_motion_score(lat), want, atol=1e-6)

    class Sched:                                                                 # the two members DDPM_forward_timesteps touches
        timesteps = torch.tensor([901, 801, 701, 601, 501, 401, 301, 201, 101, 1])

        def add_noise(self, x, noise, t):
            self.seen = (x.clone(), noise.clone(), t.clone())
            return oracle.ddpm_add_noise(x, noise, int(t[0]))
    x0 = torch.randn(1, 4, 1, 6, 7, generator=g)
    s = Sched()
    torch.manual_seed(11)
    xt, ts = C.DDPM_forward_timesteps(x0, 6, 8, s)                               # the reference draws from the global generator
    assert list(ts) == list(Sched.timesteps[4:]) and xt.shape == (1, 4, 8, 6, 7)
    xs, noise, t = s.seen
    assert int(t[0]) == 501 and torch.equal(xs, x0.repeat(1, 1, 8, 1, 1))
    xo, tso = OP.ddpm_forward_timesteps(x0, 6, 8, s, noise=noise)
    assert list(tso) == list(ts) and torch.allclose(xo, xt, atol=1e-6)
