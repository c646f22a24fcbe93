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


def test_reference_pipeline_call_drives_the_product_modules(ref_mod, emu):
    """The drop-in boundary itself (north star: "so train.py --eval still drives it"): the REFERENCE's `LatentToVideoPipeline.__call__`
    (models/pipeline.py:12-214, unmodified, on the stub `TextToVideoSDPipeline` base) given the PRODUCT's `UNet3DConditionModel` (HIP
    kernels on the SIMT emulator) and the PRODUCT's `DPMSolverMultistepScheduler` as its `unet` / `scheduler` attributes.  Every step's
    latents must be (a) equal to the product's own loop over the same module interfaces (`_denoise_generic`) up to the fp16 rounding of the
    guidance arithmetic - the reference combines the two halves in the UNet's dtype (`noise_pred_uncond + g * (text - uncond)` on fp16
    tensors, pipeline.py:181-183), the product in fp32 -, (b) within fp16 tolerance of the product's fused loop (`denoise`: session +
    guidance / solver kernel) and (c) within tolerance of the oracle loop."""
    import oracle
    import models.pipeline as P
    from animate_anything_amd.pipeline import LatentToVideoPipeline as ProductPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler as ProductScheduler
    from animate_anything_amd.unet3d import UNet3DConditionModel as ProductUNet
    assert P.__file__.startswith(REF)
    torch.manual_seed(0)
    orc = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    state = seeded_state(orc)
    orc.load_state_dict(state)
    net = ProductUNet(**TINY_UNET).eval()
    net.load_state_dict(state, strict=True)
    net = net.half()
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g)
    frames, h, w, steps = 3, 8, 8, 3
    x0, noise, pos, neg = r(1, 4, 1, h, w) * 0.5, r(1, 4, frames, h, w), r(1, 7, 64), r(1, 7, 64)
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., 2:6, 2:6] = 1
    osched = oracle.DPMSolverMultistepScheduler()
    osched.set_timesteps(steps)
    init = oracle.ddpm_add_noise(x0.repeat(1, 1, frames, 1, 1), noise, int(osched.timesteps[0]))
    for guidance in (9.0, 1.0):
        kw = dict(latents=init, prompt_embeds=pos.half(), negative_prompt_embeds=neg.half(), condition_latent=x0.half(), mask=mask.half(),
                  motion=[4.0], num_inference_steps=steps, guidance_scale=guidance, return_dict=False)
        seen_ref, seen_gen, seen_fused, seen_orc = [], [], [], []
        ref_pipe = P.LatentToVideoPipeline(None, None, None, net, ProductScheduler())
        ref_pipe.decode_latents = lambda lat: torch.zeros(1, 3, lat.shape[2], 8, 8)        # (no VAE in this test)
        with torch.no_grad():
            _, got = ref_pipe(height=h * 8, width=w * 8, output_type="pt", callback=lambda i, t, l: seen_ref.append(l.float().clone()), **kw)
        # (a) the product's loop over the same module interfaces
        gen = ProductPipeline(vae=None, unet=net, scheduler=ProductScheduler())
        gen.denoise = gen._denoise_generic
        gen(callback=lambda i, t, l: seen_gen.append(l.float().clone()), **kw)
        # (b) the product's default loop (session + fused guidance / solver kernel)
        ProductPipeline(vae=None, unet=net, scheduler=ProductScheduler())(callback=lambda i, t, l: seen_fused.append(l.float().clone()), **kw)
        # (c) the oracle
        okw = dict(kw, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0, mask=mask)
        oracle.LatentToVideoPipeline(None, orc, osched)(callback=lambda i, t, l: seen_orc.append(l.clone()), **okw)
        assert len(seen_ref) == len(seen_gen) == len(seen_fused) == len(seen_orc) == steps
        for a, b, c, d in zip(seen_ref, seen_gen, seen_fused, seen_orc):
            scale = max(1.0, d.abs().max().item())
            assert (a - b).abs().max().item() <= (2e-3 if guidance > 1.0 else 1e-6) * scale      # (no guidance: the same arithmetic, bit for bit up to fp32 round-off)
            assert (a - c).abs().max().item() <= 1e-2 * scale
            assert (a - d).abs().max().item() <= 3e-2 * scale
        assert got.shape == init.shape


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


# --------------------------------------------------------------------------------------------- SVD pipelines
def _svd_parts(in_channels):
    import oracle.svd as O
    from util import svd_state
    from test_svd import TINY_SVD_UNET, TINY_SVD_VAE
    torch.manual_seed(0)
    unet = O.UNetSpatioTemporalConditionModel(**dict(TINY_SVD_UNET, in_channels=in_channels)).eval()
    unet.load_state_dict(svd_state(unet))
    vae = O.AutoencoderKLTemporalDecoder(**TINY_SVD_VAE).eval()
    vae.load_state_dict(svd_state(vae, 1))
    return O, unet, vae


def _svd_case(seed, b=1, f=3, H=16, W=24):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(b, 3, H, W, generator=g) * 2 - 1
    emb = torch.randn(b, 1, 64, generator=g)
    latents = torch.randn(b, f, 4, H // 2, W // 2, generator=g)          # the tiny VAE scales by 2
    mask = torch.zeros(1, H // 2, W // 2)
    mask[:, 2:6, 3:9] = 1
    return image, emb, latents, mask


def _close(a, b, tol=1e-5):
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("guided", [True])
def test_reference_mask_svd_pipeline_call_equals_oracle(ref_mod, guided):
    """The reference's own `MaskStableVideoDiffusionPipeline.__call__` (models/pipeline.py:225-466: image-noise augmentation,
    `repeat(mask, '1 h w -> 2 f 1 h w')`, `cat([mask, latent_model_input, image_latents], dim=2)` (:417-422), per-frame guidance
    (:405-410, :435-440), the callback protocol) on the stub base class with oracle parts == `oracle.svd.svd_pipeline`,
    latents after EVERY step.  (The reference hard-codes the guidance pair into the mask repeat: it has no unguided form.)"""
    import models.pipeline as P
    O, unet, vae = _svd_parts(9)
    image, emb, latents, mask = _svd_case(21)
    steps, frames, seed = 3, 3, 77
    aug = torch.randn(image.shape, generator=torch.Generator().manual_seed(seed))
    seen_o, seen_r = [], []
    with torch.no_grad():
        want = O.svd_pipeline(unet, vae, O.EulerDiscreteScheduler(), image, torch.cat([torch.zeros_like(emb), emb]), mask=mask,
                              num_frames=frames, num_inference_steps=steps, latents=latents.clone(), aug_noise=aug,
                              min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=100, noise_aug_strength=0.05,
                              callback=lambda i, t, l: seen_o.append(l.clone()))
        pipe = P.MaskStableVideoDiffusionPipeline(vae, None, unet, O.EulerDiscreteScheduler())
        pipe.image_embeddings = emb

        def cb(p_, i, t, kw):
            seen_r.append(kw["latents"].clone())
            return {}
        got = pipe(image, height=16, width=24, num_frames=frames, num_inference_steps=steps, latents=latents.clone(), mask=mask,
                   generator=torch.Generator().manual_seed(seed), min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7,
                   motion_bucket_id=100, noise_aug_strength=0.05, output_type="latent", return_dict=False, callback_on_step_end=cb)
    assert len(seen_o) == len(seen_r) == steps
    for a, b in zip(seen_r, seen_o):
        assert _close(a, b)
    assert got.shape == want.shape == (1, frames, 4, 8, 12) and _close(got, want)
    # ... and the decode helper of the stub is the oracle's: frames come back [B, 3, F, H, W]
    assert pipe.decode_latents(got, frames, 2).shape == (1, 3, frames, 16, 24)


@pytest.mark.parametrize("condition_type,in_channels,max_g", [("text", 8, 2.5), ("image", 9, 3.0), ("both", 9, 2.0), ("text", 9, 1.0)])
def test_reference_text_svd_pipeline_call_equals_oracle(ref_mod, condition_type, in_channels, max_g):
    """`TextStableVideoDiffusionPipeline.__call__` (models/pipeline.py:470-731): the three `condition_type` branches (:604-616),
    `mask` doubled under guidance (:618-619), a caller-supplied `condition_latent` doubled (:647-649) or the VAE-encoded image,
    8- and 9-channel UNets (:697-702), and the unguided form (max_guidance_scale = 1)."""
    import models.pipeline as P
    O, unet, vae = _svd_parts(in_channels)
    image, emb, latents, mask = _svd_case(31)
    steps, frames, seed = 2, 3, 5
    cfg = max_g > 1.0
    g = torch.Generator().manual_seed(13)
    pe, ne = torch.randn(1, 5, 64, generator=g), torch.randn(1, 5, 64, generator=g)
    cond = torch.randn(1, frames, 4, 8, 12, generator=g) if condition_type == "text" else None
    img_e = torch.cat([torch.zeros_like(emb), emb]) if cfg else emb
    txt_e = torch.cat([ne, pe]) if cfg else pe
    ctx = {"text": txt_e, "image": img_e, "both": torch.cat([img_e, txt_e], dim=1)}[condition_type]
    aug = torch.randn(image.shape, generator=torch.Generator().manual_seed(seed))
    seen_o, seen_r = [], []
    with torch.no_grad():
        want = O.svd_pipeline(unet, vae, O.EulerDiscreteScheduler(), image, ctx, mask=mask if in_channels == 9 else None,
                              num_frames=frames, num_inference_steps=steps, latents=latents.clone(), aug_noise=aug,
                              condition_latent=cond, min_guidance_scale=1.0 if cfg else 1.0, max_guidance_scale=max_g,
                              callback=lambda i, t, l: seen_o.append(l.clone()))
        pipe = P.TextStableVideoDiffusionPipeline(vae, None, unet, O.EulerDiscreteScheduler())
        pipe.image_embeddings = emb
        mask5 = mask.reshape(1, 1, 1, *mask.shape[-2:]).repeat(1, frames, 1, 1, 1)          # [b, f, 1, h, w] (train_svd.py eval)

        def cb(p_, i, t, kw):
            seen_r.append(kw["latents"].clone())
            return {}
        got = pipe(image, prompt_embeds=pe, negative_prompt_embeds=ne, height=16, width=24, num_frames=frames,
                   num_inference_steps=steps, latents=latents.clone(), condition_type=condition_type, condition_latent=cond,
                   mask=mask5, generator=torch.Generator().manual_seed(seed), min_guidance_scale=1.0, max_guidance_scale=max_g,
                   output_type="latent", return_dict=False, callback_on_step_end=cb)
    assert len(seen_o) == len(seen_r) == steps
    for a, b in zip(seen_r, seen_o):
        assert _close(a, b)
    assert _close(got, want)


# --------------------------------------------------------------------------------------------- layerdiffuse RGBA add-ons (row f3)
def test_reference_unet384_forward_equals_oracle(ref_mod):
    """The reference's own `UNet384` (models/layerdiffuse_VAE.py:44-177: constructor channel arithmetic through
    `get_down_block` / `UNetMidBlock2D` / `get_up_block`, `latent_conv_in` added in front of the fourth down block :156-157, the
    residual-tuple bookkeeping of its forward :160-169) built on the stub 2-D blocks == `oracle.layerdiffuse.UNet384` on one state
    dict (identical key sets), default architecture, two image sizes."""
    import models.layerdiffuse_VAE as LV
    from oracle import layerdiffuse as OL
    assert LV.__file__.startswith(REF)
    torch.manual_seed(0)
    ref = LV.UNet384().eval()
    orc = OL.UNet384().eval()
    state = seeded_state(orc, rezero_std=0.05)                   # the zero-initialised latent_conv_in is re-drawn: the injection counts
    assert sorted(ref.state_dict().keys()) == sorted(state.keys())
    ref.load_state_dict(state, strict=True)
    orc.load_state_dict(state)
    g = torch.Generator().manual_seed(3)
    for (h, w) in ((32, 48), (64, 64)):
        x = torch.rand(2, 3, h, w, generator=g) * 2 - 1
        lat = torch.randn(2, 4, h // 8, w // 8, generator=g)
        with torch.no_grad():
            a, b = ref(x, lat), orc(x, lat)
            a0 = ref(x, torch.zeros_like(lat))
        assert a.shape == b.shape == (2, 4, h, w)
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
        assert (a - a0).abs().max().item() > 1e-4               # the latent really enters


def test_reference_stage2_pipeline_call_equals_oracle(ref_mod):
    """`MaskedLatentToVideoPipeline.__call__` (models/pipeline_stage2.py:173-337): guidance order `cat([prompt_embeds[1],
    prompt_embeds[0]])` over `encode_prompt`'s (positive, negative) pair (:229-231), the loop (the same body as
    LatentToVideoPipeline's), and the alpha decode (:290-318: VAE frames + latents -> `vae_alpha_decoder` -> thresholded alpha,
    (fg + 1) * 127.5, uint8 RGBA) == the oracle loop + `oracle.layerdiffuse.decode_rgba`."""
    import oracle
    import types
    absent = [m for m in ("torchvision", "torchvision.transforms") if m not in sys.modules]      # (imported at the top of the file, unused by this class)
    for m in absent:
        sys.modules[m] = types.ModuleType(m)
    try:
        import models.pipeline_stage2 as P2
    finally:
        for m in absent:
            del sys.modules[m]
    from oracle import layerdiffuse as OL
    from util import TINY_VAE
    assert P2.__file__.startswith(REF)
    torch.manual_seed(0)
    unet = oracle.UNet3DConditionModel(**TINY_UNET).eval()
    unet.load_state_dict(seeded_state(unet))
    vae = oracle.AutoencoderKL(**dict(TINY_VAE, block_out_channels=(32, 32, 32, 32))).eval()       # x8, like the real VAE: the alpha decoder
    vae.load_state_dict(seeded_state(vae, 1))                                                       # takes the latent at 1/8 of the frame
    dec = OL.UNet384(block_out_channels=(8, 16, 32, 32), layers_per_block=1, attention_head_dim=8).eval()
    dec.load_state_dict(seeded_state(dec, 2, rezero_std=0.05))
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g)
    frames, h, w, steps = 2, 4, 6, 3
    x0, noise, pos, neg = r(1, 4, 1, h, w) * 0.5, r(1, 4, frames, h, w), r(1, 7, 64), r(1, 7, 64)
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., 1:3, 2:5] = 1
    osched = oracle.DPMSolverMultistepScheduler()
    osched.set_timesteps(steps)
    init = oracle.ddpm_add_noise(x0.repeat(1, 1, frames, 1, 1), noise, int(osched.timesteps[0]))
    with torch.no_grad():
        opipe = oracle.LatentToVideoPipeline(vae, unet, osched)
        video_o, lat_o = opipe(latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0, mask=mask, motion=[4.0],
                               num_inference_steps=steps, guidance_scale=9.0, return_dict=False, output_type="pt")
        png_o, alpha_o, rgb_o = OL.decode_rgba(video_o, lat_o, dec)
        ref_pipe = P2.MaskedLatentToVideoPipeline(vae, None, None, unet, _SchedulerAdapter())
        # diffusers 0.24 `encode_prompt` returns the (positive, negative) pair un-concatenated; the reference re-orders it itself
        ref_pipe.encode_prompt = lambda prompt, device, n, cfg, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None: (prompt_embeds, negative_prompt_embeds)
        video_r, lat_r, png_r, alpha_r, rgb_r = ref_pipe(
            vae_alpha_decoder=dec, height=h * 8, width=w * 8, latents=init, prompt_embeds=pos, negative_prompt_embeds=neg, condition_latent=x0,
            mask=mask, motion=[4.0], num_inference_steps=steps, guidance_scale=9.0, return_dict=False, output_type="pt")
    assert (lat_r - lat_o).abs().max().item() <= 1e-5 * max(1.0, lat_o.abs().max().item())
    assert (video_r - video_o).abs().max().item() <= 1e-5
    assert png_r.shape == png_o.shape == (frames, h * 8, w * 8, 4) and png_r.dtype == png_o.dtype
    # uint8 after a threshold: allow the odd pixel whose pre-rounding value sits on an integer boundary to differ by one count
    diff = (png_r.astype(int) - png_o.astype(int))
    assert (abs(diff[..., :3]) <= 1).all() and (diff[..., :3] != 0).mean() < 1e-3
    assert (alpha_r == alpha_o).mean() > 0.999 and (rgb_r.astype(int) - rgb_o.astype(int)).__abs__().max() <= 1
